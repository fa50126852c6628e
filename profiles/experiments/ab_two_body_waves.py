#!/usr/bin/env python
"""Two-body stepper (BASELINE.json configs[2]): one wavefront per SIMD with all 512 registers (410 used) against two
wavefronts per SIMD with 256 registers each (amdgpu_waves_per_eu(2): 160 spilled registers) - the experiment VERDICT r4
asked for. Interleaved in one process like ab.py."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs, codegen_check

n = 4194304
sys_ = hy.model.nbody(2, masses=[1.0, 0.0])
st = configs.two_body_state(n, perturb=1e-12, seed=42)
tas, names = [], []
for w in ("0", "2"):
    os.environ["HEYOKA_AMD_UNROLLED_WAVES"] = w
    tas.append(hy.taylor_adaptive_batch(sys_, st, n))
    names.append("waves_per_simd=%s %s" % (w or "auto", codegen_check.kernel_resources(tas[-1].code_object)))
del os.environ["HEYOKA_AMD_UNROLLED_WAVES"]
rates = [[] for _ in tas]
t = 0.0
for r in range(5):
    t += 50.0
    for i, ta in enumerate(tas):
        ta.propagate_until(t)
        ns = ta.propagate_res_arrays()[3]
        ms = list(ta.kernel_ms_history(1))[-1]
        if r > 0:
            rates[i].append(float(ns.sum()) / (ms * 1e-3))
for v, rr in zip(names, rates):
    print(json.dumps({"variant": v, "rates": ["%.4g" % x for x in rr], "mean": "%.4g" % np.mean(rr)}))
