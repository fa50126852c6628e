#!/usr/bin/env python
"""Which resource binds the v5 stepper? Dummy instructions of one kind at a time are added to every order of the step
(HEYOKA_AMD_V5_PAD = chain:dep:st:ld:salu, results untouched) and the variants run interleaved in one process like ab.py:
the slope of the time per wavefront-step against the number of added instructions is the marginal cost of an instruction
of that kind in that section."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

n = 1048576
sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
variants = [("base", "0:0:0:0:0"), ("st+2", "0:0:2:0:0"), ("st+4", "0:0:4:0:0"), ("st128+1", "0:0:100:0:0"), ("st128+2", "0:0:200:0:0"), ("st128+4", "0:0:400:0:0")]
tas = []
for name, pad in variants:
    os.environ["HEYOKA_AMD_V5_PAD"] = pad
    tas.append(hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True))
del os.environ["HEYOKA_AMD_V5_PAD"]
rates = [[] for _ in tas]
t = 0.0
for r in range(5):
    t += 40.0
    for i, ta in enumerate(tas):
        ta.propagate_until(t)
        ns = ta.propagate_res_arrays()[3]
        ms = list(ta.kernel_ms_history(1))[-1]
        if r > 0:
            rates[i].append(float(ns.sum()) / (ms * 1e-3))
base = np.mean(rates[0])
# time per wavefront-step of four systems on one of the 2048 resident wavefronts, in ns
tws = lambda rate: 4.0 / rate * 2048 * 1e9
for (name, pad), rr in zip(variants, rates):
    m = np.mean(rr)
    print(json.dumps({"variant": name, "pad": pad, "rate": "%.4g" % m, "ns_per_wavefront_step": "%.1f" % tws(m),
                      "delta_ns": "%.1f" % (tws(m) - tws(base))}))
