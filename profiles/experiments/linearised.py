#!/usr/bin/env python
"""Pair-interaction systems with equal / repeated masses: the kernel the planner reaches with the accelerations flattened in
the internal program (linearise_accelerations()) against the one it reached without (HEYOKA_AMD_NO_LINEARISED_SUMS=1)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
cases = {
    "nbody(6), default masses": (6, {}),
    "nbody(6), masses [1, 1e-3, 1, 2, 1e-3, 0.5]": (6, {"masses": [1.0, 1e-3, 1.0, 2.0, 1e-3, 0.5]}),
    "nbody(8), default masses": (8, {}),
}
for name, (nb, kw) in cases.items():
    st = configs.plummer_nbody_state(nb, n, seed=5)
    for lin in (False, True):
        if not lin:
            os.environ["HEYOKA_AMD_NO_LINEARISED_SUMS"] = "1"
        else:
            os.environ.pop("HEYOKA_AMD_NO_LINEARISED_SUMS", None)
        ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, **kw), st, n)
        ta.propagate_until(0.02)
        rates = []
        for k in range(3):
            ta.propagate_until(0.02 * (k + 2))
            ns = ta.propagate_res_arrays()[3]
            rates.append(float(ns.sum()) / (list(ta.kernel_ms_history(1))[-1] * 1e-3))
        print(json.dumps({"system": name, "linearised": lin, "system_steps_per_s": "%.4g" % np.mean(rates),
                          "kernel": ta.hip_source_mode.split(":")[1][:70] if ":" in ta.hip_source_mode else ta.hip_source_mode[:80]}), flush=True)
        del ta
