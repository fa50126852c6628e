#!/usr/bin/env python
"""model::fixed_centres with a centre at the origin and a repeated coordinate: the table stepper (planner pass switched off
with HEYOKA_AMD_NO_PRIVATE_INPUTS=1) against the cluster / block kernels on the internal program with private coordinate
differences. usage: private_inputs.py [--systems N]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=65536)
args = ap.parse_args()
n = args.systems
for nc, T in ((20, 20.0), (100, 5.0)):
    rng = np.random.default_rng(5)
    pos = rng.uniform(-2.0, 2.0, (nc, 3))
    pos[0] = 0.0
    pos[3, 0] = pos[2, 0]
    m = rng.uniform(0.1, 1.0, nc)
    sys_ = hy.model.fixed_centres(Gconst=1.0, masses=[float(x) for x in m], positions=[float(x) for x in pos.reshape(-1)])
    b = np.array([6.0, 0.3, -0.2, 0.0, 2.4 if nc == 100 else 1.2, 0.1])[:, None]
    st = b + 1e-2 * (np.abs(b) + 0.05) * np.random.RandomState(7).uniform(-1, 1, (6, n))
    for off in ("1", None):
        if off:
            os.environ["HEYOKA_AMD_NO_PRIVATE_INPUTS"] = off
        else:
            os.environ.pop("HEYOKA_AMD_NO_PRIVATE_INPUTS", None)
        ta = hy.taylor_adaptive_batch(sys_, st, n)
        ta.propagate_until(T / 4)
        ta.propagate_until(T)
        ns = ta.propagate_res_arrays()[3]
        ms = list(ta.kernel_ms_history(1))[-1]
        print(json.dumps({"centres": nc, "private_inputs": off is None, "system_steps_per_s": "%.4g" % (float(ns.sum()) / (ms * 1e-3)),
                          "mode": ta.hip_source_mode[:60]}))
