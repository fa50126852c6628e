#!/usr/bin/env python
"""Kernel time of single-step launches (lock-step sweeps: step(), callbacks, propagate_grid, events) against the
propagation loop, per cluster kernel. usage: single_step.py [--systems N]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=1048576)
ap.add_argument("--kernels", default="5,3")
ap.add_argument("--steps-last", action="store_true", help="end with single-step launches (counter passes read the LAST dispatch)")
args = ap.parse_args()
n = args.systems
sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
st = configs.outer_ss_state(n, perturb=1e-6, seed=42)
for ck in [int(x) for x in args.kernels.split(",")]:
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, cluster_kernel=ck)
    ta.step()
    _ = ta.time
    ms = []
    for _ in range(4):
        ta.step()
        ms.append(list(ta.kernel_ms_history(1))[-1])
    t0 = time.perf_counter()
    for _ in range(4):
        ta.step()
    _ = ta.time
    wall = (time.perf_counter() - t0) / 4
    ta.propagate_until(float(ta.time[0]) + 40.0)
    ns = ta.propagate_res_arrays()[3]
    pms = list(ta.kernel_ms_history(1))[-1]
    if args.steps_last:
        for _ in range(3):
            ta.step()
        _ = ta.time
    print(json.dumps({"cluster_kernel": ck, "systems": n, "step_kernel_ms": ["%.3f" % x for x in ms], "step_wall_ms": "%.3f" % (wall * 1e3),
                      "single_step_rate": "%.4g" % (n / (np.mean(ms) * 1e-3)),
                      "propagate_rate": "%.4g" % (float(ns.sum()) / (pms * 1e-3)), "mode": ta.hip_source_mode[:90]}))
