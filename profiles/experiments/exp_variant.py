#!/usr/bin/env python
"""Experiment driver: outer-SS ensemble throughput of the stepper selected by the HEYOKA_AMD_* knobs in the
environment. Prints one JSON line (kernel ms per launch from HIP events, system-steps/s, registers)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=1048576)
ap.add_argument("--tol", type=float, default=0.0)
ap.add_argument("--dt", type=float, default=4.0)
ap.add_argument("--calls", type=int, default=4)
ap.add_argument("--tag", default="")
args = ap.parse_args()
sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
n = args.systems
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
kw = {"tol": args.tol} if args.tol > 0 else {}
ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, **kw)
t = 0.0
tot = []
for c in range(args.calls + 1):
    t += args.dt
    t0 = time.perf_counter()
    ta.propagate_until(t)
    oc, mn, mx, ns = ta.propagate_res_arrays()
    el = time.perf_counter() - t0
    tot.append((float(ns.sum()), el))
kms = list(ta.kernel_ms_history(args.calls))
steps = np.array([x[0] for x in tot[1:]])
rate = steps / (np.array(kms) * 1e-3)
e0 = configs.outer_ss_energy(st, n) if hasattr(configs, "outer_ss_energy") else None
print(json.dumps({"tag": args.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("HEYOKA_AMD")},
                  "order": ta.order, "mode": ta.hip_source_mode[:60], "kernel_ms": kms, "steps_per_launch": steps.tolist(),
                  "system_steps_per_s_kernel": rate.tolist(), "mean_rate": float(rate.mean()),
                  "state_checksum": float(np.abs(ta.state).sum())}))
