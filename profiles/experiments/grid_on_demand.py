#!/usr/bin/env python
"""propagate_grid() over a long horizon with few grid points, into a device buffer: the lock-step loop with the Taylor
coefficients of every step (a trivial step callback forces that) against the coefficients on demand (no callback)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import heyoka_amd as hy
from heyoka_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
T = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
sys_ = hy.model.nbody(6, masses=M, Gconst=G)
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
grid = np.outer(np.linspace(0.0, T, 5), np.ones(n))
res = {}
for name, cb in (("on_demand", None), ("every_step", lambda t: True)):
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True)
    # (No setter / mutable getter before the loop: a handed-out host pointer switches the integrator to eager
    # synchronisation - a download of the state after every sweep.)
    t0 = time.perf_counter()
    _, out = ta.propagate_grid(grid, **({"callback": cb} if cb else {}))
    el = time.perf_counter() - t0
    ns = np.array([r[3] for r in ta.propagate_res])
    res[name] = out
    print(name, "wall %.2f s, %d sweeps, %.3g system-steps/s" % (el, int(ns.max()), float(ns.sum()) / el), flush=True)
    del ta
print("identical:", np.array_equal(res["on_demand"], res["every_step"]))
