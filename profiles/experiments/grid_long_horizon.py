#!/usr/bin/env python
"""The protocol of bench.py's long-horizon leg through ONE propagate_grid() call into a device buffer: 1 048 576 outer
Solar Systems, 8 snapshots over 1e4 yr (lock-step sweeps; Taylor coefficients on demand)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import heyoka_amd as hy
from heyoka_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0e4
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
sys_ = hy.model.nbody(6, masses=M, Gconst=G)
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True)
grid = np.linspace(0.0, T, 9)
out = torch.empty((9, 36, n), dtype=torch.float64, device="cuda:0")
cf = hy.cfunc([hy.model.nbody_energy(6, masses=M, Gconst=G)], sys_.vars)
t0 = time.perf_counter()
ta.propagate_grid_device(grid, out.data_ptr())
torch.cuda.synchronize()
el = time.perf_counter() - t0
ns = ta.propagate_res_arrays()[3]
e = torch.empty((9, n), dtype=torch.float64, device="cuda:0")
for k in range(9):
    cf.eval_device(e[k].data_ptr(), out[k].data_ptr(), n)
err = float(((e[1:] - e[0]) / e[0]).abs().max())
print("propagate_grid over %.0f yr, %d systems, 8 snapshots: %.2f s, %d sweeps, %.3g system-steps/s, max rel energy error %.2e"
      % (T, n, el, int(ns.max()), float(ns.sum()) / el, err))
