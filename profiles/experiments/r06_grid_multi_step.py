"""Round 6: callback-free propagate_grid() over the headline ensemble - launches from grid point to grid point
(emitted_module::grid_multi_step) against the single-step lock-step sweeps (forced here through max_steps, which counts
lock-step iterations and keeps the old loop), same grid, results compared.
  python profiles/experiments/r06_grid_multi_step.py [n_systems] [t_end] [n_grid]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import heyoka_amd as hy
from heyoka_amd import configs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 9
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
st = configs.outer_ss_state(N, perturb=1e-12, seed=42)
grid = np.linspace(0.0, T, NG)
res = {}
for name, kw in (("grid point to grid point", {}), ("single-step sweeps", {"max_steps": 10 ** 9})):
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, N, high_accuracy=True)
    view = torch.as_tensor(ta.device_array("state"), device="cuda:0")
    view.copy_(torch.from_numpy(st))
    torch.cuda.synchronize()
    ta.mark_device_modified()
    out = torch.empty((NG, 36, N), dtype=torch.float64, device="cuda:0")
    # (Warm-up: a short grid.)
    ta.propagate_grid_device(np.array([0.0, 0.5]), out.data_ptr(), **kw)
    ta.time = 0.0
    view.copy_(torch.from_numpy(st))
    torch.cuda.synchronize()
    ta.mark_device_modified()
    t0 = time.perf_counter()
    ta.propagate_grid_device(grid, out.data_ptr(), **kw)
    ta.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    oc, mn, mx, ns = ta.propagate_res_arrays()
    res[name] = (out.cpu().numpy(), np.asarray(ns), np.asarray(mn), np.asarray(mx))
    print("%-26s %.3e system-steps/s  (%.3f s, %d system-steps, outcomes ok: %s)"
          % (name, float(ns.sum()) / el, el, int(ns.sum()), bool(np.all(oc == int(hy.taylor_outcome.time_limit)))), flush=True)
    del ta, out, view
    torch.cuda.empty_cache()
a, b = res["grid point to grid point"], res["single-step sweeps"]
print("identical grid output:", np.array_equal(a[0], b[0]), " step counts:", np.array_equal(a[1], b[1]), " min/max h:",
      np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]))
