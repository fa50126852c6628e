import os, sys
sys.path.insert(0, os.getcwd())
os.environ["HEYOKA_AMD_COMPACT_TC"] = "0"
import numpy as np, heyoka_amd as hy
from heyoka_amd import configs
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
n = 8
st = configs.outer_ss_state(n, perturb=1e-3, seed=12)
x1, y2 = hy.make_vars("x_1", "y_2")
ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=[hy.nt_event(x1 - y2, lambda *a: None)])
ta.step()
tc = np.asarray(ta.tc).reshape(36, 21, n)
for body in (0, 1):
    for c in range(3):
        xi, vi = 6 * body + c, 6 * body + 3 + c
        for k in (1, 2, 3, 5, 7, 20):
            a = tc[xi, k]; b = tc[vi, k - 1] * (1.0 / k); q = tc[vi, k - 1] / k
            print(body, c, k, np.array_equal(a, b), np.array_equal(a, q), float(np.max(np.abs(a - b) / np.abs(a))))
