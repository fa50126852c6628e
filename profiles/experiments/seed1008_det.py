#!/usr/bin/env python
"""Run-to-run determinism of the staged table stepper on one pseudo-random system of tests/test_gpu_parity.py: the Taylor
coefficients of repeated single steps (fresh integrators) against those of the first one, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import numpy as np
import heyoka_amd as hy
from test_gpu_parity import _random_system, ho
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1008
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = int(sys.argv[3]) if len(sys.argv) > 3 else 33
rs = np.random.RandomState(100 + seed)
st = rs.uniform(-0.7, 0.7, (3, n)); pars = rs.uniform(-0.5, 0.5, (2, n)); t0 = rs.uniform(0.0, 2.0, n)
os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
ref = None
for r in range(reps):
    sys_p = _random_system(hy, np.random.RandomState(seed), extended=seed >= 1000)
    ta = hy.taylor_adaptive_batch(sys_p, st, n, pars=pars, time=t0)
    ta.step(write_tc=True)
    tc = np.asarray(ta.tc).reshape(3, 21, n).copy()
    if ref is None:
        ref = tc
        print(ta.hip_source_mode[:200])
        continue
    bad = np.argwhere(tc != ref)
    if len(bad) == 0:
        print(r, "identical")
        continue
    lanes = sorted(set(int(b[2]) for b in bad))
    msg = []
    for l in lanes[:6]:
        bl = bad[bad[:, 2] == l]
        o = int(bl[:, 1].min())
        vs = sorted(set(int(b[0]) for b in bl if b[1] == o))
        rel = max(abs(tc[v, o, l] - ref[v, o, l]) / abs(ref[v, o, l]) for v in vs)
        msg.append("system %d: first order %d vars %s rel %.2e" % (l, o, vs, rel))
    print(r, len(lanes), "systems differ;", "; ".join(msg), flush=True)
