#!/usr/bin/env python
"""A few lock-step steps of an ensemble in which events fire (the events leg of bench.py, smaller): for a kernel trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
sys_ = hy.model.nbody(6, masses=M, Gconst=G)
st0 = configs.outer_ss_state(n, perturb=1e-6, seed=4243)
sp = hy.taylor_adaptive_batch(sys_, st0, n, high_accuracy=True)
sp.propagate_until(np.random.RandomState(4244).uniform(0.0, 30.0, n))
st = np.array(sp.state)
del sp
y1, y2, y3 = hy.make_vars("y_1", "y_2", "y_3")
c, ct = hy.native_event_counter(), hy.native_event_counter()
kw = {"t_events": [hy.t_event(y3, ct)]} if len(sys.argv) > 2 and sys.argv[2] == "terminal" else {}
ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=[hy.nt_event(y1, c), hy.nt_event(y2, c)], **kw)
print(ta.hip_source_mode[-150:])
for _ in range(8):
    ta.step()
_ = ta.time
t0 = time.perf_counter()
for _ in range(4):
    ta.step()
_ = ta.time
print("s per step", (time.perf_counter() - t0) / 4, "events", c.value, ta.event_stats)
