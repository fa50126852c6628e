import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import heyoka_amd as hy, heyoka_oracle as ho
from heyoka_amd import configs
EPS = np.finfo(float).eps
n = 257
st = configs.two_body_state(n, perturb=1e-2, seed=12)
def rel(a, b): return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))
for direction in (+1, -1):
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    ora = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n)
    for i in range(5):
        if direction > 0:
            ta.step(); ora.step()
        else:
            ta.step_backward(); ora.step(backward=True)
        hg = np.array([h for _, h in ta.step_res]); hr = np.array([h for _, h in ora.step_res])
        print(direction, i, "h err %.3g eps" % (np.max(np.abs(hg - hr) / np.abs(hr)) / EPS), "state err %.3g eps" % (rel(ta.state, ora.state.reshape(12, n)) / EPS), flush=True)
# lock-step with callback, backward, k iterations
for k in (1, 2, 5):
    cnt = []
    tb = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    tb.propagate_until(-5.0, callback=lambda t: cnt.append(1) or len(cnt) < k)
    orb = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n)
    orb.propagate_until(-5.0, max_steps=k)
    e = np.abs(tb.state - orb.state.reshape(12, n)) / np.maximum(1.0, np.abs(orb.state.reshape(12, n)))
    lane = np.unravel_index(np.argmax(e), e.shape)
    print("lockstep k", k, "err %.3g eps" % (e.max() / EPS), "at", lane, "time diff", np.max(np.abs(tb.time - orb.time_hi)), "t lane", tb.time[lane[1]], orb.time_hi[lane[1]], flush=True)
