import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
def mk(n, st, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return hy.taylor_adaptive_batch(hy.model.np1body(8), st, n)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
n = 8
rng = np.random.default_rng(5)
# 7 bodies around a unit central mass: rough circular-ish orbits
st = np.zeros((42, n))
for b in range(7):
    r = 1.0 + 0.7 * b
    ph = 0.9 * b
    v = 1.0 / np.sqrt(r)
    st[6*b+0] = r*np.cos(ph) + 1e-3*rng.standard_normal(n); st[6*b+1] = r*np.sin(ph); st[6*b+2] = 0.01*b
    st[6*b+3] = -v*np.sin(ph); st[6*b+4] = v*np.cos(ph); st[6*b+5] = 0.0
a = mk(n, st, {}); b = mk(n, st, {"HEYOKA_AMD_PAIR_SPLIT": "0"})
print(a.hip_source_mode[:90]); print(b.hip_source_mode[:90], flush=True)
a.step(write_tc=True); b.step(write_tc=True)
print("h", [h for _, h in a.step_res][:3], [h for _, h in b.step_res][:3])
print("state diff", np.max(np.abs(a.state - b.state)), flush=True)
ta, tb = np.asarray(a.tc).reshape(42, a.order+1, n), np.asarray(b.tc).reshape(42, a.order+1, n)
d = np.abs(ta - tb) / (np.max(np.abs(tb), axis=2, keepdims=True) + 1e-300)
print("tc diff per order", np.max(d, axis=(0, 2)))
a.propagate_until(0.5, max_steps=40); b.propagate_until(0.5, max_steps=40)
print("prop", np.max(np.abs(a.state - b.state)), [r[3] for r in a.propagate_res][:4], [r[3] for r in b.propagate_res][:4])
