import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
def mk(n, st, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
for n in (4, 64, 128):
    for pert in (1e-12, 1e-8):
        st = configs.outer_ss_state(n, perturb=pert, seed=3)
        for T in (0.5, 4.0, 20.0, 100.0):
            a = mk(n, st, {}); b = mk(n, st, {"HEYOKA_AMD_CLUSTER_V1": "1"})
            a.propagate_until(T); b.propagate_until(T)
            sa, sb = a.state, b.state
            err = np.max(np.abs(sa - sb) / np.maximum(1, np.abs(sb)))
            na = np.array([r[3] for r in a.propagate_res]); nb = np.array([r[3] for r in b.propagate_res])
            print("n=%d pert=%g T=%g err=%.3g steps v2 %d..%d v1 %d..%d finite=%s" % (n, pert, T, err, na.min(), na.max(), nb.min(), nb.max(), np.isfinite(sa).all()), flush=True)
# steps then propagate
n = 128
st = configs.outer_ss_state(n, perturb=1e-10, seed=42)
a = mk(n, st, {}); b = mk(n, st, {"HEYOKA_AMD_CLUSTER_V1": "1"})
for i in range(3):
    a.step(write_tc=True); b.step(write_tc=True)
    print("step", i, np.max(np.abs(a.state - b.state)))
a.propagate_until(10.0); b.propagate_until(10.0)
print("after prop", np.max(np.abs(a.state - b.state)), [r[3] for r in a.propagate_res][:4], [r[3] for r in b.propagate_res][:4])
