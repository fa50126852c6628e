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
for n in (4, 6, 128):
    st = configs.outer_ss_state(n, perturb=1e-8, seed=3)
    a = mk(n, st, {"HEYOKA_AMD_WAVE_ROLES": "1"}); b = mk(n, st, {"HEYOKA_AMD_CLUSTER_V1": "1"})
    if n == 4: print(a.hip_source_mode[:120], flush=True)
    a.step(write_tc=True); b.step(write_tc=True)
    ta_, tb_ = np.asarray(a.tc), np.asarray(b.tc)
    print("n=%d step: h %s state diff %.3g tc diff %.3g" % (n, np.allclose(a.last_h, b.last_h, rtol=1e-10), np.max(np.abs(a.state-b.state)), np.max(np.abs(ta_-tb_)/(np.abs(tb_)+1e-30*0+1e-300).clip(1e-20))), flush=True)
    for T in (4.0, 20.0, 100.0):
        a = mk(n, st, {"HEYOKA_AMD_WAVE_ROLES": "1"}); b = mk(n, st, {"HEYOKA_AMD_CLUSTER_V1": "1"})
        a.propagate_until(T); b.propagate_until(T)
        err = np.max(np.abs(a.state - b.state) / np.maximum(1, np.abs(b.state)))
        na = np.array([r[3] for r in a.propagate_res]); nb = np.array([r[3] for r in b.propagate_res])
        print("n=%d T=%g err=%.3g steps v4 %d..%d v1 %d..%d" % (n, T, err, na.min(), na.max(), nb.min(), nb.max()), flush=True)
