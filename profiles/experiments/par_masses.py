import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
pars = np.repeat(np.asarray(M, dtype=np.float64)[:, None], n, axis=1)
res = {}
for tag, env in (("cluster", {}), ("table", {"HEYOKA_AMD_EMIT_MODE": "table"})):
    os.environ.update(env)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=[hy.par[i] for i in range(6)], Gconst=G), st, n, pars=pars, high_accuracy=True)
    for k in env: del os.environ[k]
    ta.propagate_until(4.0)
    ta.propagate_until(24.0)
    oc, mn, mx, ns = ta.propagate_res_arrays()
    ms = ta.kernel_ms_history(1)[0]
    res[tag] = {"mode": ta.hip_source_mode[:70], "kernel_ms": ms, "system_steps": float(ns.sum()), "rate": float(ns.sum()) / (ms * 1e-3)}
res["speedup"] = res["cluster"]["rate"] / res["table"]["rate"]
print(json.dumps(res, indent=1))
