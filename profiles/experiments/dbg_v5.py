#!/usr/bin/env python
"""One-lane-per-pair kernel (v5) against the oracle: steps with Taylor coefficients, propagation."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["HEYOKA_AMD_ONE_LANE"] = "1"
import heyoka_amd as hy
import heyoka_oracle as ho
from heyoka_amd import configs
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
st = configs.outer_ss_state(n, perturb=1e-6, seed=3)
ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
print("mode:", ta.hip_source_mode, flush=True)
ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st.reshape(-1), n, high_accuracy=True)
eps = np.finfo(float).eps
def cmp(tag):
    ref = ora.state.reshape(36, n)
    err = np.max(np.abs(ta.state - ref) / np.maximum(1.0, np.abs(ref)))
    print(tag, "state err %.3g eps" % (err / eps), flush=True)
for i in range(3):
    ta.step(write_tc=True); ora.step(wtc=True)
    hg = np.array([h for _, h in ta.step_res]); ho_ = np.array([h for _, h in ora.step_res])
    print("step", i, "h err %.3g eps" % (np.max(np.abs(hg - ho_) / ho_) / eps), "oc", set(int(o) for o, _ in ta.step_res), flush=True)
    cmp("  step")
    tc_o = ora.tc.reshape(36, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    print("   tc err %.3g eps" % (np.max(np.abs(ta.tc - tc_o) / scale) / eps), flush=True)
ta.propagate_until(30.0); ora.propagate_until(30.0)
print("prop:", [(int(r[0]), r[3]) for r in ta.propagate_res][:3], [(r[0], r[3]) for r in ora.prop_res][:3], flush=True)
print("steps equal:", [r[3] for r in ta.propagate_res] == [r[3] for r in ora.prop_res])
cmp("propagate")
print("DONE", flush=True)
