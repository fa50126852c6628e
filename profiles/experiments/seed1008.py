#!/usr/bin/env python
"""Repeats the single-step comparison of tests/test_gpu_parity.py::test_random_systems_all_code_paths_vs_oracle for one seed,
per code path, several times (fresh integrator each time): run-to-run differences = a race, not rounding."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import numpy as np
import heyoka_amd as hy
from test_gpu_parity import _random_system, ho
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1008
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 33
rs = np.random.RandomState(100 + seed)
st = rs.uniform(-0.7, 0.7, (3, n)); pars = rs.uniform(-0.5, 0.5, (2, n)); t0 = rs.uniform(0.0, 2.0, n)
ext = seed >= 1000
sys_o = _random_system(ho, np.random.RandomState(seed), extended=ext)
ora = ho.OracleIntegrator(sys_o, st, n, pars=pars, time=t0)
ora.step(wtc=True)
h_o = np.array([h for _, h in ora.step_res])
tc_o = ora.tc.reshape(3, ora.order + 1, n)
scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
for mode in ("default", "table", "table_hbm"):
    os.environ.pop("HEYOKA_AMD_EMIT_MODE", None); os.environ.pop("HEYOKA_AMD_TABLE_LDS", None)
    if mode != "default":
        os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
    if mode == "table_hbm":
        os.environ["HEYOKA_AMD_TABLE_LDS"] = "0"
    for r in range(reps):
        sys_p = _random_system(hy, np.random.RandomState(seed), extended=ext)
        ta = hy.taylor_adaptive_batch(sys_p, st, n, pars=pars, time=t0)
        ta.step(write_tc=True)
        h_g = np.array([h for _, h in ta.step_res])
        eh = np.abs(h_g - h_o) / np.abs(h_o) / 2.220446049250313e-16
        etc = np.abs(np.asarray(ta.tc).reshape(3, 21, n) - tc_o) / scale / 2.220446049250313e-16
        k = np.unravel_index(np.argmax(etc), etc.shape)
        print(mode, r, "h err eps max %.3g (lane %d)" % (eh.max(), eh.argmax()), "tc err eps max %.3g at (var, order, lane) %s" % (etc.max(), k),
              "| mode:", ta.hip_source_mode[:90] if r == 0 else "", flush=True)
