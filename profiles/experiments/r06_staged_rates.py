"""Round 6: rates of the table steppers (staged: tape in LDS / one lane per system: tape in HBM) next to the default
generator, with a parity check of every leg against the oracle on a sample of lanes.
  python profiles/experiments/r06_staged_rates.py [n_systems]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import heyoka_amd as hy
import heyoka_oracle as ho
from heyoka_amd import configs

EPS = 2.220446049250313e-16
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def leg(name, sys_g, sys_o, st, T, env, reps=3, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ta = hy.taylor_adaptive_batch(sys_g, st, st.shape[1], **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    mode = ta.hip_source_mode
    t = 0.0
    ta.propagate_until(T)  # warm-up
    t += T
    tot, el = 0, 0.0
    for _ in range(reps):
        t += T
        t0 = time.perf_counter()
        ta.propagate_until(t)
        ta.synchronize()
        el += time.perf_counter() - t0
        tot += int(np.sum(ta.propagate_res_arrays()[3]))
    km = list(ta.kernel_ms_history(reps))
    # Parity of a sample of lanes: fresh integrator on 64 systems, one launch against the oracle.
    m = 64
    os.environ.update(env)
    try:
        tb = hy.taylor_adaptive_batch(sys_g, st[:, :m].copy(), m, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    kwo = {k: v for k, v in kw.items() if k in ("high_accuracy",)}
    oi = ho.OracleIntegrator(sys_o, st[:, :m].copy(), m, **kwo)
    tb.propagate_until(T)
    oi.propagate_until(T)
    ref = oi.state.reshape(st.shape[0], m)
    rows = np.max(np.abs(np.asarray(tb.state) - ref), axis=1) / (np.max(np.abs(ref), axis=1) + 1e-300)
    dn = max(abs(int(a[3]) - int(b[3])) for a, b in zip(tb.propagate_res, oi.prop_res))
    print("%-28s %.3e system-steps/s  kernel %.2f ms/launch  steps/launch %.3g | parity (64 lanes, row-scaled): %.3g eps, "
          "step counts within %d | %s" % (name, tot / el, np.mean(km), tot / reps, np.max(rows) / EPS, dn, mode[:150]), flush=True)
    return tot / el


oss_g, oss_o = hy.model.nbody(6, masses=M, Gconst=G), ho.nbody(6, masses=M, Gconst=G)
st = configs.outer_ss_state(N, perturb=1e-12, seed=42)
leg("oss default (v5)", oss_g, oss_o, st, 20.0, {}, high_accuracy=True)
leg("oss compact_mode=True", oss_g, oss_o, st, 20.0, {}, high_accuracy=True, compact_mode=True)
leg("oss table staged", oss_g, oss_o, st, 5.0, {"HEYOKA_AMD_EMIT_MODE": "table"}, high_accuracy=True)
leg("oss table HBM tape", oss_g, oss_o, st, 5.0, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "0"}, reps=1, high_accuracy=True)
np1_g, np1_o = hy.model.np1body(6, masses=M, Gconst=G), ho.np1body(6, masses=M, Gconst=G)
st1 = configs.outer_ss_state(N, perturb=1e-6, seed=11, com_shift=False)[6:]
st1 = np.ascontiguousarray(st1 - 0.0)
leg("np1body6 default", np1_g, np1_o, st1, 20.0, {})
leg("np1body6 table staged", np1_g, np1_o, st1, 5.0, {"HEYOKA_AMD_EMIT_MODE": "table"})
leg("np1body6 table HBM tape", np1_g, np1_o, st1, 5.0, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "0"}, reps=1)

# Mixed models: the multi-class wave-cluster stepper against the two table steppers on the same system.
from heyoka_amd import mixed_models as mm

sl_g, sl_o = mm.sine_lattice(hy, 16), mm.sine_lattice(ho, 16)
st2 = mm.sine_lattice_state(16, N, seed=42)
leg("sine_lattice16 multi-class", sl_g, sl_o, st2, 2.0, {})
leg("sine_lattice16 table staged", sl_g, sl_o, st2, 1.0, {"HEYOKA_AMD_MULTI_CLASS": "0"})
leg("sine_lattice16 table HBM", sl_g, sl_o, st2, 1.0, {"HEYOKA_AMD_MULTI_CLASS": "0", "HEYOKA_AMD_TABLE_LDS": "0"}, reps=1)
cen, ch = mm.lattice_centres_setup()
lc_g, lc_o = mm.lattice_centres(hy, cen, ch), mm.lattice_centres(ho, cen, ch)
st3 = mm.lattice_centres_state(N, seed=42)
leg("lattice_centres12 multi-class", lc_g, lc_o, st3, 2.0, {})
leg("lattice_centres12 table staged", lc_g, lc_o, st3, 1.0, {"HEYOKA_AMD_MULTI_CLASS": "0"})
leg("lattice_centres12 table HBM", lc_g, lc_o, st3, 1.0, {"HEYOKA_AMD_MULTI_CLASS": "0", "HEYOKA_AMD_TABLE_LDS": "0"}, reps=1)
j2_g, j2_o = mm.nbody_j2(hy, 6, M, G, 1e-7), mm.nbody_j2(ho, 6, M, G, 1e-7)
leg("nbody6_j2 table staged", j2_g, j2_o, st, 5.0, {})
leg("nbody6_j2 table HBM", j2_g, j2_o, st, 5.0, {"HEYOKA_AMD_TABLE_LDS": "0"}, reps=1)
