"""Round 6: A/B of the fusion of the staged table stepper (followers / state-variable recursion computed by the lanes of the
variables they depend on) on four decompositions: HEYOKA_AMD_TABLE_LDS = 2 (both), 3 (none), 4 (followers), 5 (recursion)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import heyoka_amd as hy
from heyoka_amd import configs, mixed_models as mm
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
N = 262144
cases = {
    "oss": (lambda: hy.model.nbody(6, masses=M, Gconst=G), configs.outer_ss_state(N, perturb=1e-12, seed=42), 5.0, dict(high_accuracy=True)),
    "np1body6": (lambda: hy.model.np1body(6, masses=M, Gconst=G), np.ascontiguousarray(configs.outer_ss_state(N, perturb=1e-6, seed=11, com_shift=False)[6:]), 5.0, {}),
    "sine_lattice16": (lambda: mm.sine_lattice(hy, 16), mm.sine_lattice_state(16, N, seed=42), 1.0, {}),
    "nbody6_j2": (lambda: mm.nbody_j2(hy, 6, M, G, 1e-7), configs.outer_ss_state(N, perturb=1e-12, seed=42), 5.0, {}),
}
hy.set_logger_level("err")
for name, (mk, st, T, kw) in cases.items():
    for sw in (os.environ.get("AB_SWITCHES", "2,3,4,5,2,3").split(",")):
        os.environ["HEYOKA_AMD_TABLE_LDS"] = sw
        os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
        os.environ["HEYOKA_AMD_MULTI_CLASS"] = "0"
        ta = hy.taylor_adaptive_batch(mk(), st, N, **kw)
        t = 0.0
        ta.propagate_until(T)
        t = T
        ms, steps = [], 0
        for _ in range(2):
            t += T
            ta.propagate_until(t)
            ta.synchronize()
            steps = int(np.sum(ta.propagate_res_arrays()[3]))
            ms.append(ta.kernel_ms_history(1)[-1])
        print("%-16s switch %s: %.3e system-steps/s (kernel %.2f ms)  %s" % (name, sw, steps / (np.mean(ms) * 1e-3), np.mean(ms),
              ta.hip_source_mode.split("tape in LDS")[0][-90:]), flush=True)
        del ta
