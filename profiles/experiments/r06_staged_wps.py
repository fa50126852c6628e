"""Round 6: wavefronts per system of the staged table stepper (HEYOKA_AMD_STAGED_WPS = 1 / 2 / 4) on four decompositions."""
import os, sys
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
os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
os.environ["HEYOKA_AMD_MULTI_CLASS"] = "0"
for name, (mk, st, T, kw) in cases.items():
    for w in ("1", "2", "4", "1", "2", "4"):
        os.environ["HEYOKA_AMD_STAGED_WPS"] = w
        ta = hy.taylor_adaptive_batch(mk(), st, N, **kw)
        ta.propagate_until(T)
        t, ms, steps = T, [], 0
        for _ in range(2):
            t += T
            ta.propagate_until(t)
            ta.synchronize()
            steps = int(np.sum(ta.propagate_res_arrays()[3]))
            ms.append(ta.kernel_ms_history(1)[-1])
        print("%-16s wps %s: %.3e system-steps/s (kernel %.2f ms)  %s" % (name, w, steps / (np.mean(ms) * 1e-3), np.mean(ms),
              ta.hip_source_mode.split("tape in LDS")[1][:40]), flush=True)
        del ta
