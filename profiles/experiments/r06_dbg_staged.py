import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import heyoka_amd as hy, heyoka_oracle as ho
from heyoka_amd import configs
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
if len(sys.argv) > 2: os.environ["HEYOKA_AMD_TABLE_LDS"] = sys.argv[2]
n = int(sys.argv[1])
st = configs.outer_ss_state(n, perturb=1e-6, seed=5)
ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
print(ta.hip_source_mode[:120], flush=True)
t0 = time.perf_counter(); ta.step(); ta.synchronize(); print("step: %.3f s" % (time.perf_counter() - t0), "kernel ms", ta.kernel_ms_history(1), flush=True)
t0 = time.perf_counter(); ta.step(); ta.synchronize(); print("step: %.3f s" % (time.perf_counter() - t0), "kernel ms", ta.kernel_ms_history(1), flush=True)
t0 = time.perf_counter(); ta.propagate_until(3.0); ta.synchronize(); print("propagate: %.3f s" % (time.perf_counter() - t0), "kernel ms", ta.kernel_ms_history(3), flush=True)
print(ta.propagate_res[0], flush=True)
