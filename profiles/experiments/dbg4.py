import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs
n = 8
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
masses = [1.0] + [1e-3 * (i + 1) for i in range(nb - 1)]
st = configs.plummer_nbody_state(nb, n, seed=5)
os.environ["HEYOKA_AMD_PAIR_SPLIT_MAX_LANES"] = "64"
a = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=masses), st, n)
del os.environ["HEYOKA_AMD_PAIR_SPLIT_MAX_LANES"]
b = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=masses), st, n)
print(a.hip_source_mode[:110]); print(b.hip_source_mode[:110], flush=True)
a.step(); b.step()
print("state diff", np.max(np.abs(a.state - b.state)), flush=True)
