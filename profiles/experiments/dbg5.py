import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import heyoka_amd as hy
import heyoka_oracle as ho
EPS = 2.0 ** -52
masses = [1.0, 1e-3, 0.0]
nb = len(masses); n = 24
rng = np.random.default_rng(5)
st = np.zeros((6 * nb, n))
for b in range(nb):
    r = 1.0 + 1.7 * b
    ph = rng.uniform(0, 2 * np.pi, n)
    vc = np.sqrt(1.0 / r) if b > 0 else 0.0
    st[6 * b + 0] = r * np.cos(ph) if b > 0 else 0.0
    st[6 * b + 1] = r * np.sin(ph) if b > 0 else 0.0
    st[6 * b + 2] = 0.01 * rng.standard_normal(n)
    st[6 * b + 3] = -vc * np.sin(ph)
    st[6 * b + 4] = vc * np.cos(ph)
    st[6 * b + 5] = 0.001 * rng.standard_normal(n)
sys_o = ho.nbody(nb, masses=masses, Gconst=1.0)
oi = ho.OracleIntegrator(sys_o, st, n, high_accuracy=True)
oi.step(wtc=True)
tc_o = oi.tc.reshape(6 * nb, oi.order + 1, n)
scale = np.max(np.abs(tc_o), axis=0, keepdims=True)
for mode in ("cluster", "unrolled"):
    os.environ["HEYOKA_AMD_EMIT_MODE"] = mode
    ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=masses, Gconst=1.0), st, n, high_accuracy=True)
    ta.step(write_tc=True)
    err = np.abs(np.asarray(ta.tc).reshape(6 * nb, oi.order + 1, n) - tc_o) / scale
    i = np.unravel_index(np.argmax(err), err.shape)
    print(mode, ta.hip_source_mode[:60], "max err / eps", err.max() / EPS, "at", i, "tc", tc_o[i], "scale", scale[0, i[1], i[2]])
    print("   per-order max err/eps:", ["%.1e" % (err[:, k, :].max() / EPS) for k in range(0, 21, 4)])
    d12 = np.sqrt((st[6] - st[12]) ** 2 + (st[7] - st[13]) ** 2)
    print("   lane dist 1-2:", "%.3f" % d12[i[2]], "min over lanes %.3f" % d12.min())
