#!/usr/bin/env python
"""Run-to-run determinism of the staged table stepper, function by function: tiny systems, repeated single steps on fresh
integrators, Taylor coefficients compared bit for bit with the first run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import heyoka_amd as hy
os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 33
x, y, z = hy.make_vars("x", "y", "z")
systems = {
    "erf": [(x, hy.erf(y) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "erf only": [(x, hy.erf(x) - 0.2 * x), (y, x), (z, 0.3 - y)],
    "exp": [(x, hy.exp(-(y * y)) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "tanh": [(x, hy.tanh(y) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "sigmoid": [(x, hy.sigmoid(y) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "atan2": [(x, hy.atan2(0.7, y) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "kepE": [(x, hy.sin(hy.kepE(0.6 * hy.sigmoid(y), z)) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "sin": [(x, hy.sin(y) - 0.2 * x), (y, x * z), (z, 0.3 - y)],
    "prod": [(x, y * z - 0.2 * x), (y, x * z), (z, 0.3 - y * x)],
    "erf + pow": [(x, hy.erf(y) - 0.2 * x + hy.pow(y, 2.0)), (y, x * z), (z, 0.3 - y)],
}
rs = np.random.RandomState(3)
st = rs.uniform(-0.7, 0.7, (3, n))
for name, s in systems.items():
    ref, nbad, first = None, 0, None
    for r in range(reps):
        ta = hy.taylor_adaptive_batch(s, st, n)
        ta.step(write_tc=True)
        tc = np.asarray(ta.tc).reshape(3, 21, n).copy()
        if ref is None:
            ref = tc
            mode = ta.hip_source_mode[:150]
            continue
        bad = np.argwhere(tc != ref)
        if len(bad):
            nbad += 1
            if first is None:
                o = int(bad[:, 1].min())
                first = "first order %d vars %s" % (o, sorted(set(int(b[0]) for b in bad if b[1] == o)))
    print("%-10s nondeterministic runs %d of %d %s | %s" % (name, nbad, reps - 1, first or "", mode), flush=True)
