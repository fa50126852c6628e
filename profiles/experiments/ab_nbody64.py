#!/usr/bin/env python
"""Interleaved A/B comparison of block-mode variants on model::nbody(n) (BASELINE config 5 by default): the variants are
environment settings applied while the integrator is constructed; a small parity check against the oracle first.
usage: ab_nbody64.py 'K1=V1,K2=V2' 'K1=V1b' [--systems 65536] [--dt 0.03] [--rounds 3] [--bodies 64]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import heyoka_amd as hy
from heyoka_amd import configs, codegen_check

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--systems", type=int, default=65536)
ap.add_argument("--dt", type=float, default=0.03)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--bodies", type=int, default=64)
ap.add_argument("--no-parity", action="store_true")
args = ap.parse_args()
nb = args.bodies
sys_ = hy.model.nbody(nb)


def with_env(v, f):
    kv = dict(x.split("=", 1) for x in v.split(",") if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        return f()
    finally:
        for k, o in old.items():
            if o is None:
                del os.environ[k]
            else:
                os.environ[k] = o


if not args.no_parity:
    import heyoka_oracle as ho

    n0 = 8
    st0 = configs.plummer_nbody_state(nb, n0, seed=77, jitter=1e-6)
    ora = ho.OracleIntegrator(ho.nbody(nb), st0, n0)
    ora.step()
    ora.step()
    ref = ora.state.reshape(6 * nb, n0)
    for v in args.variants:
        ta = with_env(v, lambda: hy.taylor_adaptive_batch(sys_, st0, n0))
        ta.step()
        h1 = np.array([h for _, h in ta.step_res])
        ta.step()
        err = np.max(np.abs(ta.state - ref) / np.maximum(1.0, np.abs(ref)))
        print(json.dumps({"parity": v, "mode": ta.hip_source_mode[-120:], "max_rel_err_eps": err / np.finfo(float).eps,
                          "h1": h1[:2].tolist(), "h1_oracle": [ora.step_res[0][1]]}), flush=True)

n = args.systems
st = configs.plummer_nbody_state(nb, n, seed=1234 + 42)
tas = []
for v in args.variants:
    t0 = time.time()
    ta = with_env(v, lambda: hy.taylor_adaptive_batch(sys_, st, n))
    print(json.dumps({"variant": v, "build_s": time.time() - t0, "res": codegen_check.kernel_resources(ta.code_object)}), flush=True)
    tas.append(ta)
rates = [[] for _ in tas]
t = 0.0
for r in range(args.rounds + 1):
    t += args.dt
    for i, ta in enumerate(tas):
        ta.propagate_until(t)
        ns = ta.propagate_res_arrays()[3]
        ms = list(ta.kernel_ms_history(1))[-1]
        if r > 0:
            rates[i].append(float(ns.sum()) / (ms * 1e-3))
for v, rr in zip(args.variants, rates):
    print(json.dumps({"variant": v, "rates": ["%.4g" % x for x in rr], "mean": "%.4g" % np.mean(rr)}), flush=True)
