#!/usr/bin/env python
"""Interleaved A/B comparison of two stepper variants in one process (same box, same thermal state): the variants are
environment settings applied while the integrator is constructed. usage: ab.py 'K1=V1,K2=V2' 'K1=V1b' [--dt 40]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--systems", type=int, default=1048576)
ap.add_argument("--dt", type=float, default=40.0)
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
n = args.systems
st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
tas = []
for v in args.variants:
    kv = dict(x.split("=", 1) for x in v.split(",") if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    tas.append(hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True))
    for k, o in old.items():
        if o is None:
            del os.environ[k]
        else:
            os.environ[k] = o
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
    print(json.dumps({"variant": v, "rates": ["%.4g" % x for x in rr], "mean": "%.4g" % np.mean(rr)}))
