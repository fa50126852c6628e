#!/usr/bin/env python
"""Interleaved A/B comparison of the arithmetic flavours of the unrolled generator on the two-body workload
(BASELINE.json configs[2]) in one process: sum_order / exact_division are constructor arguments.
usage: ab_two_body.py [--systems N] [--rounds R]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=4194304)
ap.add_argument("--dt", type=float, default=50.0)
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
n = args.systems
sys_ = hy.model.nbody(2, masses=[1.0, 0.0])
st = configs.two_body_state(n, perturb=1e-12, seed=42)
variants = [
    ("round 2 arithmetic: pairwise sums, exact division by the order", dict(sum_order="pairwise", exact_division=True)),
    ("running sums, exact division by the order", dict(sum_order="running", exact_division=True)),
    ("running sums, reciprocal division (default)", dict()),
]
tas = [hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=False, **kw) for _, kw in variants]
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
for (v, _), rr in zip(variants, rates):
    print(json.dumps({"variant": v, "rates": ["%.4g" % x for x in rr], "mean": "%.4g" % np.mean(rr)}))
