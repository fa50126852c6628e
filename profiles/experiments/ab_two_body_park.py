#!/usr/bin/env python
"""Interleaved A/B of the round-6 items of the straight-line stepper on the two-body workload (BASELINE.json configs[2]) in one
process: literal zeros and scaled copies folded / Horner steps over leading zeros collapsed (HEYOKA_AMD_UNROLLED_TRIM), one
accumulator for the sum of squares (HEYOKA_AMD_UNROLLED_MERGE_SSQ), velocity histories re-derived + one-pass evaluation
(HEYOKA_AMD_UNROLLED_DERIVE), wavefronts per SIMD (HEYOKA_AMD_UNROLLED_WAVES). The switches are read when an integrator is built.
usage: ab_two_body_park.py [--systems N] [--rounds R]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs, codegen_check

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=4194304)
ap.add_argument("--dt", type=float, default=50.0)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--variant", action="append", default=[], help="K=V,K2=V2 (environment of one more variant; repeatable)")
ap.add_argument("--only-extra", action="store_true")
args = ap.parse_args()
n = args.systems
sys_ = hy.model.nbody(2, masses=[1.0, 0.0])
st = configs.two_body_state(n, perturb=1e-12, seed=42)
OFF = {"HEYOKA_AMD_UNROLLED_TRIM": "0", "HEYOKA_AMD_UNROLLED_MERGE_SSQ": "0", "HEYOKA_AMD_UNROLLED_DERIVE": "0"}
# (HEYOKA_AMD_UNROLLED_PARK / _LICM: switches of the first runs of profiles/r06_two_body_unrolled_ab.log, removed with the code they switched.)
variants = [
    ("round 5 kernel (all three off)", dict(OFF)),
    ("+ zeros and scaled copies folded, trimmed Horner", {**OFF, "HEYOKA_AMD_UNROLLED_TRIM": "1"}),
    ("+ one accumulator for the sum of squares", {**OFF, "HEYOKA_AMD_UNROLLED_TRIM": "1", "HEYOKA_AMD_UNROLLED_MERGE_SSQ": "1"}),
    ("+ velocity histories re-derived, one wavefront per SIMD", {"HEYOKA_AMD_UNROLLED_WAVES": "1"}),
    ("default: + two wavefronts per SIMD", {}),
]
if args.only_extra:
    variants = variants[:1]
for v in args.variant:
    variants.append((v, dict(x.split("=", 1) for x in v.split(",") if x)))
tas = []
for name, env in variants:
    keys = set(OFF) | set(env)
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=False)
    tas.append(ta)
    for k, o in old.items():
        os.environ.pop(k, None)
        if o is not None:
            os.environ[k] = o
    print(json.dumps({"variant": name, "res": codegen_check.kernel_resources(ta.code_object)}), flush=True)
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
ref = np.array(tas[0].state)
for (v, _), rr, ta in zip(variants, rates, tas):
    d = np.abs(np.array(ta.state) - ref)
    sc = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-300)
    print(json.dumps({"variant": v, "rates": ["%.4g" % x for x in rr], "mean": "%.4g" % np.mean(rr),
                      "max_state_diff_vs_first_eps": float((d / sc).max() / 2.220446049250313e-16)}), flush=True)
