#!/usr/bin/env python
"""Rates of the models beside the BASELINE.json configurations on the current kernels: system-steps / kernel time of the
second of two propagate_until() launches. usage: model_rates.py [--systems N]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs, codegen_check

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=262144)
ap.add_argument("--only", default="", help="substring of the case names to run")
args = ap.parse_args()
n = args.systems
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
rs = np.random.RandomState(7)


def np1body_case():
    full = configs.outer_ss_state(n, perturb=1e-6, seed=3).reshape(6, 6, n)
    return hy.model.np1body(6, masses=M, Gconst=G), (full[1:] - full[:1]).reshape(30, n), None, 30.0


def cr3bp_case():
    st = np.array([[-0.45], [0.80], [0.0], [-0.80], [-0.45], [0.58]]) + 1e-3 * rs.uniform(-1, 1, (6, n))
    return hy.model.cr3bp(mu=0.01), st, None, 20.0


def centres_case(mascon):
    m = list(rs.uniform(0.5, 1.5, 100) / 100.0)
    pos = list(rs.uniform(-1.0, 1.0, 300))
    s = hy.model.mascon(masses=m, positions=pos, Gconst=1.0, omega=[0.0, 0.0, 0.3]) if mascon else hy.model.fixed_centres(masses=m, positions=pos)
    st = np.concatenate([rs.uniform(1.5, 2.0, (3, n)), rs.uniform(-0.3, 0.3, (3, n))])
    return s, st, None, 1.0


def par_masses_case():
    s = hy.model.nbody(6, masses=[hy.par[i] for i in range(6)], Gconst=G)
    return s, configs.outer_ss_state(n, perturb=1e-6, seed=3), np.tile(np.array(M)[:, None], (1, n)), 30.0


def pendulum_case():
    return hy.model.pendulum(), np.stack([rs.uniform(-1.5, 1.5, n), rs.uniform(-0.5, 0.5, n)]), None, 50.0


def small_nbody_case(nb, kernel):
    m = [1.0, 1e-3, 3e-4, 2e-4][:nb]
    pos = rs.uniform(-0.3, 0.3, (nb, 3, n)) + 5.0 * np.arange(nb)[:, None, None] * np.array([1.0, 0.3, -0.2])[None, :, None]
    vel = rs.uniform(-0.05, 0.05, (nb, 3, n)) + (0.4 / np.sqrt(1.0 + 5.0 * np.arange(nb)))[:, None, None] * np.array([-0.3, 1.0, 0.1])[None, :, None]
    st = np.concatenate([np.concatenate([pos[b], vel[b]], axis=0) for b in range(nb)], axis=0)
    return hy.model.nbody(nb, masses=m), st, None, 20.0, dict(high_accuracy=True, cluster_kernel=kernel)


def nbody_plummer_case(nb, default_masses):
    st = configs.plummer_nbody_state(nb, n, seed=5)
    kw = {} if default_masses else {"masses": list(1.0 / (1.0 + np.arange(nb)) ** 2 * nb / 4.0)}
    return hy.model.nbody(nb, **kw), st, None, 0.05


def two_massive_case():
    st = configs.two_body_state(n, perturb=1e-3, seed=3)
    return hy.model.nbody(2, masses=[1.0, 0.5]), st, None, 20.0


cases = {"nbody(12), default masses": lambda: nbody_plummer_case(12, True), "nbody(12), distinct masses": lambda: nbody_plummer_case(12, False),
         "nbody(16), default masses": lambda: nbody_plummer_case(16, True), "nbody(16), distinct masses": lambda: nbody_plummer_case(16, False),
         "nbody(2), both massive": two_massive_case, "nbody(3) on v5": lambda: small_nbody_case(3, "v5"), "nbody(3) on v3": lambda: small_nbody_case(3, "v3"),
         "nbody(4) on v5": lambda: small_nbody_case(4, "v5"), "nbody(4) on v3": lambda: small_nbody_case(4, "v3"),
         "np1body(6)": np1body_case, "cr3bp": cr3bp_case, "fixed_centres(100)": lambda: centres_case(False),
         "mascon(100)": lambda: centres_case(True), "nbody(6), par[] masses": par_masses_case, "pendulum": pendulum_case}
for name, mk in cases.items():
    if args.only and args.only not in name:
        continue
    try:
        s, st, pars, dt, *extra = mk()
        kw = {} if pars is None else {"pars": pars}
        kw.update(extra[0] if extra else {})
        ta = hy.taylor_adaptive_batch(s, st, n, **kw)
        rates = []
        for r in range(3):
            ta.propagate_until(dt * (r + 1))
            ns = ta.propagate_res_arrays()[3]
            ms = list(ta.kernel_ms_history(1))[-1]
            rates.append(float(ns.sum()) / (ms * 1e-3))
        res = codegen_check.kernel_resources(ta.code_object)
        print("%-24s %s system-steps/s (launches: %s) steps/system %.1f | vgpr %s spills %s waves %s | %s" % (
            name, "%.3g" % rates[-1], ", ".join("%.3g" % x for x in rates), float(ns.mean()), res["vgpr_total"], res["vgpr_spill"],
            res["waves_per_simd_by_registers"], ta.hip_source_mode[:110]), flush=True)
    except Exception as e:
        print(name, "ERROR", type(e).__name__, e, flush=True)
