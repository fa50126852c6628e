"""The oracle (oracle/) against the reference's own published known answers (tests/golden/).

This pins the oracle: the reference cannot be built here, so its tutorial console
outputs and test literals are the golden vectors."""
import numpy as np
import pytest

import heyoka_oracle as ho
from conftest import sig_close, EPS


def pendulum():
    x, v = ho.var("x"), ho.var("v")
    return [(x, v), (v, -9.8 * ho.sin(x))]


def test_pendulum_first_step_17_digits(golden):
    g = golden["pendulum_scalar"]
    ta = ho.OracleIntegrator(pendulum(), g["ic"], 1)
    assert ta.order == g["order"]
    (oc, h), = ta.step()
    assert oc == ho.OC_SUCCESS
    # Printed with 17 significant digits by the reference: must round-trip to a few ulps
    # (sin/cos/pow of the host libm are the only unpinned ingredients).
    assert abs(h - g["step1"]["h"]) <= 4 * EPS * abs(h)
    assert abs(ta.time_hi[0] - g["step1"]["time"]) <= 4 * EPS
    assert np.all(np.abs(ta.state - np.array(g["step1"]["state"])) <= 8 * EPS * np.abs(ta.state))
    (oc, h), = ta.step(backward=True)
    assert sig_close(h, g["step_backward_h_6digits"])


def test_pendulum_propagate_sequence(golden):
    g = golden["pendulum_scalar"]
    ta = ho.OracleIntegrator(pendulum(), g["ic"], 1)
    for key, call in (("propagate_for_5", lambda: ta.propagate_for(5.0)),
                      ("then_propagate_until_20", lambda: ta.propagate_until(20.0)),
                      ("then_propagate_until_0", lambda: ta.propagate_until(0.0))):
        (oc, mn, mx, ns), = call()
        assert oc == ho.OC_TIME_LIMIT
        assert ns == g[key]["steps"]
        assert sig_close(mn, g[key]["min_h"]) and sig_close(mx, g[key]["max_h"])
        assert ta.time_hi[0] == g[key]["time"]


def test_readme_pendulum(golden):
    g = golden["readme_pendulum"]
    ta = ho.OracleIntegrator(pendulum(), g["ic"], 1)
    ta.propagate_for(10.0)
    assert sig_close(ta.state[0], g["propagate_for_10"]["x"])
    assert sig_close(ta.state[1], g["propagate_for_10"]["v"])


def test_ensemble_member_9(golden):
    g = golden["ensemble_member_9"]
    ta = ho.OracleIntegrator(pendulum(), g["ic"], 1)
    (oc, mn, mx, ns), = ta.propagate_until(20.0)
    e = g["propagate_until_20"]
    assert oc == ho.OC_TIME_LIMIT and ns == e["steps"]
    assert sig_close(mn, e["min_h"]) and sig_close(mx, e["max_h"])
    # 124 steps of accumulated libm-level differences: the printed 17 digits agree to ~1e-13.
    assert np.all(np.abs(ta.state - np.array(e["state"])) <= 1e-12)


def forced(batch=4):
    x, v = ho.var("x"), ho.var("v")
    return [(x, v), (v, ho.cos(ho.TIME) - ho.par(0) * v - ho.sin(x))]


def test_batch_mode_tutorial(golden):
    g = golden["batch_mode_forced_pendulum"]
    ta = ho.OracleIntegrator(forced(), [g["x0"], g["v0"]], 4, pars=g["alpha"])
    res = ta.step()
    assert all(oc == ho.OC_SUCCESS for oc, _ in res)
    assert sig_close([h for _, h in res], g["step1"]["h"])
    st = ta.state.reshape(2, 4)
    assert sig_close(st[0], g["step1"]["x"]) and sig_close(st[1], g["step1"]["v"], 7)

    res = ta.step(g["step_clamped"]["max_delta_t"])
    assert all(oc == ho.OC_TIME_LIMIT for oc, _ in res)
    assert [h for _, h in res] == g["step_clamped"]["max_delta_t"]
    st = ta.state.reshape(2, 4)
    assert sig_close(st[0], g["step_clamped"]["x"]) and sig_close(st[1], g["step_clamped"]["v"])
    assert sig_close(ta.time_hi, g["step_clamped"]["time"])

    pf = g["propagate_for"]
    res = ta.propagate_for(pf["dt"])
    assert [r[3] for r in res] == pf["steps"]
    assert all(r[0] == ho.OC_TIME_LIMIT for r in res)
    assert sig_close([r[1] for r in res], pf["min_h"]) and sig_close([r[2] for r in res], pf["max_h"])
    st = ta.state.reshape(2, 4)
    assert sig_close(st[0], pf["x"]) and sig_close(st[1], pf["v"])
    assert sig_close(ta.time_hi, pf["time"], 7)

    pu = g["propagate_until"]
    res = ta.propagate_until(pu["t"])
    assert [r[3] for r in res] == pu["steps"]
    assert sig_close([r[1] for r in res], pu["min_h"]) and sig_close([r[2] for r in res], pu["max_h"])
    st = ta.state.reshape(2, 4)
    assert sig_close(st[0], pu["x"]) and sig_close(st[1], pu["v"], 5)
    assert list(ta.time_hi) == pu["t"]

    # Taylor coefficients of the next step (tc[var][order][lane]).
    ta.step(wtc=True)
    tc = ta.tc.reshape(2, ta.order + 1, 4)
    assert sig_close(tc[0], g["step_wtc_tc_x"], 6)
    assert sig_close(tc[1], g["step_wtc_tc_v"], 6)


def test_outer_ss_structure(golden):
    from collections import Counter

    g = golden["outer_ss_decomposition"]
    m = [1.00000597682, 1 / 1047.355, 1 / 3501.6, 1 / 22869.0, 1 / 19314.0, 7.4074074e-09]
    G = 0.01720209895 * 0.01720209895 * 365 * 365
    dc = ho.taylor_decompose_sys(ho.nbody(6, masses=m, Gconst=G))
    assert len(dc) == g["size"]
    c = Counter(e.kind for e, _ in dc[36:-36])
    assert c["sum_sq"] == g["sum_sq"] and c["sum"] == g["sum"] and c["sub"] == g["sub"]


def test_dfloat_matches_exact_arithmetic():
    from fractions import Fraction

    rng = np.random.RandomState(0)
    for _ in range(200):
        a = rng.uniform(-1e6, 1e6)
        b = rng.uniform(-1, 1) * 10.0 ** rng.randint(-20, 3)
        al = a * EPS * rng.uniform(-0.4, 0.4)
        hi, lo = ho.dfloat_add(a, al, b, 0.0)
        exact = Fraction(a) + Fraction(al) + Fraction(b)
        err = abs(Fraction(hi) + Fraction(lo) - exact)
        assert err <= abs(exact) * Fraction(EPS) ** 2 * 8 + Fraction(1, 10**320)


def test_harmonic_oscillator_long_time_analytic():
    """x' = v, v' = -x over 1e4 time units vs v0 * sin(t) (cf. test/dfloat_time.cpp:168-250)."""
    x, v = ho.var("x"), ho.var("v")
    ta = ho.OracleIntegrator([(x, v), (v, -x)], [[0.0, 0.0], [1.0, 1.0 + 1e-3]], 2)
    ta.propagate_until([1e4, 1.1e4])
    st = ta.state.reshape(2, 2)
    exact = np.array([np.sin(1e4), (1.0 + 1e-3) * np.sin(1.1e4)])
    assert np.all(np.abs(st[0] - exact) <= 1e-11)


# ---- event detection (doc/tut_events.rst) ----
def _pend():
    x, v = ho.var("x"), ho.var("v")
    return x, v, [(x, v), (v, -9.8 * ho.sin(x))]


def test_events_tutorial_nt_events(golden):
    g = golden["events_tutorial"]
    x, v, sys_ = _pend()
    times, xs = [], []

    def cb(ta, t, d_sgn, idx):
        times.append(t)

    ta = ho.OracleEventIntegrator(sys_, g["ic"], 1, nt_events=[ho.nt_event(v, cb)])
    ta.propagate_until(5.0)
    assert len(times) == len(g["event_times"])
    # "accurate to machine precision" (doc/tut_events.rst:166-170): a few ulps on times of order 1.
    assert np.max(np.abs(np.array(times) - np.array(g["event_times"]))) <= 8 * EPS
    times.clear()
    ta = ho.OracleEventIntegrator(sys_, g["ic"], 1, nt_events=[ho.nt_event(v, cb, direction=ho.DIR_POSITIVE)])
    ta.propagate_until(5.0)
    assert np.max(np.abs(np.array(times) - np.array(g["event_times_positive_direction"]))) <= 8 * EPS
    # Two close events: chronological processing within a step.
    log = []
    e0 = ho.nt_event(v, lambda ta, t, d, i: log.append((0, t)))
    e1 = ho.nt_event(v * v - 1e-12, lambda ta, t, d, i: log.append((1, t)))
    ta = ho.OracleEventIntegrator(sys_, g["ic"], 1, nt_events=[e0, e1])
    ta.propagate_until(5.0)
    seq = g["two_events"]["sequence"]
    assert [e for e, _ in log] == [e for e, _ in seq]
    # v*v - 1e-12 has two roots 4e-6 apart around each zero of v: conditioned to ~1e-11 in time.
    assert np.max(np.abs(np.array([t for _, t in log]) - np.array([t for _, t in seq]))) <= 2e-11


def test_events_tutorial_terminal_event(golden):
    g = golden["events_tutorial"]["terminal_drag_toggle"]
    x, v = ho.var("x"), ho.var("v")

    def toggle(ta, d_sgn, idx):
        ta.pars[0] = 1.0 if ta.pars[0] == 0 else 0.0
        return True

    ta = ho.OracleEventIntegrator([(x, v), (v, -9.8 * ho.sin(x) - ho.par(0) * v)], g["ic"], 1,
                                  t_events=[ho.t_event(v, toggle)], pars=[0.0])
    while True:
        (oc, h), = ta.step()
        if oc != ho.OC_SUCCESS:
            break
    assert oc == g["first_event_outcome"] and ta.pars[0] == 1.0
    assert abs(ta.state[1]) <= 1e-15  # stopped at v = 0
    ta.propagate_until(1.0)
    for tg, exp in zip(g["grid"], g["grid_states"]):
        ta.propagate_until(float(tg))
        assert ta.time_hi[0] == tg
        assert np.max(np.abs(ta.state - np.array(exp))) <= 1e-13


def stark_delaunay(m, eps):
    """Hamilton's equations of the Stark problem in Delaunay elements with the eccentric anomaly E = kepE(e, l)
    (the system of test/kepE.cpp:194-239, with the partial derivatives of the Hamiltonian written out by hand)."""
    if m is ho:
        L, G, H, l, g, h = (m.var(n) for n in ("L", "G", "H", "l", "g", "h"))
        powf, sqrt = m.pow_, m.sqrt
    else:
        L, G, H, l, g, h = m.make_vars("L", "G", "H", "l", "g", "h")
        powf, sqrt = m.pow, m.sqrt
    e = sqrt(1.0 - G * G / (L * L))
    E = m.kepE(e, l)
    sE, cE, sg, cg = m.sin(E), m.cos(E), m.sin(g), m.cos(g)
    A = sqrt(1.0 - H * H / (G * G))
    B = L * (cE - e) * sg + G * sE * cg
    den = 1.0 / (1.0 - e * cE)
    E_l, E_e = den, sE * den
    e_L, e_G = (G * G / (L * L * L)) / e, -(G / (L * L)) / e
    B_E = G * cE * cg - L * sE * sg
    A_H, A_G = -(H / (G * G)) / A, (H * H / (G * G * G)) / A
    dH_dl = -eps * L * A * B_E * E_l
    dH_dg = -eps * L * A * (L * (cE - e) * cg - G * sE * sg)
    dH_dL = powf(L, -3.0) - eps * A * B - eps * L * A * ((cE - e) * sg + B_E * E_e * e_L - L * sg * e_L)
    dH_dG = -eps * L * B * A_G - eps * L * A * (sE * cg + B_E * E_e * e_G - L * sg * e_G)
    dH_dH = -eps * L * B * A_H
    return [(L, -dH_dl), (G, -dH_dg), (H, 0.0 * L), (l, dH_dL), (g, dH_dG), (h, dH_dH)]


def test_kepE_stark_problem_known_answer(golden):
    """Known answer held by the reference for an integration through kepE (test/kepE.cpp:194-239): state after
    propagate_until(250) to 100 eps (the mean anomaly through sin / cos at 1e4 eps)."""
    g = golden["kepE_stark"]
    L0, G0, H0, E0, g0, h0 = g["init_state_LGH_E_gh"]
    l0 = E0 - np.sqrt(1 - G0 * G0 / (L0 * L0)) * np.sin(E0)
    st = np.array([L0, G0, H0, l0, g0, h0])[:, None]
    oi = ho.OracleIntegrator(stark_delaunay(ho, g["eps"]), st, 1)
    oi.propagate_until(g["t_final"])
    assert oi.prop_res[0][0] == ho.OC_TIME_LIMIT
    s = oi.state
    for got, exp in zip([s[0], s[1], s[2], s[4], s[5]], g["final_L_G_H_g_h"]):
        assert abs(got - exp) <= g["tol_eps"] * EPS * abs(exp)
    fL, fG, fE = s[0], s[1], g["final_E"]
    l_exp = fE - np.sqrt(1 - fG * fG / (fL * fL)) * np.sin(fE)
    assert abs(np.sin(s[3]) - np.sin(l_exp)) <= g["tol_eps_angle_l"] * EPS * abs(np.sin(l_exp))
    assert abs(np.cos(s[3]) - np.cos(l_exp)) <= g["tol_eps_angle_l"] * EPS * abs(np.cos(l_exp))


@pytest.mark.parametrize("which", ["outer_ss", "two_body"])
def test_compiled_cpu_baseline_is_bit_identical_to_the_interpreter(which):
    """The CPU baseline timed by bench.py (oracle/compiled_baseline.py: straight-line, 8-wide vectorised C generated from
    the oracle's decomposition, the stand-in for the reference's default-mode SIMD JIT, src/taylor_02.cpp:1339-1418) built
    strictly (-ffp-contract=off) reproduces the interpreter bit by bit: steps with Taylor coefficients and a propagation;
    the fast build (-O3, contraction on, like the reference's LLVM builder flags) stays within 1e3 eps of it."""
    import compiled_baseline as cb
    from heyoka_amd import configs

    if which == "outer_ss":
        sysd = ho.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
        st, ha, T = configs.outer_ss_state(8, perturb=1e-6, seed=1), True, 30.0
    else:
        sysd = ho.nbody(2, masses=[1.0, 0.0])
        st, ha, T = configs.two_body_state(8, perturb=1e-3, seed=2), False, 20.0
    a = ho.OracleIntegrator(sysd, st, 8, high_accuracy=ha)
    b = ho.OracleIntegrator(sysd, st, 8, high_accuracy=ha)
    c = ho.OracleIntegrator(sysd, st, 8, high_accuracy=ha)
    for _ in range(3):
        a.step(wtc=True)
    a.propagate_until(T)
    try:
        cb.install(b, fast=False)
        for _ in range(3):
            b.step(wtc=True)
        b.propagate_until(T)
        cb.install(c, fast=True)
        c.propagate_until(T)
    finally:
        cb.uninstall()
    assert np.array_equal(a.tc, b.tc) and np.array_equal(a.state, b.state)
    assert [r[3] for r in a.prop_res] == [r[3] for r in b.prop_res]
    eps = np.finfo(float).eps
    assert np.max(np.abs(c.state - a.state) / np.maximum(1.0, np.abs(a.state))) <= 1e3 * eps


def test_compiled_jet_hook_is_keyed_on_the_program():
    """The compiled jet function installed for one oracle program is not used by another program with the same batch
    width, number of u variables and order (round-2 advisor finding: the hook was matched on those three numbers only)."""
    import compiled_baseline as cb

    st = np.tile(np.array([1.0, 0.3, -0.2, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 0.9, 0.05])[:, None], (1, 8)).copy()
    st += 1e-3 * np.arange(8)[None, :]
    s1 = ho.nbody(2, masses=[1.0, 0.0])
    s2 = ho.nbody(2, masses=[1.5, 0.0])
    ref = ho.OracleIntegrator(s2, st, 8)
    ref.step(wtc=True)
    a = ho.OracleIntegrator(s1, st, 8)
    b = ho.OracleIntegrator(s2, st, 8)
    assert (a.n_u, a.order) == (b.n_u, b.order)
    try:
        cb.install(a, fast=False)
        b.step(wtc=True)
    finally:
        cb.uninstall()
    assert np.array_equal(b.tc, ref.tc) and np.array_equal(b.state, ref.state)


def test_bracket_solver_is_algorithm_748_with_the_reference_budget():
    """oracle/_bracketed_root(): TOMS 748 with Boost's eps_tolerance and 53 evaluations as the reference calls it
    (src/detail/event_detection.cpp:307-394). Roots of random polynomials with ONE root in [0, 1): agreement with plain
    bisection to the conditioning of the polynomial, far fewer evaluations, the bracket-midpoint convention, and the two
    failure flags the caller turns into "event ignored"."""
    rng = np.random.default_rng(1)
    evals = []
    orig = ho._poly_eval
    try:
        for _ in range(400):
            r = rng.uniform(0.01, 0.99)
            q = np.poly1d([1.0])
            for _k in range(rng.integers(1, 12)):
                q = q * np.poly1d([1.0, -rng.uniform(1.2, 5) * rng.choice([-1, 1])])
            a = [float(c) for c in (np.poly1d([1.0, -r]) * q).coeffs[::-1] * rng.uniform(0.1, 10)]
            a += [0.0] * (21 - len(a))
            lb, ub = 0.0, float(np.nextafter(1.0, 0.0))
            flb = orig(a, lb)
            for _i in range(200):
                mid = lb / 2 + ub / 2
                if mid <= lb or mid >= ub:
                    break
                if (orig(a, mid) < 0) == (flb < 0):
                    lb = mid
                else:
                    ub = mid
            n = [0]

            def counting(aa, x):
                n[0] += 1
                return orig(aa, x)

            ho._poly_eval = counting
            root, flag = ho._bracketed_root(a, 0.0, 1.0)
            ho._poly_eval = orig
            evals.append(n[0])
            assert flag == 0 and n[0] <= 53
            assert abs(root - (lb / 2 + ub / 2)) <= 1e3 * np.spacing(root)
            assert abs(root - r) <= 1e-12
    finally:
        ho._poly_eval = orig
    assert np.mean(evals) < 15
    # No sign change at the ends of the interval: flag 1 (Boost reports a domain error, the event is ignored).
    assert ho._bracketed_root([1.0, 0.0, 1.0], 0.0, 1.0)[1] == 1
    # A root exactly at an end is returned as it is.
    assert ho._bracketed_root([0.0, 1.0], 0.0, 1.0) == (0.0, 0)
