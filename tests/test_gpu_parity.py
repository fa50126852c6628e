"""Parity of the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Tolerances follow the reference's own tests: single step from identical
state - h within 1e4 eps, state within 1e5 eps (test/two_body_batch.cpp:118-150); after a
propagation - 1e3..1e5 eps depending on the number of steps (test/taylor_adaptive_batch.cpp:105-146)."""
import os

import numpy as np
import pytest

import heyoka_amd as hy
import heyoka_oracle as ho
from heyoka_amd import configs
from conftest import EPS, sig_close

pytestmark = pytest.mark.gpu

OC = hy.taylor_outcome


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def row_rel_err(a, b, per_row=False):
    """State comparison with a PER-ROW scale: rows = state variables, columns = systems of the ensemble; the error of a row
    is max |a - b| over the ensemble divided by max |b| over the ensemble (no floor at 1: the Sun's coordinates in the outer
    Solar System are 4e-3 AU / 2e-3 AU/yr and every z component is below 0.25 - with rel_err() '1e6 eps' is an ABSOLUTE
    2e-10 for those rows, 250 times looser than it reads). The Taylor-coefficient comparisons scale the same way."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a, b = a.reshape(b.shape[0], -1), b.reshape(b.shape[0], -1)
    scale = np.max(np.abs(b), axis=1) + 1e-300
    rows = np.max(np.abs(a - b), axis=1) / scale
    return rows if per_row else float(np.max(rows))


def nbody_err(a, b, label=""):
    """Row-scaled error of an N-body state comparison (see row_rel_err()); prints the measured error per row class."""
    b = np.asarray(b)
    if b.ndim == 2 and b.shape[0] % 6 == 0 and b.shape[0] >= 12:
        nbody_row_classes(a, b, label)
    return row_rel_err(a, b)


def nbody_row_classes(a, b, label=""):
    """Measured error in eps per row class of an N-body state (body-major rows x, y, z, vx, vy, vz): the first body (the
    Sun of the outer Solar System), the other bodies' x / y components, the z components. Printed (pytest -s / the
    captured output of a failing test) and returned."""
    rows = row_rel_err(a, b, per_row=True) / EPS
    nb = rows.size // 6
    r = rows.reshape(nb, 6)
    cls = {"first body": float(np.max(r[0])), "others x, y, vx, vy": float(np.max(r[1:][:, [0, 1, 3, 4]])) if nb > 1 else 0.0,
           "others z, vz": float(np.max(r[1:][:, [2, 5]])) if nb > 1 else 0.0}
    print("[row-scaled state error, eps]%s %s" % (" " + label if label else "", ", ".join("%s: %.3g" % kv for kv in cls.items())))
    return cls


def pendulum_p():
    x, v = hy.make_vars("x", "v")
    return [(x, v), (v, -9.8 * hy.sin(x))]


def pendulum_o():
    x, v = ho.var("x"), ho.var("v")
    return [(x, v), (v, -9.8 * ho.sin(x))]


def forced_p():
    x, v = hy.make_vars("x", "v")
    return [(x, v), (v, hy.cos(hy.time) - hy.par[0] * v - hy.sin(x))]


def forced_o():
    x, v = ho.var("x"), ho.var("v")
    return [(x, v), (v, ho.cos(ho.TIME) - ho.par(0) * v - ho.sin(x))]


def test_device_is_mi355x():
    assert hy.device_count() >= 1


def test_pendulum_golden_first_step(golden):
    g = golden["pendulum_scalar"]
    ta = hy.taylor_adaptive_batch(pendulum_p(), [[g["ic"][0]], [g["ic"][1]]], 1)
    ta.step()
    (oc, h), = ta.step_res
    assert oc == OC.success
    assert abs(h - g["step1"]["h"]) <= 1e2 * EPS
    assert ta.time[0] == h
    assert rel_err(ta.state[:, 0], g["step1"]["state"]) <= 1e3 * EPS
    ta.step_backward()
    assert sig_close(ta.step_res[0][1], g["step_backward_h_6digits"])


def test_batch_mode_tutorial_golden(golden):
    """The full doc/tut_batch_mode.rst session on the GPU (config 1 of BASELINE.json)."""
    g = golden["batch_mode_forced_pendulum"]
    ta = hy.taylor_adaptive_batch(forced_p(), [g["x0"], g["v0"]], 4, pars=g["alpha"])
    ta.step()
    res = ta.step_res
    assert all(oc == OC.success for oc, _ in res)
    assert sig_close([h for _, h in res], g["step1"]["h"])
    assert sig_close(ta.state[0], g["step1"]["x"]) and sig_close(ta.state[1], g["step1"]["v"], 7)
    assert sig_close(ta.time, g["step1"]["h"])

    ta.step(g["step_clamped"]["max_delta_t"])
    res = ta.step_res
    assert all(oc == OC.time_limit for oc, _ in res)
    assert [h for _, h in res] == g["step_clamped"]["max_delta_t"]
    assert sig_close(ta.state[0], g["step_clamped"]["x"]) and sig_close(ta.state[1], g["step_clamped"]["v"])

    pf = g["propagate_for"]
    ta.propagate_for(pf["dt"])
    res = ta.propagate_res
    assert [r[3] for r in res] == pf["steps"]
    assert all(r[0] == OC.time_limit for r in res)
    assert sig_close([r[1] for r in res], pf["min_h"]) and sig_close([r[2] for r in res], pf["max_h"])
    assert sig_close(ta.state[0], pf["x"]) and sig_close(ta.state[1], pf["v"])
    assert sig_close(ta.time, pf["time"], 7)

    pu = g["propagate_until"]
    ta.propagate_until(pu["t"])
    res = ta.propagate_res
    assert [r[3] for r in res] == pu["steps"]
    assert sig_close([r[1] for r in res], pu["min_h"]) and sig_close([r[2] for r in res], pu["max_h"])
    assert sig_close(ta.state[0], pu["x"]) and sig_close(ta.state[1], pu["v"], 5)
    assert list(ta.time) == pu["t"]

    ta.step(write_tc=True)
    tc = ta.tc
    assert sig_close(tc[0], g["step_wtc_tc_x"], 6) and sig_close(tc[1], g["step_wtc_tc_v"], 6)
    d = ta.update_d_output(g["dense_output"]["t"])
    assert sig_close(d[0], g["dense_output"]["x"]) and sig_close(d[1], g["dense_output"]["v"])


@pytest.mark.parametrize("ha", [False, True])
def test_single_step_parity_forced_pendulum(ha):
    rng = np.random.RandomState(3)
    n = 1000
    st = np.stack([rng.uniform(-1, 1, n), rng.uniform(-2, 2, n)])
    pars = rng.uniform(0.05, 0.2, n)
    t0 = rng.uniform(0, 10, n)
    ta = hy.taylor_adaptive_batch(forced_p(), st, n, pars=pars, time=t0, high_accuracy=ha)
    ora = ho.OracleIntegrator(forced_o(), st, n, pars=pars, time=t0, high_accuracy=ha)
    ta.step(write_tc=True)
    ora.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in ora.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e4 * EPS
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e5 * EPS
    assert [oc for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
    hi, lo = ta.dtime
    assert rel_err(hi, ora.time_hi) <= 1e4 * EPS
    tc_o = ora.tc.reshape(2, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(ta.tc - tc_o) / scale) <= 1e5 * EPS


def test_two_body_stepwise_and_kepler_invariants():
    """cf. test/two_body_batch.cpp:53-200: 200 steps, each compared with a re-run from the same
    pre-step state; orbital energy / angular momentum conserved."""
    n = 64
    st = configs.two_body_state(n, perturb=0.05, seed=11)
    sys_p, sys_o = hy.model.nbody(2, masses=[1.0, 0.0]), ho.nbody(2, masses=[1.0, 0.0])
    ta = hy.taylor_adaptive_batch(sys_p, st, n)
    s0 = ta.state

    def invariants(s):
        r = s[6:9] - s[0:3]
        v = s[9:12] - s[3:6]
        en = 0.5 * (v * v).sum(0) - 1.0 / np.sqrt((r * r).sum(0))
        lz = r[0] * v[1] - r[1] * v[0]
        return en, lz

    e0, l0 = invariants(s0)
    for _ in range(200):
        pre = ta.state
        pre_t = ta.time
        ora = ho.OracleIntegrator(sys_o, pre, n, time=pre_t)
        ta.step()
        ora.step()
        h_g = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        # NOTE: 1e6 eps instead of the 1e4 eps of the reference's batch-vs-scalar test: the GPU fuses
        # multiply-adds (the oracle does not), and the order-p coefficients entering h are the result
        # of cancellations for near-circular orbits.
        assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert rel_err(ta.state, ora.state.reshape(12, n)) <= 1e5 * EPS
    e1, l1 = invariants(ta.state)
    assert np.max(np.abs((e1 - e0) / e0)) <= 1e4 * EPS
    assert np.max(np.abs((l1 - l0) / l0)) <= 1e4 * EPS


@pytest.mark.parametrize("ha", [False, True])
def test_propagate_until_parity_two_body(ha):
    n = 4096
    st = configs.two_body_state(n, perturb=1e-2, seed=5)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n, high_accuracy=ha)
    ta.propagate_until(50.0)
    oc, mn, mx, ns = ta.propagate_res_arrays()
    ref, thi, tlo, oc_o, mn_o, mx_o, ns_o, tot = ho.ensemble_propagate_until(
        ho.nbody(2, masses=[1.0, 0.0]), st, n, 8, 50.0, high_accuracy=ha)
    assert np.all(oc == int(OC.time_limit)) and np.all(oc_o == ho.OC_TIME_LIMIT)
    assert np.all(ta.time == 50.0)
    # Step counts can differ by +-1 on rare lanes (h differs in the last ulps).
    assert np.max(np.abs(ns.astype(np.int64) - ns_o)) <= 1
    assert ta.last_total_steps == int(ns.sum())
    assert rel_err(ta.state, ref.reshape(12, n)) <= 2e4 * EPS
    assert np.max(np.abs(mn - mn_o) / mn_o) <= 1e-6 and np.max(np.abs(mx - mx_o) / mx_o) <= 1e-6


def test_propagate_backward_limits_and_max_steps():
    n = 256
    rng = np.random.RandomState(9)
    st = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)])
    ta = hy.taylor_adaptive_batch(pendulum_p(), st, n)
    ora = ho.OracleIntegrator(pendulum_o(), st, n)
    # Per-lane final times, forward and backward, with a max_delta_t clamp.
    tf = rng.uniform(-5, 5, n)
    ta.propagate_until(tf, max_delta_t=0.11)
    ora.propagate_until(tf, max_delta_t=0.11)
    res_g, res_o = ta.propagate_res, ora.prop_res
    assert [r[0] for r in res_g] == [r[0] for r in res_o]
    assert max(abs(a[3] - b[3]) for a, b in zip(res_g, res_o)) <= 1
    assert np.array_equal(ta.time, tf)
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e4 * EPS
    # max_steps -> step_limit for lanes that are not done.
    ta.propagate_until(tf + 100.0, max_steps=3)
    res = ta.propagate_res
    assert all(r[0] == OC.step_limit and r[3] == 3 for r in res)
    # Zero-length propagation: time_limit, zero steps.
    ta.propagate_for(0.0)
    assert all(r[0] == OC.time_limit and r[3] == 0 for r in ta.propagate_res)


def test_callback_lockstep_matches_reference_loop():
    n = 8
    rng = np.random.RandomState(1)
    st = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)])
    ta = hy.taylor_adaptive_batch(pendulum_p(), st, n)
    ora = ho.OracleIntegrator(pendulum_o(), st, n)
    calls = []

    def cb(t):
        calls.append(t.time.copy())
        return True

    ta.propagate_until(3.0, callback=cb)
    ora.propagate_until(3.0)
    assert [r[0] for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert [r[3] for r in ta.propagate_res] == [r[3] for r in ora.prop_res]
    # One callback per lock-step iteration = max step count over the batch.
    assert len(calls) == max(r[3] for r in ora.prop_res)
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e4 * EPS
    # cb returning False -> cb_stop for all lanes after one iteration.
    ta.propagate_until(6.0, callback=lambda t: False)
    assert all(r[0] == OC.cb_stop for r in ta.propagate_res)


def test_nonfinite_state_is_reported_per_lane():
    """batch_semantics = "per_lane" (the fully asynchronous device-resident path): a lane which produces a non-finite
    state stops itself, the other lanes reach their final time. (The default reproduces the reference, where the whole
    batch stops at that iteration: test_reference_batch_semantics.)"""
    x, v = hy.make_vars("x", "v")
    # x' = x^2 blows up in finite time for x0 > 0 (t* = 1/x0): lanes 0/1 explode before t = 3.
    sys = [(x, x * x), (v, -v)]
    st = [[1.0, 0.5, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0]]
    ta = hy.taylor_adaptive_batch(sys, st, 4, batch_semantics="per_lane")
    ta.propagate_until(3.0, max_steps=20000)
    res = ta.propagate_res
    assert res[2][0] == OC.time_limit and res[3][0] == OC.time_limit
    assert res[0][0] in (OC.err_nf_state, OC.step_limit) and res[1][0] in (OC.err_nf_state, OC.step_limit)


def test_outer_ss_step_selector_identity_and_energy(outer_ss_golden):
    """test/timestep_check.cpp:25-95 (h reproduces Jorba's formula from the TCs) and
    test/model_nbody.cpp:112-118 (energy conserved after propagate_until(100))."""
    n = 128
    st = configs.outer_ss_state(n, perturb=1e-10, seed=42)
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    p = ta.order
    for _ in range(3):
        ta.step(write_tc=True)
        ora.step(wtc=True)
        tc = ta.tc
        m0 = np.max(np.abs(tc[:, 0, :]), axis=0)
        mp = np.max(np.abs(tc[:, p, :]), axis=0)
        mp1 = np.max(np.abs(tc[:, p - 1, :]), axis=0)
        num = np.where(m0 <= 1, 1.0, m0)
        rho = np.minimum((num / mp) ** (1.0 / p), (num / mp1) ** (1.0 / (p - 1)))
        h_formula = rho * np.exp(-0.7 / (p - 1)) / np.e ** 2
        h_g = ta.last_h
        assert np.max(np.abs(h_g - h_formula) / h_formula) <= 100 * EPS
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_g - h_o) / h_o) <= 1e4 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e5 * EPS
    e0 = configs.nbody_energy(st, M, G)
    ta.propagate_until(100.0)
    assert all(r[0] == OC.time_limit for r in ta.propagate_res)
    e1 = configs.nbody_energy(ta.state, M, G)
    assert np.max(np.abs((e1 - e0) / e0)) <= 100 * EPS


def test_raw_step_abi():
    import torch

    n = 512
    st = configs.two_body_state(n, perturb=1e-2, seed=2)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    ora = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n)
    dev = torch.device("cuda:0")
    d_state = torch.tensor(st, device=dev, dtype=torch.float64).contiguous()
    d_time = torch.zeros(n, device=dev, dtype=torch.float64)
    d_h = torch.full((n,), float("inf"), device=dev, dtype=torch.float64)
    d_tc = torch.zeros((12, ta.order + 1, n), device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    ta.raw_step(d_state.data_ptr(), 0, d_time.data_ptr(), d_h.data_ptr(), d_tc.data_ptr(), n)
    ora.step(wtc=True)
    h_o = np.array([h for _, h in ora.step_res])
    # NOTE: h is ill-conditioned for near-circular orbits (the order-p coefficients are tiny sums of
    # cancelling terms): 1e6 eps here, while the propagated state agrees to 1e5 eps.
    h_g = d_h.cpu().numpy()
    assert np.max(np.abs(h_g - h_o) / h_o) <= 1e6 * EPS
    # The new state against the ORACLE's Taylor polynomials evaluated at the step the device took (a difference of
    # 1e6 eps in h moves the state by up to ~1e6 eps |v| h / |x| on its own), and the polynomials themselves.
    tc_o = ora.tc.reshape(12, ta.order + 1, n)
    st_o = np.zeros((12, n))
    for k in range(ta.order, -1, -1):
        st_o = st_o * h_g + tc_o[:, k, :]
    assert rel_err(d_state.cpu().numpy(), st_o) <= 1e5 * EPS
    assert rel_err(d_tc.cpu().numpy(), tc_o) <= 1e5 * EPS
    assert rel_err(d_state.cpu().numpy(), ora.state.reshape(12, n)) <= 1e6 * EPS
    assert np.array_equal(d_tc[:, 0, :].cpu().numpy(), st)


def test_raw_stepper_abi_step_e_d_out_f_and_caller_owned_tape(golden):
    """The other pointer types of the reference's stepper ABI (include/heyoka/detail/ta_jit_data.hpp:34-43) through the C
    ABI on caller-owned device buffers. step_f_e_t: jets of the state and of the event equations, step size and max |x_i|
    against the oracle's hy_oracle_step_e (the reference's stepper with events, src/taylor_00.cpp:592-710), the state left
    untouched - for the unrolled stepper (pendulum with one terminal and one non-terminal event) and for the wave-cluster
    stepper which evaluates the event equations itself (outer Solar System). d_out_f_t: the dense output of the batch-mode
    tutorial (doc/tut_batch_mode.rst golden values) and the compensated evaluation of the oracle. c_step_f_t / c_step_f_e_t:
    the compact-mode steppers (without / with events) on a caller-owned tape of hy_tab_tape_size_align() bytes give the
    results of the integrator's own tape bit for bit AND the compact-mode oracle's (hy_oracle_step_e for the stepper with
    events) to the reference's tolerances."""
    import ctypes

    import torch

    dev = torch.device("cuda:0")

    def dt(a):
        return torch.tensor(np.ascontiguousarray(a), device=dev, dtype=torch.float64)

    def oracle_step_e(ora, n):
        h = np.full(n, np.inf)
        n_ev = len(ora.t_events) + len(ora.nt_events)
        ho._lib().hy_oracle_step_e(ctypes.byref(ora._prog), n, ho._p(ora.state), ho._p(ora.pars), ho._p(ora.time_hi), ho._p(h),
                                   ho._p(ora.tc), ho._p(ora._ev_u), n_ev, ho._p(ora._ev_tc), ho._p(ora._mas), ho._p(ora._scratch))
        return h, ora.tc.copy(), ora._ev_tc.copy(), ora._mas.copy()

    # ---- step_e, unrolled stepper: pendulum, events v = 0 (terminal) and x = 0 (non-terminal).
    n = 200
    rng = np.random.RandomState(8)
    st = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)])
    x, v = hy.make_vars("x", "v")
    ox, ov = ho.var("x"), ho.var("v")
    ta = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, n, t_events=[hy.t_event(v)],
                                  nt_events=[hy.nt_event(x, lambda *a: None)])
    ora = ho.OracleEventIntegrator([(ox, ov), (ov, -9.8 * ho.sin(ox))], st, n, t_events=[ho.t_event(ov)],
                                   nt_events=[ho.nt_event(ox, lambda *a: None)])
    p = ta.order
    h_o, tc_o, evtc_o, mas_o = oracle_step_e(ora, n)
    d_state, d_time, d_h = dt(st), dt(np.zeros(n)), dt(np.full(n, np.inf))
    d_jet = torch.zeros((2 + 2) * (p + 1) * n, device=dev, dtype=torch.float64)
    d_mas = torch.zeros(n, device=dev, dtype=torch.float64)
    ta.raw_step_e(d_jet.data_ptr(), d_state.data_ptr(), 0, d_time.data_ptr(), d_h.data_ptr(), d_mas.data_ptr(), n)
    jet = d_jet.cpu().numpy()
    assert np.array_equal(d_state.cpu().numpy(), st)  # no state update
    assert np.max(np.abs(d_h.cpu().numpy() - h_o) / np.abs(h_o)) <= 1e4 * EPS
    assert np.array_equal(d_mas.cpu().numpy(), mas_o)
    tc_ref = tc_o.reshape(2, p + 1, n)
    scale = np.max(np.abs(tc_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[: 2 * (p + 1) * n].reshape(2, p + 1, n) - tc_ref) / scale) <= 1e5 * EPS
    ev_ref = evtc_o.reshape(2, p + 1, n)
    scale = np.max(np.abs(ev_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[2 * (p + 1) * n:].reshape(2, p + 1, n) - ev_ref) / scale) <= 1e5 * EPS

    # ---- step_e, the stepper which evaluates the event equations itself (outer Solar System, v5 with events inside).
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 48
    st = configs.outer_ss_state(n, perturb=1e-6, seed=9)
    x1, x2 = hy.make_vars("x_1", "x_2")
    o1, o2 = ho.var("x_1"), ho.var("x_2")
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True,
                                  nt_events=[hy.nt_event(x1 - x2, lambda *a: None)])
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, nt_events=[ho.nt_event(o1 - o2, lambda *a: None)],
                                   high_accuracy=True)
    p = ta.order
    h_o, tc_o, evtc_o, mas_o = oracle_step_e(ora, n)
    d_state, d_time, d_h = dt(st), dt(np.zeros(n)), dt(np.full(n, np.inf))
    d_jet = torch.zeros((36 + 1) * (p + 1) * n, device=dev, dtype=torch.float64)
    d_mas = torch.zeros(n, device=dev, dtype=torch.float64)
    ta.raw_step_e(d_jet.data_ptr(), d_state.data_ptr(), 0, d_time.data_ptr(), d_h.data_ptr(), d_mas.data_ptr(), n)
    jet = d_jet.cpu().numpy()
    assert np.array_equal(d_state.cpu().numpy(), st)
    assert np.max(np.abs(d_h.cpu().numpy() - h_o) / np.abs(h_o)) <= 1e6 * EPS
    assert np.max(np.abs(d_mas.cpu().numpy() - mas_o) / mas_o) <= 4 * EPS
    tc_ref = tc_o.reshape(36, p + 1, n)
    scale = np.max(np.abs(tc_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[: 36 * (p + 1) * n].reshape(36, p + 1, n) - tc_ref) / scale) <= 1e6 * EPS
    ev_ref = evtc_o.reshape(1, p + 1, n)
    scale = np.max(np.abs(ev_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[36 * (p + 1) * n:].reshape(1, p + 1, n) - ev_ref) / scale) <= 1e6 * EPS

    # ---- d_out_f: the oracle's compensated / Horner evaluation of its own Taylor coefficients at fractions of the step.
    for ha in (False, True):
        n = 64
        st = configs.two_body_state(n, perturb=1e-2, seed=12)
        ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n, high_accuracy=ha)
        ora = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n, high_accuracy=ha)
        ora.step(wtc=True)
        hs = np.array([h for _, h in ora.step_res]) * np.linspace(0.1, 1.0, n)
        d_out = torch.zeros(12 * n, device=dev, dtype=torch.float64)
        d_tc, d_hs = dt(ora.tc), dt(hs)  # (kept alive across the call)
        ta.raw_d_out_f(d_out.data_ptr(), d_tc.data_ptr(), d_hs.data_ptr(), n)
        exp = np.stack([ho.OracleEventIntegrator._dense(ora, i, hs[i]) for i in range(n)], axis=1)  # (the oracle's evaluation of its own coefficients)
        assert rel_err(d_out.cpu().numpy().reshape(12, n), exp) <= 4 * EPS

    # ---- c_step_f_t: the compact-mode stepper with its tape in HBM (kw::emitter = table; HEYOKA_AMD_TABLE_LDS=0 is the
    # developer switch for the lane-per-system variant where the tape of a system would fit in LDS) on a CALLER-OWNED tape
    # of hy_tab_tape_size_align() bytes: the results of the integrator's own tape bit for bit, and the COMPACT-MODE oracle
    # (running sums inside the convolutions, src/math/prod.cpp:686-698) to the tolerances of the reference
    # (test/two_body_batch.cpp:118-150: h 1e4 eps, coefficients 1e5 eps of the row maximum; the default build contracts
    # multiply-adds, the strict build is bit-identical: test_compact_mode_has_the_arithmetic_of_the_reference_compact_mode).
    n = 4096
    st = configs.plummer_nbody_state(4, n, seed=5)

    def hbm_table(**kw):
        old_env = os.environ.get("HEYOKA_AMD_TABLE_LDS")
        os.environ["HEYOKA_AMD_TABLE_LDS"] = "0"
        try:
            return hy.taylor_adaptive_batch(hy.model.nbody(4), st, n, compact_mode=True, emitter="table", **kw)
        finally:
            if old_env is None:
                del os.environ["HEYOKA_AMD_TABLE_LDS"]
            else:
                os.environ["HEYOKA_AMD_TABLE_LDS"] = old_env

    ta = hbm_table()
    assert "tape in HBM" in ta.hip_source_mode, ta.hip_source_mode
    size, align = ta.raw_tape_size_align(n)
    assert size > 0 and align >= 8
    tape = torch.empty(size // 8 + align // 8, device=dev, dtype=torch.float64)
    tptr = (tape.data_ptr() + align - 1) // align * align
    res = []
    for use_tape in (False, True):
        d_state, d_time, d_h = dt(st), dt(np.zeros(n)), dt(np.full(n, np.inf))
        d_tc = torch.zeros(24 * (ta.order + 1) * n, device=dev, dtype=torch.float64)
        ta.raw_step(d_state.data_ptr(), 0, d_time.data_ptr(), d_h.data_ptr(), d_tc.data_ptr(), n, d_tape=tptr if use_tape else None)
        res.append((d_state.cpu().numpy(), d_h.cpu().numpy(), d_tc.cpu().numpy()))
    for a_, b_ in zip(res[0], res[1]):
        assert np.array_equal(a_, b_)
    oc = ho.OracleIntegrator(ho.nbody(4), st, n, compact_mode=True)
    oc.step(wtc=True)
    h_o = np.array([h for _, h in oc.step_res])
    assert np.max(np.abs(res[1][1] - h_o) / np.abs(h_o)) <= 1e4 * EPS
    tc_ref = oc.tc.reshape(24, ta.order + 1, n)
    scale = np.max(np.abs(tc_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(res[1][2].reshape(tc_ref.shape) - tc_ref) / scale) <= 1e5 * EPS
    assert row_rel_err(res[1][0].reshape(24, n), oc.state.reshape(24, n)) <= 1e5 * EPS

    # ---- c_step_f_e_t: the compact-mode stepper WITH EVENTS on a caller-owned tape (hy_tab_raw_step_e_tape(),
    # include/heyoka/detail/ta_jit_data.hpp:40-43) against hy_oracle_step_e: jets of the state variables and of the event
    # equations, step size, max |x_i|; the state stays untouched; identical to the integrator's own tape.
    xa, xb = hy.make_vars("x_0", "x_1")
    oa, ob = ho.var("x_0"), ho.var("x_1")
    te = hbm_table(nt_events=[hy.nt_event(xa - xb, lambda *a: None)])
    assert "tape in HBM" in te.hip_source_mode, te.hip_source_mode
    ore = ho.OracleEventIntegrator(ho.nbody(4), st, n, nt_events=[ho.nt_event(oa - ob, lambda *a: None)])
    p = te.order
    h_o, tc_o, evtc_o, mas_o = oracle_step_e(ore, n)
    size_e, align_e = te.raw_tape_size_align(n)
    assert size_e > 0
    tape_e = torch.empty(size_e // 8 + align_e // 8, device=dev, dtype=torch.float64)
    tptr_e = (tape_e.data_ptr() + align_e - 1) // align_e * align_e
    jets = []
    for use_tape in (False, True):
        d_state, d_time, d_h = dt(st), dt(np.zeros(n)), dt(np.full(n, np.inf))
        d_jet = torch.zeros((24 + 1) * (p + 1) * n, device=dev, dtype=torch.float64)
        d_mas = torch.zeros(n, device=dev, dtype=torch.float64)
        te.raw_step_e(d_jet.data_ptr(), d_state.data_ptr(), 0, d_time.data_ptr(), d_h.data_ptr(), d_mas.data_ptr(), n,
                      d_tape=tptr_e if use_tape else None)
        assert np.array_equal(d_state.cpu().numpy(), st)
        jets.append((d_jet.cpu().numpy(), d_h.cpu().numpy(), d_mas.cpu().numpy()))
    for a_, b_ in zip(jets[0], jets[1]):
        assert np.array_equal(a_, b_)
    jet, h_g, mas_g = jets[1]
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e4 * EPS
    assert np.array_equal(mas_g, mas_o)
    tc_ref = tc_o.reshape(24, p + 1, n)
    scale = np.max(np.abs(tc_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[: 24 * (p + 1) * n].reshape(24, p + 1, n) - tc_ref) / scale) <= 1e5 * EPS
    ev_ref = evtc_o.reshape(1, p + 1, n)
    scale = np.max(np.abs(ev_ref), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(jet[24 * (p + 1) * n:].reshape(1, p + 1, n) - ev_ref) / scale) <= 1e5 * EPS
    # (A stepper which keeps its coefficients on chip needs no tape.)
    assert hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, 64).raw_tape_size_align(64)[0] == 0


def test_device_array_views_and_ensemble():
    import torch

    n = 1024
    st = configs.two_body_state(n, perturb=1e-2, seed=21)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, n)
    # Write the ICs directly into the device-resident state through a zero-copy torch view.
    view = torch.as_tensor(ta.device_array("state"), device="cuda:0")
    view.copy_(torch.tensor(st, dtype=torch.float64))
    torch.cuda.synchronize()
    ta.mark_device_modified()
    ta.propagate_until(5.0)
    ref, *_ = ho.ensemble_propagate_until(ho.nbody(2, masses=[1.0, 0.0]), st, n, 8, 5.0)
    assert rel_err(ta.state, ref.reshape(12, n)) <= 1e4 * EPS
    assert rel_err(view.cpu().numpy(), ref.reshape(12, n)) <= 1e4 * EPS

    # ensemble_propagate_until_batch: bitwise equal to a serial propagate_until on the same object
    # (test/ensemble_propagate.cpp:419-420).
    tmpl = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, 256)
    chunks = [st[:, i * 256:(i + 1) * 256] for i in range(4)]

    def gen(tc, i):
        tc.state = chunks[i]

    out = hy.ensemble_propagate_until_batch(tmpl, 5.0, 4, gen)
    # The threaded core (csrc/ensemble.cpp: one host thread per device; n_devices = -3: three workers mapped round-robin
    # onto the visible devices, i.e. all onto GPU 0 here) returns the same objects bit for bit.
    out_t = hy.ensemble_propagate_until_batch(tmpl, 5.0, 4, gen, n_devices=-3)
    assert len(out_t) == len(out) and all(np.array_equal(a.state, b.state) and a.propagate_res == b.propagate_res
                                          for a, b in zip(out_t, out))
    for i, o in enumerate(out):
        serial = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), chunks[i], 256)
        serial.propagate_until(5.0)
        assert np.array_equal(o.state, serial.state)
        assert np.array_equal(o.state, ta.state[:, i * 256:(i + 1) * 256])


def _select_cluster_kernel(monkeypatch, kernel):
    """Generator selection for the tests (construction-time): v5 = one lane per pair (default), v3 = lane pairs,
    v2 = pipelined one-lane-per-cluster kernel at one wavefront per SIMD."""
    assert kernel in ("v5", "v3", "v2")
    if kernel in ("v3", "v2"):
        monkeypatch.setenv("HEYOKA_AMD_ONE_LANE", "0")
    if kernel == "v2":
        monkeypatch.setenv("HEYOKA_AMD_PAIR_SPLIT", "0")


def _nbody_parity(n_bodies, n_sys, n_steps, expect_mode, t_final=None, env_mode=None, tol=1e5, tc_tol=1e6):
    import os

    st = configs.plummer_nbody_state(n_bodies, n_sys, seed=77, jitter=1e-6)
    old = os.environ.get("HEYOKA_AMD_EMIT_MODE")
    if env_mode is not None:
        os.environ["HEYOKA_AMD_EMIT_MODE"] = env_mode
    try:
        ta = hy.taylor_adaptive_batch(hy.model.nbody(n_bodies), st, n_sys)
    finally:
        if env_mode is not None:
            if old is None:
                del os.environ["HEYOKA_AMD_EMIT_MODE"]
            else:
                os.environ["HEYOKA_AMD_EMIT_MODE"] = old
    assert expect_mode in ta.hip_source_mode, ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(n_bodies), st, n_sys)
    for _ in range(n_steps):
        ta.step(write_tc=True)
        ora.step(wtc=True)
        h_g = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_g - h_o) / h_o) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(6 * n_bodies, n_sys)) <= tol * EPS
    tc_o = ora.tc.reshape(6 * n_bodies, ora.order + 1, n_sys)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(ta.tc - tc_o) / scale) <= tc_tol * EPS
    if t_final is not None:
        ta.propagate_until(t_final)
        ora.propagate_until(t_final)
        assert all(r[0] == OC.time_limit for r in ta.propagate_res)
        assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
        assert nbody_err(ta.state, ora.state.reshape(6 * n_bodies, n_sys)) <= 1e7 * EPS
    return ta


def test_cluster_mode_nbody8_default_masses():
    """28 isomorphic pair clusters, L = 32 lanes per system, reaction terms through sub/neg glue."""
    _nbody_parity(8, 96, 3, "cluster", t_final=0.05)


@pytest.mark.parametrize("kernel", ["v5", "v3"])
@pytest.mark.parametrize("n_bodies", [7, 8])
def test_pair_kernels_with_17_to_32_pairs(n_bodies, kernel, monkeypatch):
    """17 .. 32 pair clusters (model::nbody(7), nbody(8) with numerical masses): the one-lane-per-pair kernel with 32 lanes
    per system (two systems per wavefront) and the lane-pair kernel with ONE system per wavefront (64 lanes per system).
    In round 2 the latter did not terminate on the hardware and was gated off; with the zero-length steps of a finished
    system forced to h = 0 exactly it runs: steps, a step-limited and a complete propagation against the oracle."""
    _select_cluster_kernel(monkeypatch, kernel)
    n = 24
    rng = np.random.RandomState(3)
    masses = [1.0] + [1e-3 * (i + 1) for i in range(n_bodies - 1)]
    st = np.zeros((6 * n_bodies, n))
    for b in range(1, n_bodies):
        r = 1.0 + 0.7 * b
        ph = rng.uniform(0, 2 * np.pi, n)
        v = 1.0 / np.sqrt(r)
        st[6 * b + 0], st[6 * b + 1], st[6 * b + 2] = r * np.cos(ph), r * np.sin(ph), 0.01 * rng.randn(n)
        st[6 * b + 3], st[6 * b + 4], st[6 * b + 5] = -v * np.sin(ph), v * np.cos(ph), 0.01 * rng.randn(n)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(n_bodies, masses=masses), st, n, high_accuracy=True)
    lanes = 32 if kernel == "v5" else 64
    assert "lanes per system: %d" % lanes in ta.hip_source_mode and kernel in ta.hip_source_mode, ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(n_bodies, masses=masses), st.reshape(-1), n, high_accuracy=True)
    for _ in range(3):
        ta.step()
        ora.step()
        h_g = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_g - h_o) / h_o) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(6 * n_bodies, n)) <= 1e5 * EPS
    ta.propagate_until(40.0, max_steps=4)
    ora.propagate_until(40.0, max_steps=4)
    assert [(int(r[0]), r[3]) for r in ta.propagate_res] == [(r[0], r[3]) for r in ora.prop_res]
    ta.propagate_until(8.0)
    ora.propagate_until(8.0)
    assert all(r[0] == OC.time_limit for r in ta.propagate_res)
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(ta.state, ora.state.reshape(6 * n_bodies, n)) <= 1e6 * EPS


@pytest.mark.parametrize("variant", ["staged", "tape in HBM"])
def test_table_mode_small_dag_forced(variant, monkeypatch):
    """Table (compact-mode analogue) kernels on a DAG that would normally be unrolled, in both variants: the staged
    stepper - one system per workgroup, tape in LDS, code specialised per group of nodes: the default whenever the tape of
    a system fits in LDS - and one system per lane with the tape in HBM (HEYOKA_AMD_TABLE_LDS=0, and the automatic choice
    for decompositions whose tape is larger than the LDS of a CU)."""
    monkeypatch.setenv("HEYOKA_AMD_TABLE_LDS", "1" if variant == "staged" else "0")
    ta = _nbody_parity(3, 200, 3, "table", t_final=0.1, env_mode="table")
    assert variant in ta.hip_source_mode
    if variant == "staged":
        # The default choice does not depend on the batch size; a tape beyond the LDS of a CU goes to HBM.
        monkeypatch.delenv("HEYOKA_AMD_TABLE_LDS")
        monkeypatch.setenv("HEYOKA_AMD_EMIT_MODE", "table")
        assert "staged" in hy.taylor_adaptive_batch(hy.model.nbody(3), None, 32768).hip_source_mode
        assert "staged" in hy.taylor_adaptive_batch(hy.model.nbody(3), None, 1 << 20).hip_source_mode
        big = hy.taylor_adaptive_batch(hy.model.nbody(32), None, 64).hip_source_mode
        assert "tape in HBM" in big and "does not fit in the LDS" in big, big


def test_nbody12_block_automatic_and_table_forced():
    """66 clusters (> 64 lanes): one system per workgroup (block mode, 128 lanes); the table-driven
    one-lane-per-system kernel remains available (HEYOKA_AMD_EMIT_MODE=table) and agrees."""
    _nbody_parity(12, 64, 2, "block", t_final=0.02)
    _nbody_parity(12, 64, 2, "table", t_final=0.02, env_mode="table")


def test_config5_nbody64_table_mode():
    """BASELINE config 5 DAG (18 663 u variables) at a reduced ensemble size: parity vs the oracle
    (table-driven kernel, one system per lane)."""
    ta = _nbody_parity(64, 64, 1, "table", env_mode="table")
    assert ta.n_uvars == 18663


def test_block_mode_nbody20_forced_and_nbody64_automatic():
    """Block mode (one system per workgroup, cluster jets on a coalesced tape): 190 clusters, a forced
    instance on a DAG that normally runs in wave-cluster mode with more systems than workgroups in flight
    and the compensated update, and the BASELINE config 5 DAG (2016 clusters)."""
    ta = _nbody_parity(20, 24, 2, "block", t_final=0.02)
    assert "190 clusters" in ta.hip_source_mode
    # More systems than workgroups in flight + high accuracy (compensated summation update).
    import os

    os.environ["HEYOKA_AMD_EMIT_MODE"] = "block"
    try:
        st = configs.plummer_nbody_state(9, 700, seed=5, jitter=1e-6)
        tb = hy.taylor_adaptive_batch(hy.model.nbody(9), st, 700, high_accuracy=True)
    finally:
        del os.environ["HEYOKA_AMD_EMIT_MODE"]
    assert "block" in tb.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(9), st, 700, high_accuracy=True)
    tb.propagate_until(0.05)
    ora.propagate_until(0.05)
    assert all(r[0] == OC.time_limit for r in tb.propagate_res)
    assert max(abs(a[3] - b[3]) for a, b in zip(tb.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(tb.state, ora.state.reshape(54, 700)) <= 1e7 * EPS
    tc = _nbody_parity(64, 8, 1, "block")
    assert tc.n_uvars == 18663 and "2016 clusters" in tc.hip_source_mode


def test_unrolled_vs_cluster_v1_vs_v2_same_results():
    """The three code generators agree with each other on the outer Solar System (same inputs)."""
    import os

    n = 64
    st = configs.outer_ss_state(n, perturb=1e-8, seed=3)
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    res = {}
    for name, env in (("v2", {}), ("v1", {"HEYOKA_AMD_CLUSTER_V1": "1"}), ("table", {"HEYOKA_AMD_EMIT_MODE": "table"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        ta.propagate_until(20.0)
        res[name] = (ta.state, np.array([r[3] for r in ta.propagate_res]))
    for name in ("v1", "table"):
        assert rel_err(res[name][0], res["v2"][0]) <= 1e5 * EPS
        assert np.max(np.abs(res[name][1] - res["v2"][1])) <= 1


def test_propagate_grid():
    """propagate_grid() (src/taylor_adaptive_batch.cpp:1546-2055): doc/tut_adaptive.rst:324-325 known answer,
    agreement with direct propagation to every grid point, per-lane grids, error paths."""
    ta = hy.taylor_adaptive_batch(pendulum_p(), [[0.05] * 3, [0.025] * 3], 3)
    grid = np.linspace(0.0, 1.0, 11)
    _, out = ta.propagate_grid(grid)
    assert out.shape == (11, 2, 3)
    assert sig_close(out[4, 0, 0], 0.0232578) and sig_close(out[4, 1, 0], -0.14078)
    assert np.array_equal(out[0], [[0.05] * 3, [0.025] * 3])
    assert all(r[0] == OC.time_limit for r in ta.propagate_res)
    assert np.array_equal(ta.time, [1.0] * 3)
    assert rel_err(ta.state, out[-1]) <= 10 * EPS
    # Every grid point agrees with a direct propagate_until() from the initial conditions.
    for k in (1, 5, 10):
        tb = hy.taylor_adaptive_batch(pendulum_p(), [[0.05] * 3, [0.025] * 3], 3)
        tb.propagate_until(float(grid[k]))
        assert rel_err(out[k], tb.state) <= 100 * EPS
    # Per-lane grids (different spacing per lane), backward in time, with the oracle as cross-check.
    n = 4
    rng = np.random.RandomState(4)
    st = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)])
    g2 = -np.outer(np.arange(6.0), np.array([0.3, 0.5, 0.7, 1.1]))
    tc_ = hy.taylor_adaptive_batch(pendulum_p(), st, n)
    _, out2 = tc_.propagate_grid(g2)
    for k in range(1, 6):
        ora = ho.OracleIntegrator(pendulum_o(), st, n)
        ora.propagate_until(g2[k])
        assert rel_err(out2[k], ora.state.reshape(2, n)) <= 1e3 * EPS
    # Error paths (messages of the reference).
    with pytest.raises(ValueError, match="non-monotonic time grid"):
        tc_.propagate_grid(np.outer([-5.5, -6.0, -5.8], np.ones(n)) * np.array([0.3, 0.5, 0.7, 1.1]) / 0.3 * 0.3)
    with pytest.raises(ValueError, match="must match the current time coordinate"):
        tc_.propagate_grid(np.outer([0.0, 1.0], np.ones(n)))
    # max_steps -> step_limit, unreached grid points are NaN.
    td = hy.taylor_adaptive_batch(pendulum_p(), [[0.05], [0.025]], 1)
    _, out3 = td.propagate_grid(np.linspace(0.0, 10.0, 11), max_steps=2)
    assert td.propagate_res[0][0] == OC.step_limit
    assert np.isnan(out3[-1]).all() and not np.isnan(out3[0]).any()


def test_continuous_output():
    """kw::c_output (src/taylor_adaptive_batch.cpp:1243-1346, src/continuous_output.cpp:602-1236), modelled on
    test/c_output.cpp:289-560: agreement with the oracle's propagation to arbitrary times, with
    propagate_grid(), bounds / padding / tcs bookkeeping, error messages, device-side evaluation."""
    import copy

    n = 6
    rng = np.random.RandomState(11)
    st = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)])
    tf = 3.0 + 0.25 * np.arange(n)
    for ha in (False, True):
        ta = hy.taylor_adaptive_batch(pendulum_p(), st, n, high_accuracy=ha)
        co, cb = ta.propagate_until(tf, c_output=True)
        assert cb is None and co is not None
        assert co.batch_size == n and co.dim == 2 and co.order == 20
        ns = co.n_steps
        assert ns == max(r[3] for r in ta.propagate_res)
        times = co.times
        assert times.shape == (ns + 2, n) and np.all(times[0] == 0.0) and np.all(np.isposinf(times[-1]))
        lb, ub = co.bounds
        assert np.array_equal(lb, np.zeros(n)) and np.array_equal(ub, tf)
        # Lanes which finish early repeat their final time (zero-length steps) up to the padding row.
        assert np.all(np.diff(times[:-1], axis=0) >= 0.0)
        tcs = co.tcs
        assert tcs.shape == (ns, 2, 21, n)
        assert np.array_equal(tcs[0, :, 0, :], st)
        # Arbitrary target times vs an independent oracle propagation.
        for frac in (0.0, 0.21, 0.5, 0.83, 1.0):
            tm = frac * tf
            out = co(tm)
            ora = ho.OracleIntegrator(pendulum_o(), st, n, high_accuracy=ha)
            ora.propagate_until(tm)
            assert rel_err(out, ora.state.reshape(2, n)) <= 1e3 * EPS
            assert co.output is out
        # Scalar time and agreement with propagate_grid().
        tg = hy.taylor_adaptive_batch(pendulum_p(), st, n, high_accuracy=ha)
        grid = np.linspace(0.0, 3.0, 7)
        _, gout = tg.propagate_grid(grid)
        for k, t in enumerate(grid):
            assert rel_err(co(float(t)), gout[k]) <= 100 * EPS
        # Copies share the device data.
        co2 = copy.copy(co)
        assert np.array_equal(co2(1.25), co(1.25))
        # Device-side evaluation on caller-owned buffers.
        import torch

        d_tm = torch.full((n,), 1.25, dtype=torch.float64, device="cuda")
        d_out = torch.empty((2, n), dtype=torch.float64, device="cuda")
        co.eval_device(d_tm.data_ptr(), d_out.data_ptr())
        ta.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), co(1.25))
        with pytest.raises(ValueError, match="the vector size is 7, but a size of 6 was expected instead"):
            co(np.zeros(n + 1))
        with pytest.raises(ValueError, match="at the non-finite time"):
            co(float("nan"))
        assert "forward" in repr(co) and "N of steps  : %d" % ns in repr(co)
        # Backward, with a callback and max_delta_t at the same time.
        calls = []
        co_b, cb_b = ta.propagate_for(-tf, c_output=True, max_delta_t=0.05, callback=lambda t: calls.append(1) or True)
        assert co_b.n_steps == len(calls) and co_b.n_steps >= 60 and "backward" in repr(co_b)
        assert rel_err(co_b(0.0), st) <= 1e3 * EPS
    # No step taken -> no continuous output (empty optional in the reference).
    bad = st.copy()
    bad[1, 0] = np.inf
    tb = hy.taylor_adaptive_batch(pendulum_p(), bad, n)
    co_n, _ = tb.propagate_until(1.0, c_output=True)
    assert co_n is None and tb.propagate_res[0][0] == OC.err_nf_state


def test_cfunc_values_and_device_side_energy_monitor():
    """cfunc<double> on the device: values against numpy, parameters and time,
    and model::nbody_energy evaluated directly on the integrator's device-resident state
    (test/model_nbody.cpp:112-118: relative energy drift <= 100 eps over propagate_until(100))."""
    import torch

    rng = np.random.RandomState(5)
    x, v = hy.make_vars("x", "v")
    cf = hy.cfunc([x * hy.par[1] + hy.cos(hy.time) - hy.par[0], hy.exp(v) * hy.sin(x) / (1.0 + v * v),
                   hy.pow(x * x + 1.0, -1.5) + hy.log(2.0 + v * v), x], [x, v])
    nev = 1000
    inp, pars, tm = rng.uniform(-1, 1, (2, nev)), rng.uniform(-1, 1, (2, nev)), rng.uniform(0, 10, nev)
    out = cf(inp, pars=pars, time=tm)
    X, V = inp
    exp = np.stack([X * pars[1] + np.cos(tm) - pars[0], np.exp(V) * np.sin(X) / (1.0 + V * V),
                    (X * X + 1.0) ** -1.5 + np.log(2.0 + V * V), X])
    assert out.shape == (4, nev) and rel_err(out, exp) <= 8 * EPS
    # Single evaluation overload.
    o1 = cf(inp[:, 0], pars=pars[:, 0], time=tm[0])
    assert o1.shape == (4,) and np.array_equal(o1, out[:, 0])
    with pytest.raises(ValueError, match="An array of parameter values must be passed"):
        cf(inp, time=tm)
    with pytest.raises(ValueError, match="time value"):
        cf(inp[:, 0], pars=pars[:, 0])
    with pytest.raises(ValueError, match="Invalid inputs array passed to a cfunc"):
        cf(inp[:1, 0], pars=pars[:, 0], time=tm[0])

    # Energy monitor on the outer Solar System ensemble, no host round trip for the state.
    n = 256
    st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
    masses, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sys_ = hy.model.nbody(6, masses=masses, Gconst=G)
    en = hy.cfunc([hy.model.nbody_energy(6, masses=masses, Gconst=G)], sys_.vars)
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True)
    d_e0 = torch.empty(n, dtype=torch.float64, device="cuda")
    d_e1 = torch.empty(n, dtype=torch.float64, device="cuda")
    en.eval_device(d_e0.data_ptr(), ta.device_array("state").ptr, n)
    ta.propagate_until(100.0)
    en.eval_device(d_e1.data_ptr(), ta.device_array("state").ptr, n)
    ta.synchronize()
    torch.cuda.synchronize()
    e0, e1 = d_e0.cpu().numpy(), d_e1.cpu().numpy()
    assert rel_err(e0, configs.nbody_energy(st, masses, G)) <= 16 * EPS
    assert np.max(np.abs((e1 - e0) / e0)) <= 100 * EPS
    assert rel_err(e1, configs.nbody_energy(ta.state, masses, G)) <= 16 * EPS


@pytest.mark.parametrize("fn", "tan tanh sinh cosh asin acos atan asinh acosh atanh erf sigmoid".split())
def test_unary_functions_full_order_step_parity(fn):
    """Order-20 recurrences of the elementary functions beyond the N-body set: one full step and a short
    propagation of a coupled system vs the oracle (the order <= 3 coefficients are pinned by closed forms in
    tests/test_node_jets.py)."""
    n = 37
    rng = np.random.RandomState(sum(map(ord, fn)))
    lo, hi = (1.2, 1.8) if fn == "acosh" else (-0.6, 0.6)
    st = np.stack([rng.uniform(lo, hi, n), rng.uniform(lo, hi, n)])
    x, y = hy.make_vars("x", "y")
    ox, oy = ho.var("x"), ho.var("y")
    c = 1.5 if fn == "acosh" else 0.0
    # Damped coupling keeps the arguments inside the domains of asin/acos/atanh/acosh.
    sys_p = [(x, 0.3 * getattr(hy, fn)(y) - 0.4 * (x - c)), (y, -0.3 * getattr(hy, fn)(x) * hy.cos(y) - 0.4 * (y - c))]
    sys_o = [(ox, 0.3 * getattr(ho, fn)(oy) - 0.4 * (ox - c)), (oy, -0.3 * getattr(ho, fn)(ox) * ho.cos(oy) - 0.4 * (oy - c))]
    ta = hy.taylor_adaptive_batch(sys_p, st, n)
    ora = ho.OracleIntegrator(sys_o, st, n)
    ta.step(write_tc=True)
    ora.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in ora.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = ora.tc.reshape(2, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(2, 21, n) - tc_o) / scale) <= 1e6 * EPS
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e5 * EPS
    ta.propagate_until(0.8)
    ora.propagate_until(0.8)
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e6 * EPS


@pytest.mark.parametrize("mode", ["unrolled", "table"])
def test_atan2_kepE_full_order_step_parity(mode):
    """atan2 and kepE in all the argument combinations (variable / number / parameter), order 20: one full step and a
    propagation vs the oracle, in the unrolled and in the interpreted (table) stepper. The order <= 3 coefficients
    are pinned by closed forms in tests/test_node_jets.py; the Kepler solver is the restatement of
    llvm_add_inv_kep_E() (src/detail/llvm_helpers_celmec.cpp:181-466) on both sides (device function hy_kepE in the
    generated module vs oracle/taylor_oracle.c)."""
    import os

    def build(m, x, y, t, par):
        return [
            (x, 0.3 * m.atan2(y, 1.0 + x * x) - 0.4 * x + 0.1 * m.sin(m.kepE(0.3 + 0.2 * m.sin(y), 2.0 * x + t))
             + 0.1 * m.atan2(y, 1.5) + 0.1 * m.atan2(par, x + 2.0)),
            (y, -0.3 * m.atan2(x, y + 3.0) - 0.4 * y + 0.2 * m.cos(m.kepE(0.6, y)) - 0.2 * m.kepE(0.1 + 0.05 * m.cos(x), 0.7)),
        ]

    n = 45
    rng = np.random.RandomState(77)
    st = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.8, 0.8, n)])
    pars = rng.uniform(0.2, 0.5, (1, n))
    x, y = hy.make_vars("x", "y")
    os.environ["HEYOKA_AMD_EMIT_MODE"] = mode
    try:
        ta = hy.taylor_adaptive_batch(build(hy, x, y, hy.time, hy.par[0]), st, n, pars=pars)
    finally:
        del os.environ["HEYOKA_AMD_EMIT_MODE"]
    assert ta.hip_source_mode.startswith(mode)
    ora = ho.OracleIntegrator(build(ho, ho.var("x"), ho.var("y"), ho.func("time", []), ho.par(0)), st, n, pars=pars)
    ta.step(write_tc=True)
    ora.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in ora.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = ora.tc.reshape(2, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(2, 21, n) - tc_o) / scale) <= 1e6 * EPS
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e5 * EPS
    ta.propagate_until(2.5)
    ora.propagate_until(2.5)
    assert [r[3] for r in ta.propagate_res] == [r[3] for r in ora.prop_res]
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e6 * EPS

    # The Kepler solver itself: E - e sin E = M mod 2 pi to a few ulps, nan for invalid eccentricities
    # (test/kepE.cpp:296-373 checks the same residual at 1000 eps).
    e, M = hy.make_vars("e", "M")
    cf = hy.cfunc([hy.kepE(e, M)], [e, M])
    ecc = np.concatenate([rng.uniform(0, 0.99, 500), [0.0, 0.999999, 1.0, -0.1, np.nan]])
    Mv = np.concatenate([rng.uniform(-40, 40, 500), [0.3, 1e-9, 0.3, 0.3, 0.3]])
    E = cf(np.stack([ecc, Mv]))[0]
    assert np.all(np.isnan(E[-3:])) and np.all(np.isfinite(E[:-3]))
    ok = slice(0, -3)
    Mr = np.mod(Mv[ok], 2 * np.pi)
    res = E[ok] - ecc[ok] * np.sin(E[ok]) - Mr
    res = np.minimum(np.abs(res), np.abs(np.abs(res) - 2 * np.pi))
    assert np.max(res) <= 1000 * EPS * 2 * np.pi
    E_o = np.array([ho.inv_kep_E(a, b) for a, b in zip(ecc[ok], Mv[ok])])
    assert np.max(np.abs(E[ok] - E_o)) <= 64 * EPS * 2 * np.pi


@pytest.mark.parametrize("mode", ["unrolled", "table"])
def test_piecewise_functions_full_order_step_parity(mode):
    """relu / relup / select / comparisons / logical_and / logical_or at order 20, in the unrolled and in the table
    stepper: one full step (h, Taylor coefficients, state) and a propagation vs the oracle. Lanes whose branch
    conditions flip inside the propagation are part of the test (both sides pick the branch from the order-0 values
    at the beginning of each step, as the reference does)."""
    import os

    from test_decomposition import piecewise_system

    n = 64
    rng = np.random.RandomState(3)
    st = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)])
    pars = rng.uniform(-0.2, 0.2, (1, n))
    x, y = hy.make_vars("x", "y")
    os.environ["HEYOKA_AMD_EMIT_MODE"] = mode
    try:
        ta = hy.taylor_adaptive_batch(piecewise_system(hy, x, y, hy.time, hy.par[0]), st, n, pars=pars)
    finally:
        del os.environ["HEYOKA_AMD_EMIT_MODE"]
    assert ta.hip_source_mode.startswith(mode)
    ora = ho.OracleIntegrator(piecewise_system(ho, ho.var("x"), ho.var("y"), ho.func("time", []), ho.par(0)), st, n, pars=pars)
    ta.step(write_tc=True)
    ora.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in ora.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = ora.tc.reshape(2, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(2, 21, n) - tc_o) / scale) <= 1e6 * EPS
    assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e5 * EPS
    # A propagation across branch switches: a lane whose state lands within rounding of a switching surface may take
    # a different branch on the two sides; all the other lanes must agree.
    ta.propagate_until(1.0)
    ora.propagate_until(1.0)
    err = np.max(np.abs(ta.state - ora.state.reshape(2, n)) / np.maximum(1.0, np.abs(ora.state.reshape(2, n))), axis=0)
    assert np.sum(err > 1e6 * EPS) <= 1


def test_kepE_stark_problem_known_answer_on_gpu(golden):
    """The reference's known answer for an integration through kepE (test/kepE.cpp:194-239, Stark problem in Delaunay
    elements, propagate_until(250)): lane 0 carries the reference's initial state and must reproduce its final state
    to 100 eps (mean anomaly through sin / cos at 1e4 eps); the other lanes are perturbed copies compared with the
    oracle."""
    from test_oracle_golden import stark_delaunay

    g = golden["kepE_stark"]
    n = 24
    L0, G0, H0, E0, g0, h0 = g["init_state_LGH_E_gh"]
    l0 = E0 - np.sqrt(1 - G0 * G0 / (L0 * L0)) * np.sin(E0)
    rng = np.random.RandomState(8)
    st = np.array([L0, G0, H0, l0, g0, h0])[:, None] * (1.0 + 1e-4 * rng.uniform(-1, 1, (6, n)))
    st[:, 0] = [L0, G0, H0, l0, g0, h0]
    ta = hy.taylor_adaptive_batch(stark_delaunay(hy, g["eps"]), st, n)
    ora = ho.OracleIntegrator(stark_delaunay(ho, g["eps"]), st, n)
    ta.propagate_until(g["t_final"])
    ora.propagate_until(g["t_final"])
    assert all(r[0] == OC.time_limit for r in ta.propagate_res)
    s = ta.state
    for got, exp in zip([s[0, 0], s[1, 0], s[2, 0], s[4, 0], s[5, 0]], g["final_L_G_H_g_h"]):
        assert abs(got - exp) <= g["tol_eps"] * EPS * abs(exp)
    fL, fG, fE = s[0, 0], s[1, 0], g["final_E"]
    l_exp = fE - np.sqrt(1 - fG * fG / (fL * fL)) * np.sin(fE)
    assert abs(np.sin(s[3, 0]) - np.sin(l_exp)) <= g["tol_eps_angle_l"] * EPS * abs(np.sin(l_exp))
    assert abs(np.cos(s[3, 0]) - np.cos(l_exp)) <= g["tol_eps_angle_l"] * EPS * abs(np.cos(l_exp))
    so = ora.state.reshape(6, n)
    # Actions and slow angles: 1e4 eps; the fast angle l accumulates the rounding of L over 250 time units.
    assert rel_err(s[[0, 1, 2, 4, 5]], so[[0, 1, 2, 4, 5]]) <= 1e4 * EPS
    assert np.max(np.abs(s[3] - so[3])) <= 1e5 * EPS * np.max(np.abs(so[3]))


def test_propagate_grid_device_loop_with_and_without_callback():
    """The device-resident propagate_grid() loop (step kernel + post-step kernel, no per-lane host work; the only
    implementation: the host transcription of the reference's loop is gone): samples against independent
    propagate_until() calls to the grid times, forward and backward, with max_delta_t, max_steps (step_limit, NaN rows)
    and with a step callback - one invocation per sweep, samples bit-identical to the run without callback, cb_stop."""
    n = 300
    st = configs.outer_ss_state(n, perturb=1e-6, seed=9)
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sysd = hy.model.nbody(6, masses=M, Gconst=G)
    grid = np.outer(np.linspace(0.0, 30.0, 13), np.ones(n)) * (1.0 + 0.01 * np.arange(n) / n)
    ta = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    _, out = ta.propagate_grid(grid, max_delta_t=3.0)
    pr1 = ta.propagate_res
    assert all(r[0] == OC.time_limit for r in pr1) and not np.isnan(out).any()
    # Against independent propagations to three of the grid times.
    tu = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    for k in (4, 9, 12):
        tu.propagate_until(grid[k])
        assert rel_err(out[k], tu.state) <= 1e5 * EPS
    # With a callback: one call per sweep, same samples bit for bit.
    calls = []
    tb = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    _, out_cb = tb.propagate_grid(grid, max_delta_t=3.0, callback=lambda t: calls.append(float(t.time[0])) or True)
    assert np.array_equal(out_cb, out) and tb.propagate_res == pr1 and np.array_equal(tb.state, ta.state)
    assert len(calls) >= max(r[3] for r in pr1) and calls == sorted(calls)
    # (Without a callback the one-lane-per-pair stepper stores the Taylor coefficients only of the steps which reach a grid
    # point - hy_kargs::pad bit 2 -, with one of every step: the coefficients of the LAST step of every lane are the same.)
    assert np.array_equal(np.asarray(tb.tc), np.asarray(ta.tc))
    # Backward, back to the start: energy-level agreement with the initial state.
    _, out_b = ta.propagate_grid(grid[::-1].copy())
    assert all(r[0] == OC.time_limit for r in ta.propagate_res) and rel_err(ta.state, st) <= 1e-9
    assert rel_err(out_b[::-1][4], out[4]) <= 1e-9
    # max_steps: step_limit in every lane, unreached grid points are NaN.
    tc = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    _, out_c = tc.propagate_grid(grid, max_steps=3)
    assert all(r[0] == OC.step_limit for r in tc.propagate_res) and np.isnan(out_c[-1]).all() and not np.isnan(out_c[0]).any()
    # A callback which stops after two sweeps: cb_stop, and the state of the run limited to two iterations.
    n_cb = []
    td = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    _, out_d = td.propagate_grid(grid, callback=lambda t: n_cb.append(1) or len(n_cb) < 2)
    te = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)
    _, out_e = te.propagate_grid(grid, max_steps=2)
    assert all(r[0] == OC.cb_stop for r in td.propagate_res) and len(n_cb) == 2
    assert np.array_equal(td.state, te.state) and np.array_equal(np.nan_to_num(out_d), np.nan_to_num(out_e))
    # A callback altering the time coordinate is rejected.
    tf = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True)

    def bad(t):
        t.time = np.zeros(n)
        return True

    with pytest.raises(RuntimeError, match="alteration of the time coordinate"):
        tf.propagate_grid(grid, callback=bad)


def test_propagate_grid_device_output():
    """hy_tab_propagate_grid_device(): samples written straight to a caller-owned device buffer."""
    import torch

    n = 128
    st = configs.two_body_state(n, perturb=1e-3, seed=4)
    grid = np.linspace(0.0, 6.0, 9)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    _, ref = ta.propagate_grid(grid)
    tb = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    d_out = torch.empty((9, 12, n), dtype=torch.float64, device="cuda")
    tb.propagate_grid_device(grid, d_out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), ref)
    assert tb.propagate_res == ta.propagate_res and np.array_equal(tb.state, ta.state)


# ---- event detection (doc/tut_events.rst; test/taylor_t_event.cpp, taylor_nt_event.cpp) ----
@pytest.mark.parametrize("mode", ["unrolled", "table"])
def test_events_tutorial_known_answers(golden, mode, monkeypatch):
    """The tutorial session of doc/tut_events.rst on the HIP path (both steppers with events: fully unrolled and
    table-driven): non-terminal event times to machine precision, direction filter, two close events in
    chronological order, terminal event with a parameter-toggling callback followed by propagate_grid()."""
    if mode == "table":
        monkeypatch.setenv("HEYOKA_AMD_EMIT_MODE", "table")
    g = golden["events_tutorial"]
    x, v = hy.make_vars("x", "v")
    sys_ = [(x, v), (v, -9.8 * hy.sin(x))]
    times, xs = [], []

    def cb(ta, t, d_sgn, idx):
        times.append(t)
        xs.append(ta.update_d_output(t)[0, 0])

    ta = hy.taylor_adaptive_batch(sys_, [[g["ic"][0]], [g["ic"][1]]], 1, nt_events=[hy.nt_event(v, cb)])
    assert ta.with_events and mode in ta.hip_source_mode
    ta.propagate_until(5.0)
    assert np.max(np.abs(np.array(times) - np.array(g["event_times"]))) <= 8 * EPS
    assert np.max(np.abs(np.array(xs) - np.array(g["x_at_events"]))) <= 1e-15
    assert all(r[0] == OC.time_limit for r in ta.propagate_res)
    times.clear()
    ta = hy.taylor_adaptive_batch(sys_, [[g["ic"][0]], [g["ic"][1]]], 1,
                                  nt_events=[hy.nt_event(v, lambda ta, t, d, i: times.append(t),
                                                         direction=hy.event_direction.positive)])
    ta.propagate_until(5.0)
    assert np.max(np.abs(np.array(times) - np.array(g["event_times_positive_direction"]))) <= 8 * EPS
    log = []
    ta = hy.taylor_adaptive_batch(sys_, [[g["ic"][0]], [g["ic"][1]]], 1, nt_events=[
        hy.nt_event(v, lambda ta, t, d, i: log.append((0, t))),
        hy.nt_event(v * v - 1e-12, lambda ta, t, d, i: log.append((1, t)))])
    ta.propagate_until(5.0)
    seq = g["two_events"]["sequence"]
    assert [e for e, _ in log] == [e for e, _ in seq]
    assert np.max(np.abs(np.array([t for _, t in log]) - np.array([t for _, t in seq]))) <= 2e-11

    # Terminal event toggling the drag coefficient.
    gt = g["terminal_drag_toggle"]

    def toggle(ta, d_sgn, idx):
        p = ta.pars
        ta.pars = np.where(p == 0, 1.0, 0.0)
        return True

    ta = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x) - hy.par[0] * v)], [[gt["ic"][0]], [gt["ic"][1]]], 1,
                                  t_events=[hy.t_event(v, toggle)], pars=[0.0])
    while True:
        ta.step()
        oc, h = ta.step_res[0]
        if oc != OC.success:
            break
    assert int(oc) == gt["first_event_outcome"] and ta.pars[0, 0] == 1.0 and abs(ta.state[1, 0]) <= 1e-15
    assert ta.te_cooldowns[0][0] is not None
    ta.propagate_until(1.0)
    _, out = ta.propagate_grid(np.array(gt["grid"], dtype=float))
    assert ta.time[0] == gt["final_time"]
    assert np.max(np.abs(out[:, :, 0] - np.array(gt["grid_states"]))) <= 1e-13


@pytest.mark.parametrize("contract", [True, False])
def test_events_batch_vs_oracle(contract, monkeypatch):
    """Batch of pendulums with different amplitudes: terminal + non-terminal events, lane by lane against the oracle
    (event times, signs, order, step outcomes, cooldowns, stopping terminal events in propagate_until). contract = False:
    built without FMA contraction and with true quotients, the reference's tolerance on the step size (1e4 eps)."""
    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    xkw = {} if contract else {"exact_division": True, "sum_order": "pairwise"}
    h_tol = 1e6 if contract else 1e4
    n = 7
    amp = np.linspace(0.05, 1.2, n)
    st = np.stack([-amp, np.zeros(n)])
    x, v = hy.make_vars("x", "v")
    ox, ov = ho.var("x"), ho.var("v")
    log_p, log_o = [], []
    te_p, te_o = [], []
    ta = hy.taylor_adaptive_batch(
        [(x, v), (v, -9.8 * hy.sin(x))], st, n,
        nt_events=[hy.nt_event(v, lambda ta, t, d, i: log_p.append((i, 0, t, d))),
                   hy.nt_event(x, lambda ta, t, d, i: log_p.append((i, 1, t, d)), direction=hy.event_direction.negative)],
        t_events=[hy.t_event(x * x + v * v - 1e-3, lambda ta, d, i: te_p.append((i, d)) or True, cooldown=0.05)], **xkw)
    ora = ho.OracleEventIntegrator(
        [(ox, ov), (ov, -9.8 * ho.sin(ox))], st, n,
        nt_events=[ho.nt_event(ov, lambda ta, t, d, i: log_o.append((i, 0, t, d))),
                   ho.nt_event(ox, lambda ta, t, d, i: log_o.append((i, 1, t, d)), direction=ho.DIR_NEGATIVE)],
        t_events=[ho.t_event(ox * ox + ov * ov - 1e-3, lambda ta, d, i: te_o.append((i, d)) or True, cooldown=0.05)])
    for _ in range(12):
        ta.step()
        ora.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= h_tol * EPS
        assert rel_err(ta.state, ora.state.reshape(2, n)) <= 1e5 * EPS
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o] and len(log_p) > 10
    assert np.max(np.abs(np.array([a[2] for a in log_p]) - np.array([a[2] for a in log_o]))) <= 1e-12
    assert te_p == te_o
    cd_p, cd_o = ta.te_cooldowns, ora.te_cooldowns
    for i in range(n):
        assert (cd_p[i][0] is None) == (cd_o[i][0] is None)
    # A stopping terminal event interrupts propagate_until() for the whole batch (outcome = -index - 1).
    stop = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, n, t_events=[hy.t_event(v - 0.3)])
    stop_o = ho.OracleEventIntegrator([(ox, ov), (ov, -9.8 * ho.sin(ox))], st, n, t_events=[ho.t_event(ov - 0.3)])
    stop.propagate_until(10.0)
    stop_o.propagate_until(10.0)
    assert [int(r[0]) for r in stop.propagate_res] == [r[0] for r in stop_o.prop_res]
    assert any(int(r[0]) == -1 for r in stop.propagate_res)
    assert rel_err(stop.state, stop_o.state.reshape(2, n)) <= 1e5 * EPS
    assert np.max(np.abs(stop.time - stop_o.time_hi)) <= 1e-13
    # Exceptions raised by Python callbacks surface after the call.
    def boom(ta, t, d, i):
        raise RuntimeError("boom")

    tb = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, n, nt_events=[hy.nt_event(v, boom)])
    with pytest.raises(RuntimeError, match="boom"):
        tb.step()


def test_events_more_than_sixteen_roots_in_one_step_and_failure_counters():
    """The lists of detected events are sized from the Taylor order and the number of events (round 2: 16 per class and
    lane, which this case overflowed): x' = 1 and three event equations which are polynomials with 7 roots each inside
    ONE step (21 non-terminal events per lane and step), per lane against the oracle; the working list of the root
    isolation has the reference's limit of 250 intervals (src/detail/event_detection.cpp:2082), the bracket solver is
    TOMS 748 (:361-363) on both sides."""
    n = 5
    x, y = hy.make_vars("x", "y")
    ox, oy = ho.var("x"), ho.var("y")
    roots = [[0.05 + 0.13 * i + 0.011 * e for i in range(7)] for e in range(3)]

    def poly(v, one, rts):
        g = one
        for r in rts:
            g = g * (v - r)
        return g

    st = np.stack([np.linspace(0.0, 0.004, n), np.zeros(n)])
    log_p, log_o = [], []
    ta = hy.taylor_adaptive_batch(
        [(x, 1.0 + 0.0 * y), (y, 0.0 * x)], st, n,
        nt_events=[hy.nt_event(poly(x, 1.0 + 0.0 * y, roots[e]), lambda ta, t, d, i, e=e: log_p.append((i, e, t, d)))
                   for e in range(3)])
    ora = ho.OracleEventIntegrator(
        [(ox, 1.0 + 0.0 * oy), (oy, 0.0 * ox)], st, n,
        nt_events=[ho.nt_event(poly(ox, 1.0 + 0.0 * oy, roots[e]), lambda ta, t, d, i, e=e: log_o.append((i, e, t, d)))
                   for e in range(3)])
    ta.step(max_delta_t=[1.0] * n)
    ora.step(max_delta_ts=[1.0] * n)
    assert [h for _, h in ta.step_res] == [1.0] * n == [h for _, h in ora.step_res]
    assert len(log_o) == n * 21
    # Same events in the same (chronological, per lane) order, same directions.
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o]
    tp, to = np.array([a[2] for a in log_p]), np.array([a[2] for a in log_o])
    assert np.max(np.abs(tp - to)) <= 1e-12
    exact = {(i, e, k): roots[e][k] - st[0, i] for i in range(n) for e in range(3) for k in range(7)}
    got = sorted((a[0], a[1], a[2]) for a in log_o)
    assert np.max(np.abs(np.array([g[2] for g in got]) - np.array([exact[k] for k in sorted(exact)]))) <= 1e-11
    assert ta.event_detection_failures == 0


def test_block_mode_nonuniform_masses_and_massless_bodies():
    """Block mode with per-cluster constants (distinct masses -> G*m_j factors in the hy_cst table) and with
    massless particles (clusters of two shapes are not isomorphic -> table-mode fallback), vs the oracle."""
    nb, n = 13, 40
    masses = [1.0 + 0.1 * i for i in range(nb)]
    st = configs.plummer_nbody_state(nb, n, seed=21, jitter=1e-6)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=masses, Gconst=0.9), st, n)
    assert "block" in ta.hip_source_mode and "78 clusters" in ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(nb, masses=masses, Gconst=0.9), st, n)
    for _ in range(2):
        ta.step()
        ora.step()
        assert nbody_err(ta.state, ora.state.reshape(6 * nb, n)) <= 1e5 * EPS
    ta.propagate_until(0.02)
    ora.propagate_until(0.02)
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(ta.state, ora.state.reshape(6 * nb, n)) <= 1e7 * EPS
    # 12 massive + 2 massless bodies: the massive-massive and massive-massless pairs have different shapes.
    nb2 = 14
    m2 = [1.0] * 12
    st2 = configs.plummer_nbody_state(nb2, 16, seed=22, jitter=1e-6)
    tb = hy.taylor_adaptive_batch(hy.model.nbody(nb2, masses=m2), st2, 16)
    ob = ho.OracleIntegrator(ho.nbody(nb2, masses=m2), st2, 16)
    tb.step()
    ob.step()
    assert nbody_err(tb.state, ob.state.reshape(6 * nb2, 16)) <= 1e5 * EPS


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [12, 16])
def test_block_v2_with_distinct_masses_vs_oracle(nb):
    """model::nbody() with DISTINCT masses and more than 64 pairs: the scalings of the pair products are moved out of the
    clusters in the internal program (externalise_scalings()), which takes the system to the v2 cluster phase of block mode
    like its equal-mass sibling (before round 6: the first-generation cluster phase with 150 ... 186 spilled registers, 2.7x
    slower). One step with Taylor coefficients and a short propagation against the oracle on the USER's decomposition."""
    n = 24
    masses = list(1.0 / (1.0 + np.arange(nb)) ** 2 * nb / 4.0)
    st = configs.plummer_nbody_state(nb, n, seed=31, jitter=1e-6)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=masses), st, n)
    assert "v2 cluster phase" in ta.hip_source_mode and "scalings of the pair products moved out" in ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(nb, masses=masses), st, n)
    ta.step(write_tc=True)
    ora.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in ora.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = ora.tc.reshape(6 * nb, ora.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(tc_o.shape) - tc_o) / scale) <= 1e6 * EPS
    assert nbody_err(ta.state, ora.state.reshape(6 * nb, n)) <= 1e5 * EPS
    ta.propagate_until(0.02)
    ora.propagate_until(0.02)
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(ta.state, ora.state.reshape(6 * nb, n)) <= 1e7 * EPS


def _random_system(m, rng, n_var=3, extended=False):
    """The same pseudo-random ODE system through expression module m (product or oracle)."""
    if m is ho:
        vs = [m.var("x%d" % i) for i in range(n_var)]
        par, tm, powf = m.par, m.func("time", []), m.pow_
    else:
        vs = list(m.make_vars(*["x%d" % i for i in range(n_var)]))
        par, tm, powf = (lambda i: m.par[i]), m.time, m.pow
    unary = [m.sin, m.cos, lambda e: m.exp(0.3 * e), lambda e: m.log(2.0 + e * e), m.tanh, m.atan, m.sigmoid,
             lambda e: powf(1.5 + e * e, -1.5), lambda e: powf(e, 2.0), lambda e: m.sqrt(1.0 + e * e), m.sinh, m.erf]
    if extended:
        # Two-argument and piecewise functions (bounded / smooth enough for a step-by-step comparison).
        unary += [lambda e: m.relu(e, 0.1), lambda e: m.relup(e, 0.2) * e]
        binary = [lambda a, b: m.atan2(a, 1.5 + b * b), lambda a, b: m.atan2(0.7, a) * b,
                  lambda a, b: m.sin(m.kepE(0.6 * m.sigmoid(a), b)), lambda a, b: m.cos(m.kepE(0.3, a + b)),
                  lambda a, b: m.select(m.gt(a, b), a, 0.5 * b), lambda a, b: m.select(m.logical_and([m.lt(a, 0.3), m.gte(b, -0.3)]), a * b, a - b),
                  lambda a, b: m.logical_or([m.lte(a, b), m.neq(a, 1.0)]) * a + m.eq(b, 2.0)]

    def leaf():
        k = rng.randint(0, 6)
        if k < 4:
            return vs[rng.randint(0, n_var)]
        if k == 4:
            return par(rng.randint(0, 2))
        return tm

    def tree(depth):
        if depth == 0:
            return leaf()
        k = rng.randint(0, 7)
        if k == 0:
            return tree(depth - 1) + tree(depth - 1)
        if k == 1:
            return tree(depth - 1) - float(rng.randint(1, 4)) * tree(depth - 1)
        if k == 2:
            return tree(depth - 1) * tree(depth - 1)
        if k == 3:
            return tree(depth - 1) / (2.0 + powf(tree(depth - 1), 2.0))
        if k == 4:
            return float(rng.uniform(-1.5, 1.5)) * tree(depth - 1)
        if extended and k == 5:
            return binary[rng.randint(0, len(binary))](tree(depth - 1), tree(depth - 1))
        return unary[rng.randint(0, len(unary))](tree(depth - 1))

    # NOTE: both runtime parameters appear in every system (fixed size of the pars array).
    return [(v, 0.3 * tree(3) - 0.2 * v + 1e-3 * (par(0) - par(1))) for v in vs]


EXT_SEEDS = [1000 + i for i in range(12)]


@pytest.mark.gpu
def test_unrolled_kernel_and_its_tc_variant_take_the_same_steps():
    """The straight-line stepper with register-resident jets and its variant which streams the Taylor coefficients out
    (write_tc) share their arithmetic - the re-derived velocity histories and the one-multiplication x'' = u recursion of
    round 6 included: the same states and step sizes bit for bit; two wavefronts per SIMD on the test-particle system."""
    n = 256
    st = configs.two_body_state(n, perturb=1e-3, seed=5)
    sys_ = hy.model.nbody(2, masses=[1.0, 0.0])
    ta, tb = hy.taylor_adaptive_batch(sys_, st, n), hy.taylor_adaptive_batch(sys_, st, n)
    assert "two wavefronts per SIMD" in ta.hip_source_mode
    for _ in range(5):
        ta.step()
        tb.step(write_tc=True)
        assert [h for _, h in ta.step_res] == [h for _, h in tb.step_res]
        assert np.array_equal(np.asarray(ta.state), np.asarray(tb.state))
    # The coefficients written are those of the step: order 0 = the state before it, order 1 of a position = the velocity.
    tc = np.asarray(tb.tc).reshape(12, tb.order + 1, n)
    assert np.array_equal(tc[6:9, 1, :], tc[9:12, 0, :])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["v5", "v3", "v2", "v2_aliased", "unrolled_two_waves", "staged", "table_hbm", "multi_class",
                                  "block_v2", "block_centres"])
def test_every_stepper_is_deterministic_run_to_run(path, monkeypatch):
    """The same propagation twice on fresh integrators, one per code path, through the device-side work queue (which hands
    the systems to the lanes in an order which differs from run to run): states, times, step counts and extreme step
    sizes are identical bit for bit - no result depends on which lane, workgroup or neighbour a system happened to get."""
    from heyoka_amd import mixed_models as mm

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    kw, t_end = {}, 30.0
    if path in ("v3", "v2"):
        _select_cluster_kernel(monkeypatch, path)
    if path in ("v5", "v3", "v2", "staged", "table_hbm"):
        n = 2048 if path in ("staged", "table_hbm") else 8192
        sys_, st = hy.model.nbody(6, masses=M, Gconst=G), configs.outer_ss_state(n, perturb=1e-3, seed=3)
        kw = dict(high_accuracy=True)
        if path in ("staged", "table_hbm"):
            kw["emitter"] = "table"
            t_end = 5.0
        if path == "table_hbm":
            monkeypatch.setenv("HEYOKA_AMD_TABLE_LDS", "0")
    elif path == "unrolled_two_waves":
        n = 16384
        sys_, st = hy.model.nbody(2, masses=[1.0, 0.0]), configs.two_body_state(n, perturb=1e-3, seed=3)
    elif path == "v2_aliased":
        n = 4096
        sys_ = hy.model.np1body(6, masses=M, Gconst=G)
        full = configs.outer_ss_state(n, perturb=1e-3, seed=3).reshape(6, 6, n)
        st = (full[1:] - full[:1]).reshape(30, n)
    elif path == "block_centres":
        n, t_end = 512, 0.5
        rs = np.random.RandomState(11)
        sys_ = hy.model.fixed_centres(masses=list(rs.uniform(0.5, 1.5, 100) / 100.0), positions=list(rs.uniform(-1.0, 1.0, 300)))
        st = np.concatenate([rs.uniform(1.5, 2.0, (3, n)), rs.uniform(-0.3, 0.3, (3, n))])
    elif path == "multi_class":
        n, t_end = 4096, 2.0
        sys_, st = mm.sine_lattice(hy, 16), mm.sine_lattice_state(16, n, seed=3)
    else:
        n, t_end = 128, 0.02
        sys_, st = hy.model.nbody(64), configs.plummer_nbody_state(64, n, seed=3)
    # (Final times which differ from system to system: the lanes of a wavefront finish at different moments.)
    tf = t_end * np.random.RandomState(1).uniform(0.5, 1.5, n)
    res = []
    for _ in range(2):
        ta = hy.taylor_adaptive_batch(sys_, st, n, **kw)
        want = {"v5": "v5", "v3": "v3", "v2": "v2", "v2_aliased": "aliased", "block_centres": "block", "unrolled_two_waves": "two wavefronts per SIMD", "staged": "staged", "table_hbm": "tape in HBM",
                "multi_class": "classes of clusters", "block_v2": "v2 cluster phase"}[path]
        assert want in ta.hip_source_mode, ta.hip_source_mode
        ta.propagate_until(tf)
        arr = ta.propagate_res_arrays()
        res.append([np.asarray(ta.state).copy(), np.asarray(ta.time).copy()] + [np.asarray(a).copy() for a in arr])
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_staged_table_stepper_is_deterministic_run_to_run():
    """The staged table stepper on the pseudo-random system which exposed run-to-run differences on the GPU (a kernel
    compiled for the occupancy of its LDS tapes with 637 spilled registers: profiles/r06_staged_spill_nondeterminism.log):
    repeated single steps on fresh integrators return the same Taylor coefficients bit for bit, and the occupancy
    attribute leaves room for the table registers of a lane."""
    import os
    import re

    seed, n = 1008, 33
    rs = np.random.RandomState(100 + seed)
    st = rs.uniform(-0.7, 0.7, (3, n))
    pars = rs.uniform(-0.5, 0.5, (2, n))
    t0 = rs.uniform(0.0, 2.0, n)
    os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
    try:
        ref = None
        for _ in range(6):
            ta = hy.taylor_adaptive_batch(_random_system(hy, np.random.RandomState(seed), extended=True), st, n, pars=pars, time=t0)
            assert "staged" in ta.hip_source_mode
            w = int(re.search(r"amdgpu_waves_per_eu\((\d+)\)", ta.hip_source).group(1))
            t_regs = int(re.search(r"(\d+) table registers per lane", ta.hip_source_mode).group(1))
            assert w * (t_regs + 56) <= 512 or w == 1
            ta.step(write_tc=True)
            tc = np.asarray(ta.tc).copy()
            if ref is None:
                ref = tc
            assert np.array_equal(tc, ref)
    finally:
        os.environ.pop("HEYOKA_AMD_EMIT_MODE", None)


@pytest.mark.parametrize("seed,contract", [(sd, True) for sd in [0, 1, 2, 3, 4, 5, 7, 8, 9] + EXT_SEEDS]
                         + [(sd, False) for sd in [0, 1, 2, 3, 5, 8] + EXT_SEEDS[:6]])
def test_random_systems_all_code_paths_vs_oracle(seed, contract, monkeypatch):
    """Pseudo-random ODE systems over the whole function set (shared subexpressions, parameters, explicit time): the
    default code generator and the table-driven one against the oracle - decomposition, one full step with
    Taylor coefficients, a short propagation. contract = False: the same kernels built without FMA contraction and with
    true quotients, at the reference's tolerances (h 1e4 eps, coefficients 1e5 eps: test/two_body_batch.cpp:118-150)."""
    import os

    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    h_tol, tc_tol = (1e6, 1e6) if contract else (1e4, 1e5)

    n = 33
    rs = np.random.RandomState(100 + seed)
    st = rs.uniform(-0.7, 0.7, (3, n))
    pars = rs.uniform(-0.5, 0.5, (2, n))
    t0 = rs.uniform(0.0, 2.0, n)
    ext = seed >= 1000
    sys_o = _random_system(ho, np.random.RandomState(seed), extended=ext)
    for mode in ("default", "table"):
        if mode == "table":
            os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
        try:
            sys_p = _random_system(hy, np.random.RandomState(seed), extended=ext)
            ta = hy.taylor_adaptive_batch(sys_p, st, n, pars=pars, time=t0, exact_division=not contract)
        finally:
            os.environ.pop("HEYOKA_AMD_EMIT_MODE", None)
        assert hy.taylor_decompose_sys(sys_p) == ho.dc_to_strings(ho.taylor_decompose_sys(sys_o))
        ora = ho.OracleIntegrator(sys_o, st, n, pars=pars, time=t0)
        ta.step(write_tc=True)
        ora.step(wtc=True)
        h_g = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= h_tol * EPS
        tc_o = ora.tc.reshape(3, ora.order + 1, n)
        scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
        assert np.max(np.abs(np.asarray(ta.tc).reshape(3, 21, n) - tc_o) / scale) <= tc_tol * EPS
        assert rel_err(ta.state, ora.state.reshape(3, n)) <= 1e5 * EPS
        ta.propagate_for(0.5)
        ora.propagate_for(0.5)
        assert rel_err(ta.state, ora.state.reshape(3, n)) <= 1e6 * EPS


def test_full_size_invariants_baseline_configs():
    """BASELINE.json's full sizes through size-independent properties (no oracle at this scale): energy conservation
    evaluated on the device by compiled functions (test/model_nbody.cpp:112-118: <= 100 eps over the propagation),
    time reversibility (propagate forth and back), every lane ends exactly on the requested time."""
    import torch

    def energy_monitor(sys_, expr, n):
        cf = hy.cfunc([expr], sys_.vars)
        buf = torch.empty(n, dtype=torch.float64, device="cuda")
        return cf, buf

    def run(name, sys_, energy_expr, st, t_final, drift_tol, rev_tol, **kw):
        n = st.shape[1]
        ta = hy.taylor_adaptive_batch(sys_, None, n, **kw)
        view = torch.as_tensor(ta.device_array("state"), device="cuda")
        view.copy_(torch.from_numpy(st))
        torch.cuda.synchronize()
        ta.mark_device_modified()
        cf, e0 = energy_monitor(sys_, energy_expr, n)
        e1 = torch.empty_like(e0)
        cf.eval_device(e0.data_ptr(), ta.device_array("state").ptr, n)
        ta.propagate_until(t_final)
        cf.eval_device(e1.data_ptr(), ta.device_array("state").ptr, n)
        ta.synchronize()
        torch.cuda.synchronize()
        drift = float(((e1 - e0) / e0).abs().max())
        oc = torch.as_tensor(ta.device_array("outcome"), device="cuda")
        assert bool((oc == int(OC.time_limit)).all()), name
        thi = torch.as_tensor(ta.device_array("time_hi"), device="cuda")
        assert bool((thi == t_final).all()), name
        steps = int(torch.as_tensor(ta.device_array("n_steps"), device="cuda").sum())
        assert drift <= drift_tol, (name, drift)
        ta.propagate_until(0.0)
        ta.synchronize()
        back = torch.as_tensor(ta.device_array("state"), device="cuda")
        ref = torch.from_numpy(st).cuda()
        rev = float(((back - ref).abs() / ref.abs().clamp(min=1.0)).max())
        assert rev <= rev_tol, (name, rev)
        return steps

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 1048576
    steps = run("outer_ss", hy.model.nbody(6, masses=M, Gconst=G), hy.model.nbody_energy(6, masses=M, Gconst=G),
                configs.outer_ss_state(n, perturb=1e-12, seed=42), 20.0, 100 * EPS, 1e-11, high_accuracy=True)
    assert steps > 20 * n
    n = 4194304
    run("two_body", hy.model.nbody(2, masses=[1.0, 0.0]), hy.model.nbody_energy(2, masses=[1.0, 1e-300]),
        configs.two_body_state(n, perturb=1e-12, seed=42), 50.0, 1e3 * EPS, 1e-11)
    n = 65536
    run("nbody64", hy.model.nbody(64), hy.model.nbody_energy(64), configs.plummer_nbody_state(64, n, seed=1234 + 42),
        0.03, 1e3 * EPS, 1e-10)


def test_step_callback_protocol_pre_hook_and_callback_sets():
    """kw::callback in full (include/heyoka/step_callback.hpp:46-62, :139-185): a callback object with a pre_hook() method
    - invoked once, after the validation of the arguments and before the first step, allowed to touch the state but not
    the time -, and a list of callbacks (a set: every member runs at every step, the results are and-ed); the same
    objects are handed back. Through hy_tab_propagate_{until,for,grid}_cbs of the C ABI."""
    n = 8
    st = configs.two_body_state(n, perturb=1e-2, seed=3)

    class Cb:
        def __init__(self, stop_after=None):
            self.n_pre, self.n_call, self.stop_after = 0, 0, stop_after

        def pre_hook(self, ta):
            self.n_pre += 1
            self.t_at_pre = ta.time.copy()

        def __call__(self, ta):
            self.n_call += 1
            return self.stop_after is None or self.n_call < self.stop_after

    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    c = Cb()
    _, back = ta.propagate_until(2.0, callback=c)
    assert back is c and c.n_pre == 1 and c.n_call == max(r[3] for r in ta.propagate_res) and np.all(c.t_at_pre == 0.0)
    # A set: three callbacks, the second one stops after 2 calls - all of them have run twice.
    a, b, d = Cb(), Cb(stop_after=2), Cb()
    plain = []
    _, back = ta.propagate_for(50.0, callback=[a, b, d, lambda t: plain.append(1) or True])
    assert [x.n_call for x in (a, b, d)] == [2, 2, 2] and len(plain) == 2 and [x.n_pre for x in (a, b, d)] == [1, 1, 1]
    assert all(r[0] == OC.cb_stop for r in ta.propagate_res) and back[0] is a
    # propagate_grid() runs the hook as well.
    g = Cb()
    t0 = float(ta.time[0])
    grid = np.repeat(np.array([0.0, 0.3, 0.6])[:, None], n, axis=1) + np.asarray(ta.time)[None, :]
    ta.propagate_grid(grid, callback=g)
    assert g.n_pre == 1 and g.n_call >= 1 and np.all(g.t_at_pre == np.asarray(grid[0])) and t0 == grid[0, 0]
    # A pre_hook which moves the time coordinate is rejected; an empty member of a set is an error.
    class Bad(Cb):
        def pre_hook(self, ta):
            ta.time = np.zeros(n)

    with pytest.raises(RuntimeError, match="alteration of the time coordinate"):
        ta.propagate_for(1.0, callback=Bad())
    with pytest.raises(ValueError, match="empty callbacks"):
        ta.propagate_for(1.0, callback=[Cb(), None])
    # A pre_hook may set the state up: the propagation starts from what it left.
    tb = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    tc = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    st2 = configs.two_body_state(n, perturb=1e-2, seed=4)

    class Setup(Cb):
        def pre_hook(self, ta):
            ta.state = st2

    tb.propagate_until(1.0, callback=Setup())
    tc.state = st2
    tc.propagate_until(1.0, callback=lambda t: True)
    assert np.array_equal(tb.state, tc.state)


def test_lockstep_device_loop_against_the_oracle_lockstep_loop():
    """propagate_until() with a callback / continuous output: the device-driven lock-step loop (post-step kernel, three
    counters per sweep; the only implementation - the host transcription of the reference's loop is gone) against the
    oracle's lock-step loop: states, times, outcomes, step counters, one callback invocation per iteration of the batch,
    cb_stop == the state after that many iterations, step_limit."""
    n = 257
    st = configs.two_body_state(n, perturb=1e-2, seed=12)
    tf = 3.0 + 0.01 * np.arange(n)
    osys = ho.nbody(2, masses=[1.0, 0.0])
    calls = []
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    co, _ = ta.propagate_until(tf, callback=lambda t: calls.append(1) or True, max_delta_t=0.4, c_output=True)
    ora = ho.OracleIntegrator(osys, st, n)
    ora.propagate_until(tf, max_delta_t=0.4)
    assert [int(r[0]) for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert [r[3] for r in ta.propagate_res] == [r[3] for r in ora.prop_res]
    assert len(calls) == max(r[3] for r in ora.prop_res) and co.n_steps == len(calls)
    assert np.array_equal(ta.time, tf) and rel_err(ta.state, ora.state.reshape(12, n)) <= 1e4 * EPS
    # Continuous output in the middle of the range against an independent propagation.
    tm = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    tm.propagate_until(1.7)
    assert rel_err(co(1.7), tm.state) <= 1e4 * EPS
    # cb_stop after k invocations == the oracle's loop limited to k iterations (outcome aside).
    k = 5
    cnt = []
    tb = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    tb.propagate_until(-5.0, callback=lambda t: cnt.append(1) or len(cnt) < k)
    orb = ho.OracleIntegrator(osys, st, n)
    orb.propagate_until(-5.0, max_steps=k)
    assert all(r[0] == OC.cb_stop for r in tb.propagate_res) and len(cnt) == k
    assert [r[3] for r in tb.propagate_res] == [r[3] for r in orb.prop_res]
    # (Compared after the same number of STEPS, not at the same time: the step size of a near-circular orbit is only
    # determined to ~1e5 eps, and the state inherits that through the end time.)
    assert rel_err(tb.state, orb.state.reshape(12, n)) <= 1e6 * EPS
    assert np.max(np.abs(tb.time - orb.time_hi)) <= 1e-9
    # max_steps with a callback.
    tc = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)
    tc.propagate_for(50.0, callback=lambda t: True, max_steps=4)
    orc = ho.OracleIntegrator(osys, st, n)
    orc.propagate_for(50.0, max_steps=4)
    assert all(r[0] == OC.step_limit for r in tc.propagate_res)
    assert [(int(r[0]), r[3]) for r in tc.propagate_res] == [(r[0], r[3]) for r in orc.prop_res]
    assert rel_err(tc.state, orc.state.reshape(12, n)) <= 1e6 * EPS
    # A callback altering the time coordinate is rejected.
    te = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n)

    def bad(t):
        t.time = np.zeros(n)
        return True

    with pytest.raises(RuntimeError, match="alteration of the time coordinate"):
        te.propagate_until(1.0, callback=bad)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["two_body_unrolled", "outer_ss_cluster_v5", "outer_ss_cluster_v3", "outer_ss_cluster_v2"])
def test_contraction_off_build_meets_the_reference_tolerances(which, monkeypatch):
    """The stated slack of the parity tests on h (1e6 eps) and on the Taylor coefficients is FMA contraction (allowed
    by the reference too, src/llvm_state.cpp:843-845) and nothing else: the same kernels built with
    -ffp-contract=off (test-only build, HEYOKA_AMD_HIPRTC_FLAGS) meet the reference's own tolerances against the
    strict-IEEE oracle from identical states - h to 1e4 eps, every Taylor coefficient to 1e5 eps of the largest
    coefficient of its order IN ITS LANE (not of the row maximum over the lanes), states to 1e5 eps
    (test/two_body_batch.cpp:118-150)."""
    monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    n = 64
    if which == "two_body_unrolled":
        st = configs.two_body_state(n, perturb=1e-3, seed=21)
        sys_g, sys_o, ha = hy.model.nbody(2, masses=[1.0, 0.0]), ho.nbody(2, masses=[1.0, 0.0]), False
    else:
        M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
        st = configs.outer_ss_state(n, perturb=1e-6, seed=22)
        sys_g, sys_o, ha = hy.model.nbody(6, masses=M, Gconst=G), ho.nbody(6, masses=M, Gconst=G), True
        _select_cluster_kernel(monkeypatch, which[-2:])
    # (Two-body: with true quotients as well - the step size of a near-circular orbit is conditioned like 1e5, and the
    # reciprocal form of the division by the order, within 1 ulp per coefficient, would move it by more than 1e4 eps.)
    ta = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=ha, exact_division=(which == "two_body_unrolled"))
    if which.startswith("outer_ss"):
        assert which[-2:] in ta.hip_source_mode, ta.hip_source_mode
    # (The unrolled generator adds the terms of a convolution in the order of the reference's compact mode by default,
    # kw::sum_order: the oracle of the same flavour. The step size of a near-circular orbit is conditioned like 1e5: the
    # ulp-level differences between the two flavours alone would move it by more than the 1e4 eps asserted here.)
    ora = ho.OracleIntegrator(sys_o, st, n, high_accuracy=ha, compact_mode=(which == "two_body_unrolled"))
    n_eq, p = st.shape[0], ta.order
    for _ in range(4):
        # Identical states at the beginning of every step.
        ta.state = ora.state.reshape(n_eq, n)
        ta.step(write_tc=True)
        ora.step(wtc=True)
        h_g = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e4 * EPS
        tc_g = np.asarray(ta.tc).reshape(n_eq, p + 1, n)
        tc_o = ora.tc.reshape(n_eq, p + 1, n)
        scale = np.max(np.abs(tc_o), axis=0, keepdims=True)  # per order and per lane
        assert np.max(np.abs(tc_g - tc_o) / scale) <= 1e5 * EPS
        assert rel_err(ta.state, ora.state.reshape(n_eq, n)) <= 1e5 * EPS


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", [None, "lockstep"])
def test_reference_batch_semantics(semantics):
    """Default (batch_semantics = "reference", device-resident lanes + fix-ups) and "lockstep" (always the lock-step
    loop): a non-finite lane stops the whole batch at that iteration and max_steps counts iterations of the batch
    (src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467, :1516) - outcomes, step counters and times of every lane are
    those of the oracle's lock-step loop, also for the healthy lanes of a batch with a diverging one. No environment
    variable involved: the switch is a constructor argument (hy_tab_config::batch_semantics, kw::batch_semantics)."""
    x, v = hy.make_vars("x", "v")
    ox, ov = ho.var("x"), ho.var("v")
    st = np.array([[1.0, 0.5, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0]])
    for max_steps, t_end in ((0, 3.0), (7, 3.0), (0, 0.4)):
        ta = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], st, 4, batch_semantics=semantics)
        ora = ho.OracleIntegrator([(ox, ox * ox), (ov, -1.0 * ov)], st, 4)
        ta.propagate_until(t_end, max_steps=max_steps)
        ora.propagate_until(t_end, max_steps=max_steps)
        got, ref = ta.propagate_res, ora.prop_res
        assert [int(r[0]) for r in got] == [int(r[0]) for r in ref]
        assert [int(r[3]) for r in got] == [int(r[3]) for r in ref]
        # (Times of the healthy lanes: in the step which produces the non-finite state the step size of the diverging
        # lane itself depends on the order in which NaNs enter the max() reductions of the selector.)
        ok_l = np.all(np.isfinite(ora.state.reshape(2, 4)), axis=0)
        assert np.allclose(np.asarray(ta.time)[ok_l], ora.time_hi[ok_l], rtol=1e-12, atol=0)
        assert np.all(np.isfinite(ta.state[:, ok_l]))
        assert np.all(ok_l) or not np.all(np.isfinite(ta.state[:, ~ok_l]))
        assert np.allclose(ta.state[:, ok_l], ora.state.reshape(2, 4)[:, ok_l], rtol=1e-10)
    # Every lane reports the step limit (per-batch counter) ...
    ta = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], [[-1.0, -2.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0]], 4,
                                  batch_semantics=semantics)
    ta.propagate_until(50.0, max_steps=3)
    assert all(r[0] == OC.step_limit for r in ta.propagate_res)
    if semantics is None:
        # (The device-side outcome array is brought in line too.)
        import torch

        assert bool(torch.all(torch.as_tensor(ta.device_array("outcome"), device="cuda") == int(OC.step_limit)))
    # ... unlike the per-lane results of the fully asynchronous device-resident path (batch_semantics = "per_lane"): lanes
    # which reach their final time report time_limit, the others the step limit.
    st2 = np.array([[-1.0, -2.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0]])
    tb = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], st2, 4, batch_semantics="per_lane")
    tb.propagate_until([50.0, 50.0, 50.0, 1e-3], max_steps=3)
    oc = [r[0] for r in tb.propagate_res]
    assert oc[3] == OC.time_limit and all(o == OC.step_limit for o in oc[:3])
    tc = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], st2, 4, batch_semantics=semantics)
    tc.propagate_until([50.0, 50.0, 50.0, 1e-3], max_steps=3)
    assert all(r[0] == OC.step_limit for r in tc.propagate_res)
    assert [r[3] for r in tc.propagate_res] == [r[3] for r in tb.propagate_res]
    assert np.array_equal(tc.state, tb.state) and np.array_equal(tc.time, tb.time)


@pytest.mark.gpu
def test_rollback_of_a_batch_with_a_nonfinite_lane_when_the_state_was_set_through_the_setter():
    """Advisor finding (round 3): with the state written through the mutable host pointer (get_state_data() /
    ta.state_data(), which keeps the host mirrors in step with the device after every launch) the rollback of a batch with a diverging lane restored
    the device buffers only - the forced lock-step re-run then uploaded the END state of the rolled-back propagation over
    the snapshot. The healthy lanes must come out as in the oracle's lock-step loop, exactly like with the state passed to
    the constructor."""
    x, v = hy.make_vars("x", "v")
    ox, ov = ho.var("x"), ho.var("v")
    st = np.array([[1.0, 0.5, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0]])
    ref = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], st, 4)
    ref.propagate_until(3.0)
    ta = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], np.zeros((2, 4)), 4)
    ta.state_data()[:] = st  # get_state_data(): a handed-out host pointer, eager synchronisation from here on
    ta.propagate_until(3.0)
    # (The setter by value does not switch the integrator to eager synchronisation - and gives the same results.)
    tv = hy.taylor_adaptive_batch([(x, x * x), (v, -v)], np.zeros((2, 4)), 4)
    tv.state = st
    tv.propagate_until(3.0)
    assert tv.propagate_res == ta.propagate_res and np.array_equal(np.nan_to_num(tv.state), np.nan_to_num(ta.state))
    ora = ho.OracleIntegrator([(ox, ox * ox), (ov, -1.0 * ov)], st, 4)
    ora.propagate_until(3.0)
    assert [int(r[0]) for r in ta.propagate_res] == [int(r[0]) for r in ora.prop_res]
    assert [int(r[3]) for r in ta.propagate_res] == [int(r[3]) for r in ora.prop_res]
    ok_l = np.all(np.isfinite(ora.state.reshape(2, 4)), axis=0)
    assert np.array_equal(np.asarray(ta.time)[ok_l], np.asarray(ref.time)[ok_l])
    assert np.array_equal(ta.state[:, ok_l], ref.state[:, ok_l])
    assert np.allclose(ta.state[:, ok_l], ora.state.reshape(2, 4)[:, ok_l], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("sum_order", ["pairwise", "running"])
def test_unrolled_kernel_is_bit_identical_to_the_oracle_without_contraction(sum_order, monkeypatch):
    """The unrolled generator keeps the reference's operation order inside the convolutions, in either of its two forms
    (kw::sum_order): "pairwise" = the default mode (products first, pairwise sums: src/math/prod.cpp:386-395), "running" =
    the compact mode (running sums from 0: src/math/prod.cpp:686-698; the generator's default - one FMA per term once
    contraction is allowed). With kw::exact_division = true (true quotients instead of the reciprocal forms) and built
    without FMA contraction, its Taylor coefficients of the two-body problem are those of the strict-IEEE oracle of the
    same flavour BIT FOR BIT from identical states - the 1e5 eps of the other parity tests are contraction and the
    reciprocal forms only."""
    monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    n = 64
    st = configs.two_body_state(n, perturb=1e-3, seed=21)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), st, n, exact_division=True, sum_order=sum_order)
    assert ta.hip_source_mode.startswith("unrolled"), ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n, compact_mode=(sum_order == "running"))
    for _ in range(4):
        ta.state = ora.state.reshape(12, n)
        ta.step(write_tc=True)
        ora.step(wtc=True)
        tc_g, tc_o = np.asarray(ta.tc).reshape(12, ta.order + 1, n), ora.tc.reshape(12, ta.order + 1, n)
        assert np.array_equal(tc_g, tc_o), (np.count_nonzero(tc_g != tc_o), np.max(np.abs(tc_g - tc_o) / (np.abs(tc_o) + 1e-300)))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["staged", "tape in HBM"])
def test_compact_mode_has_the_arithmetic_of_the_reference_compact_mode(variant, monkeypatch):
    """kw::compact_mode = true is a flavour of the ARITHMETIC too: running sums inside the convolutions
    (src/math/prod.cpp:686-698, src/math/pow.cpp:905-925, src/detail/sum_sq.cpp:330-345), pairwise sums over the arguments
    of sum() / sum_sq() (src/math/sum.cpp:355). The oracle has the same flavour (OracleIntegrator(compact_mode=True)); built
    without FMA contraction, the table stepper reproduces its Taylor coefficients BIT FOR BIT from identical states (the
    N-body right-hand side contains no library function: sqrt and the divisions are correctly rounded on both sides),
    and differs from the default-mode flavour in the last bits - for both variants of the table kernel."""
    monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    monkeypatch.setenv("HEYOKA_AMD_TABLE_LDS", "1" if variant == "staged" else "0")
    # (kw::compact_mode keeps the on-chip kernels where the planner can shape the decomposition - see
    # test_compact_mode_keeps_the_on_chip_kernels -: the table steppers are selected explicitly here.)
    monkeypatch.setenv("HEYOKA_AMD_EMIT_MODE", "table")
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 64
    for sys_g, sys_o, st, ha in (
        (hy.model.nbody(6, masses=M, Gconst=G), ho.nbody(6, masses=M, Gconst=G), configs.outer_ss_state(n, perturb=1e-6, seed=2), True),
        (hy.model.nbody(4), ho.nbody(4), configs.plummer_nbody_state(4, n, seed=5), False),
    ):
        ta = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=ha, compact_mode=True)
        assert ta.hip_source_mode.startswith("table") and variant in ta.hip_source_mode, ta.hip_source_mode
        oc = ho.OracleIntegrator(sys_o, st, n, high_accuracy=ha, compact_mode=True)
        od = ho.OracleIntegrator(sys_o, st, n, high_accuracy=ha)
        n_eq, p = st.shape[0], ta.order
        diff_default = 0
        for _ in range(3):
            ta.state = oc.state.reshape(n_eq, n)
            od.state[:] = oc.state
            ta.step(write_tc=True)
            oc.step(wtc=True)
            od.step(wtc=True)
            tc_g = np.asarray(ta.tc).reshape(n_eq, p + 1, n)
            assert np.array_equal(tc_g, oc.tc.reshape(n_eq, p + 1, n))
            diff_default += np.count_nonzero(tc_g != od.tc.reshape(n_eq, p + 1, n))
            # (The step size goes through exp(log()) on the device and pow() in the oracle: a few ulps.)
            h_g = np.array([h for _, h in ta.step_res])
            h_o = np.array([h for _, h in oc.step_res])
            assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 16 * EPS
        assert diff_default > 0


@pytest.mark.gpu
def test_compact_mode_keeps_the_on_chip_kernels():
    """kw::compact_mode = true (include/heyoka/kw.hpp; benchmark/outer_ss_long_term_batch.cpp:113 exposes it on the headline
    workload) is a code-size knob in the reference which changes the order of the additions inside the convolutions
    (running sums, src/math/prod.cpp:686-698) and costs little at run time. Here it keeps the wave-cluster kernel of the
    outer Solar System (its convolutions are FMA chains - running sums - already), keeps straight-line code only for small
    decompositions - with the compact order of the additions -, and sends everything else to the table steppers. Parity
    against the COMPACT-mode oracle: one step (h 1e6 eps, Taylor coefficients 1e6 eps of the row maximum over the
    ensemble) and a propagation of ~80 steps (identical outcomes / end times, step counts within one step, states
    1e6 eps of the row maximum)."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 256
    st = configs.outer_ss_state(n, perturb=1e-6, seed=9)
    sys_g, sys_o = hy.model.nbody(6, masses=M, Gconst=G), ho.nbody(6, masses=M, Gconst=G)
    a = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=True, compact_mode=True)
    b = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=True)
    assert a.compact_mode and not b.compact_mode
    assert a.hip_source_mode.startswith("cluster") and "v5" in a.hip_source_mode, a.hip_source_mode
    oc = ho.OracleIntegrator(sys_o, st, n, high_accuracy=True, compact_mode=True)
    a.step(write_tc=True)
    oc.step(wtc=True)
    h_g, h_o = np.array([h for _, h in a.step_res]), np.array([h for _, h in oc.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = oc.tc.reshape(36, a.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(a.tc).reshape(tc_o.shape) - tc_o) / scale) <= 1e6 * EPS
    a.propagate_until(35.0)
    oc.propagate_until(35.0)
    assert all(r[0] == OC.time_limit for r in a.propagate_res)
    assert max(abs(x[3] - y[3]) for x, y in zip(a.propagate_res, oc.prop_res)) <= 1
    assert np.array_equal(np.asarray(a.time), oc.time_hi)
    assert row_rel_err(a.state, oc.state.reshape(36, n)) <= 1e6 * EPS

    # A small decomposition: straight-line code with the compact order of the additions (kw::sum_order = running), bit
    # for bit the kernel the default mode builds with that order - and the oracle's compact flavour to the usual
    # tolerances after a propagation.
    n = 32
    st = configs.two_body_state(n, perturb=1e-3, seed=5)
    sys_g = hy.model.nbody(2, masses=[1.0, 0.0])
    a = hy.taylor_adaptive_batch(sys_g, st, n, compact_mode=True)
    b = hy.taylor_adaptive_batch(sys_g, st, n, sum_order="running")
    assert a.hip_source_mode.startswith("unrolled") and a.hip_source == b.hip_source
    ora = ho.OracleIntegrator(ho.nbody(2, masses=[1.0, 0.0]), st, n, compact_mode=True)
    a.propagate_until(7.0)
    ora.propagate_until(7.0)
    assert row_rel_err(a.state, ora.state.reshape(12, n)) <= 1e5 * EPS

    # A decomposition the planners cannot shape and which is not small: the table steppers (staged: tape in LDS).
    x, y, z = hy.make_vars("x", "y", "z")
    big = [(x, y * z + hy.sin(x) * hy.cos(y) + hy.exp(-x * x) * z + hy.log(1.5 + y * y) - x),
           (y, z * hy.sin(x * y) - y * hy.cos(z) + hy.pow(1.2 + x * x + y * y, -1.5) * x),
           (z, x * y - z * hy.sqrt(1.0 + z * z) + hy.sin(z) * hy.exp(-y * y))]
    assert len(hy.taylor_decompose_sys(big)) - 6 > 40
    c = hy.taylor_adaptive_batch(big, None, 64, compact_mode=True)
    d = hy.taylor_adaptive_batch(big, None, 64)
    assert c.hip_source_mode.startswith("table") and "staged" in c.hip_source_mode, c.hip_source_mode
    assert d.hip_source_mode.startswith("unrolled"), d.hip_source_mode


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["unrolled", "table"])
def test_event_equations_take_part_in_the_step_size_selector(mode, monkeypatch):
    """taylor_determine_h() of the stepper with events iterates over the state variables AND the event equations
    (src/taylor_00.cpp:209-219, :693): with a fast-oscillating event function g = sin(40 x) the step must stay inside the
    convergence radius of g's Taylor series - the step sizes are those of the oracle (which follows the reference) and
    about 40 times smaller than those of the same system without the event; every zero crossing is reported."""
    if mode == "table":
        monkeypatch.setenv("HEYOKA_AMD_EMIT_MODE", "table")
    n = 4
    st = np.stack([np.linspace(0.1, 0.4, n), np.linspace(1.0, 1.3, n)])
    x, v = hy.make_vars("x", "v")
    ox, ov = ho.var("x"), ho.var("v")
    log_p, log_o = [], []
    ta = hy.taylor_adaptive_batch([(x, v), (v, -x)], st, n,
                                  nt_events=[hy.nt_event(hy.sin(40.0 * x), lambda ta, t, d, i: log_p.append((i, t, d)))])
    ora = ho.OracleEventIntegrator([(ox, ov), (ov, -1.0 * ox)], st, n,
                                   nt_events=[ho.nt_event(ho.sin(40.0 * ox), lambda ta, t, d, i: log_o.append((i, t, d)))])
    plain = hy.taylor_adaptive_batch([(x, v), (v, -x)], st, n)
    plain.step()
    h_plain = np.array([h for _, h in plain.step_res])
    for _ in range(30):
        ta.step()
        ora.step()
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert np.all(h_p < 0.2 * h_plain)
    assert len(log_p) == len(log_o) and len(log_p) >= 8
    assert [(a[0], a[2]) for a in log_p] == [(a[0], a[2]) for a in log_o]
    assert np.max(np.abs(np.array([a[1] for a in log_p]) - np.array([a[1] for a in log_o]))) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,contract", [("v5", True), ("v5", False), ("v3", True), ("v3", False), ("v2", True)])
def test_bench_length_parity_on_4096_systems(kernel, contract, monkeypatch):
    """The headline kernels over one bench-sized launch (propagate_until(60 yr), ~80 Taylor steps per system) on 4 096
    perturbed outer Solar Systems against the oracle's ensemble driver, lane by lane: identical outcomes, step counts
    within +-1 (and equal for all but a handful of lanes), final states, and the smallest / largest step of every lane.
    Default build: 1e6 eps on the states (test/taylor_adaptive_batch.cpp:105-146 style), 1e-6 on the step sizes; built with
    -ffp-contract=off (the oracle's arithmetic): 1e5 eps (test/two_body_batch.cpp:118-150)."""
    _select_cluster_kernel(monkeypatch, kernel)
    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 4096
    st = configs.outer_ss_state(n, perturb=1e-6, seed=77)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    assert kernel in ta.hip_source_mode, ta.hip_source_mode
    ta.propagate_until(60.0)
    ref, thi, tlo, oc, mn, mx, ns, total = ho.ensemble_propagate_until(ho.nbody(6, masses=M, Gconst=G), st, n, 8, 60.0,
                                                                       high_accuracy=True)
    oc_g, mn_g, mx_g, ns_g = ta.propagate_res_arrays()
    assert np.array_equal(np.asarray(oc_g, dtype=np.int64), oc) and np.all(oc == int(OC.time_limit))
    assert np.array_equal(ta.time, thi) and np.all(thi == 60.0)
    dn = np.abs(np.asarray(ns_g, dtype=np.int64) - ns)
    assert dn.max() <= 1 and np.count_nonzero(dn) <= n // 100, (dn.max(), np.count_nonzero(dn))
    assert 70 <= ns.mean() <= 95
    tol = 1e6 if contract else 1e5
    assert nbody_err(ta.state, ref.reshape(36, n)) <= tol * EPS
    same = dn == 0
    assert np.max(np.abs(np.asarray(mn_g)[same] - mn[same]) / mn[same]) <= 1e-6
    assert np.max(np.abs(np.asarray(mx_g)[same] - mx[same]) / mx[same]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("contract", [True, False], ids=["default-build", "no-contraction"])
def test_bench_length_parity_nbody64_block_v2(contract, monkeypatch):
    """Config 5's benchmarked kernel (block mode, v2 cluster phase) over a bench-sized launch: 768 Plummer spheres of 64
    bodies - three times the workgroups a launch keeps in flight, so that the work queue hands every workgroup several
    systems - through propagate_until(0.055) (~34 Taylor steps per system, what a bench launch runs), lane by lane against
    the oracle's ensemble driver: identical outcomes and end times, step counts within +-1 on at most 1 % of the systems,
    states to 1e6 eps (1e5 eps built with -ffp-contract=off, the oracle's arithmetic), smallest / largest steps to 1e-6.
    (Tolerances: test/two_body_batch.cpp:118-150; the tape: src/taylor_02.cpp:1194-1260.)"""
    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    n, t_final = 768, 0.055
    st = configs.plummer_nbody_state(64, n, seed=77, jitter=1e-6)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(64), st, n)
    assert "block" in ta.hip_source_mode and "v2 cluster phase" in ta.hip_source_mode, ta.hip_source_mode
    ta.propagate_until(t_final)
    ref, thi, tlo, oc, mn, mx, ns, total = ho.ensemble_propagate_until(ho.nbody(64), st, n, 8, t_final)
    oc_g, mn_g, mx_g, ns_g = ta.propagate_res_arrays()
    assert np.array_equal(np.asarray(oc_g, dtype=np.int64), oc) and np.all(oc == int(OC.time_limit))
    assert np.array_equal(ta.time, thi) and np.all(thi == t_final)
    dn = np.abs(np.asarray(ns_g, dtype=np.int64) - ns)
    assert dn.max() <= 1 and np.count_nonzero(dn) <= n // 100, (dn.max(), np.count_nonzero(dn))
    assert 25 <= ns.mean() <= 45, ns.mean()
    assert nbody_err(ta.state, ref.reshape(384, n)) <= (1e6 if contract else 1e5) * EPS
    same = dn == 0
    assert np.max(np.abs(np.asarray(mn_g)[same] - mn[same]) / mn[same]) <= 1e-6
    assert np.max(np.abs(np.asarray(mx_g)[same] - mx[same]) / mx[same]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["v5", "v3", "v2"])
def test_cluster_kernels_loop_control_semantics(kernel, monkeypatch):
    """The wave-uniform step loop of the cluster kernels (a finished system keeps taking zero-length steps with frozen
    bookkeeping until the other systems of its wavefront are done): per-lane final times forward and backward with a
    max_delta_t clamp, max_steps, zero-length propagation, the raw step - against the oracle, for the one-lane-per-pair
    kernel (v5, the default), the lane-pair kernel (v3) and the pipelined one-lane-per-cluster kernel (v2)."""
    _select_cluster_kernel(monkeypatch, kernel)
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 22  # ragged: the last wavefront / workgroup holds replicas
    rng = np.random.RandomState(31)
    st = configs.outer_ss_state(n, perturb=1e-7, seed=23)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    assert kernel in ta.hip_source_mode
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    # Very different final times inside one wavefront (systems finish after 0 ... ~40 steps), both directions.
    tf = rng.uniform(-30.0, 30.0, n)
    tf[3] = 0.0
    ta.propagate_until(tf, max_delta_t=0.9)
    ora.propagate_until(tf, max_delta_t=0.9)
    assert [int(r[0]) for r in ta.propagate_res] == [int(r[0]) for r in ora.prop_res]
    assert max(abs(int(a[3]) - int(b[3])) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert np.array_equal(ta.time, tf)
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
    mn_g = np.array([r[1] for r in ta.propagate_res])[tf != 0]
    mn_o = np.array([r[1] for r in ora.prop_res])[tf != 0]
    assert np.max(np.abs(mn_g - mn_o) / mn_o) <= 1e-6
    # max_steps: every lane which is not done reports the step limit after exactly 3 steps (per-lane counter).
    ta.propagate_until(tf + 200.0, max_steps=3)
    assert all(r[0] == OC.step_limit and r[3] == 3 for r in ta.propagate_res)
    # Zero-length propagation.
    ta.propagate_for(0.0)
    assert all(r[0] == OC.time_limit and r[3] == 0 for r in ta.propagate_res)
    # Single steps with per-lane limits (some negative, one zero).
    ora2 = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), ta.state, n, high_accuracy=True, time=ta.time)
    lims = rng.uniform(-0.5, 0.5, n)
    lims[5] = 0.0
    ta.step(lims)
    ora2.step(max_delta_ts=lims)
    assert [int(o) for o, _ in ta.step_res] == [int(o) for o, _ in ora2.step_res]
    assert np.allclose([h for _, h in ta.step_res], [h for _, h in ora2.step_res], rtol=1e-9, atol=0)
    assert nbody_err(ta.state, ora2.state.reshape(36, n)) <= 1e5 * EPS


@pytest.mark.gpu
def test_per_system_refill_of_the_one_lane_per_pair_stepper(monkeypatch):
    """A finished system of the one-lane-per-pair stepper (v5) is retired on the spot and its 16 lanes take the next system
    of the work queue while the other three systems of the wavefront keep stepping (hip_emit_cluster2.cpp, "refill"; the
    reference's lanes idle until the slowest lane of the batch is done, src/taylor_adaptive_batch.cpp:1378-1460). With more
    systems than resident slots (8 192) and very different per-lane final times every slot is refilled several times, at
    different steps of its neighbours. Independent systems: the results must not depend on the schedule - bit-identical to the
    same kernel without refill (whole groups of four from the queue) - and agree with the oracle on a sample of lanes drawn from
    the whole queue."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 40000 + 3  # ragged tail
    rng = np.random.RandomState(5)
    st = configs.outer_ss_state(n, perturb=1e-3, seed=91)
    tf = rng.uniform(2.0, 45.0, n)
    tf[rng.randint(0, n, 50)] = 0.0  # systems which are done before their first step
    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True)
    assert "v5" in ta.hip_source_mode and "u64 snew" in ta.hip_source
    ta.propagate_until(tf)
    monkeypatch.setenv("HEYOKA_AMD_NO_REFILL", "1")
    tb = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True)
    monkeypatch.delenv("HEYOKA_AMD_NO_REFILL")
    assert "u64 snew" not in tb.hip_source
    tb.propagate_until(tf)
    assert np.array_equal(ta.state, tb.state) and np.array_equal(ta.time, tb.time) and np.array_equal(ta.time, tf)
    ra, rb = ta.propagate_res_arrays(), tb.propagate_res_arrays()
    for a, b in zip(ra, rb):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert np.all(np.asarray(ra[0]) == int(OC.time_limit))
    ns = np.asarray(ra[3])
    assert ns.max() >= 4 * max(1, int(np.median(ns))) // 3 and ns.min() == 0
    # The oracle on 512 lanes spread over the whole queue (first groups, refilled slots, the ragged tail).
    idx = np.unique(np.concatenate([np.arange(16), rng.randint(0, n, 480), np.arange(n - 16, n)]))
    m = len(idx)
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st.reshape(36, n)[:, idx].copy(), m, high_accuracy=True)
    ora.propagate_until(tf[idx])
    assert [int(r[0]) for r in ora.prop_res] == [int(OC.time_limit)] * m
    dn = np.abs(ns[idx].astype(np.int64) - np.array([int(r[3]) for r in ora.prop_res]))
    assert dn.max() <= 1 and np.count_nonzero(dn) <= max(2, m // 50)
    assert nbody_err(ta.state[:, idx], ora.state.reshape(36, m)) <= 1e6 * EPS
    # A second call reuses the queue from the start.
    ta.propagate_until(tf + 3.0)
    tb.propagate_until(tf + 3.0)
    assert np.array_equal(ta.state, tb.state)


def _outer_ss_event_setup(m, log, te_log):
    """Events on the outer Solar System through expression module m (product or oracle): radial velocity of Jupiter
    (non-terminal, both directions), Saturn crossing y = 0 upwards (non-terminal), Jupiter - Saturn distance falling
    below 9 AU (terminal, callback keeps going)."""
    mk = (lambda s: m.var(s)) if m is ho else (lambda s: m.make_vars(s)[0] if isinstance(m.make_vars(s), (list, tuple)) else m.make_vars(s))
    x1, y1, z1, vx1, vy1, vz1 = [mk(s + "_1") for s in ("x", "y", "z", "vx", "vy", "vz")]
    x2, y2, z2 = [mk(s + "_2") for s in ("x", "y", "z")]
    pos = m.DIR_POSITIVE if m is ho else m.event_direction.positive
    neg = m.DIR_NEGATIVE if m is ho else m.event_direction.negative
    nt = [m.nt_event(x1 * vx1 + y1 * vy1 + z1 * vz1, lambda ta, t, d, i: log.append((i, 0, t, d))),
          m.nt_event(y2, lambda ta, t, d, i: log.append((i, 1, t, d)), direction=pos)]
    d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 81.0
    te = [m.t_event(d2, lambda ta, d, i: te_log.append((i, d)) or True, direction=neg)]
    return nt, te


@pytest.mark.gpu
def test_event_equations_inside_the_stepper_vs_oracle_and_vs_the_event_jet_kernel(monkeypatch):
    """Integrators WITHOUT terminal events on the one-lane-per-pair stepper: the stepper evaluates the event equations itself
    (jets of the state variables from LDS), extends the selector norms to them, takes the final step size and updates the
    state - the reference's step_e (src/taylor_00.cpp:592-710) in one kernel; hy_ev_jets and the dense-output pass are not
    launched. Against the oracle's stepper with events, step by step, and against the same integrator with the event
    equations in their own kernel (HEYOKA_AMD_NO_EVENTS_IN_STEPPER=1): outcomes, step sizes, states, events in order, the
    Taylor coefficients / dense output of the last step (still written by every step), lock-step propagation and grid. An
    integrator with a terminal event keeps the two-kernel path (its steps can be truncated)."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 70  # ragged: three wavefronts, the last one partly filled
    st = configs.outer_ss_state(n, perturb=1e-3, seed=12)
    logs = {"p": [], "o": [], "q": []}

    def events(m, key):
        # Saturn crossing y = 0 upwards, Jupiter overtaking Saturn in x, their distance falling below 9 AU: two linear event
        # equations and one with three products (the budget of the stepper: three nonlinear nodes).
        log = logs[key]
        mk = (lambda s_: m.var(s_)) if m is ho else (lambda s_: m.make_vars(s_, "dummy__")[0])
        x1, y1, z1, x2, y2, z2 = [mk(s_) for s_ in ("x_1", "y_1", "z_1", "x_2", "y_2", "z_2")]
        d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 81.0
        pos = m.DIR_POSITIVE if m is ho else m.event_direction.positive
        neg = m.DIR_NEGATIVE if m is ho else m.event_direction.negative
        return [m.nt_event(y2, lambda ta, t, d, i: log.append((i, 0, t, d)), direction=pos),
                m.nt_event(x1 - x2, lambda ta, t, d, i: log.append((i, 1, t, d))),
                m.nt_event(d2, lambda ta, t, d, i: log.append((i, 2, t, d)), direction=neg)]

    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=events(hy, "p"))
    assert "v5" in ta.hip_source_mode and "inside the stepper" in ta.hip_source_mode, ta.hip_source_mode
    monkeypatch.setenv("HEYOKA_AMD_NO_EVENTS_IN_STEPPER", "1")
    tq = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=events(hy, "q"))
    monkeypatch.delenv("HEYOKA_AMD_NO_EVENTS_IN_STEPPER")
    assert "v5" in tq.hip_source_mode and "inside the stepper" not in tq.hip_source_mode
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=events(ho, "o"))
    for it in range(40):
        ta.step()
        ora.step()
        tq.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        assert [int(oc) for oc, _ in ta.step_res] == [int(oc) for oc, _ in tq.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        h_q = np.array([h for _, h in tq.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert np.max(np.abs(h_p - h_q) / np.abs(h_q)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
        assert nbody_err(ta.state, np.asarray(tq.state)) <= 1e6 * EPS
        assert np.max(np.abs(ta.time - ora.time_hi)) <= 1e-11
        if it in (0, 17):
            # The Taylor coefficients of the step which was just taken: dense output half-way back.
            tm = ta.time - 0.5 * h_p
            assert rel_err(np.asarray(ta.update_d_output(tm)), np.asarray(tq.update_d_output(tm))) <= 1e6 * EPS
            assert rel_err(np.asarray(ta.tc), np.asarray(tq.tc)) <= 1e6 * EPS
    assert len(logs["p"]) >= n
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["q"]]
    assert np.max(np.abs(np.array([a[2] for a in logs["p"]]) - np.array([a[2] for a in logs["o"]]))) <= 1e-10
    t_end = float(np.max(ora.time_hi)) + 10.0
    ta.propagate_until(t_end)
    ora.propagate_until(t_end)
    tq.propagate_until(t_end)
    assert [int(r[0]) for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e7 * EPS
    assert nbody_err(ta.state, np.asarray(tq.state)) <= 1e7 * EPS
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]
    grid = np.repeat(t_end + np.array([0.0, 1.5, 3.0, 7.0])[:, None], n, axis=1)
    _, out_p = ta.propagate_grid(grid)
    _, out_q = tq.propagate_grid(grid)
    assert rel_err(np.asarray(out_p), np.asarray(out_q)) <= 1e7 * EPS
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["q"]]
    # Taylor coefficients on demand: the stepper stores them only for workgroups in which an event may have happened and the
    # integrator regenerates the rest from its snapshot of the state before the step. (a) The regeneration stores nothing
    # but the coefficients: a state which was set after the step survives it, and the coefficients are those of the step;
    # (b) continuous output (every step's coefficients are consumed: stored by every step).
    ta.step()
    tq.step()
    mod = ta.state * (1.0 + 1e-3)
    ta.state = mod
    assert rel_err(np.asarray(ta.tc), np.asarray(tq.tc)) <= 1e6 * EPS
    assert np.array_equal(ta.state, mod)
    assert rel_err(np.asarray(ta.tc)[:, 0, :], np.asarray(tq.tc)[:, 0, :]) <= 1e6 * EPS
    ta.state = tq.state
    t_c = float(np.max(ta.time)) + 6.0
    co_p, _ = ta.propagate_until(t_c, c_output=True)
    co_q, _ = tq.propagate_until(t_c, c_output=True)
    for frac in (0.1, 0.5, 0.9):
        tm = t_c - 6.0 * frac
        assert rel_err(np.asarray(co_p(tm)), np.asarray(co_q(tm))) <= 1e7 * EPS
    # Radial velocity + squared distance: each a sum of three terms of one shape, evaluated side by side by three lanes of
    # the system - both fit the budget (two convolutions instead of six). Beyond the budget (two more products of unrelated
    # shape on top): the three-kernel path.
    nt_t, te_t = _outer_ss_event_setup(hy, [], [])
    tt = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=nt_t + events(hy, "q")[2:])
    assert "v5" in tt.hip_source_mode and "inside the stepper" in tt.hip_source_mode
    xa, ya, vxa = hy.make_vars("x_3", "y_3", "vx_3")
    more = [hy.nt_event(xa * ya - 1.0, lambda *a: None), hy.nt_event(xa * vxa * ya, lambda *a: None)]
    tt = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=nt_t + events(hy, "q")[2:] + more)
    assert "v5" in tt.hip_source_mode and "inside the stepper" not in tt.hip_source_mode


@pytest.mark.gpu
def test_close_encounter_events_from_the_pair_lanes_vs_oracle(monkeypatch):
    """An event equation |r_i - r_j|^2 - R^2 is the squared distance whose Taylor coefficients the lane of the pair (i, j) of
    the one-lane-per-pair stepper holds anyway (the history of its pow recurrence): such events are not evaluated, the lane
    contributes its history (order p from one more convolution), tests ITS event and stores ITS row - all pairs at the same
    time. Sun - Jupiter, Jupiter - Saturn (twice: two radii, the second one goes through the generic statements) and
    Saturn - Uranus distances next to a linear event, against the oracle step by step and against the same integrator with the
    events evaluated by the generic statements (HEYOKA_AMD_NO_PAIR_EVENTS=1); a terminal close-encounter event."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 70
    st = configs.outer_ss_state(n, perturb=1e-3, seed=15)
    logs = {"p": [], "o": [], "q": []}

    def events(m, key, terminal=False):
        log = logs[key]
        mk = (lambda s_: m.var(s_)) if m is ho else (lambda s_: m.make_vars(s_, "dummy__")[0])

        def d2(a, b, r2):
            pa = [mk("%s_%d" % (c, a)) for c in "xyz"]
            pb = [mk("%s_%d" % (c, b)) for c in "xyz"]
            return (pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1]) + (pa[2] - pb[2]) * (pa[2] - pb[2]) - r2

        neg = m.DIR_NEGATIVE if m is ho else m.event_direction.negative
        nt = [m.nt_event(d2(0, 1, 27.0), lambda ta, t, d, i: log.append((i, 0, t, d))),
              m.nt_event(d2(2, 1, 81.0), lambda ta, t, d, i: log.append((i, 1, t, d)), direction=neg),
              m.nt_event(mk("y_2"), lambda ta, t, d, i: log.append((i, 2, t, d))),
              m.nt_event(d2(1, 2, 100.0), lambda ta, t, d, i: log.append((i, 3, t, d))),
              m.nt_event(d2(2, 3, 160.0), lambda ta, t, d, i: log.append((i, 4, t, d))),
              # (R^2 - d^2: the squares are subtracted - the lane switch carries the sign.)
              m.nt_event(900.0 - d2(4, 5, 0.0), lambda ta, t, d, i: log.append((i, 5, t, d)))]
        te = [m.t_event(d2(1, 2, 64.0), lambda ta, d, i: log.append((i, 9, 0.0, d)) or True, direction=neg)] if terminal else []
        return nt, te

    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    nt_p, _ = events(hy, "p")
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=nt_p)
    assert "inside the stepper" in ta.hip_source_mode and "pe_on" in ta.hip_source, ta.hip_source_mode
    monkeypatch.setenv("HEYOKA_AMD_NO_PAIR_EVENTS", "1")
    monkeypatch.setenv("HEYOKA_AMD_EV_INLINE_MAX_NONLINEAR", "8")
    nt_q, _ = events(hy, "q")
    tq = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=nt_q)
    monkeypatch.delenv("HEYOKA_AMD_NO_PAIR_EVENTS")
    monkeypatch.delenv("HEYOKA_AMD_EV_INLINE_MAX_NONLINEAR")
    assert "inside the stepper" in tq.hip_source_mode and "pe_on" not in tq.hip_source
    nt_o, _ = events(ho, "o")
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_o)
    for it in range(45):
        ta.step()
        ora.step()
        tq.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        h_q = np.array([h for _, h in tq.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert np.max(np.abs(h_p - h_q) / np.abs(h_q)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
        if it == 20:
            assert rel_err(np.asarray(ta.tc), np.asarray(tq.tc)) <= 1e6 * EPS
    t_end = float(np.max(ora.time_hi)) + 8.0
    ta.propagate_until(t_end)
    ora.propagate_until(t_end)
    tq.propagate_until(t_end)
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e7 * EPS
    kinds = {a[1] for a in logs["p"]}
    assert {0, 1, 2}.issubset(kinds) and len(logs["p"]) >= 2 * n, (kinds, len(logs["p"]))
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["q"]]
    assert np.max(np.abs(np.array([a[2] for a in logs["p"]]) - np.array([a[2] for a in logs["o"]]))) <= 1e-9
    # A terminal close-encounter event: the step of its lane is truncated there.
    logs["p"], logs["o"] = [], []
    nt_p, te_p = events(hy, "p", terminal=True)
    tt = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, nt_events=nt_p[:1], t_events=te_p)
    assert "pe_on" in tt.hip_source
    nt_o, te_o = events(ho, "o", terminal=True)
    ot = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_o[:1], t_events=te_o)
    for _ in range(50):
        tt.step()
        ot.step()
        assert [int(oc) for oc, _ in tt.step_res] == [oc for oc, _ in ot.step_res]
        assert nbody_err(tt.state, ot.state.reshape(36, n)) <= 1e6 * EPS
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]


@pytest.mark.gpu
def test_terminal_events_with_the_event_equations_inside_the_stepper_vs_oracle():
    """A terminal event truncates the step of ITS lane at the event (src/taylor_adaptive_batch.cpp:771-781): the stepper which
    evaluates the event equations itself takes the full step everywhere, and the lanes with a terminal event are redone from
    the Taylor coefficients (their workgroup stored them: a detected event is one the stepper's exclusion test could not
    rule out). Jupiter - Saturn distance below 9 AU as a terminal event whose callback keeps going (cooldown deduced
    automatically), two linear non-terminal events; step by step against the oracle."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 40
    st = configs.outer_ss_state(n, perturb=1e-3, seed=14)
    logs = {"p": ([], []), "o": ([], [])}

    def events(m, key):
        log, te_log = logs[key]
        mk = (lambda s_: m.var(s_)) if m is ho else (lambda s_: m.make_vars(s_, "dummy__")[0])
        x1, y1, z1, x2, y2, z2 = [mk(s_) for s_ in ("x_1", "y_1", "z_1", "x_2", "y_2", "z_2")]
        d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 81.0
        pos = m.DIR_POSITIVE if m is ho else m.event_direction.positive
        neg = m.DIR_NEGATIVE if m is ho else m.event_direction.negative
        nt = [m.nt_event(y2, lambda ta, t, d, i: log.append((i, 0, t, d)), direction=pos),
              m.nt_event(x1 - x2, lambda ta, t, d, i: log.append((i, 1, t, d)))]
        te = [m.t_event(d2, lambda ta, d, i: te_log.append((i, d)) or True, direction=neg)]
        return nt, te

    nt_p, te_p = events(hy, "p")
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_p, t_events=te_p)
    assert "v5" in ta.hip_source_mode and "inside the stepper" in ta.hip_source_mode, ta.hip_source_mode
    nt_o, te_o = events(ho, "o")
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_o, t_events=te_o)
    seen_te = 0
    for _ in range(60):
        ta.step()
        ora.step()
        oc_p = [int(oc) for oc, _ in ta.step_res]
        assert oc_p == [oc for oc, _ in ora.step_res]
        seen_te += sum(1 for oc in oc_p if oc >= 0)
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
        assert np.max(np.abs(ta.time - ora.time_hi)) <= 1e-10
    assert seen_te >= n // 2 and logs["p"][1] == logs["o"][1]
    assert [(a[0], a[1], a[3]) for a in logs["p"][0]] == [(a[0], a[1], a[3]) for a in logs["o"][0]]
    t_end = float(np.max(ora.time_hi)) + 8.0
    ta.propagate_until(t_end)
    ora.propagate_until(t_end)
    assert [int(r[0]) for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e7 * EPS
    assert logs["p"][1] == logs["o"][1]


@pytest.mark.gpu
def test_time_dependent_events_on_the_cluster_stepper_vs_oracle():
    """Time-triggered events (time - t0, x_1 - 5 cos(time / 3)) on the outer Solar System: mode-4 cluster stepper +
    hy_ev_jets against the oracle's stepper with events."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 6
    st = configs.outer_ss_state(n, perturb=1e-3, seed=5)
    logs = {}

    def setup(m, key):
        log = logs.setdefault(key, [])
        x1 = m.var("x_1") if m is ho else hy.make_vars("x_1")
        x1 = x1[0] if isinstance(x1, (list, tuple)) else x1
        t = m.TIME if m is ho else m.time
        return [m.nt_event(t - 0.7, lambda ta, tm, d, i: log.append((i, 0, tm, d))),
                m.nt_event(x1 - 5.0 * m.cos(t / 3.0), lambda ta, tm, d, i: log.append((i, 1, tm, d)))]

    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=setup(hy, "p"))
    assert ta.hip_source_mode.startswith("cluster") and "events:" in ta.hip_source_mode, ta.hip_source_mode
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=setup(ho, "o"))
    for _ in range(25):
        ta.step()
        ora.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
    lp, lo = logs["p"], logs["o"]
    assert len(lp) >= n and [(a[0], a[1], a[3]) for a in lp] == [(a[0], a[1], a[3]) for a in lo]
    assert np.max(np.abs(np.array([a[2] for a in lp]) - np.array([a[2] for a in lo]))) <= 1e-10
    assert all(abs(a[2] - 0.7) <= 1e-14 for a in lp if a[1] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("high_accuracy,exact_division", [(True, False), (False, False), (True, True)])
def test_compact_taylor_coefficients_of_the_stepper_with_events_change_nothing(high_accuracy, exact_division, monkeypatch):
    """The wave-cluster stepper with events writes a compact set of Taylor coefficients (only the order-0 row of the
    positions: x^[k] = v^[k-1] / k is derived by hy_ev_jets, by the dense output of the state update and - on demand - by
    hy_tc_expand). Against the same integrator built with HEYOKA_AMD_COMPACT_TC=0: states, times, step sizes, event
    records, get_tc(), update_d_output() and propagate_grid() are IDENTICAL bit for bit; get_tc() also against the oracle."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 37
    st = configs.outer_ss_state(n, perturb=1e-3, seed=12)

    def build(log, te):
        nt, tev = _outer_ss_event_setup(hy, log, te)
        return hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=high_accuracy,
                                        nt_events=nt, t_events=tev, exact_division=exact_division)

    log_c, te_c, log_f, te_f, log_o, te_o = [], [], [], [], [], []
    # (Both on the lane-pair kernel v3: the one-lane kernel v5, the default for this system, has no full variant.)
    monkeypatch.setenv("HEYOKA_AMD_V5_EVENTS", "0")
    tc_ = build(log_c, te_c)
    monkeypatch.setenv("HEYOKA_AMD_COMPACT_TC", "0")
    tf = build(log_f, te_f)
    monkeypatch.delenv("HEYOKA_AMD_COMPACT_TC")
    monkeypatch.delenv("HEYOKA_AMD_V5_EVENTS")
    assert tc_.hip_source_mode.startswith("cluster") and "v3" in tc_.hip_source_mode and "hy_tc_src" in tc_.hip_source
    # (The full build lists every row in its table, the compact one 18 x 21 + 18 of the 36 x 21.)
    count = lambda src: int(src.split("hy_tc_src[")[1].split("]")[0])
    assert count(tf.hip_source) == 36 * 21 and count(tc_.hip_source) == 18 * 21 + 18
    nt_o, te_ev_o = _outer_ss_event_setup(ho, log_o, te_o)
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=high_accuracy, nt_events=nt_o,
                                   t_events=te_ev_o)
    for it in range(30):
        tc_.step()
        tf.step()
        ora.step()
        assert tc_.step_res == tf.step_res
        assert np.array_equal(tc_.state, tf.state) and np.array_equal(tc_.time, tf.time)
        if it % 7 == 0:
            a, b = np.asarray(tc_.tc), np.asarray(tf.tc)
            assert np.array_equal(a, b)
            tco = ora.tc.reshape(36, ora.order + 1, n)
            scale = np.max(np.abs(tco), axis=2, keepdims=True) + 1e-300
            assert np.max(np.abs(a.reshape(36, ora.order + 1, n) - tco) / scale) <= 1e6 * EPS
        if it % 5 == 0:
            tq = np.asarray(tc_.time) - 0.3 * np.array([h for _, h in tc_.step_res])
            assert np.array_equal(np.asarray(tc_.update_d_output(tq)), np.asarray(tf.update_d_output(tq)))
    assert log_c == log_f and te_c == te_f and len(log_c) > 0
    t0 = float(np.max(tc_.time)) + 5.0
    tc_.propagate_until(t0)
    tf.propagate_until(t0)
    assert np.array_equal(tc_.state, tf.state)
    grid = np.repeat(t0 + np.array([0.0, 1.5, 3.0, 7.0])[:, None], n, axis=1)
    _, out_c = tc_.propagate_grid(grid)
    _, out_f = tf.propagate_grid(grid)
    assert np.array_equal(np.asarray(out_c), np.asarray(out_f))
    c_c, _ = tc_.propagate_until(t0 + 12.0, c_output=True)
    c_f, _ = tf.propagate_until(t0 + 12.0, c_output=True)
    assert np.array_equal(np.asarray(c_c(t0 + 9.1)), np.asarray(c_f(t0 + 9.1)))
    assert log_c == log_f and te_c == te_f


def test_events_on_the_pipelined_cluster_stepper_vs_oracle():
    """Events on a system which runs on the pipelined (v2) wave-cluster stepper - model::np1body(6), heliocentric
    coordinates: mode 4 with the compact Taylor coefficients and the cooperative store, hy_ev_jets, detection - step by step
    against the oracle's stepper with events; get_tc() (expanded on demand) against the oracle's coefficients."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 9
    st = configs.outer_ss_state(n, perturb=1e-3, seed=21, com_shift=False)
    st = st[6:] - np.tile(st[:6], (5, 1))
    log_p, log_o = [], []

    def evs(m, log, var):
        x1, y1, x2, y2, z1, z2 = [var(s) for s in ("x_1", "y_1", "x_2", "y_2", "z_1", "z_2")]
        d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 81.0
        return [m.nt_event(y1, lambda ta, t, d, i: log.append((i, 0, t, d))),
                m.nt_event(d2, lambda ta, t, d, i: log.append((i, 1, t, d)))]

    pv = lambda s_: (lambda v: v[0] if isinstance(v, (list, tuple)) else v)(hy.make_vars(s_))
    ta = hy.taylor_adaptive_batch(hy.model.np1body(6, masses=M, Gconst=G), st, n, high_accuracy=True,
                                  nt_events=evs(hy, log_p, pv))
    assert ta.hip_source_mode.startswith("cluster") and "events:" in ta.hip_source_mode and "lds_tcsrc" in ta.hip_source
    ora = ho.OracleEventIntegrator(ho.np1body(6, masses=M, Gconst=G), st, n, high_accuracy=True,
                                   nt_events=evs(ho, log_o, ho.var))
    for it in range(40):
        ta.step()
        ora.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(30, n)) <= 1e6 * EPS
        if it % 13 == 0:
            tco = ora.tc.reshape(30, ora.order + 1, n)
            scale = np.max(np.abs(tco), axis=2, keepdims=True) + 1e-300
            assert np.max(np.abs(np.asarray(ta.tc).reshape(30, ora.order + 1, n) - tco) / scale) <= 1e6 * EPS
    assert len(log_p) > n and [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o]
    assert np.max(np.abs(np.array([a[2] for a in log_p]) - np.array([a[2] for a in log_o]))) <= 1e-10


def test_events_on_the_cluster_stepper_vs_oracle(monkeypatch):
    """Integrators with events whose system runs on a wave-cluster stepper: the stepper computes the jets of the state
    variables only (mode 4, no update), hy_ev_jets derives the jets of the event equations and the final step size from
    them. Step by step against the oracle's stepper with events (outcomes, step sizes, states, event times / signs /
    order), and against the one-system-per-lane stepper with events of the product."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 10
    st = configs.outer_ss_state(n, perturb=1e-3, seed=11)
    log_p, te_p, log_o, te_o, log_q, te_q = [], [], [], [], [], []
    nt_p, te_ev_p = _outer_ss_event_setup(hy, log_p, te_p)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_p,
                                  t_events=te_ev_p)
    assert ta.hip_source_mode.startswith("cluster") and "events:" in ta.hip_source_mode, ta.hip_source_mode
    # (The stepper with events of this system is the one-lane-per-pair kernel v5 since round 3.)
    assert "v5" in ta.hip_source_mode
    nt_o, te_ev_o = _outer_ss_event_setup(ho, log_o, te_o)
    ora = ho.OracleEventIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_o,
                                   t_events=te_ev_o)
    monkeypatch.setenv("HEYOKA_AMD_EVENTS_ON_CLUSTER", "0")
    nt_q, te_ev_q = _outer_ss_event_setup(hy, log_q, te_q)
    tq = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, nt_events=nt_q,
                                  t_events=te_ev_q)
    assert not tq.hip_source_mode.startswith("cluster")
    for _ in range(45):
        ta.step()
        ora.step()
        tq.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        assert [int(oc) for oc, _ in ta.step_res] == [int(oc) for oc, _ in tq.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
        assert nbody_err(ta.state, np.asarray(tq.state)) <= 1e6 * EPS
    assert len(log_p) >= 3 * n and len(te_p) >= n
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o]
    assert np.max(np.abs(np.array([a[2] for a in log_p]) - np.array([a[2] for a in log_o]))) <= 1e-10
    assert te_p == te_o and te_p == te_q
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_q]
    # propagate_until() / propagate_grid() with events: the device-driven lock-step loops with the step with events as
    # their sweep, against the oracle's loop (propagate_until) and on the one-system-per-lane stepper with events.
    t_end = float(np.max(ora.time_hi)) + 12.0
    ta.propagate_until(t_end)
    ora.propagate_until(t_end)
    tq.propagate_until(t_end)
    assert [int(r[0]) for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert [int(r[0]) for r in ta.propagate_res] == [int(r[0]) for r in tq.propagate_res]
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, ora.prop_res)) <= 1
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e7 * EPS
    assert nbody_err(ta.state, np.asarray(tq.state)) <= 1e7 * EPS
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o]
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_q]
    assert te_p == te_o and te_p == te_q
    grid = np.repeat(t_end + np.array([0.0, 1.5, 3.0, 7.0])[:, None], n, axis=1)
    _, out_p = ta.propagate_grid(grid)
    _, out_q = tq.propagate_grid(grid)
    assert rel_err(np.asarray(out_p), np.asarray(out_q)) <= 1e7 * EPS
    assert [int(r[0]) for r in ta.propagate_res] == [int(r[0]) for r in tq.propagate_res]
    assert [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_q]


@pytest.mark.gpu
def test_events_on_systems_with_default_masses_reach_the_cluster_stepper(monkeypatch):
    """model::nbody(6) with its DEFAULT masses and events: until round 5 such an integrator fell to the table stepper (the
    accelerations of equal masses are sum / sub / negation trees which no stepper with events of the wave-cluster family
    took). With the accelerations flattened in the internal program (linearise_accelerations()) it runs on the
    one-lane-per-pair stepper with the event equations inside. Step by step against the oracle's stepper with events and
    against the one-system-per-lane stepper with events: outcomes, step sizes, states, events in order; then a lock-step
    propagation."""
    n = 70
    st = configs.plummer_nbody_state(6, n, seed=21, jitter=1e-3)
    logs = {"p": [], "o": [], "q": []}
    tes = {"p": [], "o": [], "q": []}

    def events(m, key):
        mk = (lambda s_: m.var(s_)) if m is ho else (lambda s_: m.make_vars(s_))
        x1, y1, z1, vx1, vy1, vz1 = [mk(s_ + "_1") for s_ in ("x", "y", "z", "vx", "vy", "vz")]
        x2, y2, z2 = [mk(s_ + "_2") for s_ in ("x", "y", "z")]
        nt = [m.nt_event(y1, lambda ta, t, d, i: logs[key].append((i, 0, t, d))),
              m.nt_event(x1 * vx1 + y1 * vy1 + z1 * vz1, lambda ta, t, d, i: logs[key].append((i, 1, t, d)))]
        # (Bodies 1 and 2 start 2.5 apart and approach: the squared distance falls through 6.2 within the first steps.)
        d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 6.2
        te = [m.t_event(d2, lambda ta, d, i: tes[key].append((i, d)) or True)]
        return nt, te

    nt_p, te_p = events(hy, "p")
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6), st, n, nt_events=nt_p, t_events=te_p)
    mode = ta.hip_source_mode
    assert mode.startswith("cluster") and "v5" in mode and "accelerations rewritten as flat sums" in mode, mode
    nt_o, te_o = events(ho, "o")
    ora = ho.OracleEventIntegrator(ho.nbody(6), st, n, nt_events=nt_o, t_events=te_o)
    monkeypatch.setenv("HEYOKA_AMD_EVENTS_ON_CLUSTER", "0")
    nt_q, te_q = events(hy, "q")
    tq = hy.taylor_adaptive_batch(hy.model.nbody(6), st, n, nt_events=nt_q, t_events=te_q)
    assert not tq.hip_source_mode.startswith("cluster")
    for _ in range(40):
        ta.step()
        ora.step()
        tq.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        assert [int(oc) for oc, _ in ta.step_res] == [int(oc) for oc, _ in tq.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e6 * EPS
        assert nbody_err(ta.state, np.asarray(tq.state)) <= 1e6 * EPS
    assert len(logs["p"]) >= n // 2 and len(tes["p"]) >= n // 2
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]
    assert np.max(np.abs(np.array([a[2] for a in logs["p"]]) - np.array([a[2] for a in logs["o"]]))) <= 1e-10
    assert tes["p"] == tes["o"] and tes["p"] == tes["q"]
    t_end = float(np.max(ora.time_hi)) + 0.3
    ta.propagate_until(t_end)
    ora.propagate_until(t_end)
    assert [int(r[0]) for r in ta.propagate_res] == [r[0] for r in ora.prop_res]
    assert nbody_err(ta.state, ora.state.reshape(36, n)) <= 1e7 * EPS
    assert [(a[0], a[1], a[3]) for a in logs["p"]] == [(a[0], a[1], a[3]) for a in logs["o"]]


@pytest.mark.gpu
def test_python_exceptions_in_pre_hook_and_in_a_callback_set_stop_the_propagation():
    """Advisor findings (round 3): an exception raised by pre_hook() must stop propagate_*() BEFORE the first step (the
    reference lets it out of propagate_*(), src/taylor_adaptive_batch.cpp:1356-1365) - state and time untouched -, and after
    an exception in one member of a callback set the other members are not run any more. Also: get_tc() keeps the
    coefficients of the last step taken with write_tc when later steps run without it (the steppers which use the
    coefficient buffer as their scratch included)."""
    x, v = hy.make_vars("x", "v")
    st = np.array([[0.05, 0.06], [0.025, 0.03]])
    ta = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, 2)

    class Hooked:
        def __init__(self):
            self.calls = 0

        def pre_hook(self, t):
            raise RuntimeError("pre_hook says no")

        def __call__(self, t):
            self.calls += 1
            return True

    h = Hooked()
    with pytest.raises(RuntimeError, match="pre_hook says no"):
        ta.propagate_until(1.0, callback=h)
    assert h.calls == 0 and np.array_equal(ta.state, st) and np.all(np.asarray(ta.time) == 0.0)

    seen = []

    def bad(t):
        seen.append("bad")
        raise ValueError("callback failed")

    def other(t):
        seen.append("other")
        return True

    with pytest.raises(ValueError, match="callback failed"):
        ta.propagate_until(1.0, callback=[bad, other])
    assert seen == ["bad"]

    tb = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, 2)
    tb.step(write_tc=True)
    tb2 = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], st, 2)
    tb2.step(write_tc=True)
    ref_tc = np.array(tb2.tc).copy()
    tb.step()          # no write_tc: must not disturb what get_tc() returns
    assert np.array_equal(np.asarray(tb.tc), ref_tc)


@pytest.mark.gpu
def test_code_objects_are_reference_counted_and_unloaded_beyond_the_bound():
    """Code objects are shared between integrators of the same system, reference counted and - once idle - kept as a bounded
    cache (HEYOKA_AMD_KEEP_MODULES, default 256; earlier rounds never unloaded anything). With a bound of 2 a process which
    builds and drops ten distinct integrators unloads eight code objects on the way; every integrator steps correctly, an
    integrator of a system seen before works again (reload), and the kernels of another HIP client (torch) run afterwards."""
    import subprocess
    import sys

    code = r"""
import numpy as np, torch
import heyoka_amd as hy
x, v = hy.make_vars("x", "v")
res = []
for k in list(range(10)) + [0, 3]:
    ta = hy.taylor_adaptive_batch([(x, v), (v, -(1.0 + 0.1 * k) * hy.sin(x))], np.array([[0.3] * 64, [0.0] * 64]), 64)
    ta.propagate_until(1.0)
    res.append((k, float(ta.state[0, 0])))
    del ta
assert res[10][1] == res[0][1] and res[11][1] == res[3][1]
assert len({r[1] for r in res[:10]}) == 10
t = torch.arange(1024, device="cuda", dtype=torch.float64)
assert float((t * 2).sum()) == 1023 * 1024.0
print("OK")
"""
    env = dict(os.environ, HEYOKA_AMD_KEEP_MODULES="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.gpu
def test_library_side_counting_callbacks_are_applied_on_the_device():
    """Events whose callbacks are ALL the library's counting callbacks (hy_event_counter_nt / hy_event_counter_t: nothing
    of the caller's runs) are applied by the post-step kernel itself - counts per event, cooldown and "continuing" outcome
    of the first terminal event of a lane - instead of the host loop over the records of the step
    (src/taylor_adaptive_batch.cpp:837-1030). Against the same integrator with Python callbacks which count (the host
    loop): identical counts per event, outcomes, step sizes, states, times and cooldowns after every step, on an ensemble
    in which events fire in most steps."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 1024
    rng = np.random.RandomState(5)
    sysd = hy.model.nbody(6, masses=M, Gconst=G)
    spread = hy.taylor_adaptive_batch(sysd, configs.outer_ss_state(n, perturb=1e-6, seed=31), n, high_accuracy=True)
    spread.propagate_until(rng.uniform(0.0, 30.0, n))
    st = np.array(spread.state)
    y1, y2, y3 = hy.make_vars("y_1", "y_2", "y_3")
    c_nt = [hy.native_event_counter(), hy.native_event_counter()]
    c_t = hy.native_event_counter()
    a = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True, nt_events=[hy.nt_event(y1, c_nt[0]), hy.nt_event(y2, c_nt[1])],
                                 t_events=[hy.t_event(y3, c_t, cooldown=0.05)])
    p_nt, p_t = [0, 0], [0]

    def mk_nt(k):
        def cb(ta, t, d_sgn, idx):
            p_nt[k] += 1
        return cb

    def cb_t(ta, d_sgn, idx):
        p_t[0] += 1
        return True

    b = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True, nt_events=[hy.nt_event(y1, mk_nt(0)), hy.nt_event(y2, mk_nt(1))],
                                 t_events=[hy.t_event(y3, cb_t, cooldown=0.05)])
    for _ in range(12):
        a.step()
        b.step()
        assert a.step_res == b.step_res
        assert np.array_equal(a.state, b.state) and np.array_equal(np.asarray(a.time), np.asarray(b.time))
        assert [c.value for c in c_nt] == p_nt and c_t.value == p_t[0]
    assert sum(p_nt) > 100 and p_t[0] > 10
    assert any(int(o) >= 0 for o, _ in a.step_res) or p_t[0] > 0
    assert a.te_cooldowns == b.te_cooldowns
    # With an automatic cooldown as well (taylor_deduce_cooldown(), src/detail/event_detection.cpp:519-550).
    c2, cnt = hy.native_event_counter(), [0]

    def cb2(ta, d_sgn, idx):
        cnt[0] += 1
        return True

    a2 = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True, t_events=[hy.t_event(y3, c2)])
    b2 = hy.taylor_adaptive_batch(sysd, st, n, high_accuracy=True, t_events=[hy.t_event(y3, cb2)])
    for _ in range(6):
        a2.step()
        b2.step()
        assert a2.step_res == b2.step_res and np.array_equal(a2.state, b2.state)
    assert c2.value == cnt[0] > 0 and a2.te_cooldowns == b2.te_cooldowns
