"""The point-mass models beyond model::nbody: np1body, cr3bp, fixed_centres, rotating,
mascon. CPU part: the product's expression system / decomposition / compiled-function decomposition against the
sizes the reference's own tests pin (tests/golden/model_structure_pins.json, transcribed from
test/model_{nbody,cr3bp,fixed_centres,rotating,mascon}.cpp), against the independent Python restatement in
oracle/heyoka_oracle.py, and the verbatim error messages. GPU part: one full-order step and a propagation vs the
oracle, and the conservation checks of the reference's tests (energy / Jacobi constant) with the device-side
compiled functions."""
import json
import os

import numpy as np
import pytest

import heyoka_amd as hy
import heyoka_oracle as ho

EPS = 2.220446049250313e-16
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pins():
    with open(os.path.join(HERE, "golden", "model_structure_pins.json")) as f:
        return json.load(f)


def dc_size(sys_):
    return len(hy.taylor_decompose_sys(sys_))


def cfunc_dc_size(ex, vars_):
    return len(hy.cfunc([ex], vars_).dc)


def _masses(kind, M, par, num):
    return {
        "numeric": lambda: list(M),
        "numeric_first5": lambda: list(M[:5]),
        "par6": lambda: [par(i) for i in range(6)],
        "par5": lambda: [par(i) for i in range(5)],
        "par3_num3": lambda: [par(0), par(1), par(2)] + list(M[3:]),
        "par3_zero_num2": lambda: [par(0), par(1), par(2), 0.0] + list(M[4:]),
        "default": lambda: None,
        "zeros": lambda: [0.0] * 6,
    }[kind]()


def _energy_masses(kind, M):
    # The reference's tests build the energy with the numerical values of the masses.
    return {
        "numeric": list(M), "numeric_first5": list(M[:5]), "par6": list(M), "par5": list(M[:5]),
        "par3_num3": list(M), "par3_zero_num2": list(M[:3]) + [0.0] + list(M[4:]), "default": None, "zeros": [0.0] * 6,
    }[kind]


def _G(pins):
    a, b = pins["outer_ss"]["Gconst_factors"]
    return a * a * b * b


def test_nbody_and_np1body_structure_pins(pins):
    M, G = pins["outer_ss"]["masses"], _G(pins)
    par = lambda i: hy.par[i]
    vars6 = hy.model.nbody(6).vars
    for case in pins["nbody6"]:
        kw = {} if case["masses"] == "default" else {"Gconst": G}
        if "dc" in case:
            assert dc_size(hy.model.nbody(6, masses=_masses(case["masses"], M, par, None), **kw)) == case["dc"], case
        en = hy.model.nbody_energy(6, masses=_energy_masses(case["masses"], M), **kw)
        assert cfunc_dc_size(en, vars6) == case["cfunc_dc"], case
    vars5 = hy.model.np1body(6).vars
    assert [str(v) for v in vars5[:6]] == ["x_1", "y_1", "z_1", "vx_1", "vy_1", "vz_1"] and len(vars5) == 30
    for case in pins["np1body6"]:
        kw = {} if case["masses"] == "default" else {"Gconst": G}
        if "dc" in case:
            assert dc_size(hy.model.np1body(6, masses=_masses(case["masses"], M, par, None), **kw)) == case["dc"], case
        en = hy.model.np1body_energy(6, masses=_energy_masses(case["masses"], M), **kw)
        assert cfunc_dc_size(en, vars5) == case["cfunc_dc"], case
    # nbody_potential() without massive particles (test/model_nbody.cpp:449-450).
    # (the reference compares numbers by value: -G * sum({}) is the number -0 == 0).
    assert float(str(hy.model.nbody_potential(2, masses=[]))) == 0 and float(str(hy.model.nbody_potential(10, masses=[]))) == 0


def test_cr3bp_fixed_centres_rotating_mascon_structure_pins(pins):
    c = pins["cr3bp"]
    for mu in (None, c["other_mu"]):
        kw = {} if mu is None else {"mu": mu}
        s = hy.model.cr3bp(**kw)
        assert [str(v) for v in s.vars] == c["state_vars"]
        assert dc_size(s) == c["dc"] and cfunc_dc_size(hy.model.cr3bp_jacobi(**kw), s.vars) == c["cfunc_dc"]

    f = pins["fixed_centres"]
    rng = np.random.RandomState(0)
    m = rng.uniform(*f["value_range"], f["n_masses"])
    pos = rng.uniform(*f["value_range"], 3 * f["n_masses"])
    s = hy.model.fixed_centres(masses=m, positions=pos)
    assert dc_size(s) == f["dc"]
    assert cfunc_dc_size(hy.model.fixed_centres_energy(masses=m, positions=pos), s.vars) == f["cfunc_dc"]

    r = pins["rotating"]
    s = hy.model.rotating(omega=r["omega"])
    x, y, z, vx, vy, vz = s.vars
    kin = 0.5 * (vx * vx + vy * vy + vz * vz)
    assert dc_size(s) == r["dc"]
    assert cfunc_dc_size(kin + hy.model.rotating_potential(omega=r["omega"]), s.vars) == r["cfunc_dc"]
    P = [hy.par[0], hy.par[1], hy.par[2]]
    assert dc_size(hy.model.rotating(omega=P)) == r["dc_par"]
    assert cfunc_dc_size(kin + hy.model.rotating_potential(omega=P), s.vars) == r["cfunc_dc_par"]
    # No rotation: free particle, zero potential (test/model_rotating.cpp:41-62).
    assert str(hy.model.rotating_potential()) == "0"
    assert hy.taylor_decompose_sys(hy.model.rotating())[-3:] == ["0", "0", "0"]

    k = pins["mascon"]
    m1, p1 = m[:1], pos[:3]
    s = hy.model.mascon(masses=m1, positions=p1, Gconst=k["Gconst"], omega=k["omega"])
    assert dc_size(s) == k["dc"]
    en = hy.model.mascon_energy(masses=m1, positions=p1, Gconst=k["Gconst"], omega=k["omega"])
    assert cfunc_dc_size(en, s.vars) == k["cfunc_dc_energy"]
    kp = kin + hy.model.mascon_potential(masses=m1, positions=p1, Gconst=k["Gconst"], omega=k["omega"])
    assert cfunc_dc_size(kp, s.vars) == k["cfunc_dc_kin_plus_potential"]


def test_model_error_messages(pins):
    for mu, msg in pins["cr3bp"]["errors"]:
        for fn in (hy.model.cr3bp, hy.model.cr3bp_jacobi):
            with pytest.raises(ValueError) as ei:
                fn(mu=mu)
            assert str(ei.value) == msg
    for m, pos, msg in pins["fixed_centres"]["errors"]:
        for fn in (hy.model.fixed_centres, hy.model.fixed_centres_energy, hy.model.fixed_centres_potential):
            with pytest.raises(ValueError) as ei:
                fn(masses=m, positions=pos)
            assert str(ei.value) == msg
    for om, msg in pins["rotating"]["errors"]:
        for fn in (hy.model.rotating, hy.model.rotating_energy, hy.model.rotating_potential):
            with pytest.raises(ValueError) as ei:
                fn(omega=om)
            assert str(ei.value) == msg


_M8 = [1.0, 1e-3, 2e-3, 3e-4, 5e-4, 1e-5, 2e-5, 3e-5]


def _model_cases():
    M = [1.00000597682, 1 / 1047.355, 1 / 3501.6, 1 / 22869.0, 1 / 19314.0, 7.4074074e-09]
    G = 0.01720209895 * 0.01720209895 * 365 * 365
    rng = np.random.RandomState(3)
    m = rng.uniform(0.2, 1.0, 7)
    m /= m.sum()  # total mass 1: the reference tests' initial state (1, 0, 0, 0, 1, 0) is then a near-circular orbit
    pos = rng.uniform(-0.3, 0.3, 21)
    om = [0.1, 0.11, 0.12]
    return {
        "cr3bp": (lambda: hy.model.cr3bp(), lambda: ho.cr3bp()),
        "cr3bp_par": (lambda: hy.model.cr3bp(mu=hy.par[0]), lambda: ho.cr3bp(mu=ho.par(0))),
        "np1body6": (lambda: hy.model.np1body(6, masses=M, Gconst=G), lambda: ho.np1body(6, masses=M, Gconst=G)),
        "np1body4_par": (lambda: hy.model.np1body(4, masses=[hy.par[0], 1e-3, hy.par[1]]),
                         lambda: ho.np1body(4, masses=[ho.par(0), 1e-3, ho.par(1)])),
        "np1body5_massless": (lambda: hy.model.np1body(5, masses=[1.0, 1e-3]), lambda: ho.np1body(5, masses=[1.0, 1e-3])),
        "np1body8_default": (lambda: hy.model.np1body(8), lambda: ho.np1body(8)),
        "np1body13_default": (lambda: hy.model.np1body(13), lambda: ho.np1body(13)),
        "np1body8_masses": (lambda: hy.model.np1body(8, masses=_M8), lambda: ho.np1body(8, masses=_M8)),
        "fixed_centres7": (lambda: hy.model.fixed_centres(masses=m, positions=pos, Gconst=1.02),
                           lambda: ho.fixed_centres(masses=m, positions=pos, Gconst=1.02)),
        "rotating": (lambda: hy.model.rotating(omega=om), lambda: ho.rotating(omega=om)),
        "rotating_par": (lambda: hy.model.rotating(omega=[hy.par[0], hy.par[1], hy.par[2]]),
                         lambda: ho.rotating(omega=[ho.par(0), ho.par(1), ho.par(2)])),
        "rotating_none": (lambda: hy.model.rotating(), lambda: ho.rotating()),
        "mascon7": (lambda: hy.model.mascon(masses=m, positions=pos, Gconst=1.01, omega=om),
                    lambda: ho.mascon(masses=m, positions=pos, Gconst=1.01, omega=om)),
    }


@pytest.mark.parametrize("name", sorted(_model_cases()))
def test_model_decomposition_identical_to_oracle(name):
    prod, ora = _model_cases()[name]
    assert hy.taylor_decompose_sys(prod()) == ho.dc_to_strings(ho.taylor_decompose_sys(ora()))


# ------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path vs the oracle and the reference tests' conservation checks.
# ------------------------------------------------------------------------------------------------------------------
def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def row_rel_err(a, b):
    """State comparison with a per-row scale (rows = state variables, columns = systems): max |a - b| over the ensemble
    divided by max |b| over the ensemble, no floor at 1 (round 6: rows well below 1 - velocities of massive bodies,
    out-of-plane components - were compared at an ABSOLUTE tolerance otherwise)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a = a.reshape(b.shape)
    return float(np.max(np.max(np.abs(a - b), axis=1) / (np.max(np.abs(b), axis=1) + 1e-300)))


def _lanes(base, n, rel, seed):
    """(len(base), n) initial conditions: lane 0 is the reference test's state, the others are perturbed copies."""
    rng = np.random.RandomState(seed)
    b = np.asarray(base, dtype=np.float64)[:, None]
    st = b + (np.abs(b) + 0.05) * rel * rng.uniform(-1, 1, (b.shape[0], n))
    st[:, 0] = b[:, 0]
    return np.ascontiguousarray(st)


def _relative_plummer(n_bodies, n):
    """Positions / velocities of bodies 1..n-1 relative to body 0 of the seeded Plummer-like cloud of the N-body tests."""
    from heyoka_amd import configs

    st = configs.plummer_nbody_state(n_bodies, n, seed=77, jitter=1e-6).reshape(n_bodies, 6, n)
    return np.ascontiguousarray((st[1:] - st[:1]).reshape(6 * (n_bodies - 1), n))


def _gpu_cases(pins):
    from heyoka_amd import configs

    M, G = pins["outer_ss"]["masses"], _G(pins)
    n = 40
    cr_st = _lanes(pins["cr3bp"]["init_state"], n, 1e-3, 1)
    fc_st = _lanes(pins["mascon"]["init_state"], n, 1e-2, 2)
    rot_st = _lanes(pins["rotating"]["init_state"], n, 1e-2, 4)
    oss = configs.outer_ss_state(n, perturb=1e-6, seed=11, com_shift=False)[6:]
    npar = lambda vals: np.repeat(np.asarray(vals, dtype=np.float64)[:, None], n, axis=1) * (1 + 1e-3 * np.arange(n) / n)
    cases = _model_cases()
    return {
        # name: (product system, oracle system, state, pars, horizon)
        "cr3bp": (*cases["cr3bp"], cr_st, None, 5.0),
        "cr3bp_par": (*cases["cr3bp_par"], cr_st, npar([1e-2]), 5.0),
        "np1body6": (lambda: hy.model.np1body(6, masses=M, Gconst=G), lambda: ho.np1body(6, masses=M, Gconst=G), oss, None, 30.0),
        "np1body4_par": (*cases["np1body4_par"], oss[:18], npar([1.0, 3e-4]), 30.0),
        # Unit masses: the planner needs its second attempt (no absorption of the scaling products) on top of the aliases.
        "np1body8_default": (*cases["np1body8_default"], _relative_plummer(8, n), None, 0.5),
        # The first-generation cluster kernel (separate state-variable rounds) on aliased programs.
        "np1body8_masses": (*cases["np1body8_masses"], _relative_plummer(8, n), None, 0.5),
        "np1body5_massless": (*cases["np1body5_massless"], _relative_plummer(5, n), None, 0.5),
        # 78 clusters (> 64 lanes): one system per workgroup (block mode) on the aliased program.
        "np1body13_default": (*cases["np1body13_default"], _relative_plummer(13, n), None, 0.2),
        "fixed_centres7": (*cases["fixed_centres7"], fc_st, None, 5.0),
        "rotating": (*cases["rotating"], rot_st, None, 5.0),
        "rotating_par": (*cases["rotating_par"], rot_st, npar([0.1, 0.2, 0.3]), 5.0),
        "mascon7": (*cases["mascon7"], fc_st, None, 5.0),
        # Centres at the origin / with repeated and zero coordinates / with G m = 1: shared and negated coordinate
        # differences, which the planner makes private per cluster (privatise_cluster_inputs()): wave-cluster kernel
        # (12 centres) and block mode (70 centres) instead of the table stepper. A particle well outside the centres.
        "fixed_centres12_special": (lambda: hy.model.fixed_centres(Gconst=1.0, masses=_fixed_centres_special(12)[0],
                                                                   positions=_fixed_centres_special(12)[1]),
                                    lambda: ho.fixed_centres(Gconst=1.0, masses=_fixed_centres_special(12)[0],
                                                             positions=_fixed_centres_special(12)[1]),
                                    _lanes([5.0, 0.3, -0.2, 0.0, 1.0, 0.1], n, 1e-2, 6), None, 30.0),
        "fixed_centres70_special": (lambda: hy.model.fixed_centres(Gconst=1.0, masses=_fixed_centres_special(70)[0],
                                                                   positions=_fixed_centres_special(70)[1]),
                                    lambda: ho.fixed_centres(Gconst=1.0, masses=_fixed_centres_special(70)[0],
                                                             positions=_fixed_centres_special(70)[1]),
                                    _lanes([6.0, 0.3, -0.2, 0.0, 2.4, 0.1], n, 1e-2, 7), None, 10.0),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cr3bp", "cr3bp_par", "np1body6", "np1body4_par", "np1body8_default", "np1body13_default",
                                  "np1body8_masses", "np1body5_massless", "fixed_centres7",
                                  "rotating", "rotating_par", "mascon7", "fixed_centres12_special",
                                  "fixed_centres70_special"])
@pytest.mark.parametrize("contract", [True, False])
def test_models_step_and_propagate_vs_oracle(name, pins, contract, monkeypatch):
    """One full-order step (h, Taylor coefficients, state) and a propagation of every model against the oracle.
    Tolerances: those of the N-body parity tests (h 1e6 eps, coefficients 1e6 eps of the row maximum, state 1e5 eps
    after one step, 1e7 eps after the propagation of ~25-150 steps); built WITHOUT FMA contraction and with true
    quotients (contract = False: the oracle's arithmetic up to the re-association of the cluster kernels and the library
    functions) the reference's own tolerances: h 1e4 eps, coefficients 1e5 eps (test/two_body_batch.cpp:118-150)."""
    prod, ora, st, pars, T = _gpu_cases(pins)[name]
    n = st.shape[1]
    kw = {} if pars is None else {"pars": pars}
    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
        kw["exact_division"] = True
    h_tol, tc_tol = (1e6, 1e6) if contract else (1e4, 1e5)
    ta = hy.taylor_adaptive_batch(prod(), st, n, **kw)
    kw.pop("exact_division", None)
    if name in ("np1body8_masses", "np1body5_massless"):
        m_ = ta.hip_source_mode
        assert m_.startswith("cluster") and "v2" not in m_.split(";")[0] and "aliased" in m_
    if name == "np1body13_default":
        assert ta.hip_source_mode.startswith("block") and "aliased" in ta.hip_source_mode
    if name.endswith("_special"):
        assert ta.hip_source_mode.startswith("cluster" if name == "fixed_centres12_special" else "block")
        assert "private coordinate differences" in ta.hip_source_mode
    if name == "np1body4_par":
        # Runtime masses: constant u variables (m_0 + m_i, -m_i ...) are recognised, products with them are linear, and
        # the system leaves the 21 000-statement unrolled kernel for the wave-cluster stepper.
        assert ta.hip_source_mode.startswith("cluster")
    if name in ("np1body6", "np1body8_default"):
        # State variables in history-operand position (|r_i|^2 = sum_sq(x_i, y_i, z_i)) are aliased by u variables so that
        # the wave-cluster stepper applies (add_state_aliases(), heyoka_amd/csrc/hip_emit_cluster.cpp).
        assert ta.hip_source_mode.startswith("cluster") and "aliased" in ta.hip_source_mode
    oi = ho.OracleIntegrator(ora(), st, n, **kw)
    n_eq = st.shape[0]
    ta.step(write_tc=True)
    oi.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in oi.step_res])
    assert all(o == hy.taylor_outcome.success for o, _ in ta.step_res)
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= h_tol * EPS
    tc_o = oi.tc.reshape(n_eq, oi.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(n_eq, oi.order + 1, n) - tc_o) / scale) <= tc_tol * EPS
    assert row_rel_err(ta.state, oi.state.reshape(n_eq, n)) <= 1e5 * EPS
    ta.propagate_until(T)
    oi.propagate_until(T)
    assert all(r[0] == hy.taylor_outcome.time_limit for r in ta.propagate_res)
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, oi.prop_res)) <= (0 if contract else 1)
    assert row_rel_err(ta.state, oi.state.reshape(n_eq, n)) <= 1e7 * EPS


def _invariant_drift(sys_, inv_ex, st, T, pars=None, high_accuracy=False):
    """Relative drift of a compiled-function invariant evaluated on the device-resident state before and after
    propagate_until(T) (the pattern of test/model_*.cpp: cf(outs, init_state); propagate; cf(outs, state))."""
    import torch

    n = st.shape[1]
    kw = {} if pars is None else {"pars": pars}
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=high_accuracy, **kw)
    cf = hy.cfunc([inv_ex], sys_.vars)
    e0 = cf(st, **kw)[0]
    ta.propagate_until(T)
    assert all(r[0] == hy.taylor_outcome.time_limit for r in ta.propagate_res)
    e1 = cf(ta.state, **kw)[0]
    # The same through the device-side entry point (no host round trip of the state).
    if pars is None:
        d_e = torch.empty(n, dtype=torch.float64, device="cuda")
        cf.eval_device(d_e.data_ptr(), ta.device_array("state").ptr, n)
        ta.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(d_e.cpu().numpy(), e1)
    return np.max(np.abs((e1 - e0) / e0)), ta


@pytest.mark.gpu
def test_models_conservation_checks_of_the_reference_tests(pins):
    """The invariants the reference's model tests check, with their horizons and tolerances ('approximately()' = 100
    eps unless stated): Jacobi constant of the CR3BP (test/model_cr3bp.cpp:41-91), energy in a rotating frame (1000
    eps, test/model_rotating.cpp:65-126), mascon energy (test/model_mascon.cpp:311-343), energy with 100 fixed
    centres (test/model_fixed_centres.cpp:112-153), N+1-body energy of the outer Solar System
    (test/model_nbody.cpp:490-524), and the equivalence of one fixed centre with the two-body problem with one
    massive body (1000 eps, test/model_fixed_centres.cpp:66-110). Lane 0 holds the reference's initial state, the
    other lanes perturbed copies."""
    from heyoka_amd import configs

    n = 32
    c = pins["cr3bp"]
    for mu in (c["default_mu"], c["other_mu"]):
        drift, _ = _invariant_drift(hy.model.cr3bp(mu=mu), hy.model.cr3bp_jacobi(mu=mu), _lanes(c["init_state"], n, 1e-3, 5),
                                    c["t_final"])
        assert drift <= c["invariant_tol_eps"] * EPS

    r = pins["rotating"]
    st = _lanes(r["init_state"], n, 1e-2, 6)
    drift, _ = _invariant_drift(hy.model.rotating(omega=r["omega"]), hy.model.rotating_energy(omega=r["omega"]), st, r["t_final"])
    assert drift <= r["invariant_tol_eps"] * EPS
    P = [hy.par[0], hy.par[1], hy.par[2]]
    pars = np.repeat(np.asarray(r["omega"])[:, None], n, axis=1)
    drift, _ = _invariant_drift(hy.model.rotating(omega=P), hy.model.rotating_energy(omega=P), st, r["t_final"], pars=pars)
    assert drift <= r["invariant_tol_eps"] * EPS

    k = pins["mascon"]
    rng = np.random.RandomState(0)
    m1, p1 = rng.uniform(*k["value_range"], 1), rng.uniform(*k["value_range"], 3)
    args = dict(masses=m1, positions=p1, Gconst=k["Gconst"], omega=k["omega"])
    drift, _ = _invariant_drift(hy.model.mascon(**args), hy.model.mascon_energy(**args), _lanes(k["init_state"], n, 1e-2, 7),
                                k["t_final"])
    assert drift <= k["invariant_tol_eps"] * EPS

    f = pins["fixed_centres"]
    m = rng.uniform(*f["value_range"], f["n_masses"])
    pos = rng.uniform(*f["value_range"], 3 * f["n_masses"])
    drift, ta = _invariant_drift(hy.model.fixed_centres(masses=m, positions=pos),
                                 hy.model.fixed_centres_energy(masses=m, positions=pos), _lanes(f["init_state"], n, 1e-2, 8),
                                 f["t_final"])
    assert drift <= 100 * EPS

    M, G = pins["outer_ss"]["masses"], _G(pins)
    st = configs.outer_ss_state(n, perturb=1e-9, seed=12, com_shift=False)[6:]
    drift, _ = _invariant_drift(hy.model.np1body(6, masses=M, Gconst=G), hy.model.np1body_energy(6, masses=M, Gconst=G), st, 100.0)
    assert drift <= 100 * EPS

    e = f["two_body_equivalence"]
    cx, cy, cz = e["position"]
    rel = _lanes([1.0, 0.0, 0.0, 0.1, 1.0, 0.2], n, 1e-2, 9)  # the reference's state relative to the centre
    fix_st = rel + np.array([cx, cy, cz, 0, 0, 0])[:, None]
    two_st = np.concatenate([np.repeat(np.array([cx, cy, cz, 0.0, 0.0, 0.0])[:, None], n, axis=1), fix_st])
    ta_fix = hy.taylor_adaptive_batch(hy.model.fixed_centres(Gconst=e["Gconst"], masses=[e["mass"]], positions=e["position"]), fix_st, n)
    ta_2bp = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[e["mass"]], Gconst=e["Gconst"]), two_st, n)
    ta_fix.propagate_until(e["t_final"])
    ta_2bp.propagate_until(e["t_final"])
    a, b = ta_fix.state, ta_2bp.state[6:]
    # Lane 0 (the reference's initial state): the reference's per-component criterion; the perturbed lanes: relative
    # to the norms of the position and velocity vectors (a component may pass close to zero).
    assert np.max(np.abs(a[:, 0] - b[:, 0]) / np.maximum(np.abs(a[:, 0]), np.abs(b[:, 0]))) <= e["tol_eps"] * EPS
    for sl in (slice(0, 3), slice(3, 6)):
        assert np.max(np.linalg.norm(a[sl] - b[sl], axis=0) / np.linalg.norm(b[sl], axis=0)) <= e["tol_eps"] * EPS


@pytest.mark.gpu
def test_np1body_cluster_stepper_equals_table_stepper(pins, monkeypatch):
    """model::np1body through the wave-cluster stepper (with the state-variable aliases) and through the table stepper
    (HEYOKA_AMD_NO_STATE_ALIASES=1): same step counts, states within 1e5 eps, with the compensated update as well."""
    from heyoka_amd import configs

    M, G = pins["outer_ss"]["masses"], _G(pins)
    n = 128
    st = configs.outer_ss_state(n, perturb=1e-6, seed=5, com_shift=False)[6:]
    for ha in (False, True):
        res = {}
        for which in ("cluster", "table"):
            if which == "table":
                monkeypatch.setenv("HEYOKA_AMD_NO_STATE_ALIASES", "1")
            else:
                monkeypatch.delenv("HEYOKA_AMD_NO_STATE_ALIASES", raising=False)
            ta = hy.taylor_adaptive_batch(hy.model.np1body(6, masses=M, Gconst=G), st, n, high_accuracy=ha)
            assert ta.hip_source_mode.startswith(which), ta.hip_source_mode
            ta.propagate_until(50.0)
            res[which] = (ta.state.copy(), [r[3] for r in ta.propagate_res])
        assert res["cluster"][1] == res["table"][1]
        assert rel_err(res["cluster"][0], res["table"][0]) <= 1e5 * EPS


@pytest.mark.gpu
def test_nbody_with_parameter_masses_runs_on_the_cluster_kernel():
    """model::nbody with kw::masses = par[...] (the non-grouped branch of src/model/nbody.cpp:131-150): the pair clusters
    differ only by the indices of the parameters they read (per-lane parameter tables) and by constant u variables
    (-par[i]) in linear position: wave-cluster stepper instead of the table-driven one; results = those of the oracle and of
    the same system with numerical masses."""
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 48
    st = configs.outer_ss_state(n, perturb=1e-6, seed=13)
    pars = np.repeat(np.asarray(M, dtype=np.float64)[:, None], n, axis=1)
    pars[1:] *= 1 + 1e-3 * np.arange(n) / n  # different planet masses in every system
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=[hy.par[i] for i in range(6)], Gconst=G), st, n, pars=pars,
                                  high_accuracy=True)
    assert ta.hip_source_mode.startswith("cluster") and "v2" in ta.hip_source_mode, ta.hip_source_mode
    oi = ho.OracleIntegrator(ho.nbody(6, masses=[ho.par(i) for i in range(6)], Gconst=G), st, n, pars=pars,
                             high_accuracy=True)
    ta.step(write_tc=True)
    oi.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in oi.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = oi.tc.reshape(36, oi.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=0, keepdims=True)
    assert np.max(np.abs(np.asarray(ta.tc).reshape(36, oi.order + 1, n) - tc_o) / scale) <= 1e6 * EPS
    assert rel_err(ta.state, oi.state.reshape(36, n)) <= 1e5 * EPS
    ta.propagate_until(25.0)
    oi.propagate_until(25.0)
    assert rel_err(ta.state, oi.state.reshape(36, n)) <= 1e7 * EPS
    # Lane 0 carries the nominal masses: same trajectory as the numerical-mass system (cluster v3).
    tn = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    tn.step()
    tn.propagate_until(25.0)
    assert rel_err(ta.state[:, 0], tn.state[:, 0]) <= 1e7 * EPS


@pytest.mark.gpu
def test_test_particles_next_to_parameter_masses_padded_clusters():
    """model::nbody(6, masses = par[0..3]): two test particles next to four massive bodies. The massive - test-particle pair
    clusters lack the products acting on the (massless) partner: they are padded in the internal program to the shape of
    the massive - massive clusters (pad_clusters(), subgraph embedding) and the system runs on the wave-cluster stepper
    instead of the table-driven one; results = the oracle's on the user-visible (unpadded) decomposition."""
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n = 40
    st = configs.outer_ss_state(n, perturb=1e-6, seed=19)
    pars = np.repeat(np.asarray(M[:4], dtype=np.float64)[:, None], n, axis=1)
    sys_g = hy.model.nbody(6, masses=[hy.par[i] for i in range(4)], Gconst=G)
    sys_o = ho.nbody(6, masses=[ho.par(i) for i in range(4)], Gconst=G)
    assert hy.taylor_decompose_sys(sys_g) == ho.dc_to_strings(ho.taylor_decompose_sys(sys_o))
    ta = hy.taylor_adaptive_batch(sys_g, st, n, pars=pars, high_accuracy=True)
    assert ta.hip_source_mode.startswith("cluster") and "padded" in ta.hip_source_mode, ta.hip_source_mode
    oi = ho.OracleIntegrator(sys_o, st, n, pars=pars, high_accuracy=True)
    ta.step(write_tc=True)
    oi.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in oi.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = oi.tc.reshape(36, oi.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=0, keepdims=True)
    assert np.max(np.abs(np.asarray(ta.tc).reshape(36, oi.order + 1, n) - tc_o) / scale) <= 1e6 * EPS
    ta.propagate_until(25.0)
    oi.propagate_until(25.0)
    assert rel_err(ta.state, oi.state.reshape(36, n)) <= 1e7 * EPS


@pytest.mark.gpu
@pytest.mark.parametrize("masses,t_end,expect", [([1.0, 3e-4, 1e-4, 5e-5, 0.0, 0.0], 10.0, ""), ([1.0, 1e-3, 0.0], 10.0, ""),
                                                  # Unit factors G m_j = 1 are elided by the model (the products read r^-3
                                                  # directly) next to pairs which keep their scaling: insert_unit_scalings().
                                                  ([1.0, 1.0, 1.0, 1.0, 0.0, 0.0], 2.0, "unit scalings"),
                                                  ([1.0, 2.0, 1.0, 0.0], 2.0, "unit scalings")])
def test_test_particles_next_to_numeric_masses_with_a_unit_mass(masses, t_end, expect):
    """model::nbody() with numeric masses, G = 1 and a unit-mass primary: -G m_0 is then the literal -1, which the
    decomposition shows as a negation in the pairs of body 0 with the test particles, next to ordinary factors in the
    other pairs. The planner treats the factor as a per-cluster constant (not as part of the cluster shape), so that
    the clusters are isomorphic and the system runs on the wave-cluster stepper."""
    nb = len(masses)
    n = 24
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
    sys_g = hy.model.nbody(nb, masses=masses, Gconst=1.0)
    sys_o = ho.nbody(nb, masses=masses, Gconst=1.0)
    assert hy.taylor_decompose_sys(sys_g) == ho.dc_to_strings(ho.taylor_decompose_sys(sys_o))
    ta = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=True)
    assert ta.hip_source_mode.startswith("cluster") and expect in ta.hip_source_mode, ta.hip_source_mode
    oi = ho.OracleIntegrator(sys_o, st, n, high_accuracy=True)
    ta.step(write_tc=True)
    oi.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in oi.step_res])
    assert np.max(np.abs(h_g - h_o) / np.abs(h_o)) <= 1e6 * EPS
    tc_o = oi.tc.reshape(6 * nb, oi.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=0, keepdims=True)
    # NOTE: 1e7, not the 1e6 of the other model tests: in the three-body case the coefficients fall to 1e-26 at order 18 and
    # the rounding differences of the FMA contraction grow by ~100x every four orders through the pow recurrence - the
    # same 1.2e6 - 1.4e6 eps at order 18 on the wave-cluster and on the fully unrolled stepper
    # (profiles/experiments/dbg5.py), the step sizes agree to 1e6 eps.
    assert np.max(np.abs(np.asarray(ta.tc).reshape(6 * nb, oi.order + 1, n) - tc_o) / scale) <= 1e7 * EPS
    ta.propagate_until(t_end)
    oi.propagate_until(t_end)
    assert rel_err(ta.state, oi.state.reshape(6 * nb, n)) <= 1e7 * EPS


def _oracle_on_program(monkeypatch, lines, n_eq, st, n, **kw):
    """The oracle's interpreter on an explicit flattened program (the text of taylor_adaptive_batch.internal_program:
    one node per line, then the definitions of the state derivatives)."""
    import re

    def operand(t):
        t = t.strip()
        if t.startswith("u_"):
            return ho.var(t)
        if re.fullmatch(r"p\d+", t):
            return ho.par(int(t[1:]))
        return ho.num(float(t))

    dc = [(ho.var("u_%d" % i), []) for i in range(n_eq)]
    for ln in lines[: len(lines) - n_eq]:
        m = re.fullmatch(r"(\w+)\((.*?)\)((?: \[dep \d+\])*)", ln)
        assert m, ln
        deps = [int(d) for d in re.findall(r"\[dep (\d+)\]", m.group(3))]
        dc.append((ho.func(m.group(1), [operand(t) for t in m.group(2).split(",")]), deps))
    for ln in lines[len(lines) - n_eq :]:
        dc.append((operand(ln), []))
    monkeypatch.setattr(ho, "taylor_decompose_sys", lambda sys, sv_funcs=None: dc)
    try:
        return ho.OracleIntegrator([None] * n_eq, st, n, **kw)
    finally:
        monkeypatch.undo()


def _fixed_centres_special(nc):
    """Fixed centres with one centre at the origin (0 - x becomes -1 * x), repeated coordinates (shared differences), a
    zero coordinate elsewhere and one product G m = 1 (elided scaling)."""
    rng = np.random.default_rng(5)
    pos = rng.uniform(-2.0, 2.0, (nc, 3))
    pos[0] = 0.0
    pos[3, 0] = pos[2, 0]
    pos[5, 1] = 0.0
    pos[6, 2] = pos[1, 2]
    m = rng.uniform(0.1, 1.0, nc)
    m[0] = 1.0
    return [float(v) for v in m], [float(v) for v in pos.reshape(-1)]


@pytest.mark.parametrize("case", ["unit_scalings", "unit_scalings_padded", "padded_par_masses", "state_aliases",
                                  "private_inputs_cluster", "private_inputs_block"])
def test_planner_rewrites_of_the_internal_program_do_not_change_the_jets(case, monkeypatch):
    """The planner of the wave-cluster kernels may rewrite the INTERNAL program (never the decomposition the user sees):
    alias u variables for state variables in history-operand position (add_state_aliases()), padding of clusters which
    are sub-shapes of the largest one (pad_clusters()), unit scalings restored where the model elided a factor 1
    (insert_unit_scalings()). The rewritten program is exposed as text; the oracle's interpreter runs it next to the
    original decomposition: the Taylor coefficients of the state variables, the step size and the new state must be
    IDENTICAL, bit for bit (the added nodes are exact copies / x - 0 / x * 1.0, or are read by nobody)."""
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    n, pars = 6, None
    if case == "unit_scalings":
        sys_g, sys_o = hy.model.nbody(4, masses=[1.0, 2.0, 1.0, 0.0]), ho.nbody(4, masses=[1.0, 2.0, 1.0, 0.0])
    elif case == "unit_scalings_padded":
        m = [1.0, 1.0, 1.0, 1.0, 0.0, 0.0]
        sys_g, sys_o = hy.model.nbody(6, masses=m), ho.nbody(6, masses=m)
    elif case == "padded_par_masses":
        sys_g = hy.model.nbody(6, masses=[hy.par[i] for i in range(4)], Gconst=G)
        sys_o = ho.nbody(6, masses=[ho.par(i) for i in range(4)], Gconst=G)
        pars = np.repeat(np.asarray(M[:4], dtype=np.float64)[:, None], n, axis=1)
    elif case.startswith("private_inputs"):
        # privatise_cluster_inputs(): private copies of the coordinate differences per distance cluster, -1 * x as
        # -0.0 - x (+ unit scalings on top): 12 centres -> wave-cluster kernel, 70 -> block mode.
        mm, pp = _fixed_centres_special(12 if case.endswith("cluster") else 70)
        sys_g = hy.model.fixed_centres(Gconst=1.0, masses=mm, positions=pp)
        sys_o = ho.fixed_centres(Gconst=1.0, masses=mm, positions=pp)
    else:
        sys_g, sys_o = hy.model.np1body(6, masses=M, Gconst=G), ho.np1body(6, masses=M, Gconst=G)
    n_eq = len(sys_o)
    rng = np.random.default_rng(3)
    st = rng.uniform(-1.0, 1.0, (n_eq, n)) + np.arange(n_eq)[:, None] * 0.37
    kw = {"high_accuracy": True}
    if pars is not None:
        kw["pars"] = pars
    ta = hy.taylor_adaptive_batch(sys_g, st, n, **kw)
    prog = ta.internal_program
    mode = ta.hip_source_mode
    assert mode.startswith("block" if case == "private_inputs_block" else "cluster"), mode
    assert len(prog) > len(ta.decomposition) - n_eq, mode
    if case.startswith("private_inputs"):
        assert "private coordinate differences" in mode and any(ln.startswith("sub(-0, u_") for ln in prog)
    plain = ho.OracleIntegrator(sys_o, st, n, **kw)
    rewritten = _oracle_on_program(monkeypatch, prog, n_eq, st, n, **kw)
    assert rewritten.n_u > plain.n_u
    plain.step(wtc=True)
    rewritten.step(wtc=True)
    assert plain.step_res == rewritten.step_res
    assert np.array_equal(plain.tc, rewritten.tc) and np.array_equal(plain.state, rewritten.state)
    assert np.array_equal(plain.time_hi, rewritten.time_hi)


@pytest.mark.gpu
def test_events_on_a_program_with_private_cluster_inputs():
    """model::fixed_centres with zero / repeated coordinates (internal program rewritten by privatise_cluster_inputs()) AND
    events whose equations are exactly the shared / negated differences: the wave-cluster stepper with events on the
    rewritten program + hy_ev_jets on the decomposition, step by step against the oracle's stepper with events."""
    mm, pp = _fixed_centres_special(12)
    n = 11
    st = _lanes([5.0, 0.3, -0.2, 0.0, 1.0, 0.1], n, 1e-2, 6)
    log_p, log_o = [], []
    xg, yg = hy.make_vars("x", "y")
    xo, yo = ho.var("x"), ho.var("y")
    ta = hy.taylor_adaptive_batch(hy.model.fixed_centres(Gconst=1.0, masses=mm, positions=pp), st, n,
                                  nt_events=[hy.nt_event(pp[6] - xg, lambda ta, t, d, i: log_p.append((i, 0, t, d))),
                                             hy.nt_event(-1.0 * yg, lambda ta, t, d, i: log_p.append((i, 1, t, d)))])
    assert ta.hip_source_mode.startswith("cluster") and "private coordinate differences" in ta.hip_source_mode
    assert "events:" in ta.hip_source_mode
    ora = ho.OracleEventIntegrator(ho.fixed_centres(Gconst=1.0, masses=mm, positions=pp), st, n,
                                   nt_events=[ho.nt_event(ho.num(pp[6]) - xo, lambda ta, t, d, i: log_o.append((i, 0, t, d))),
                                              ho.nt_event(ho.num(-1.0) * yo, lambda ta, t, d, i: log_o.append((i, 1, t, d)))])
    for _ in range(60):
        ta.step()
        ora.step()
        assert [int(oc) for oc, _ in ta.step_res] == [oc for oc, _ in ora.step_res]
        h_p = np.array([h for _, h in ta.step_res])
        h_o = np.array([h for _, h in ora.step_res])
        assert np.max(np.abs(h_p - h_o) / np.abs(h_o)) <= 1e6 * EPS
        assert rel_err(ta.state, ora.state.reshape(6, n)) <= 1e6 * EPS
    assert len(log_p) >= n and [(a[0], a[1], a[3]) for a in log_p] == [(a[0], a[1], a[3]) for a in log_o]
    assert np.max(np.abs(np.array([a[2] for a in log_p]) - np.array([a[2] for a in log_o]))) <= 1e-10


@pytest.mark.parametrize("case", ["default_masses_6", "default_masses_8", "repeated_masses_6", "equal_masses_G_5"])
def test_linearised_accelerations_of_the_internal_program(case, monkeypatch):
    """linearise_accelerations(): with equal (default) or repeated masses model::nbody() writes the accelerations as sum / sub
    / negation trees, or with the reactions as glue nodes in front of the sums - shapes which kept those systems off the
    one-lane-per-pair kernel. The planner flattens them in the INTERNAL program into plain sums of scaled pair products (the
    shape of the distinct-mass decomposition; unit scalings restored on top where unit masses sit next to others). Unlike the
    other rewrites this one RE-ASSOCIATES additions: the oracle's interpreter on the rewritten program agrees with the
    decomposition to rounding (1e3 eps on the Taylor coefficients, 1e4 eps on the step size), not bit for bit."""
    if case == "default_masses_6":
        sys_g, sys_o = hy.model.nbody(6), ho.nbody(6)
    elif case == "default_masses_8":
        sys_g, sys_o = hy.model.nbody(8), ho.nbody(8)
    elif case == "repeated_masses_6":
        m = [1.0, 1e-3, 1.0, 2.0, 1e-3, 0.5]
        sys_g, sys_o = hy.model.nbody(6, masses=m), ho.nbody(6, masses=m)
    else:
        sys_g, sys_o = hy.model.nbody(5, masses=[0.3] * 5, Gconst=2.5), ho.nbody(5, masses=[0.3] * 5, Gconst=2.5)
    n_eq, n = len(sys_o), 6
    rng = np.random.default_rng(11)
    nb = n_eq // 6
    st = rng.uniform(-1.0, 1.0, (n_eq, n)) * 0.3
    st[0::6] += 5.0 * np.arange(nb)[:, None]
    st[1::6] += 2.0 * np.arange(nb)[:, None] ** 2 % 7
    ta = hy.taylor_adaptive_batch(sys_g, st, n, high_accuracy=True)
    mode, prog = ta.hip_source_mode, ta.internal_program
    assert "cluster mode v5" in mode and "accelerations rewritten as flat sums" in mode, mode
    # The shape of the distinct-mass decomposition: one sum of N - 1 terms per acceleration, reactions as prod(c, product).
    sums = [ln for ln in prog if ln.startswith("sum(")]
    assert len(sums) == n_eq // 2 and all(ln.count("u_") == nb - 1 for ln in sums)
    assert not any(ln.startswith("sub(u_") and int(ln[6:].split(",")[0].rstrip(")")) >= n_eq for ln in prog)  # (no sub over u variables)
    plain = ho.OracleIntegrator(sys_o, st, n, high_accuracy=True)
    rewritten = _oracle_on_program(monkeypatch, prog, n_eq, st, n, high_accuracy=True)
    plain.step(wtc=True)
    rewritten.step(wtc=True)
    h_p, h_r = np.array([h for _, h in plain.step_res]), np.array([h for _, h in rewritten.step_res])
    assert np.max(np.abs(h_p - h_r) / np.abs(h_p)) <= 1e4 * 2.220446049250313e-16
    tc_p, tc_r = plain.tc.reshape(n_eq, -1, n), rewritten.tc.reshape(n_eq, -1, n)
    scale = np.max(np.abs(tc_p), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(tc_p - tc_r) / scale) <= 1e3 * 2.220446049250313e-16
    # (Re-associated sums: default masses differ in the last bits; with repeated masses only the reactions moved - same bits.)
    if case == "repeated_masses_6":
        assert np.array_equal(tc_p, tc_r)


def test_externalised_scalings_of_the_internal_program(monkeypatch):
    """externalise_scalings(): model::nbody() with DISTINCT masses and more than 64 pairs keeps scaling, scaled products and
    reactions inside the pair clusters - not the shape of block mode's v2 cluster phase. The planner moves them out in the
    INTERNAL program (every product reads r^-3 directly, c * (d * r^-3) for d * (c * r^-3), reactions (c' c) * (d * r^-3)): the
    clusters are those of the equal-mass system. The oracle's interpreter on the rewritten program agrees with the
    decomposition to rounding (1e3 eps on the Taylor coefficients, 1e4 eps on the step size)."""
    nb, n = 12, 5
    masses = list(1.0 / (1.0 + np.arange(nb)) ** 2 * nb / 4.0)
    sys_g, sys_o = hy.model.nbody(nb, masses=masses), ho.nbody(nb, masses=masses)
    n_eq = 6 * nb
    rng = np.random.default_rng(12)
    st = rng.uniform(-1.0, 1.0, (n_eq, n)) * 0.3
    st[0::6] += 5.0 * np.arange(nb)[:, None]
    st[1::6] += 2.0 * np.arange(nb)[:, None] ** 2 % 7
    ta = hy.taylor_adaptive_batch(sys_g, st, n)
    mode, prog = ta.hip_source_mode, ta.internal_program
    assert "v2 cluster phase" in mode and "scalings of the pair products moved out of the clusters" in mode, mode
    n_pairs = nb * (nb - 1) // 2
    # No scaled power any more (number * pow), six scaled unit products per pair.
    u_of = {n_eq + i: ln for i, ln in enumerate(prog[: len(prog) - n_eq])}
    scaled = [ln for ln in prog if ln.startswith("prod(") and not ln.startswith("prod(u_")]
    assert len(scaled) == 6 * n_pairs
    for ln in scaled:
        src = int(ln.split("u_")[1].rstrip(")"))
        assert u_of[src].startswith("prod(u_"), (ln, u_of[src])
    plain = ho.OracleIntegrator(sys_o, st, n)
    rewritten = _oracle_on_program(monkeypatch, prog, n_eq, st, n)
    plain.step(wtc=True)
    rewritten.step(wtc=True)
    h_p, h_r = np.array([h for _, h in plain.step_res]), np.array([h for _, h in rewritten.step_res])
    assert np.max(np.abs(h_p - h_r) / np.abs(h_p)) <= 1e4 * 2.220446049250313e-16
    tc_p, tc_r = plain.tc.reshape(n_eq, -1, n), rewritten.tc.reshape(n_eq, -1, n)
    scale = np.max(np.abs(tc_p), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(tc_p - tc_r) / scale) <= 1e3 * 2.220446049250313e-16


# ------------------------------------------------------------------------------------------------------------------
# Mixed models (round 6): nonlinear sub-DAGs of SEVERAL shapes in one system. The planner groups the clusters into
# classes of identical shape and dependency level and runs every class as its own section of straight-line code, cluster i
# of a class on lane i of the system's lane group (heyoka_amd/csrc/hip_emit_cluster.cpp, cluster_class); what does not fit
# the register file goes to the staged table stepper (tape in LDS, hip_emit_staged.cpp). The reference runs any
# decomposition through its segment / block tables at full SIMD rate (src/taylor_02.cpp:105-207, :983-1189).
# ------------------------------------------------------------------------------------------------------------------
from heyoka_amd import mixed_models as mm  # noqa: E402


def _mixed_cases():
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    cen, ch = mm.lattice_centres_setup()
    return {
        # name: (system builder over an expression module, state builder, horizon, expected stepper)
        "sine_lattice16": (lambda m: mm.sine_lattice(m, 16), lambda n: mm.sine_lattice_state(16, n), 4.0, "classes of clusters"),
        "lattice_centres12": (lambda m: mm.lattice_centres(m, cen, ch), mm.lattice_centres_state, 6.0, "classes of clusters"),
        # (17 histories per lane: beyond the register file - the staged table stepper.)
        "nbody6_j2": (lambda m: mm.nbody_j2(m, 6, M, G, 1e-7), lambda n: configs.outer_ss_state(n, perturb=1e-6, seed=3), 15.0,
                      "table mode (staged)"),
    }


@pytest.mark.parametrize("name", sorted(_mixed_cases()))
def test_mixed_models_decomposition_and_planner(name, monkeypatch):
    """Decomposition identical to the oracle's, and the stepper the planner picks: the multi-class wave-cluster stepper where
    the histories of one cluster of every class fit the register file of a lane, the staged table stepper otherwise (and
    with HEYOKA_AMD_MULTI_CLASS=0)."""
    build, _, _, expect = _mixed_cases()[name]
    sg = build(hy)
    assert hy.taylor_decompose_sys(sg) == ho.dc_to_strings(ho.taylor_decompose_sys(build(ho)))
    mode = hy.taylor_adaptive_batch(sg, None, 1 << 18).hip_source_mode
    assert expect in mode, mode
    if expect == "classes of clusters":
        assert mode.startswith("cluster")
        monkeypatch.setenv("HEYOKA_AMD_MULTI_CLASS", "0")
        assert "table mode (staged)" in hy.taylor_adaptive_batch(sg, None, 1 << 18).hip_source_mode
    else:
        assert "multi-class plan: the jets of the cluster classes do not fit in the register file" in mode


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_mixed_cases()))
@pytest.mark.parametrize("contract", [True, False])
def test_mixed_models_step_and_propagate_vs_oracle(name, contract, monkeypatch):
    """One full-order step (h, Taylor coefficients, state) and a propagation of the mixed models against the oracle, at the
    tolerances of test_models_step_and_propagate_vs_oracle (default build: h 1e6 eps, coefficients 1e6 eps of the row
    maximum, states 1e5 eps of the row maximum after one step and 1e7 eps after the propagation; without FMA contraction
    the reference's own tolerances, test/two_body_batch.cpp:118-150)."""
    build, state, T, expect = _mixed_cases()[name]
    n = 48
    st = state(n)
    if not contract:
        monkeypatch.setenv("HEYOKA_AMD_HIPRTC_FLAGS", "-ffp-contract=off")
    h_tol, tc_tol = (1e6, 1e6) if contract else (1e4, 1e5)
    ta = hy.taylor_adaptive_batch(build(hy), st, n)
    assert expect in ta.hip_source_mode, ta.hip_source_mode
    oi = ho.OracleIntegrator(build(ho), st, n)
    n_eq = st.shape[0]
    ta.step(write_tc=True)
    oi.step(wtc=True)
    h_g = np.array([h for _, h in ta.step_res])
    h_o = np.array([h for _, h in oi.step_res])
    assert all(o == hy.taylor_outcome.success for o, _ in ta.step_res)
    # (The step size is a ratio of norms of the last two coefficients. A first version of the pendulum chain wrote the cube
    # of a bond as pow(d, 3), whose recurrence divides by d^[0]: with bonds swinging through d = 0 the oracle's own default and
    # compact flavours disagreed by parts in a thousand on h and by several steps per propagation - the model now multiplies.
    # Lanes in which the two flavours of the oracle disagree are ill-conditioned by that very fact and only compared on the
    # coefficients / state; there should be none.)
    oc = ho.OracleIntegrator(build(ho), st, n, compact_mode=True)
    oc.step()
    h_c = np.array([h for _, h in oc.step_res])
    well = np.abs(h_c - h_o) / np.abs(h_o) <= 1e3 * EPS
    assert np.count_nonzero(well) >= n - 4
    assert np.max(np.abs(h_g - h_o)[well] / np.abs(h_o)[well]) <= h_tol * EPS
    tc_o = oi.tc.reshape(n_eq, oi.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(ta.tc).reshape(n_eq, oi.order + 1, n) - tc_o) / scale) <= tc_tol * EPS

    def row_err(a, b):
        return np.max(np.max(np.abs(a - b), axis=1) / (np.max(np.abs(b), axis=1) + 1e-300))

    assert row_err(np.asarray(ta.state)[:, well], oi.state.reshape(n_eq, n)[:, well]) <= 1e5 * EPS
    ta.propagate_until(T)
    oi.propagate_until(T)
    assert all(r[0] == hy.taylor_outcome.time_limit for r in ta.propagate_res)
    assert max(abs(a[3] - b[3]) for a, b in zip(ta.propagate_res, oi.prop_res)) <= 1
    assert row_err(np.asarray(ta.state), oi.state.reshape(n_eq, n)) <= 1e7 * EPS
