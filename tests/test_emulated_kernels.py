"""CPU tests (no GPU): the generated stepper kernels, compiled for the HOST and run under the wavefront emulator of
tests/emu (fibres with rendezvous at the cross-lane operations), against the oracle. TEST INFRASTRUCTURE: the emulator is
a checker of the generators' arithmetic and exchange logic in the authoring container - the proof for the hardware stays
with the -m gpu parity tests, which run the same comparisons through the C ABI on an MI355X."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

import emu  # noqa: E402
import heyoka_oracle as ho  # noqa: E402

import heyoka_amd as hy  # noqa: E402
from heyoka_amd import configs  # noqa: E402

EPS = 2.220446049250313e-16
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))


def _outer_ss(kernel, monkeypatch=None, **kw):
    # ("v5-nofrx": the one-lane-per-pair kernel with the reactions exported by the pair lanes and the consumer-arranged
    # operand arrays read with 128-bit loads - the layout the stepper with events keeps.)
    opts = None
    if kernel.endswith("-nofrx"):
        kernel, opts = kernel[:-6], "nofrx"
    old = os.environ.get("HEYOKA_AMD_V5_OPTS")
    if opts is not None:
        os.environ["HEYOKA_AMD_V5_OPTS"] = opts
    try:
        ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, 64, high_accuracy=True, cluster_kernel=kernel, **kw)
    finally:
        if opts is not None:
            if old is None:
                del os.environ["HEYOKA_AMD_V5_OPTS"]
            else:
                os.environ["HEYOKA_AMD_V5_OPTS"] = old
    assert kernel in ta.hip_source_mode, ta.hip_source_mode
    if kernel == "v5":
        assert ("const hy_d2 w" in ta.hip_source) == (opts == "nofrx")  # (128-bit operand reads: only in the nofrx layout)
    return ta


@pytest.mark.parametrize("kernel", ["v5", "v5-nofrx", "v3", "v2"])
def test_emulated_cluster_kernels_single_step_vs_oracle(kernel):
    """One Taylor step of 11 perturbed outer Solar Systems (ragged: the last wavefront holds replicas) with the strict
    (no contraction) host build of the generated source: step sizes to 1e4 eps, states and Taylor coefficients to 1e5 eps
    of the oracle's - the tolerances of the GPU parity tests built with -ffp-contract=off."""
    n = 11
    st = configs.outer_ss_state(n, perturb=1e-6, seed=5)
    ta = _outer_ss(kernel)
    k = emu.EmulatedKernel(ta.hip_source)
    r = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=36 * (ta.order + 1))
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    ora.step(wtc=True)
    h_o = np.array([h for _, h in ora.step_res])
    assert rel_err(r["last_h"], h_o) <= 1e4 * EPS
    assert rel_err(r["state"], ora.state.reshape(36, n)) <= 1e5 * EPS
    assert rel_err(r["time_hi"], ora.time_hi) <= 1e4 * EPS
    tc_o = ora.tc.reshape(36, ta.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(r["tc"].reshape(tc_o.shape) - tc_o) / scale) <= 1e5 * EPS


def test_emulated_v5_propagation_through_the_work_queue_with_refill():
    """propagate_until() of 37 systems with per-system final times through ONE workgroup (32 systems in flight): the
    device-side queue, the retire / refill path of the one-lane-per-pair kernel and the frozen bookkeeping of finished
    systems, lane by lane against the oracle's ensemble driver."""
    n = 37
    rng = np.random.RandomState(3)
    st = configs.outer_ss_state(n, perturb=1e-6, seed=77)
    tf = 4.0 * rng.uniform(0.2, 1.5, n)
    ta = _outer_ss("v5")
    k = emu.EmulatedKernel(ta.hip_source)
    r = k.run(st, np.zeros(n), np.zeros(n), mode=1, tfin=tf, max_grid=1)
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    ora.propagate_until(tf)
    assert np.array_equal(r["outcome"], np.array([int(p[0]) for p in ora.prop_res]))
    assert np.array_equal(r["time_hi"], tf)
    ns_o = np.array([int(p[3]) for p in ora.prop_res])
    assert np.abs(r["n_steps"].astype(np.int64) - ns_o).max() <= 1
    assert rel_err(r["state"], ora.state.reshape(36, n)) <= 1e5 * EPS


def test_emulated_v5_taylor_coefficients_by_threshold():
    """Single-step launch with hy_kargs::pad bit 2 (the lock-step loop of propagate_grid() without a callback): a.tfin_hi holds a
    time per system, and only the systems whose step reaches it store their Taylor coefficients - bit for bit the
    coefficients of the launch which stores them all; state, step sizes and times do not depend on it."""
    n = 11
    st = configs.outer_ss_state(n, perturb=1e-6, seed=5)
    ta = _outer_ss("v5")
    k = emu.EmulatedKernel(ta.hip_source)
    rows = 36 * (ta.order + 1)
    full = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=rows)
    thr = np.where(np.arange(n) % 2 == 0, 0.01, 1e9)  # (a step of the outer Solar System is ~0.4 yr)
    part = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=rows, tfin=thr, pad=4)
    for key in ("state", "last_h", "time_hi", "time_lo"):
        assert np.array_equal(full[key], part[key]), key
    reached = full["time_hi"] >= thr
    assert reached.any() and not reached.all()
    assert np.array_equal(part["tc"][:, reached], full["tc"][:, reached])
    assert np.all(part["tc"][:, ~reached] == 0.0)
    # Backward in time: the comparison turns around.
    fullb = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, -np.inf), want_tc_rows=rows)
    partb = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, -np.inf), want_tc_rows=rows, tfin=-thr, pad=4)
    assert np.array_equal(partb["tc"][:, reached], fullb["tc"][:, reached])
    assert np.all(partb["tc"][:, ~reached] == 0.0)
    # The LAST step of a lane is clamped to the remaining time rem.hi and ends at tlast - rem.lo: with a grid ending at 0
    # (or crossing 0) the high part of the new time lands on either side of the last grid time, while hy_grid_post treats
    # the lane as done (h == rem.hi) and evaluates every remaining grid point from these coefficients - a clamped step
    # always stores (round-5 advisor finding: t = (-0.05, -+3e-18), limit 0.05, last grid time 0).
    for lo in (-3e-18, 3e-18):
        t_hi, t_lo = np.full(n, -0.05), np.full(n, lo)
        fullc = k.run(st, t_hi, t_lo, mode=0, lim=np.full(n, 0.05), want_tc_rows=rows)
        partc = k.run(st, t_hi, t_lo, mode=0, lim=np.full(n, 0.05), want_tc_rows=rows, tfin=np.zeros(n), pad=4)
        assert np.all(fullc["last_h"] == 0.05)
        assert np.all(np.sign(fullc["time_hi"]) == np.sign(lo))  # (the new time is -+3e-18: both sides of the grid time)
        assert np.array_equal(partc["tc"], fullc["tc"]), lo


@pytest.mark.parametrize("nb,masses", [(3, None), (4, None), (3, [1.0, 1e-3, 3e-4]), (5, None), (6, None), (6, [3.0, 1e-3, 0.7, 2.0, 4e-3, 0.5]), (6, "default"),
                                       (6, [1.0, 1e-3, 1.0, 2.0, 1e-3, 0.5]), (7, "default"), (9, "default")])
def test_emulated_v5_other_systems_single_step_vs_oracle(nb, masses):
    """The one-lane-per-pair kernel with the reactions fused into the sums on other pair-interaction systems: 5 bodies (10
    pairs on 16 lanes, 15 sums: ONE glue round; 3 and 4 bodies - round 6: 3 pairs on 8 lanes, 6 pairs on 16, more lanes per
    system than pairs so that the jets of the systems of a CU fit in its LDS) and 6 bodies with other mass ratios than the outer Solar System's (comparable
    masses: reaction coefficients of order one). (Equal masses take model::nbody() to its grouped branch, whose clusters
    have another shape: those systems run on the lane-pair / first-generation kernels.)"""
    rng = np.random.RandomState(40 + nb)
    if masses is None:
        masses = list(1.0 / (1.0 + np.arange(nb)) ** 2)
    kw_g = {} if masses == "default" else {"masses": masses}
    n = 13
    pos = rng.uniform(-3.0, 3.0, (nb, 3, n)) + 6.0 * np.arange(nb)[:, None, None] * np.array([1.0, 0.3, -0.2])[None, :, None]
    vel = rng.uniform(-0.3, 0.3, (nb, 3, n))
    st = np.concatenate([np.concatenate([pos[b], vel[b]], axis=0) for b in range(nb)], axis=0)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, **kw_g), None, 64, high_accuracy=True, cluster_kernel="v5")
    assert "v5" in ta.hip_source_mode, ta.hip_source_mode
    assert "frc" in ta.hip_source  # (coefficient registers of the fused sums)
    k = emu.EmulatedKernel(ta.hip_source)
    rows = 6 * nb * (ta.order + 1)
    r = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=rows)
    ora = ho.OracleIntegrator(ho.nbody(nb, **kw_g), st, n, high_accuracy=True)
    ora.step(wtc=True)
    h_o = np.array([h for _, h in ora.step_res])
    assert rel_err(r["last_h"], h_o) <= 1e4 * EPS
    assert rel_err(r["state"], ora.state.reshape(6 * nb, n)) <= 1e5 * EPS
    tc_o = ora.tc.reshape(6 * nb, ta.order + 1, n)
    scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(r["tc"].reshape(tc_o.shape) - tc_o) / scale) <= 1e5 * EPS


def test_emulated_multi_class_cluster_stepper_and_staged_table_stepper_vs_oracle():
    """Round 6. (a) A system whose clusters come in two shapes - a chain of pendula with cubic bonds: 16 sin / cos pairs of
    state variables, 15 cubes (as products) of differences - on the multi-class wave-cluster stepper (one section of code per class of
    clusters): built without contraction it keeps the reference's operation order, the Taylor coefficients of one step are
    the default-mode oracle's BIT FOR BIT. (b) The staged table stepper (tape in LDS, lanes
    over the nodes of a group) on the outer Solar System: with the strict order of the additions (kw::compact_mode) the
    COMPACT-mode oracle's coefficients bit for bit, with the terms of the convolutions dealt to several lanes 1e5 eps; a
    propagation through the device-side queue against the oracle."""
    from heyoka_amd import mixed_models as mm

    n, ns = 9, 16
    st = mm.sine_lattice_state(ns, n)
    ta = hy.taylor_adaptive_batch(mm.sine_lattice(hy, ns), None, 64)
    assert "2 classes of clusters" in ta.hip_source_mode, ta.hip_source_mode
    p = ta.order
    k = emu.EmulatedKernel(ta.hip_source)
    r = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=2 * ns * (p + 1),
              scratch_per_wave=(p + 1) * 4 * 64)
    ora = ho.OracleIntegrator(mm.sine_lattice(ho, ns), st, n)
    ora.step(wtc=True)
    # (The Taylor coefficients bit for bit; the step size goes through exp(log()) in the kernel and pow() in the oracle: a few
    # ulps, and the new state with it.)
    assert np.array_equal(r["tc"].reshape(2 * ns, p + 1, n), ora.tc.reshape(2 * ns, p + 1, n))
    assert rel_err(r["last_h"], np.array([h for _, h in ora.step_res])) <= 16 * EPS
    assert rel_err(r["state"], ora.state.reshape(2 * ns, n)) <= 1e3 * EPS

    n = 7
    st = configs.outer_ss_state(n, perturb=1e-6, seed=5)
    for compact in (True, False):
        ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, 64, high_accuracy=True, compact_mode=compact,
                                      emitter="table")
        assert "table mode (staged)" in ta.hip_source_mode and ("strict order" in ta.hip_source_mode) == compact
        k = emu.EmulatedKernel(ta.hip_source)
        r = k.run(st, np.zeros(n), np.zeros(n), mode=0, lim=np.full(n, np.inf), want_tc_rows=36 * 21)
        ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, compact_mode=True)
        ora.step(wtc=True)
        tc_o = ora.tc.reshape(36, 21, n)
        if compact:
            assert np.array_equal(r["tc"].reshape(tc_o.shape), tc_o)
        else:
            scale = np.max(np.abs(tc_o), axis=2, keepdims=True) + 1e-300
            assert np.max(np.abs(r["tc"].reshape(tc_o.shape) - tc_o) / scale) <= 1e5 * EPS
        tf = np.linspace(1.0, 3.0, n)
        r = k.run(st, np.zeros(n), np.zeros(n), mode=1, tfin=tf, max_grid=2)
        ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True, compact_mode=True)
        ora.propagate_until(tf)
        assert np.array_equal(r["outcome"], np.array([int(q[0]) for q in ora.prop_res]))
        assert np.array_equal(r["n_steps"].astype(np.int64), np.array([int(q[3]) for q in ora.prop_res]))
        assert np.array_equal(r["time_hi"], tf)
        assert rel_err(r["state"], ora.state.reshape(36, n)) <= 1e5 * EPS


def test_emulated_v5_propagation_from_grid_point_to_grid_point():
    """Round 6: propagate-mode launches with hy_kargs::pad bit 2 (emitted_module::grid_multi_step) - the lock-step loop of
    propagate_grid() without a callback takes every system from one grid point to the next in ONE launch: the system leaves
    the step loop after the first step which reaches a.tc_thr[s] (not clamped there: only the last grid time a.tfin clamps),
    with the Taylor coefficients of THAT step stored. Checked: the exit step brackets the threshold, counters and outcomes,
    and the dense output of the stored coefficients at the threshold against the oracle propagated to that time; through the
    retire / refill path of the work queue (37 systems on one workgroup) with per-system thresholds."""
    n = 37
    rng = np.random.RandomState(11)
    st = configs.outer_ss_state(n, perturb=1e-6, seed=21)
    thr = rng.uniform(0.3, 2.5, n)
    ta = _outer_ss("v5")
    k = emu.EmulatedKernel(ta.hip_source)
    p = ta.order
    r = k.run(st, np.zeros(n), np.zeros(n), mode=1, tfin=np.full(n, 100.0), pad=4, tc_thr=thr, want_tc_rows=36 * (p + 1), max_grid=1)
    t1, h = r["time_hi"], r["last_h"]
    assert np.all(r["outcome"] == int(hy.taylor_outcome.success))
    assert np.all(t1 >= thr) and np.all(t1 - h < thr)  # the exit step is the first one which reaches the threshold
    assert np.all(r["n_steps"] >= 1) and np.all(r["n_steps"] == np.ceil(np.round(r["n_steps"])))
    # Dense output at the threshold from the stored coefficients (Horner in extended precision is not needed at 1e-13).
    tc = r["tc"].reshape(36, p + 1, n)
    hd = thr - (t1 - h)
    dense = np.zeros((36, n))
    for kk in range(p, -1, -1):
        dense = dense * hd + tc[:, kk, :]
    ora = ho.OracleIntegrator(ho.nbody(6, masses=M, Gconst=G), st, n, high_accuracy=True)
    ora.propagate_until(thr)
    ref = ora.state.reshape(36, n)
    assert np.max(np.abs(dense - ref) / (np.max(np.abs(ref), axis=1, keepdims=True))) <= 1e5 * EPS
    # The clamp at the last grid time still holds: a threshold beyond it ends with time_limit exactly there.
    r2 = k.run(st[:, :4], np.zeros(4), np.zeros(4), mode=1, tfin=np.full(4, 1.0), pad=4, tc_thr=np.full(4, np.inf), want_tc_rows=36 * (p + 1))
    assert np.all(r2["time_hi"] == 1.0) and np.all(r2["outcome"] == int(hy.taylor_outcome.time_limit))
    # hy_kargs::grid_done: 1 for the systems whose last step was clamped to the remaining time, 0 for a grid-point exit.
    assert np.all(r2["grid_done"] == 1.0) and np.all(r["grid_done"] == 0.0)
