"""C-ABI library: loads, exports every symbol declared in include/heyoka_amd.h, and its host-side
logic (construction, validation, code generation, hiprtc compilation for gfx950) works without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import heyoka_amd as hy
from heyoka_amd import _lib, configs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "heyoka_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hy_[A-Za-z_0-9]+)\s*\(", txt)) - {"hy_step_callback", "hy_ensemble_gen"})


def test_every_declared_symbol_is_exported():
    syms = header_symbols()
    assert len(syms) > 60
    declared_in_py = {s[0] for s in _lib.SIGNATURES}
    for s in syms:
        assert hasattr(_lib.lib, s), "missing export: " + s
        assert s in declared_in_py, "ctypes signature missing for " + s
    assert declared_in_py <= set(syms)


def test_version_and_no_gpu_behaviour():
    assert "gfx950" in hy.version()
    if hy.device_count() == 0:
        x, v = hy.make_vars("x", "v")
        ta = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], [[0.05] * 2, [0.025] * 2], 2)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ta.step()


def test_construction_getters_and_codegen():
    x, v = hy.make_vars("x", "v")
    sys = [(x, v), (v, hy.cos(hy.time) - hy.par[0] * v - hy.sin(x))]
    ta = hy.taylor_adaptive_batch(sys, [[0.01, 0.02, 0.03, 0.04], [1.85, 1.86, 1.87, 1.88]], 4,
                                  pars=[0.10, 0.11, 0.12, 0.13], opt_level=3, fast_math=False)
    assert ta.order == 20 and ta.dim == 2 and ta.batch_size == 4 and ta.n_pars == 1
    assert ta.tol == np.finfo(float).eps and not ta.high_accuracy and not ta.compact_mode
    assert np.array_equal(ta.state, [[0.01, 0.02, 0.03, 0.04], [1.85, 1.86, 1.87, 1.88]])
    assert np.array_equal(ta.pars, [[0.10, 0.11, 0.12, 0.13]])
    assert np.array_equal(ta.time, np.zeros(4))
    src = ta.hip_source
    assert "extern \"C\" __global__" in src and "hy_taylor" in src and "hy_dout" in src
    assert ta.compile_seconds > 0  # hiprtc produced a gfx950 code object
    # Orders from tolerances (reference: taylor_order_from_tol, taylor_common.hpp:165-191).
    for tol, order in ((1.0, 2), (0.5, 2), (0.1, 3), (1e-9, 12), (1e-15, 19)):
        t2 = hy.taylor_adaptive_batch([(x, v), (v, -x)], [[0.0], [1.0]], 1, tol=tol)
        assert t2.order == order
    # Zero-initialised state, scalar / vector time.
    t3 = hy.taylor_adaptive_batch([(x, v), (v, -x)], None, 3, time=2.5)
    assert np.array_equal(t3.state, np.zeros((2, 3))) and np.array_equal(t3.time, [2.5] * 3)
    t3.time = [1.0, 2.0, 3.0]
    assert np.array_equal(t3.time, [1.0, 2.0, 3.0])
    t3.dtime = ([1.0, 2.0, 3.0], [1e-20, 0.0, -1e-20])
    assert np.array_equal(t3.dtime[1], [1e-20, 0.0, -1e-20])
    t3.state = np.arange(6.0).reshape(2, 3)
    c = t3.copy()
    assert np.array_equal(c.state, np.arange(6.0).reshape(2, 3)) and np.array_equal(c.time, [1.0, 2.0, 3.0])


def test_constructor_error_messages():
    """Messages checked verbatim by the reference's tests (test/taylor_adaptive_batch.cpp:955-1016)."""
    x, v = hy.make_vars("x", "v")
    sys = [(x, v), (v, -x)]
    with pytest.raises(ValueError, match="The batch size in an adaptive Taylor integrator cannot be zero"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0], 0)
    with pytest.raises(ValueError, match=r"the state vector has a size of 3, which is not a multiple of the batch size \(2\)"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0, 2.0], 2)
    with pytest.raises(ValueError, match="the state vector has a dimension of 1 and a batch size of 2, while the number of equations is 2"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0], 2)
    with pytest.raises(ValueError, match=r"the time vector has a size of 3, which is not equal to the batch size \(2\)"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0, 2.0, 3.0], 2, time=[0.0, 1.0, 2.0])
    with pytest.raises(ValueError, match="The tolerance in an adaptive Taylor integrator must be finite and positive, but it is -1 instead"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0], 1, tol=-1.0)
    with pytest.raises(ValueError, match="Parallel mode can be activated only in conjunction with compact mode"):
        hy.taylor_adaptive_batch(sys, [0.0, 1.0], 1, parallel_mode=True)
    with pytest.raises(ValueError, match=r"3 parameter value\(s\) were passed, but the ODE system contains 1 parameter\(s\) \(in batches of 2\)"):
        hy.taylor_adaptive_batch([(x, v), (v, -hy.par[0] * x)], [0.0, 0.0, 1.0, 1.0], 2, pars=[1.0, 2.0, 3.0])


def test_codegen_compiles_for_benchmark_dags():
    """HIP generation + hiprtc compilation (no GPU needed) for the small benchmark DAGs."""
    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, 64)
    assert ta.n_uvars == 21 and ta.dim == 12
    ta = hy.taylor_adaptive_batch(hy.model.pendulum(gconst=9.8), None, 64, high_accuracy=True)
    assert ta.n_uvars == 5


def test_cfunc_host_logic_without_gpu():
    """cfunc<double> construction (function_decompose(), src/expression_cfunc.cpp:723-900): decomposition
    structure, properties, error messages, and the generated kernel compiles for gfx950."""
    x, v = hy.make_vars("x", "v")
    cf = hy.cfunc([x * hy.par[1] + hy.cos(hy.time), v * v, x], [x, v])
    assert (cf.nvars, cf.nouts, cf.nparams, cf.is_time_dependent) == (2, 3, 2, True)
    dc = cf.dc
    assert dc[:2] == ["x", "v"] and len(dc) == 2 + 5 + 3
    # Outputs: u variables or (here) a direct reference to an input variable.
    assert dc[-1] == "u_0" and dc[-3].startswith("u_")
    assert "hy_cfunc" in cf.hip_source and "cos(" in cf.hip_source
    # model::nbody_energy: 3 bodies -> 3 pairs.
    sys3 = hy.model.nbody(3)
    en = hy.cfunc([hy.model.nbody_energy(3, masses=[1.0, 2.0, 3.0], Gconst=0.5)], sys3.vars)
    assert en.nvars == 18 and en.nouts == 1 and en.nparams == 0 and not en.is_time_dependent
    assert sum("sum_sq" in l for l in en.dc) == 3 + 3
    with pytest.raises(ValueError, match="appears in the function but not in the user-provided list of variables"):
        hy.cfunc([x + v], [x])
    with pytest.raises(ValueError, match="appears in the user-provided list of variables twice"):
        hy.cfunc([x + v], [x, v, x])
    with pytest.raises(ValueError, match="which is not a variable"):
        hy.cfunc([x + v], [x, v + 1.0])
    with pytest.raises(ValueError, match="Cannot decompose a function with no outputs"):
        hy.cfunc([], [x])
    # Runtime masses switch model::nbody to the non-grouped branch (src/model/nbody.cpp:131-150).
    sp = hy.model.nbody(2, masses=[hy.par[0], hy.par[1]])
    dcs = hy.taylor_decompose_sys(sp)
    assert any("p0" in l for l in dcs) and any("p1" in l for l in dcs)


def test_aux_kernels_compile_for_gfx950():
    """The auxiliary kernels compiled at first use on a GPU (continuous output, propagate_grid post-step, event
    detection) build for gfx950 with hiprtc on the CPU-only box."""
    for order, dim, ha in ((20, 36, 1), (3, 2, 0)):
        _lib.raise_for(_lib.lib.hy_compile_aux_kernels(order, dim, ha))


def test_propagate_until_rejects_time_vectors_of_the_wrong_size():
    """Any number of final times other than the batch size throws like the reference (src/taylor_adaptive_batch.cpp:
    propagate_until_impl()); in particular 2 * batch_size values are NOT read as double-length times (that form is private
    to propagate_for())."""
    import heyoka_amd as hy

    ta = hy.taylor_adaptive_batch(hy.model.pendulum(), None, 4)
    for n in (3, 8, 5):
        with pytest.raises(ValueError, match="the number of specified time limits is %d" % n):
            ta.propagate_until([1.0] * n)


def test_event_integrators_pick_the_cluster_stepper_when_the_event_equations_are_small(monkeypatch):
    """Integrators with events: the wave-cluster stepper (built from the system alone, mode-4 specialisation) + hy_ev_jets
    when the right-hand side qualifies and the event equations depend on a small part of the decomposition - including an
    event equation which IS a state variable; otherwise (event equations which need most of the decomposition, systems
    without a cluster structure, HEYOKA_AMD_EVENTS_ON_CLUSTER=0) the one-system-per-lane steppers with events."""
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    x1, y1, vx1, x2 = hy.make_vars("x_1", "y_1", "vx_1", "x_2")
    cb = lambda *a: None
    ta = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True,
                                  nt_events=[hy.nt_event((x1 - x2) * (x1 - x2) - 4.0, cb), hy.nt_event(y1, cb)],
                                  t_events=[hy.t_event(x1 * vx1)])
    mode = ta.hip_source_mode
    assert mode.startswith("cluster") and "events: 3 event equation(s) evaluated by the stepper" in mode, mode
    src = ta.hip_source
    # Round 4: event equations of up to three nonlinear nodes are evaluated by the stepper itself (the jets of the event
    # equations leave from there, the selector norms stay in registers); the cooperative store of the Taylor coefficients runs
    # for the workgroups which may have an event.
    # (a.sel_norms carries the per-system verdict of the exclusion test to the detection kernel on this path.)
    assert "inside the stepper" in mode and "a.ev_tc[" in src and "__syncthreads_or" in src
    assert "a.sel_norms[s] = ev_possible" in src and "a.sel_norms[(u64)" not in src
    monkeypatch.setenv("HEYOKA_AMD_NO_EVENTS_IN_STEPPER", "1")
    ta2 = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True,
                                   nt_events=[hy.nt_event((x1 - x2) * (x1 - x2) - 4.0, cb), hy.nt_event(y1, cb)],
                                   t_events=[hy.t_event(x1 * vx1)])
    monkeypatch.delenv("HEYOKA_AMD_NO_EVENTS_IN_STEPPER")
    src = ta2.hip_source
    assert "inside the stepper" not in ta2.hip_source_mode and "events: jets of 3 event equation(s)" in ta2.hip_source_mode
    assert "a.sel_norms[" in src and "__syncthreads" in src  # the mode-4 specialisation, cooperative tc store
    # The plain integrator of the same system keeps the propagation kernel (no mode-4 code in it).
    tb = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True)
    assert "sel_norms[s]" not in tb.hip_source and tb.hip_source_mode.startswith("cluster")
    # An event equation that needs most of the decomposition: the energy of the system.
    en = hy.model.nbody_energy(6, masses=M, Gconst=G)
    tc = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True, nt_events=[hy.nt_event(en + 1.0, cb)])
    assert not tc.hip_source_mode.startswith("cluster"), tc.hip_source_mode
    # No cluster structure: the pendulum stays on the unrolled stepper with events.
    x, v = hy.make_vars("x", "v")
    tp = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], None, 4, nt_events=[hy.nt_event(v, cb)])
    assert tp.hip_source_mode.startswith("unrolled")
    monkeypatch.setenv("HEYOKA_AMD_EVENTS_ON_CLUSTER", "0")
    td = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True, nt_events=[hy.nt_event(y1, cb)])
    assert not td.hip_source_mode.startswith("cluster")


def test_isomorphic_terms_of_event_equations_are_evaluated_side_by_side_on_the_lanes(monkeypatch):
    """(CPU: generated source.) Event equations inside the one-lane-per-pair stepper: a sum of terms of one shape over state
    variables of one access class - a squared distance, a radial velocity, also written as a chain of binary sums - is
    evaluated once, term c by lane c (per-lane offsets evo<p> / evz<p>, lane broadcasts in the order of the arguments); terms
    of different shapes, or which share a node, are evaluated one after the other."""
    from heyoka_amd import configs

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    x1, y1, z1, x2, y2, z2, vx1, vy1, vz1 = hy.make_vars("x_1", "y_1", "z_1", "x_2", "y_2", "z_2", "vx_1", "vy_1", "vz_1")
    cb = lambda *a: None

    def src_of(ev):
        ta = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True, nt_events=[hy.nt_event(ev, cb)])
        assert "inside the stepper" in ta.hip_source_mode, ta.hip_source_mode
        return ta.hip_source

    d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 81.0
    s = src_of(d2)
    # The squared distance of two bodies is not evaluated at all: the lane of their pair holds its Taylor coefficients (the
    # history of the pow recurrence) and contributes them - per-lane switch, constant and row of the event.
    assert "__shfl(" not in s and "const double pe_on =" in s and "pe_row" in s and "evo0" not in s
    monkeypatch.setenv("HEYOKA_AMD_NO_PAIR_EVENTS", "1")
    s = src_of(d2)
    monkeypatch.delenv("HEYOKA_AMD_NO_PAIR_EVENTS")
    # Two leaf positions (the two bodies), both position-type variables: offsets of the parent columns + current values.
    assert s.count("__shfl(") == 3 * 21 and "const unsigned evo1 =" in s and "const unsigned evz1 =" in s and "evo2" not in s
    assert "pe_on" not in s
    # A distance in the plane (two squares) is not a pair distance: two terms side by side.
    s = src_of((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) - 4.0)
    assert s.count("__shfl(") == 2 * 21 and "pe_on" not in s
    # Other spellings of the same event - pow(., 2) (a sum of squares becomes sum_sq in the decomposition), a flat sum, the
    # bodies the other way round with the constant spread over the expression, R^2 - d^2 (the lane switch carries the sign) -
    # are recognised; squares of mixed signs and a "distance" which mixes three bodies are not.
    z3 = hy.make_vars("z_3", "dummy__")[0]
    for ev, lane in ((hy.pow(x1 - x2, 2.0) + hy.pow(y1 - y2, 2.0) + hy.pow(z1 - z2, 2.0) - 81.0, True),
                     (hy.sum([(x1 - x2) * (x1 - x2), (y1 - y2) * (y1 - y2), (z1 - z2) * (z1 - z2)]) - 81.0, True),
                     (81.0 + ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1)) - 162.0, True),
                     (81.0 - ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2)), True),
                     ((x1 - x2) * (x1 - x2) - (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - 1.0, False),
                     ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z3) * (z1 - z3) - 81.0, False)):
        assert ("const double pe_on =" in src_of(ev)) == lane
    # All 15 pair distances of the six bodies at once: one event per lane, nothing evaluated.
    def pos(b_):
        return hy.make_vars("x_%d" % b_, "y_%d" % b_, "z_%d" % b_)
    evs = []
    for a_ in range(6):
        for b_ in range(a_ + 1, 6):
            pa, pb = pos(a_), pos(b_)
            evs.append(hy.nt_event((pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1])
                                   + (pa[2] - pb[2]) * (pa[2] - pb[2]) - 1.0, cb))
    ta15 = hy.taylor_adaptive_batch(sys_, None, 8, high_accuracy=True, nt_events=evs)
    assert "inside the stepper" in ta15.hip_source_mode and "15 event equation(s) evaluated by the stepper" in ta15.hip_source_mode
    assert ta15.hip_source.count("l == ") >= 45 and "__shfl(" not in ta15.hip_source
    s = src_of(x1 * vx1 + y1 * vy1 + z1 * vz1)
    assert s.count("__shfl(") == 3 * 21 and "const unsigned evz0 =" in s and "const unsigned evz1 =" not in s
    # Different shapes (a position times a velocity, a position times a position): one after the other.
    s = src_of(x1 * vx1 + y1 * y2 - 1.0)
    assert "__shfl(" not in s and "evo0" not in s
    # Terms which share a node (x_1 - x_2 in both): one after the other.
    dx = x1 - x2
    s = src_of(dx * (y1 - y2) + dx * (z1 - z2))
    assert "__shfl(" not in s
    # Linear terms only: nothing to gain.
    s = src_of((x1 - x2) + (y1 - y2) + (z1 - z2))
    assert "__shfl(" not in s


def test_time_dependent_event_on_the_cluster_event_stepper_builds():
    """(CPU: hiprtc cross-compiles.) An event equation which depends on the time coordinate next to a system which runs
    on the wave-cluster stepper: hy_ev_jets evaluates func_kind::time and needs the time of the lane (round-2 advisor
    finding: the constructor threw 'use of undeclared identifier t_hi')."""
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    x1 = hy.make_vars("x_1")
    x1 = x1[0] if isinstance(x1, (list, tuple)) else x1
    for ev in (hy.time - 0.5, x1 - hy.cos(hy.time)):
        ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, 8, high_accuracy=True,
                                      nt_events=[hy.nt_event(ev, lambda *a: None)])
        assert ta.hip_source_mode.startswith("cluster") and "events:" in ta.hip_source_mode, ta.hip_source_mode


def test_time_setters_follow_the_reference_checks_without_a_gpu():
    """set_dtime() through the C ABI: normalisation of (hi, lo), the reference's argument checks with their messages
    (include/heyoka/detail/taylor_common.hpp:232-249, test/taylor_adaptive_batch.cpp:1864-1942), nothing modified when a
    check fails; get_tc() reads zeros before any write_tc step."""
    import heyoka_amd as hy

    x, v = hy.make_vars("x", "v")
    ta = hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], [[0.0, 0.01], [0.1, 0.11]], 2)
    eps = np.finfo(float).eps
    ta.set_dtime([3.0, -7.0], [2.0, 5.0])
    hi, lo = ta.dtime
    assert list(hi) == [5.0, -2.0] and list(lo) == [0.0, 0.0]
    ta.set_dtime([3.0, -3.0], [eps, eps])
    hi, lo = ta.dtime
    assert list(hi) == [3.0, -3.0] and list(lo) == [eps, eps]
    ta.set_dtime([3.0, 4.0], [1.0, 2.0])
    with pytest.raises(ValueError, match="must both be finite, but they are inf and 1 instead"):
        ta.set_dtime([np.inf, 1.0], [1.0, 1.0])
    with pytest.raises(ValueError, match=r"coordinate \(3\) must not be smaller in magnitude than the second component \(4\)"):
        ta.set_dtime([3.0, 3.0], [4.0, 4.0])
    hi, lo = ta.dtime
    assert list(hi) == [4.0, 6.0] and list(lo) == [0.0, 0.0]
    assert not np.any(ta.tc)


def test_stage_logger_levels_callback_and_build_id():
    """The minimal stage logger (include/heyoka/logging.hpp:19-24): levels, a callback sink, and what a
    construction logs - the size of the decomposition (debug), the planner's verdict with the reasons the other generators
    did not apply (info), the stage times (trace) - plus the build id which ties the library to the sources of the tree."""
    import heyoka_amd as hy
    from heyoka_amd import _lib

    assert hy.build_id() == _lib.tree_build_id() and len(hy.build_id()) == 16
    msgs = []
    hy.set_log_callback(lambda lvl, m: msgs.append((lvl, m)))
    try:
        old = _lib.lib.hy_get_logger_level()
        assert old == 3  # warn, the reference's default
        hy.set_logger_level_trace()
        x, v = hy.make_vars("x", "v")
        hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], None, 8)
        text = "\n".join(m for _, m in msgs)
        assert "Taylor decomposition of 2 equations: 7 entries" in text
        assert "Taylor decomposition construction runtime:" in text and "hiprtc compilation runtime:" in text
        assert any(lvl == 2 and "Taylor batch code generation: unrolled" in m and "fewer than 2 clusters" in m for lvl, m in msgs)
        del msgs[:]
        hy.set_logger_level("err")
        hy.taylor_adaptive_batch([(x, v), (v, -9.8 * hy.sin(x))], None, 8)
        assert msgs == []
        with pytest.raises(ValueError):
            hy.set_logger_level(9)
    finally:
        hy.set_logger_level("warn")
        hy.set_log_callback(None)
