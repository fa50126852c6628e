"""Per-function Taylor coefficients against closed-form derivatives (tests/golden/node_jets.json), in the
style of the reference's per-function tests (test/taylor_pow.cpp:573-700 etc.: batch 3, tol = .1 -> order 3,
`approximately()` = 100 eps). Pins the recurrences of the oracle (CPU) and of the HIP path (GPU) for every
elementary function of the hot path independently of each other."""
import json
import os

import numpy as np
import pytest

import heyoka_oracle as ho
from conftest import EPS

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "node_jets.json")) as _f:
    G = json.load(_f)


def build(m, name):
    """The systems of make_node_jets.py for expression module m (oracle or product)."""
    if m is ho:
        x, y, t = m.var("x"), m.var("y"), m.func("time", [])
        P = m.par
        powf, sqrt = m.pow_, m.sqrt
    else:
        x, y = m.make_vars("x", "y")
        t, P = m.time, lambda i: m.par[i]
        powf, sqrt = m.pow, m.sqrt
    if name.startswith("unary_"):
        fn = name[len("unary_"):]
        a, c = (1.0, 0.5) if fn == "acosh" else (0.1, 0.0)
        f = getattr(m, fn)
        return [(x, f(a * y + c) if c else f(a * y)), (y, f(a * x + c) if c else f(a * x))]
    rhs = {
        "pow_3_2__m1_3": (powf(y, 1.5), powf(x, -1.0 / 3.0)),
        "pow_par_exponents": (powf(y, P(0)), powf(x, P(1))),
        "pow_m3_2__2": (powf(y, -1.5), powf(x, 2.0)),
        "sqrt": (sqrt(y), sqrt(x)),
        "prod_var_var": (x * y, y * x * x),
        "prod_num_var_neg": (-2.0 * y, -x),
        "div": (x / y, 1.5 / x),
        "sum_sub": (x - y, x + y + 2.0),
        "sum_sq": (x * x + y * y, x * x + y * y + 9.0),
        "sin_cos": (m.sin(y), m.cos(x)),
        "exp_log": (m.exp(0.1 * y), m.log(x)),
        "time": (t + y, x * t),
        "atan2_var_var": (m.atan2(y, x), m.atan2(x, y)),
        "atan2_var_num": (m.atan2(y, 1.5), m.atan2(0.3, x)),
        "kepE_var_var": (m.kepE(0.1 * x, y), m.kepE(0.05 * y, x)),
        "kepE_num_var": (m.kepE(0.3, y), m.kepE(0.05 * x, 0.4)),
        "relu_leaky": (m.relu(y - 4.5), m.relu(x - 3.0, 0.01)),
        "relup_select": (m.relup(y - 4.5, 0.1) * x, m.select(m.gt(x, y), x * y, x + y)),
        "logical_rel": (m.logical_and([m.lt(x, 3.0), m.gte(y, 3.5)]) + 0.5 * y,
                        m.logical_or([m.eq(x, 5.0), m.lte(y, 3.0)]) * x + m.neq(x, 1.0)),
    }[name]
    return [(x, rhs[0]), (y, rhs[1])]


def check(tc, case):
    # tc[var][order][lane] vs jets[lane][order][var]
    exp = np.transpose(np.array(case["jets"]), (2, 1, 0))
    err = np.abs(tc - exp) / np.maximum(np.abs(exp), 1e-300)
    err[exp == 0] = np.abs(tc)[exp == 0]
    assert np.max(err) <= 100 * EPS, (case["name"], float(np.max(err)))


@pytest.mark.parametrize("case", G["cases"], ids=[c["name"] for c in G["cases"]])
def test_oracle_node_jets(case):
    ta = ho.OracleIntegrator(build(ho, case["name"]), G["state"], 3, tol=G["tol"], pars=case["pars"],
                             time=case["time"])
    assert ta.order == G["order"]
    ta.step(wtc=True)
    check(ta.tc.reshape(2, G["order"] + 1, 3), case)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "table"])
def test_gpu_node_jets(mode):
    import heyoka_amd as hy

    old = os.environ.get("HEYOKA_AMD_EMIT_MODE")
    if mode == "table":
        os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
    try:
        for case in G["cases"]:
            kw = {}
            if case["pars"] is not None:
                kw["pars"] = case["pars"]
            if case["time"] is not None:
                kw["time"] = case["time"]
            ta = hy.taylor_adaptive_batch(build(hy, case["name"]), G["state"], 3, tol=G["tol"], **kw)
            assert ta.order == G["order"]
            ta.step(write_tc=True)
            check(np.asarray(ta.tc).reshape(2, G["order"] + 1, 3), case)
    finally:
        if mode == "table":
            if old is None:
                del os.environ["HEYOKA_AMD_EMIT_MODE"]
            else:
                os.environ["HEYOKA_AMD_EMIT_MODE"] = old


# ---- The reference's own literal expectations (tests/golden/reference_node_tests.json, transcribed from
# test/taylor_pow.cpp:575-640, test/taylor_sum_sq.cpp:435-497, test/taylor_prod.cpp:977-1022 by
# tests/golden/make_reference_node_tests.py) ----
with open(os.path.join(HERE, "golden", "reference_node_tests.json")) as _f:
    R = json.load(_f)


def build_ref(m, name):
    if m is ho:
        x, y = m.var("x"), m.var("y")
        powf, sum_sq = m.pow_, lambda a: m.func("sum_sq", [m.as_ex(v) for v in a])
    else:
        x, y = m.make_vars("x", "y")
        powf, sum_sq = m.pow, getattr(m, "sum_sq", None)
    if name == "pow_frac":
        return [(x, powf(y, 3.0 / 2.0)), (y, powf(x, -1.0 / 3.0))]
    if name == "prod_vars":
        return [(x, x * y), (y, y * x)]
    if name == "sincos_vars":
        return [(x, m.sin(y)), (y, m.cos(x))]
    if name == "div_vars":
        return [(x, x / y), (y, y / x)]
    if name == "sub_vars":
        return [(x, x - y), (y, y - x)]
    tm = m.TIME if m is ho else m.time
    par0 = m.par(0) if m is ho else m.par[0]
    if name == "time_vars":
        return [(x, tm + x), (y, x + y)]
    if name == "sum_vars":
        return [(x, 2.0 + x + par0 + y), (y, x + y)]
    if name == "no_decomp":
        return [(x, y), (y, x)]
    if name == "const_pars":
        z = m.var("z") if m is ho else m.make_vars("z")
        z = z[0] if isinstance(z, (list, tuple)) else z
        pr = (lambda i: m.par(i)) if m is ho else (lambda i: m.par[i])
        return [(x, pr(0)), (y, pr(1)), (z, pr(2))]
    if name == "sum_sq_vars":
        # (sum_to_sum_sq() builds the sum_sq nodes out of the sums of squares, src/math/sum_sq.cpp.)
        return [(x, y * y + x * x + 1.0), (y, x * x + y * y + 4.0)]
    raise KeyError(name)


def check_ref(tc, case, n_ord):
    # tc[var][order][lane] vs jet[(k * n_eq + var) * 3 + lane]
    exp = np.array(case["jet"]).reshape(n_ord, case.get("n_eq", 2), 3).transpose(1, 0, 2)
    err = np.abs(tc[:, :n_ord, :] - exp) / np.maximum(np.abs(exp), 1e-300)
    assert np.max(err) <= 100 * EPS, (case["system"], float(np.max(err)))


@pytest.mark.parametrize("case", R["cases"], ids=[c["system"] for c in R["cases"]])
def test_oracle_reference_literal_node_expectations(case):
    st = np.array(case["state"])
    kw = {}
    if "pars" in case:
        kw["pars"] = np.array(case["pars"])
    if "time" in case:
        kw["time"] = np.array(case["time"])
    ta = ho.OracleIntegrator(build_ref(ho, case["system"]), st, case["batch"], tol=case["tol"], **kw)
    ta.step(wtc=True)
    ne = case.get("n_eq", 2)
    check_ref(ta.tc.reshape(ne, ta.order + 1, 3), case, len(case["jet"]) // (3 * ne))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "table"])
def test_gpu_reference_literal_node_expectations(mode):
    import heyoka_amd as hy

    old = os.environ.get("HEYOKA_AMD_EMIT_MODE")
    if mode == "table":
        os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
    try:
        for case in R["cases"]:
            kw = {}
            if "pars" in case:
                kw["pars"] = np.array(case["pars"])
            if "time" in case:
                kw["time"] = np.array(case["time"])
            ta = hy.taylor_adaptive_batch(build_ref(hy, case["system"]), np.array(case["state"]), case["batch"], tol=case["tol"],
                                          **kw)
            ta.step(write_tc=True)
            ne = case.get("n_eq", 2)
            check_ref(np.asarray(ta.tc).reshape(ne, ta.order + 1, 3), case, len(case["jet"]) // (3 * ne))
    finally:
        if mode == "table":
            if old is None:
                del os.environ["HEYOKA_AMD_EMIT_MODE"]
            else:
                os.environ["HEYOKA_AMD_EMIT_MODE"] = old
