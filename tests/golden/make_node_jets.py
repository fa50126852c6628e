#!/usr/bin/env python
"""Generates tests/golden/node_jets.json: closed-form Taylor coefficients (orders 0..3) of small coupled
systems exercising every elementary function of the hot path, in the style of the reference's per-function
tests (test/taylor_pow.cpp:573-700, test/taylor_sum_sq.cpp, test/taylor_sincos.cpp, test/taylor_div.cpp, ...:
batch size 3, tol = .1 -> order 3, state {2, 5, 1, 3, 4, 6}, jets compared with the analytic derivatives).

The expected values come from the calculus identities for z' = Phi(z):
    z1 = Phi,  z2 = J z1 / 2,  z3 = (H[z1, z1] + 2 J z2) / 6
with hand-written Jacobians J and Hessians H - no recurrence of the oracle or of the product is involved.
The inputs are the reference's literals; where a reference test uses the same system its file:line is given.

    python tests/golden/make_node_jets.py > tests/golden/node_jets.json
"""
import json
import math

import numpy as np

X0 = [2.0, 5.0, 1.0]
Y0 = [3.0, 4.0, 6.0]


def jets(phi, jac, hess, z0):
    z0 = np.array(z0, dtype=float)
    z1 = np.array(phi(z0))
    J = np.array(jac(z0))
    H = np.array(hess(z0))
    z2 = J @ z1 / 2.0
    z3 = (np.einsum("ijk,j,k->i", H, z1, z1) + 2.0 * (J @ z2)) / 6.0
    return [z0.tolist(), z1.tolist(), z2.tolist(), z3.tolist()]


def Z(n):
    return [[[0.0] * n for _ in range(n)] for _ in range(n)]


CASES = []


def case(name, sys, source, phi, jac, hess, pars=None, time=None):
    CASES.append(dict(name=name, sys=sys, source=source, phi=phi, jac=jac, hess=hess, pars=pars, time=time))


# 1. pow with fractional exponents (test/taylor_pow.cpp:573-641).
def _pow(a, b):
    def hess(z):
        h = Z(2)
        h[0][1][1] = a * (a - 1) * z[1] ** (a - 2)
        h[1][0][0] = b * (b - 1) * z[0] ** (b - 2)
        return h
    return (lambda z: [z[1] ** a, z[0] ** b],
            lambda z: [[0, a * z[1] ** (a - 1)], [b * z[0] ** (b - 1), 0]], hess)


case("pow_3_2__m1_3", ["pow(y, 3/2)", "pow(x, -1/3)"], "test/taylor_pow.cpp:573-641", *_pow(1.5, -1.0 / 3.0))
case("pow_par_exponents", ["pow(y, par[0])", "pow(x, par[1])"], "test/taylor_pow.cpp:643-700", *_pow(1.5, -1.0 / 3.0),
     pars=[[1.5] * 3, [-1.0 / 3.0] * 3])
case("pow_m3_2__2", ["pow(y, -3/2)", "pow(x, 2)"], "src/math/pow.cpp:292-355 (neg_small_half / square paths)",
     *_pow(-1.5, 2.0))
case("sqrt", ["sqrt(y)", "sqrt(x)"], "test/taylor_sqrt.cpp", *_pow(0.5, 0.5))

# 2. Products, quotients, linear combinations.
case("prod_var_var", ["x * y", "y * x * x"], "test/taylor_mul.cpp / taylor_prod.cpp",
     lambda z: [z[0] * z[1], z[1] * z[0] * z[0]],
     lambda z: [[z[1], z[0]], [2 * z[0] * z[1], z[0] * z[0]]],
     lambda z: [[[0, 1], [1, 0]], [[2 * z[1], 2 * z[0]], [2 * z[0], 0]]])
case("prod_num_var_neg", ["-2 * y", "-x"], "test/taylor_neg.cpp",
     lambda z: [-2 * z[1], -z[0]], lambda z: [[0, -2], [-1, 0]], lambda z: Z(2))
case("div", ["x / y", "1.5 / x"], "test/taylor_div.cpp",
     lambda z: [z[0] / z[1], 1.5 / z[0]],
     lambda z: [[1 / z[1], -z[0] / z[1] ** 2], [-1.5 / z[0] ** 2, 0]],
     lambda z: [[[0, -1 / z[1] ** 2], [-1 / z[1] ** 2, 2 * z[0] / z[1] ** 3]], [[3.0 / z[0] ** 3, 0], [0, 0]]])
case("sum_sub", ["x - y", "x + y + 2"], "test/taylor_sub.cpp / taylor_sum.cpp",
     lambda z: [z[0] - z[1], z[0] + z[1] + 2], lambda z: [[1, -1], [1, 1]], lambda z: Z(2))
case("sum_sq", ["x*x + y*y", "x*x + y*y + 9"], "test/taylor_sum_sq.cpp",
     lambda z: [z[0] ** 2 + z[1] ** 2, z[0] ** 2 + z[1] ** 2 + 9],
     lambda z: [[2 * z[0], 2 * z[1]], [2 * z[0], 2 * z[1]]],
     lambda z: [[[2, 0], [0, 2]], [[2, 0], [0, 2]]])

# 3. Transcendental functions.
case("sin_cos", ["sin(y)", "cos(x)"], "test/taylor_sincos.cpp",
     lambda z: [math.sin(z[1]), math.cos(z[0])],
     lambda z: [[0, math.cos(z[1])], [-math.sin(z[0]), 0]],
     lambda z: [[[0, 0], [0, -math.sin(z[1])]], [[-math.cos(z[0]), 0], [0, 0]]])
case("exp_log", ["exp(0.1 * y)", "log(x)"], "test/taylor_exp.cpp / taylor_log.cpp",
     lambda z: [math.exp(0.1 * z[1]), math.log(z[0])],
     lambda z: [[0, 0.1 * math.exp(0.1 * z[1])], [1 / z[0], 0]],
     lambda z: [[[0, 0], [0, 0.01 * math.exp(0.1 * z[1])]], [[-1 / z[0] ** 2, 0], [0, 0]]])

# 3b. Elementary functions beyond the N-body set (test/taylor_{tan,tanh,sinhcosh,asin,acos,atan,asinh,acosh,
# atanh,erf,sigmoid}.cpp): x' = f(a*y + c), y' = f(a*x + c).
def _unary(f, d1, d2, a=0.1, c=0.0):
    def hess(z):
        h = Z(2)
        h[0][1][1] = a * a * d2(a * z[1] + c)
        h[1][0][0] = a * a * d2(a * z[0] + c)
        return h
    return (lambda z: [f(a * z[1] + c), f(a * z[0] + c)],
            lambda z: [[0, a * d1(a * z[1] + c)], [a * d1(a * z[0] + c), 0]], hess)


_sig = lambda x: 1.0 / (1.0 + math.exp(-x))
_SQPI = 2.0 / math.sqrt(math.pi)
UNARY = {
    "tan": (math.tan, lambda x: 1 + math.tan(x) ** 2, lambda x: 2 * math.tan(x) * (1 + math.tan(x) ** 2), 0.1, 0.0),
    "tanh": (math.tanh, lambda x: 1 - math.tanh(x) ** 2, lambda x: -2 * math.tanh(x) * (1 - math.tanh(x) ** 2), 0.1, 0.0),
    "sinh": (math.sinh, math.cosh, math.sinh, 0.1, 0.0),
    "cosh": (math.cosh, math.sinh, math.cosh, 0.1, 0.0),
    "asin": (math.asin, lambda x: (1 - x * x) ** -0.5, lambda x: x * (1 - x * x) ** -1.5, 0.1, 0.0),
    "acos": (math.acos, lambda x: -((1 - x * x) ** -0.5), lambda x: -x * (1 - x * x) ** -1.5, 0.1, 0.0),
    "atan": (math.atan, lambda x: 1 / (1 + x * x), lambda x: -2 * x / (1 + x * x) ** 2, 0.1, 0.0),
    "asinh": (math.asinh, lambda x: (1 + x * x) ** -0.5, lambda x: -x * (1 + x * x) ** -1.5, 0.1, 0.0),
    "acosh": (math.acosh, lambda x: (x * x - 1) ** -0.5, lambda x: -x * (x * x - 1) ** -1.5, 1.0, 0.5),
    "atanh": (math.atanh, lambda x: 1 / (1 - x * x), lambda x: 2 * x / (1 - x * x) ** 2, 0.1, 0.0),
    "erf": (math.erf, lambda x: _SQPI * math.exp(-x * x), lambda x: -2 * x * _SQPI * math.exp(-x * x), 0.1, 0.0),
    "sigmoid": (_sig, lambda x: _sig(x) * (1 - _sig(x)), lambda x: _sig(x) * (1 - _sig(x)) * (1 - 2 * _sig(x)), 0.1, 0.0),
}
for _n, (_f, _d1, _d2, _a, _c) in UNARY.items():
    case("unary_" + _n, ["%s(%g*y + %g)" % (_n, _a, _c), "%s(%g*x + %g)" % (_n, _a, _c)],
         "test/taylor_%s.cpp (closed forms)" % _n, *_unary(_f, _d1, _d2, _a, _c))

# 3b. Two-argument functions: atan2 (test/taylor_atan2.cpp) and the eccentric anomaly kepE (test/taylor_kepE.cpp).
# g(p, q) with partials (g_p, g_q, g_pp, g_pq, g_qq); each right-hand side is g(a*z[i] + c, b*z[j] + d) with linear
# inner maps (a == 0 or b == 0: constant argument).
def _two_arg(g, parts, rows):
    # rows[r] = (i, a, c, j, b, d): phi_r = g(a*z[i] + c, b*z[j] + d)
    def args(z, r):
        i, a, c, j, b, d = rows[r]
        return a * z[i] + c, b * z[j] + d

    def phi(z):
        return [g(*args(z, r)) for r in range(2)]

    def jac(z):
        J = [[0.0, 0.0], [0.0, 0.0]]
        for r in range(2):
            i, a, c, j, b, d = rows[r]
            gp, gq = parts(*args(z, r))[:2]
            J[r][i] += a * gp
            J[r][j] += b * gq
        return J

    def hess(z):
        H = Z(2)
        for r in range(2):
            i, a, c, j, b, d = rows[r]
            _, _, gpp, gpq, gqq = parts(*args(z, r))
            H[r][i][i] += a * a * gpp
            H[r][j][j] += b * b * gqq
            H[r][i][j] += a * b * gpq
            H[r][j][i] += a * b * gpq
        return H

    return phi, jac, hess


def _atan2_parts(y, x):
    r2 = x * x + y * y
    return x / r2, -y / r2, -2 * x * y / r2 ** 2, (y * y - x * x) / r2 ** 2, 2 * x * y / r2 ** 2


def _kep_solve(e, M):
    # Independent of the oracle's solver: bracketing root finder on the reduced anomaly + Newton polishing.
    from scipy.optimize import brentq

    Mr = math.fmod(M, 2 * math.pi)
    Mr = Mr + 2 * math.pi if Mr < 0 else Mr
    E = brentq(lambda E: E - e * math.sin(E) - Mr, 0.0, 2 * math.pi, xtol=1e-15, rtol=1e-15)
    for _ in range(3):
        E -= (E - e * math.sin(E) - Mr) / (1 - e * math.cos(E))
    return E


def _kep_parts(e, M):
    E = _kep_solve(e, M)
    s, c = math.sin(E), math.cos(E)
    D = 1.0 / (1.0 - e * c)
    # (E_e, E_M, E_ee, E_eM, E_MM)
    return s * D, D, D * D * s * (2 * c - e * s * s * D), D * D * (c - e * s * s * D), -e * s * D ** 3


case("atan2_var_var", ["atan2(y, x)", "atan2(x, y)"], "test/taylor_atan2.cpp (closed forms)",
     *_two_arg(math.atan2, _atan2_parts, [(1, 1.0, 0.0, 0, 1.0, 0.0), (0, 1.0, 0.0, 1, 1.0, 0.0)]))
case("atan2_var_num", ["atan2(y, 1.5)", "atan2(0.3, x)"], "test/taylor_atan2.cpp (closed forms)",
     *_two_arg(math.atan2, _atan2_parts, [(1, 1.0, 0.0, 0, 0.0, 1.5), (1, 0.0, 0.3, 0, 1.0, 0.0)]))
case("kepE_var_var", ["kepE(0.1*x, y)", "kepE(0.05*y, x)"], "test/taylor_kepE.cpp (closed forms)",
     *_two_arg(_kep_solve, _kep_parts, [(0, 0.1, 0.0, 1, 1.0, 0.0), (1, 0.05, 0.0, 0, 1.0, 0.0)]))
case("kepE_num_var", ["kepE(0.3, y)", "kepE(0.05*x, 0.4)"], "test/taylor_kepE.cpp (closed forms)",
     *_two_arg(_kep_solve, _kep_parts, [(0, 0.0, 0.3, 1, 1.0, 0.0), (0, 0.05, 0.0, 1, 0.0, 0.4)]))

# 3c. Piecewise functions (test/relu.cpp, test/select.cpp, test/relational.cpp, test/logical.cpp): the branch is fixed
# by the state, inside a branch the right-hand side is a polynomial of degree <= 2.
def _relu_case():
    s0 = lambda z: 1.0 if z[1] - 4.5 > 0 else 0.0
    s1 = lambda z: 1.0 if z[0] - 3.0 > 0 else 0.01
    return (lambda z: [s0(z) * (z[1] - 4.5), s1(z) * (z[0] - 3.0)],
            lambda z: [[0, s0(z)], [s1(z), 0]], lambda z: Z(2))


def _relup_select_case():
    s = lambda z: 1.0 if z[1] - 4.5 > 0 else 0.1
    prod = lambda z: z[0] > z[1]

    def hess(z):
        h = Z(2)
        if prod(z):
            h[1][0][1] = h[1][1][0] = 1.0
        return h
    return (lambda z: [s(z) * z[0], z[0] * z[1] if prod(z) else z[0] + z[1]],
            lambda z: [[s(z), 0], [z[1], z[0]] if prod(z) else [1, 1]], hess)


def _logical_rel_case():
    c0 = lambda z: 1.0 if (z[0] < 3 and z[1] >= 3.5) else 0.0
    c1 = lambda z: 1.0 if (z[0] == 5 or z[1] <= 3) else 0.0
    c2 = lambda z: 1.0 if z[0] != 1 else 0.0
    return (lambda z: [c0(z) + 0.5 * z[1], c1(z) * z[0] + c2(z)],
            lambda z: [[0, 0.5], [c1(z), 0]], lambda z: Z(2))


case("relu_leaky", ["relu(y - 4.5)", "leaky_relu(0.01)(x - 3)"], "test/relu.cpp (closed forms)", *_relu_case())
case("relup_select", ["relup(y - 4.5, 0.1) * x", "select(gt(x, y), x * y, x + y)"],
     "test/relu.cpp, test/select.cpp (closed forms)", *_relup_select_case())
case("logical_rel", ["logical_and({lt(x, 3), gte(y, 3.5)}) + 0.5 * y", "logical_or({eq(x, 5), lte(y, 3)}) * x + neq(x, 1)"],
     "test/logical.cpp, test/relational.cpp (closed forms)", *_logical_rel_case())

# 4. Explicit time dependence (z = (x, y, t), t' = 1; test/taylor_time.cpp).
case("time", ["time + y", "x * time"], "test/taylor_time.cpp",
     lambda z: [z[2] + z[1], z[0] * z[2], 1.0],
     lambda z: [[0, 1, 1], [z[2], 0, z[0]], [0, 0, 0]],
     lambda z: [Z(3)[0], [[0, 0, 1], [0, 0, 0], [1, 0, 0]], Z(3)[0]],
     time=[0.5, -1.25, 3.0])


def main():
    out = {"note": "closed-form jets, see make_node_jets.py; jets[lane][order][variable]", "tol": 0.1, "order": 3,
           "state": [X0, Y0], "cases": []}
    for c in CASES:
        lanes = []
        for l in range(3):
            z0 = [X0[l], Y0[l]] + ([c["time"][l]] if c["time"] else [])
            j = jets(c["phi"], c["jac"], c["hess"], z0)
            lanes.append([row[:2] for row in j])
        out["cases"].append({"name": c["name"], "sys": c["sys"], "source": c["source"], "pars": c["pars"],
                             "time": c["time"], "jets": lanes})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
