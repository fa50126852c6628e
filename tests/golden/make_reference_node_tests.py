#!/usr/bin/env python
"""Writes tests/golden/reference_node_tests.json: the per-node known answers the reference's own tests assert,
transcribed case by case (system, initial state, batch size, tolerance -> order, and the REQUIRE(jet[i] == ...) expressions
evaluated here in double precision exactly as written in the test). Data only: no reference source is copied.

jet layout of the reference's tc_to_jet(): jet[(k * n_eq + var) * batch + lane].

  pow_frac      test/taylor_pow.cpp:575-640       x' = pow(y, 3/2), y' = pow(x, -1/3), batch 3, tol .1 (order 3)
  sum_sq_vars   test/taylor_sum_sq.cpp:435-497    x' = sum_sq(y, x, 1), y' = sum_sq(x, y, 2), batch 3, tol .1
  prod_vars     test/taylor_prod.cpp:977-1022     x' = x * y, y' = y * x, batch 3, tol .1
  sincos_vars   test/taylor_sincos.cpp:459-507    x' = sin(y), y' = cos(x), batch 3, tol .1
  div_vars      test/taylor_div.cpp:894-962       x' = x / y, y' = y / x, batch 3, tol .1
  sub_vars      test/taylor_sub.cpp:859-893       x' = x - y, y' = y - x, batch 3, tol .1
  time_vars     test/taylor_time.cpp:196-230      x' = t + x, y' = x + y, batch 3, tol .1, times (-5, 6, -1)
  sum_vars      test/taylor_sum.cpp:154-189       x' = sum(2, x, par[0], y), y' = x + y, batch 3, tol .1, pars (2, -1, 3)
  no_decomp     test/taylor_no_decomp_sys.cpp:149-184  x' = y, y' = x (no u variables), batch 3, tol .1
  const_pars    test/taylor_const_sys.cpp:285-318   x' = par[0], y' = par[1], z' = par[2] (n_eq = 3), batch 3, tol .1
"""
import json
import os
from math import cos, sin
from math import pow as P


def pow_frac():
    j = [2., 5., 1., 3., 4., 6.] + [0.] * 18
    for l in range(3):
        j[6 + l] = P(j[3 + l], 3. / 2)
        j[9 + l] = P(j[l], -1. / 3)
        j[12 + l] = 1. / 2 * 3. / 2 * P(j[3 + l], 1. / 2) * j[9 + l]
        j[15 + l] = 1. / 2 * -1. / 3 * P(j[l], -4. / 3) * j[6 + l]
        j[18 + l] = 1. / 6 * 3. / 2 * (1. / 2 * P(j[3 + l], -1. / 2) * j[9 + l] * j[9 + l] + P(j[3 + l], 1. / 2) * 2 * j[15 + l])
        j[21 + l] = 1. / 6 * -1. / 3 * (-4. / 3 * P(j[l], -7. / 3) * j[6 + l] * j[6 + l] + P(j[l], -4. / 3) * 2 * j[12 + l])
    return {"source": "test/taylor_pow.cpp:575-640", "system": "pow_frac", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def sum_sq_vars():
    j = [2., 4., 3., 3., 5., 6.] + [0.] * 18
    for l in range(3):
        j[6 + l] = j[3 + l] * j[3 + l] + j[l] * j[l] + 1
        j[9 + l] = j[3 + l] * j[3 + l] + j[l] * j[l] + 4
        j[12 + l] = j[3 + l] * j[9 + l] + j[l] * j[6 + l]
        j[15 + l] = j[3 + l] * j[9 + l] + j[l] * j[6 + l]
        j[18 + l] = 1. / 3 * (j[9 + l] * j[9 + l] + j[3 + l] * 2 * j[15 + l] + j[6 + l] * j[6 + l] + j[l] * 2 * j[12 + l])
        j[21 + l] = j[18 + l]
    return {"source": "test/taylor_sum_sq.cpp:435-497", "system": "sum_sq_vars", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def prod_vars():
    j = [2., 1., 3., 3., -4., 6.] + [0.] * 18
    x0, y0 = j[:3], j[3:6]
    for l in range(3):
        j[6 + l] = x0[l] * y0[l]
        j[9 + l] = j[6 + l]
        j[12 + l] = 1. / 2 * (j[6 + l] * y0[l] + j[9 + l] * x0[l])
        j[15 + l] = j[12 + l]
        j[18 + l] = 1 / 6. * (2 * j[12 + l] * y0[l] + 2 * j[6 + l] * j[9 + l] + 2 * x0[l] * j[15 + l])
        j[21 + l] = j[18 + l]
    return {"source": "test/taylor_prod.cpp:977-1022", "system": "prod_vars", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def sincos_vars():
    j = [2., -1., -5., 3., -4., 6.] + [0.] * 18
    for l in range(3):
        j[6 + l] = sin(j[3 + l])
        j[9 + l] = cos(j[l])
        j[12 + l] = 1. / 2 * j[9 + l] * cos(j[3 + l])
        j[15 + l] = 1. / 2 * -j[6 + l] * sin(j[l])
        j[18 + l] = 1. / 6 * (2 * j[15 + l] * cos(j[3 + l]) - j[9 + l] * j[9 + l] * sin(j[3 + l]))
        j[21 + l] = 1. / 6 * (-2 * j[12 + l] * sin(j[l]) - j[6 + l] * j[6 + l] * cos(j[l]))
    return {"source": "test/taylor_sincos.cpp:459-507", "system": "sincos_vars", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def div_vars():
    j = [2., -5., 1., 3., 4., -2.] + [0.] * 18
    x0, y0 = j[:3], j[3:6]
    for l in range(3):
        x, y = x0[l], y0[l]
        j[6 + l] = x / y
        j[9 + l] = y / x
        j[12 + l] = 1 / 2. * (j[6 + l] * y - j[9 + l] * x) / (y * y)
        j[15 + l] = 1 / 2. * (j[9 + l] * x - j[6 + l] * y) / (x * x)
        # (The reference writes these out lane by lane with the numbers plugged in; the cross terms x' y' - y' x' it
        # keeps in lane 0 cancel identically.)
        j[18 + l] = 1 / 6. * ((2 * j[12 + l] * y + j[6 + l] * j[9 + l] - 2 * j[15 + l] * x - j[9 + l] * j[6 + l]) * y * y
                              - 2 * y * j[9 + l] * (j[6 + l] * y - j[9 + l] * x)) / (y * y * y * y)
        j[21 + l] = 1 / 6. * ((2 * j[15 + l] * x + j[9 + l] * j[6 + l] - 2 * j[12 + l] * y - j[9 + l] * j[6 + l]) * x * x
                              - 2 * x * j[6 + l] * (j[9 + l] * x - j[6 + l] * y)) / (x * x * x * x)
    return {"source": "test/taylor_div.cpp:894-962", "system": "div_vars", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def sub_vars():
    j = [2., 1., 3., 3., -4., 6.] + [0.] * 18
    for l in range(3):
        j[6 + l] = j[l] - j[3 + l]
        j[9 + l] = -j[l] + j[3 + l]
        j[12 + l] = 1. / 2 * (j[6 + l] - j[9 + l])
        j[15 + l] = 1. / 2 * (-j[6 + l] + j[9 + l])
        j[18 + l] = 1 / 3. * (j[12 + l] - j[15 + l])
        j[21 + l] = 1 / 3. * (-j[12 + l] + j[15 + l])
    return {"source": "test/taylor_sub.cpp:859-893", "system": "sub_vars", "state": j[:6], "batch": 3, "tol": .1, "jet": j}


def time_vars():
    j = [2., -2., 1., 3., -3., 0.] + [0.] * 18
    tm = [-5., 6., -1.]
    for l in range(3):
        j[6 + l] = tm[l] + j[l]
        j[9 + l] = j[l] + j[3 + l]
        j[12 + l] = 1. / 2 * (1 + j[6 + l])
        j[15 + l] = 1. / 2 * (j[9 + l] + j[6 + l])
        j[18 + l] = 1. / 6 * 2 * j[12 + l]
        j[21 + l] = 1. / 6 * (2 * j[15 + l] + 2 * j[12 + l])
    return {"source": "test/taylor_time.cpp:196-230", "system": "time_vars", "state": j[:6], "batch": 3, "tol": .1, "time": tm,
            "jet": j}


def sum_vars():
    j = [2., -2., 1., 3., -3., 2.] + [0.] * 18
    pars = [2., -1., 3.]
    for l in range(3):
        j[6 + l] = 2 + j[l] + pars[l] + j[3 + l]
        j[9 + l] = j[l] + j[3 + l]
        j[12 + l] = .5 * (j[6 + l] + j[9 + l])
        j[15 + l] = .5 * (j[6 + l] + j[9 + l])
        j[18 + l] = (j[12 + l] + j[15 + l]) / 3
        j[21 + l] = (j[12 + l] + j[15 + l]) / 3
    return {"source": "test/taylor_sum.cpp:154-189", "system": "sum_vars", "state": j[:6], "batch": 3, "tol": .1, "pars": pars,
            "jet": j}


def no_decomp():
    j = [2., 1., 0., -3., 5., 4.] + [0.] * 18
    for l in range(3):
        j[6 + l] = j[3 + l]
        j[9 + l] = j[l]
        j[12 + l] = j[9 + l] / 2
        j[15 + l] = j[6 + l] / 2
        j[18 + l] = 1. / 6 * j[15 + l] * 2
        j[21 + l] = 1. / 6 * j[12 + l] * 2
    return {"source": "test/taylor_no_decomp_sys.cpp:149-184", "system": "no_decomp", "state": j[:6], "batch": 3, "tol": .1,
            "jet": j}


def const_pars():
    st = [2., -2., 0., 3., -3., 0., 4., -4., 0.]
    pars = [1., 1., 1., -2., -2., -2., 0., 0., 0.]
    j = st + pars + [0.] * 18
    return {"source": "test/taylor_const_sys.cpp:285-318", "system": "const_pars", "state": st, "batch": 3, "tol": .1,
            "pars": pars, "n_eq": 3, "jet": j}


if __name__ == "__main__":
    out = {"note": __doc__, "cases": [pow_frac(), sum_sq_vars(), prod_vars(), sincos_vars(), div_vars(), sub_vars(), time_vars(), sum_vars(), no_decomp(), const_pars()],
           "tolerance": "the reference's approximately(): 100 eps relative"}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_node_tests.json"), "w") as f:
        json.dump(out, f, indent=1)
