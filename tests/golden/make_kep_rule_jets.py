#!/usr/bin/env python
"""Generates tests/golden/kep_rule_jets.json: Taylor coefficients (orders 0..3) of small coupled systems built on kepF
and kepDE - the two functions which exist only as registered node rules (heyoka_amd/csrc/builtin_rules.cpp) - in the
format of node_jets.json (batch 3, tol = .1 -> order 3, state {2, 5, 1, 3, 4, 6}).

Independent of every recurrence: for z' = Phi(z)
    z1 = Phi,  z2 = J z1 / 2,  z3 = (H[z1, z1] + 2 J z2) / 6
with the Jacobian J and the Hessian H obtained by 40-digit numerical differentiation (mpmath) of Phi, in which the
implicit functions are evaluated by root finding on their defining equations
    kepF(h, k, lam):    F + h cos F - k sin F = lam          (reference: src/math/kepF.cpp)
    kepDE(s0, c0, DM):  DE - c0 sin DE + s0 (1 - cos DE) = DM (reference: src/math/kepDE.cpp)
reduced to [0, 2 pi) like the reference's solvers (src/detail/llvm_helpers_celmec.cpp:540-856, :857-1170).

    python tests/golden/make_kep_rule_jets.py > tests/golden/kep_rule_jets.json
"""
import json

import mpmath as mp

mp.mp.dps = 40
X0 = [2.0, 5.0, 1.0]
Y0 = [3.0, 4.0, 6.0]
TWO_PI = 2 * mp.pi


def kepF(h, k, lam):
    F = mp.findroot(lambda F: F + h * mp.cos(F) - k * mp.sin(F) - lam, lam, tol=mp.mpf(10) ** -35)
    return F - TWO_PI * mp.floor(F / TWO_PI)


def kepDE(s0, c0, DM):
    D = mp.findroot(lambda D: D - c0 * mp.sin(D) + s0 * (1 - mp.cos(D)) - DM, DM, tol=mp.mpf(10) ** -35)
    return D - TWO_PI * mp.floor(D / TWO_PI)


CASES = {
    # every argument a variable expression
    "kepF_var_var_var": lambda x, y: (kepF(0.05 * x, 0.04 * y, 0.3 * x + 0.1 * y), kepF(0.03 * y, 0.06 * x, 0.2 * y)),
    # numerical arguments in each position (the reference's specialised overloads, src/math/kepF.cpp:160-632)
    "kepF_num_mixed": lambda x, y: (kepF(0.3, 0.04 * y, 0.25 * x), kepF(0.03 * y, 0.2, 1.5) + 0.1 * x),
    "kepF_num_num_var": lambda x, y: (kepF(0.3, -0.2, 0.2 * y), kepF(0.05 * x, 0.04 * y, 0.7)),
    "kepDE_var_var_var": lambda x, y: (kepDE(0.05 * x, 0.04 * y, 0.3 * x + 0.1 * y), kepDE(0.03 * y, 0.06 * x, 0.2 * y)),
    "kepDE_num_mixed": lambda x, y: (kepDE(0.3, 0.04 * y, 0.25 * x), kepDE(0.03 * y, -0.2, 0.9) + 0.1 * x),
    # both functions in one system, one feeding the other
    "kepF_of_kepDE": lambda x, y: (kepF(0.05 * x, 0.1, kepDE(0.1, 0.04 * y, 0.2 * x)), 0.2 * y - 0.1 * x),
}


def jets(phi, z0):
    z0 = [mp.mpf(v) for v in z0]
    n = len(z0)
    comp = lambda i: (lambda *z: phi(*z)[i])
    z1 = [comp(i)(*z0) for i in range(n)]
    J = [[mp.diff(comp(i), z0, tuple(1 if q == j else 0 for q in range(n))) for j in range(n)] for i in range(n)]

    def second(i, j, k):
        order = [0] * n
        order[j] += 1
        order[k] += 1
        return mp.diff(comp(i), z0, tuple(order))

    H = [[[second(i, j, k) for k in range(n)] for j in range(n)] for i in range(n)]
    z2 = [sum(J[i][j] * z1[j] for j in range(n)) / 2 for i in range(n)]
    z3 = [(sum(H[i][j][k] * z1[j] * z1[k] for j in range(n) for k in range(n)) + 2 * sum(J[i][j] * z2[j] for j in range(n))) / 6
          for i in range(n)]
    return [[float(v) for v in row] for row in (z0, z1, z2, z3)]


def main():
    out = {"state": [X0, Y0], "tol": 0.1, "order": 3, "cases": []}
    for name, phi in CASES.items():
        # (The values of the implicit functions are folded into [0, 2 pi) like the reference's solvers do; the inputs keep
        # away from the jump.)
        out["cases"].append({"name": name, "pars": None, "time": None,
                             "jets": [jets(phi, [X0[l], Y0[l]]) for l in range(3)]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
