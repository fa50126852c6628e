"""Algorithmic work of one Taylor step, derived from the decomposition the integrator actually built.

The counts follow the reference's default-mode formulas term by term (a product and an addition per term of a
convolution, pairwise sums; citations below) and are independent of how the HIP kernels evaluate them (FMA
contraction, lane splitting, history / finish splitting): they are the numerator of the `roofline` figures of
bench.py (F_alg flop and B_tape bytes per system-step).

  taylor_compute_jet()            src/taylor_02.cpp:1339-1418
  prod                            src/math/prod.cpp:352-395      (k + 1 products, k additions at order k)
  sum_sq                          src/detail/sum_sq.cpp:120-245
  pow                             src/math/pow.cpp:517-549        (k terms of 2 products, k - 1 additions, 1 product + 1 division)
  sin / cos pair                  src/math/sin.cpp:152-192        (k terms of 2 products, k - 1 additions, 1 division)
  state variables                 src/taylor_02.cpp:245-287       (one division per order)
  step-size selector              src/taylor_00.cpp:102-273
  Horner / compensated update     src/taylor_00.cpp:279-460
"""
import re

_CALL = re.compile(r"^(\w+)\((.*)\)$")


def _parse(entry):
    m = _CALL.match(entry.strip())
    if not m:
        return None, []
    args = [a.strip() for a in m.group(2).split(",")] if m.group(2).strip() else []
    return m.group(1), args


def _is_var(a):
    return a.startswith("u_")


def node_flops(kind, args, k):
    """Floating-point operations of the order-k coefficient of one elementary subexpression."""
    nv = sum(1 for a in args if _is_var(a))
    if kind in ("sum",):
        # Numbers / parameters only enter at order 0.
        return max((len(args) if k == 0 else nv) - 1, 0)
    if kind == "sub":
        return 1 if (nv == 2 or k == 0) else 0
    if kind == "prod":
        if nv == 2:
            return (k + 1) + k
        return 1 if nv == 1 else (1 if k == 0 else 0)
    if kind == "div":
        if len(args) == 2 and _is_var(args[1]):
            return 1 if k == 0 else (2 * k - 1) + 2
        return 1 if nv else 0
    if kind == "sum_sq":
        if k % 2 == 1:
            t = (k - 1) // 2 + 1
            return nv * (2 * t - 1) + (nv - 1) + 1
        t = k // 2  # cross terms
        per = 1 + ((2 * t - 1) + 2 if t > 0 else 0)
        return nv * per + (nv - 1)
    if kind == "pow":
        if k == 0:
            return 4  # sqrt / products / division of the order-0 evaluation
        return 2 * k + (k - 1) + 2
    if kind in ("sin", "cos", "exp", "log", "tan", "tanh", "sinh", "cosh", "erf", "sigmoid", "asin", "acos", "atan",
                "asinh", "acosh", "atanh", "atan2", "kepE"):
        if k == 0:
            return 20  # order-0 libm evaluation (nominal)
        return 2 * k + (k - 1) + 2
    if kind in ("time", "num_identity"):
        return 0
    return 1


def algorithmic_counts(decomposition, n_eq, order, high_accuracy, n_par=0):
    """(flop, tape bytes) per system-step for a decomposition given as the list of strings of
    taylor_adaptive_batch.decomposition (state variables, elementary subexpressions, trailing definitions)."""
    n_u = len(decomposition) - n_eq
    nodes = [_parse(e) for e in decomposition[n_eq:n_u]]
    defs = decomposition[n_u:]
    flops = 0
    for kind, args in nodes:
        if kind is None:
            continue
        for k in range(order):
            flops += node_flops(kind, args, k)
    # State variables: x^[k] = rhs^[k-1] / k for k = 1 .. order, when the definition is a u variable.
    flops += sum(order for d in defs if d.strip().startswith("u_"))
    # Step-size selector: 3 infinity norms over the state variables, 2 divisions, 2 roots (~20 flop each), min, scaling.
    flops += 3 * n_eq + 50
    # State update: Horner (1 product + 1 addition per order) or the compensated summation (6 per order, :430-456).
    flops += n_eq * order * (6 if high_accuracy else 2)
    b_min = 8 * (2 * n_eq + 6 + n_par)
    b_tape = b_min + 16 * n_u * order
    return flops, b_tape


def counts_for(ta):
    return algorithmic_counts(ta.decomposition, ta.dim, ta.order, bool(ta.high_accuracy))
