"""
TEST INFRASTRUCTURE - NOT PRODUCT CODE.

CPU oracle for the taylor_adaptive_batch<double> hot path of bluescarni/heyoka (v7.12.0).

This module is an *independent* restatement (pure Python) of the symbolic half of the path --
expression construction with the reference's constant folding, the rewrite pipeline, the Taylor
decomposition, CSE and the breadth-first re-sort -- and a ctypes driver for the numerical half
(oracle/taylor_oracle.c). It shares no code with the product (heyoka_amd/): tests compare the
product's decomposition and numerical results against it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Oracle pinning: the reference cannot be built in this environment (no LLVM/Boost/fmt/spdlog dev
files), so the oracle is pinned against the reference's published known
answers instead (tests/golden/*.json, transcribed from the reference's docs and tests):
doc/tut_adaptive.rst:96-229, doc/tut_ensemble.rst:120-139, doc/tut_batch_mode.rst:160-330,
test/model_nbody.cpp:92-118, test/taylor_*.cpp, test/timestep_check.cpp. sin/cos/pow/sqrt come
from the host libm (<= 1 ulp class): bitwise parity for those is unpinned.

Reference citations (file:line relative to /root/reference):
  operators ............... src/expression_ops.cpp:34-91
  sum()/prod()/pow() ...... src/math/sum.cpp:548-601, src/math/prod.cpp:913-973, src/math/pow.cpp:1024-1062
  sin()/cos() folding ..... src/math/sin.cpp:381-395, src/math/cos.cpp:381-395
  nbody / pendulum ........ src/model/nbody.cpp:53-174, src/model/pendulum.cpp:23-28
  traversal order ......... src/detail/ex_traversal.cpp:35-180
  sum_to_sub .............. src/math/sum.cpp:461-544
  sum/prod split .......... include/heyoka/detail/udf_split.hpp:49-100, src/expression_basic.cpp:1177-1213
  sum -> sum_sq ........... src/math/sum.cpp:385-455
  prod -> div ............. src/math/prod.cpp:753-908
  decomposition ........... src/expression_decompose.cpp:43-210, src/func.cpp:392-420
  sin/cos pairs ........... src/math/sin.cpp:115-133, src/math/cos.cpp:116-134
  taylor_decompose_sys .... src/taylor_01.cpp:848-1008
  CSE ..................... src/taylor_01.cpp:315-443
  BFS topological sort .... src/taylor_01.cpp:454-645
  order from tolerance .... include/heyoka/detail/taylor_common.hpp:165-191
"""

import ctypes
import math
import os
import subprocess
from collections import deque

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KIND_IDS = {
    "sum": 0,
    "prod": 1,
    "pow": 2,
    "sub": 3,
    "div": 4,
    "sum_sq": 5,
    "sin": 6,
    "cos": 7,
    "exp": 8,
    "log": 9,
    "time": 10,
    "num_identity": 11,
    "tan": 12,
    "tanh": 13,
    "sinh": 14,
    "cosh": 15,
    "erf": 16,
    "sigmoid": 17,
    "asin": 18,
    "acos": 19,
    "atan": 20,
    "asinh": 21,
    "acosh": 22,
    "atanh": 23,
    "atan2": 24,
    "kepE": 25,
    "relu": 26,
    "relup": 27,
    "select": 28,
    "logical_and": 29,
    "logical_or": 30,
    "rel_eq": 31,
    "rel_neq": 32,
    "rel_lt": 33,
    "rel_gt": 34,
    "rel_lte": 35,
    "rel_gte": 36,
    "kepF": 37,
    "kepDE": 38,
    "pi": 39,
}

OC_SUCCESS = -4294967296 - 1
OC_STEP_LIMIT = -4294967296 - 2
OC_TIME_LIMIT = -4294967296 - 3
OC_ERR_NF_STATE = -4294967296 - 4
OC_CB_STOP = -4294967296 - 5


# ----------------------------------------------------------------------------------------------
# Expressions.
# ----------------------------------------------------------------------------------------------
class Ex:
    """Expression node. tag in {'num', 'var', 'par', 'func'}."""

    __slots__ = ("tag", "val", "kind", "args", "_key")

    def __init__(self, tag, val=None, kind=None, args=()):
        self.tag = tag
        self.val = val
        self.kind = kind
        self.args = tuple(args)
        self._key = None

    # Structural key (used for CSE equality).
    def key(self):
        if self._key is None:
            if self.tag == "func":
                self._key = ("f", self.kind, tuple(a.key() for a in self.args))
            elif self.tag == "num":
                v = self.val
                self._key = ("n", "nan" if v != v else (0.0 if v == 0 else v))
            else:
                self._key = (self.tag, self.val)
        return self._key

    def is_num(self):
        return self.tag == "num"

    def is_func(self, kind=None):
        return self.tag == "func" and (kind is None or self.kind == kind)

    def __repr__(self):
        if self.tag == "func":
            return "%s(%s)" % (self.kind, ", ".join(map(repr, self.args)))
        if self.tag == "par":
            return "p%d" % self.val
        return repr(self.val) if self.tag == "num" else self.val

    # Operators.
    def __neg__(self):
        if self.is_num():
            return num(-self.val)
        return prod([num(-1.0), self])

    def __add__(self, o):
        o = as_ex(o)
        if self.is_num() and o.is_num():
            return num(self.val + o.val)
        return sum_([self, o])

    def __radd__(self, o):
        return as_ex(o) + self

    def __sub__(self, o):
        o = as_ex(o)
        if self.is_num() and o.is_num():
            return num(self.val - o.val)
        return self + (-o)

    def __rsub__(self, o):
        return as_ex(o) - self

    def __mul__(self, o):
        o = as_ex(o)
        if self.is_num() and o.is_num():
            return num(self.val * o.val)
        return prod([self, o])

    def __rmul__(self, o):
        return as_ex(o) * self

    def __truediv__(self, o):
        o = as_ex(o)
        if self.is_num() and o.is_num():
            return num(np.float64(self.val) / np.float64(o.val))
        return prod([self, pow_(o, num(-1.0))])

    def __rtruediv__(self, o):
        return as_ex(o) / self


def num(v):
    return Ex("num", float(v))


def var(name):
    return Ex("var", name)


def par(i):
    return Ex("par", int(i))


def func(kind, args):
    return Ex("func", kind=kind, args=args)


def as_ex(x):
    return x if isinstance(x, Ex) else num(x)


TIME = func("time", [])
# heyoka::pi: a function without arguments with its own u variable (include/heyoka/math/constants.hpp:117; order 0 = the
# value, 0 beyond: src/math/constants.cpp:258-273).
PI = func("pi", [])


def _stable_partition(lst, pred):
    a = [x for x in lst if pred(x)]
    b = [x for x in lst if not pred(x)]
    return a + b, len(a)


def sum_(args):
    args = [as_ex(a) for a in args]
    args, n_nonnum = _stable_partition(args, lambda e: not e.is_num())
    if n_nonnum != len(args):
        acc = args[n_nonnum].val
        for e in args[n_nonnum + 1 :]:
            acc = acc + e.val
        args = args[:n_nonnum] + [num(acc)]
        if acc == 0:
            if len(args) == 1:
                return args[0]
            args.pop()
    if not args:
        return num(0.0)
    if len(args) == 1:
        return args[0]
    args, _ = _stable_partition(args, lambda e: e.is_num())
    return func("sum", args)


def prod(args):
    args = [as_ex(a) for a in args]
    args, n_nonnum = _stable_partition(args, lambda e: not e.is_num())
    if n_nonnum != len(args):
        acc = args[n_nonnum].val
        for e in args[n_nonnum + 1 :]:
            acc = acc * e.val
        args = args[:n_nonnum] + [num(acc)]
        if acc == 1:
            if len(args) == 1:
                return args[0]
            args.pop()
        elif acc == 0:
            return args[-1]
    if not args:
        return num(1.0)
    if len(args) == 1:
        return args[0]
    args, _ = _stable_partition(args, lambda e: e.is_num())
    return func("prod", args)


def pow_(b, e):
    b, e = as_ex(b), as_ex(e)
    if b.is_num() and e.is_num():
        return num(math.pow(b.val, e.val))
    if e.is_num():
        if e.val == 0:
            return num(1.0)
        if e.val == 1:
            return b
    return func("pow", [b, e])


def sqrt(e):
    return pow_(e, num(0.5))


def sin(e):
    e = as_ex(e)
    return num(math.sin(e.val)) if e.is_num() else func("sin", [e])


def cos(e):
    e = as_ex(e)
    return num(math.cos(e.val)) if e.is_num() else func("cos", [e])


def exp(e):
    e = as_ex(e)
    return num(math.exp(e.val)) if e.is_num() else func("exp", [e])


def log(e):
    e = as_ex(e)
    return num(math.log(e.val)) if e.is_num() else func("log", [e])


def _unary(name, ev):
    def f(e):
        e = as_ex(e)
        return num(ev(e.val)) if e.is_num() else func(name, [e])

    f.__name__ = name
    return f


# Elementary functions beyond the N-body set (reference: src/math/{tan,tanh,...}.cpp).
tan = _unary("tan", math.tan)
tanh = _unary("tanh", math.tanh)
sinh = _unary("sinh", math.sinh)
cosh = _unary("cosh", math.cosh)
asin = _unary("asin", math.asin)
acos = _unary("acos", math.acos)
atan = _unary("atan", math.atan)
asinh = _unary("asinh", math.asinh)
acosh = _unary("acosh", math.acosh)
atanh = _unary("atanh", math.atanh)
erf = _unary("erf", math.erf)
sigmoid = _unary("sigmoid", lambda x: 1.0 / (1.0 + math.exp(-x)))


def atan2(y, x):
    """Reference: src/math/atan2.cpp:763-786 (two numbers fold)."""
    y, x = as_ex(y), as_ex(x)
    if y.is_num() and x.is_num():
        return num(math.atan2(y.val, x.val))
    return func("atan2", [y, x])


def relu(x, slope=0.0):
    """relu(x, slope) = x > 0 ? x : slope * x; the slope is kept as a second, numerical argument
    (reference: src/math/relu.cpp:580-590)."""
    x = as_ex(x)
    if not (math.isfinite(slope) and slope >= 0):
        raise ValueError("invalid slope")
    if x.is_num():
        return num(x.val if x.val > 0 else slope * x.val)
    return func("relu", [x, num(float(slope))])


def relup(x, slope=0.0):
    """Derivative of relu (src/math/relu.cpp:592-602)."""
    x = as_ex(x)
    if not (math.isfinite(slope) and slope >= 0):
        raise ValueError("invalid slope")
    if x.is_num():
        return num(1.0 if x.val > 0 else float(slope))
    return func("relup", [x, num(float(slope))])


def select(cond, t, f):
    """select(c, t, f) = c != 0 ? t : f (src/math/select.cpp:267-270)."""
    return func("select", [as_ex(cond), as_ex(t), as_ex(f)])


def logical_and(args):
    """src/math/logical.cpp:314-325."""
    args = [as_ex(a) for a in args]
    return num(1.0) if not args else (args[0] if len(args) == 1 else func("logical_and", args))


def logical_or(args):
    """src/math/logical.cpp:327-338."""
    args = [as_ex(a) for a in args]
    return num(0.0) if not args else (args[0] if len(args) == 1 else func("logical_or", args))


def _rel(name):
    def f(a, b):
        return func("rel_" + name, [as_ex(a), as_ex(b)])

    f.__name__ = name
    return f


# Comparisons returning 1 / 0 (src/math/relational.cpp:343-354).
eq, neq, lt, gt, lte, gte = (_rel(n) for n in ("eq", "neq", "lt", "gt", "lte", "gte"))


def inv_kep_E(ecc, M):
    """The oracle's Kepler solver (C restatement of llvm_add_inv_kep_E())."""
    return _lib().hy_oracle_inv_kep_E(float(ecc), float(M))


def kepE(e, M):
    """Eccentric anomaly (src/math/kepE.cpp:801-809): zero eccentricity gives M, nothing else folds."""
    e, M = as_ex(e), as_ex(M)
    if e.is_num() and e.val == 0:
        return M
    return func("kepE", [e, M])


def kepF(h, k, lam):
    """Eccentric longitude F(h, k, lam), F + h cos F - k sin F = lam (src/math/kepF.cpp:1689-1699): h = k = 0 gives lam."""
    h, k, lam = as_ex(h), as_ex(k), as_ex(lam)
    if h.is_num() and k.is_num() and h.val == 0 and k.val == 0:
        return lam
    return func("kepF", [h, k, lam])


def kepDE(s0, c0, DM):
    """DE(s0, c0, DM), DE - c0 sin DE + s0 (1 - cos DE) = DM (src/math/kepDE.cpp:113-123): s0 = c0 = 0 gives DM."""
    s0, c0, DM = as_ex(s0), as_ex(c0), as_ex(DM)
    if s0.is_num() and c0.is_num() and s0.val == 0 and c0.val == 0:
        return DM
    return func("kepDE", [s0, c0, DM])


def inv_kep_F(h, k, lam):
    return _lib().hy_oracle_inv_kep_F(float(h), float(k), float(lam))


def inv_kep_DE(s0, c0, DM):
    return _lib().hy_oracle_inv_kep_DE(float(s0), float(c0), float(DM))


# ----------------------------------------------------------------------------------------------
# Models.
# ----------------------------------------------------------------------------------------------
def nbody(n, masses=None, Gconst=1.0):
    masses = [as_ex(1.0)] * n if masses is None else [as_ex(m) for m in masses]
    G = as_ex(Gconst)
    if n < 2 or len(masses) > n:
        raise ValueError("invalid N-body configuration")
    x = [var("x_%d" % i) for i in range(n)]
    y = [var("y_%d" % i) for i in range(n)]
    z = [var("z_%d" % i) for i in range(n)]
    vx = [var("vx_%d" % i) for i in range(n)]
    vy = [var("vy_%d" % i) for i in range(n)]
    vz = [var("vz_%d" % i) for i in range(n)]
    xa = [[] for _ in range(n)]
    ya = [[] for _ in range(n)]
    za = [[] for _ in range(n)]
    sys = []
    nm = len(masses)
    for i in range(nm):
        sys += [(x[i], vx[i]), (y[i], vy[i]), (z[i], vz[i])]
        for j in range(i + 1, n):
            dx, dy, dz = x[j] - x[i], y[j] - y[i], z[j] - z[i]
            r_m3 = pow_(sum_([pow_(dx, 2.0), pow_(dy, 2.0), pow_(dz, 2.0)]), num(-3.0 / 2))
            j_massive = j < nm
            opt = j_massive and masses[j].is_num() and masses[j].val != 0 and G.is_num()
            if opt:
                fac_j = G * masses[j] * r_m3
                c_ij = -masses[i] / masses[j]
                xa[i].append(dx * fac_j)
                ya[i].append(dy * fac_j)
                za[i].append(dz * fac_j)
                xa[j].append(xa[i][-1] * c_ij)
                ya[j].append(ya[i][-1] * c_ij)
                za[j].append(za[i][-1] * c_ij)
            else:
                G_r_m3 = G * r_m3
                fac_i = -masses[i] * G_r_m3
                xa[j].append(dx * fac_i)
                ya[j].append(dy * fac_i)
                za[j].append(dz * fac_i)
                if j_massive:
                    fac_j = masses[j] * G_r_m3
                    xa[i].append(dx * fac_j)
                    ya[i].append(dy * fac_j)
                    za[i].append(dz * fac_j)
        sys += [(vx[i], sum_(xa[i])), (vy[i], sum_(ya[i])), (vz[i], sum_(za[i]))]
    for i in range(nm, n):
        sys += [(x[i], vx[i]), (y[i], vy[i]), (z[i], vz[i])]
        sys += [(vx[i], sum_(xa[i])), (vy[i], sum_(ya[i])), (vz[i], sum_(za[i]))]
    return sys


def pendulum(gconst=1.0, length=1.0):
    x, v = var("x"), var("v")
    return [(x, v), (v, -as_ex(gconst) / as_ex(length) * sin(x))]


def _r2(dx, dy, dz):
    return sum_([pow_(dx, 2.0), pow_(dy, 2.0), pow_(dz, 2.0)])


def np1body(n, masses=None, Gconst=1.0):
    """N+1 bodies in the frame of body 0 (reference: src/model/nbody.cpp:236-325)."""
    masses = [as_ex(1.0)] * n if masses is None else [as_ex(m) for m in masses]
    G = as_ex(Gconst)
    if n < 2 or len(masses) > n:
        raise ValueError("invalid N-body configuration")
    rng = range(1, n)
    x = [var("x_%d" % i) for i in rng]
    y = [var("y_%d" % i) for i in rng]
    z = [var("z_%d" % i) for i in rng]
    vx = [var("vx_%d" % i) for i in rng]
    vy = [var("vy_%d" % i) for i in rng]
    vz = [var("vz_%d" % i) for i in rng]
    rm3 = [pow_(_r2(x[i], y[i], z[i]), num(-3.0 / 2)) for i in range(n - 1)]
    xr3 = [x[i] * rm3[i] for i in range(n - 1)]
    yr3 = [y[i] * rm3[i] for i in range(n - 1)]
    zr3 = [z[i] * rm3[i] for i in range(n - 1)]
    nm = len(masses)
    sys = []
    for i in range(n - 1):
        sys += [(x[i], vx[i]), (y[i], vy[i]), (z[i], vz[i])]
        m0 = masses[0] if nm > 0 else num(0.0)
        mi = masses[i + 1] if i + 1 < nm else num(0.0)
        mu0i = -G * (m0 + mi)
        ax, ay, az = [mu0i * xr3[i]], [mu0i * yr3[i]], [mu0i * zr3[i]]
        for j in range(max(nm - 1, 0)):
            if j == i:
                continue
            fwd = j > i
            dx = x[j] - x[i] if fwd else x[i] - x[j]
            dy = y[j] - y[i] if fwd else y[i] - y[j]
            dz = z[j] - z[i] if fwd else z[i] - z[j]
            drm3 = pow_(_r2(dx, dy, dz), num(-1.5))
            mu = G * masses[j + 1]
            tx, ty, tz = mu * (dx * drm3), mu * (dy * drm3), mu * (dz * drm3)
            ax.append(tx if fwd else -tx)
            ay.append(ty if fwd else -ty)
            az.append(tz if fwd else -tz)
            ax.append(-mu * xr3[j])
            ay.append(-mu * yr3[j])
            az.append(-mu * zr3[j])
        sys += [(vx[i], sum_(ax)), (vy[i], sum_(ay)), (vz[i], sum_(az))]
    return sys


def _cr3bp_check(mu):
    if mu.is_num() and not (np.isfinite(mu.val) and 0 < mu.val < 0.5):
        raise ValueError("The 'mu' parameter in a CR3BP must be in the range (0, 0.5)")


def cr3bp(mu=1e-3):
    """Reference: src/model/cr3bp.cpp (state x, y, z, px, py, pz)."""
    mu = as_ex(mu)
    _cr3bp_check(mu)
    px, py, pz, x, y, z = (var(n) for n in ("px", "py", "pz", "x", "y", "z"))
    d1 = x - mu
    d2 = d1 + 1.0
    yz2 = pow_(y, 2.0) + pow_(z, 2.0)
    r1_2 = pow_(d1, 2.0) + yz2
    r2_2 = pow_(d2, 2.0) + yz2
    g1 = (1.0 - mu) * pow_(r1_2, num(-3.0 / 2))
    g2 = mu * pow_(r2_2, num(-3.0 / 2))
    g12 = g1 + g2
    return [(x, px + y), (y, py - x), (z, pz), (px, py - g1 * d1 - g2 * d2), (py, -px - g12 * y), (pz, -g12 * z)]


def fixed_centres(Gconst=1.0, masses=(), positions=()):
    """Reference: src/model/fixed_centres.cpp."""
    G = as_ex(Gconst)
    masses = [as_ex(m) for m in masses]
    positions = [as_ex(p) for p in positions]
    if len(positions) % 3 != 0 or len(positions) // 3 != len(masses):
        raise ValueError("invalid fixed centres configuration")
    x, y, z, vx, vy, vz = (var(n) for n in ("x", "y", "z", "vx", "vy", "vz"))
    ax, ay, az = [], [], []
    for i, m in enumerate(masses):
        dx, dy, dz = positions[3 * i] - x, positions[3 * i + 1] - y, positions[3 * i + 2] - z
        mrm3 = m * pow_(_r2(dx, dy, dz), num(-1.5))
        ax.append(dx * mrm3)
        ay.append(dy * mrm3)
        az.append(dz * mrm3)
    return [(x, vx), (y, vy), (z, vz), (vx, G * sum_(ax)), (vy, G * sum_(ay)), (vz, G * sum_(az))]


def rotating(omega=()):
    """Reference: src/model/rotating.cpp."""
    omega = [as_ex(o) for o in omega]
    if omega and len(omega) != 3:
        raise ValueError("invalid angular velocity")
    x, y, z, vx, vy, vz = (var(n) for n in ("x", "y", "z", "vx", "vy", "vz"))
    ax, ay, az = [], [], []
    if omega:
        p, q, r = omega
        qx, rx, qy, rz = q * x, r * x, q * y, r * z
        ax = [q * qx, r * rx, -(p * qy), -(p * rz)]
        ay = [pow_(p, 2.0) * y, pow_(r, 2.0) * y, -(p * qx), -(q * rz)]
        az = [pow_(p, 2.0) * z, pow_(q, 2.0) * z, -(p * rx), -(r * qy)]
        ax.append(num(-2.0) * (q * vz - r * vy))
        ay.append(num(-2.0) * (r * vx - p * vz))
        az.append(num(-2.0) * (p * vy - q * vx))
    return [(x, vx), (y, vy), (z, vz), (vx, sum_(ax)), (vy, sum_(ay)), (vz, sum_(az))]


def mascon(Gconst=1.0, masses=(), positions=(), omega=()):
    """Reference: src/model/mascon.cpp."""
    fc = fixed_centres(Gconst, masses, positions)
    rot = rotating(omega)
    return fc[:3] + [(fc[i][0], fc[i][1] + rot[i][1]) for i in range(3, 6)]


# ----------------------------------------------------------------------------------------------
# Rewrites + decomposition.
# ----------------------------------------------------------------------------------------------
def _transform(cache, e, leaf_f, branch_f):
    """Post-order transform; shared nodes (by identity) transformed once; children visited
    last-argument-first like the reference (only matters for side effects of the callbacks)."""
    if e.tag != "func":
        return leaf_f(e) if leaf_f else e
    hit = cache.get(id(e))
    if hit is not None:
        return hit[1]
    new_args = [None] * len(e.args)
    for i in range(len(e.args) - 1, -1, -1):
        new_args[i] = _transform(cache, e.args[i], leaf_f, branch_f)
    ne = e if all(a is b for a, b in zip(new_args, e.args)) else func(e.kind, new_args)
    if branch_f:
        ne = branch_f(ne)
    cache[id(e)] = (e, ne)  # keep e alive so that ids are not recycled
    return ne


def _transform_all(v_ex, branch_f):
    cache = {}
    return [_transform(cache, e, None, branch_f) for e in v_ex]


def _pow_to_explog(ex):
    if ex.kind == "pow" and not ex.args[1].is_num():
        return exp(ex.args[1] * func("log", [ex.args[0]]))
    return ex


def _sum_to_sub(ex):
    if ex.kind != "sum":
        return ex

    def keep(a):
        if a.is_func("prod") and len(a.args) >= 2 and a.args[0].is_num():
            return a.args[0].val != -1
        return True

    new_args, n_keep = _stable_partition(list(ex.args), keep)
    if n_keep == len(new_args):
        return ex
    sub_args = [prod(list(a.args[1:])) for a in new_args[n_keep:]]
    st = sum_(sub_args)
    if n_keep == 0:
        return prod([num(-1.0), st])
    return func("sub", [sum_(new_args[:n_keep]), st])


def _udf_split(ex, kind, split):
    while ex.is_func(kind) and len(ex.args) > split:
        seq, tmp = [], []
        for a in ex.args:
            tmp.append(a)
            if len(tmp) == split:
                seq.append(func(kind, tmp))
                tmp = []
        if tmp:
            seq.append(tmp[0] if len(tmp) == 1 else func(kind, tmp))
        ex = func(kind, seq)
    return ex


def _sum_to_sum_sq(ex):
    if ex.kind != "sum":
        return ex
    bases = []
    for a in ex.args:
        if a.is_func("pow") and a.args[1].is_num() and a.args[1].val == 2:
            bases.append(a.args[0])
        else:
            return ex
    return func("sum_sq", bases)


def _prod_to_div(ex):
    if ex.kind != "prod":
        return ex

    def keep(a):
        return not (a.is_func("pow") and a.args[1].is_num() and a.args[1].val == -1)

    new_args, n_keep = _stable_partition(list(ex.args), keep)
    if n_keep == len(new_args):
        return ex
    divisor = prod([pow_(a.args[0], num(-a.args[1].val)) for a in new_args[n_keep:]])
    return func("div", [prod(new_args[:n_keep]), divisor])


def _uname(i):
    return "u_%d" % i


def _uidx(name):
    return int(name[2:])


def _rename(e, m):
    return _transform({}, e, lambda l: var(m[l.val]) if l.tag == "var" and l.val in m else l, None)


def _get_vars(e, out):
    if e.tag == "var":
        out.add(e.val)
    elif e.tag == "func":
        for a in e.args:
            _get_vars(a, out)


def taylor_decompose_sys(sys, sv_funcs=None):
    """Returns dc = list of (Ex, deps); with sv_funcs (extra functions of the state, e.g. event equations,
    src/taylor_01.cpp:848-1008) returns (dc, sv_funcs_dc) with the u-variable index of each function."""
    n_eq = len(sys)
    n_sv = 0 if sv_funcs is None else len(sv_funcs)
    repl = {lhs.val: _uname(i) for i, (lhs, _) in enumerate(sys)}
    all_ex = [rhs for _, rhs in sys] + ([] if sv_funcs is None else [as_ex(f) for f in sv_funcs])
    all_ex = _transform_all(all_ex, _pow_to_explog)
    all_ex = _transform_all(all_ex, _sum_to_sub)
    all_ex = _transform_all(all_ex, lambda e: _udf_split(e, "sum", 8))
    all_ex = _transform_all(all_ex, _sum_to_sum_sq)
    all_ex = _transform_all(all_ex, _prod_to_div)
    all_ex = _transform_all(all_ex, lambda e: _udf_split(e, "prod", 2))
    cache = {}
    all_ex = [_transform(cache, e, lambda l: var(repl[l.val]) if l.tag == "var" else l, None) for e in all_ex]

    dc = [(lhs, []) for lhs, _ in sys]
    fmap = {}

    def decomp(e):
        # Returns the index of the u variable, or None for non-functions.
        if e.tag != "func":
            return None
        hit = fmap.get(id(e))
        if hit is not None:
            return hit[1]
        idxs = [None] * len(e.args)
        for i in range(len(e.args) - 1, -1, -1):  # last argument first
            idxs[i] = decomp(e.args[i])
        new_args = [var(_uname(ix)) if ix is not None else a for ix, a in zip(idxs, e.args)]
        f = func(e.kind, new_args)
        pairs = {"sin": "cos", "cos": "sin", "sinh": "cosh", "cosh": "sinh"}
        if e.kind in pairs:
            dc.append((func(pairs[e.kind], [new_args[0]]), []))
            dc.append((f, []))
            dc[-2][1].append(len(dc) - 1)
            dc[-1][1].append(len(dc) - 2)
            ret = len(dc) - 1
        elif e.kind in ("tan", "tanh", "sigmoid"):
            # f(b), then its square on which it depends (src/math/tan.cpp:69-85).
            dc.append((f, []))
            ret = len(dc) - 1
            dc.append((pow_(var(_uname(ret)), num(2.0)), []))
            dc[ret][1].append(ret + 1)
        elif e.kind in ("asin", "acos", "asinh", "acosh"):
            # b^2 -> 1 - b^2 | 1 + b^2 | b^2 - 1 -> sqrt -> f(b) depending on the square root (src/math/asin.cpp:77-108).
            dc.append((pow_(new_args[0], num(2.0)), []))
            sq = var(_uname(len(dc) - 1))
            if e.kind in ("asin", "acos"):
                dc.append((func("sub", [num(1.0), sq]), []))
            elif e.kind == "asinh":
                dc.append((num(1.0) + sq, []))
            else:
                dc.append((sq - num(1.0), []))
            dc.append((sqrt(var(_uname(len(dc) - 1))), []))
            dc.append((f, [len(dc) - 1]))
            ret = len(dc) - 1
        elif e.kind in ("atan", "atanh"):
            dc.append((pow_(new_args[0], num(2.0)), []))
            dc.append((f, [len(dc) - 1]))
            ret = len(dc) - 1
        elif e.kind == "atan2":
            # y^2 + x^2 -> atan2(y, x) depending on it (src/math/atan2.cpp:92-108).
            dc.append((func("sum_sq", [new_args[0], new_args[1]]), []))
            dc.append((f, [len(dc) - 1]))
            ret = len(dc) - 1
        elif e.kind == "kepE":
            # E -> sin E -> cos E -> e cos E; E depends on (e cos E, sin E) (src/math/kepE.cpp:100-135).
            dc.append((f, []))
            ret = len(dc) - 1
            uE = var(_uname(ret))
            dc.append((sin(uE), []))
            dc.append((cos(uE), []))
            dc.append((new_args[0] * var(_uname(ret + 2)), []))
            dc[ret][1].extend([ret + 3, ret + 1])
            dc[ret + 1][1].append(ret + 2)
            dc[ret + 2][1].append(ret + 1)
        elif e.kind in ("kepF", "kepDE"):
            # a -> sin a -> cos a -> (h sin a, k cos a) for kepF (src/math/kepF.cpp:110-156), (c0 cos a, s0 sin a) for
            # kepDE; a depends on (c, d, sin a, cos a) in this order, sin and cos on each other.
            dc.append((f, []))
            ret = len(dc) - 1
            ua = var(_uname(ret))
            dc.append((sin(ua), [ret + 2]))
            dc.append((cos(ua), [ret + 1]))
            # (Product nodes as they stand: the folding operator* of the reference, src/math/kepF.cpp:131-134, turns
            # 0 * x / 1 * x into a number / the bare variable - not a valid entry of a decomposition.)
            if e.kind == "kepF":
                dc.append((func("prod", [new_args[0], var(_uname(ret + 1))]), []))
                dc.append((func("prod", [new_args[1], var(_uname(ret + 2))]), []))
            else:
                dc.append((func("prod", [new_args[1], var(_uname(ret + 2))]), []))
                dc.append((func("prod", [new_args[0], var(_uname(ret + 1))]), []))
            dc[ret][1].extend([ret + 3, ret + 4, ret + 1, ret + 2])
        elif e.kind == "erf":
            dc.append((pow_(new_args[0], num(2.0)), []))
            dc.append((-var(_uname(len(dc) - 1)), []))
            dc.append((exp(var(_uname(len(dc) - 1))), []))
            dc.append((f, [len(dc) - 1]))
            ret = len(dc) - 1
        else:
            ret = len(dc)
            dc.append((f, []))
        fmap[id(e)] = (e, ret)
        return ret

    outs = []
    for e in all_ex[:n_eq]:
        r = decomp(e)
        outs.append((var(_uname(r)) if r is not None else e, []))
    # Extra functions: decomposed after the right-hand sides; they ride along as additional trailing
    # entries through CSE and sorting (which renumber them) and are stripped at the end.
    for e in all_ex[n_eq:]:
        if e.tag == "var":
            outs.append((e, []))
        else:
            r = decomp(e)
            if r is None:
                raise ValueError("The extra functions in a Taylor decomposition cannot be constants or parameters")
            outs.append((var(_uname(r)), []))
    dc += outs
    n_outs = n_eq + n_sv

    # CSE.
    new_dc = list(dc[:n_eq])
    ex_map = {}
    ren = {_uname(i): _uname(i) for i in range(n_eq)}
    for i in range(n_eq, len(dc) - n_outs):
        ex, deps = dc[i]
        ne = _rename(ex, ren)
        j = ex_map.get(ne.key())
        if j is None:
            new_dc.append((ne, list(deps)))
            ex_map[ne.key()] = len(new_dc) - 1
            ren[_uname(i)] = _uname(len(new_dc) - 1)
        else:
            ren[_uname(i)] = _uname(j)
    for i in range(len(dc) - n_outs, len(dc)):
        new_dc.append((_rename(dc[i][0], ren), []))
    new_dc = [(ex, [_uidx(ren[_uname(d)]) for d in deps]) for ex, deps in new_dc]
    dc = new_dc

    # Kahn BFS sort. Vertex 0 = root, vertex i+1 = u_i.
    n_vert = len(dc) - n_outs + 1
    out_edges = [[] for _ in range(n_vert)]
    indeg = [0] * n_vert
    for i in range(n_eq):
        out_edges[0].append(i + 1)
        indeg[i + 1] += 1
    for i in range(n_eq, len(dc) - n_outs):
        vs = set()
        _get_vars(dc[i][0], vs)
        if not vs:
            out_edges[0].append(i + 1)
            indeg[i + 1] += 1
        else:
            for v in vs:
                out_edges[_uidx(v) + 1].append(i + 1)
                indeg[i + 1] += 1
    order_v = []
    q = deque([0])
    while q:
        v = q.popleft()
        order_v.append(v)
        for t in sorted(out_edges[v]):
            indeg[t] -= 1
            if indeg[t] == 0:
                q.append(t)
    assert len(order_v) == n_vert
    v_idx = [v - 1 for v in order_v[1:]] + list(range(len(dc) - n_outs, len(dc)))
    remap = {_uname(v_idx[i]): _uname(i) for i in range(len(dc) - n_outs)}
    dc = [(_rename(dc[ix][0], remap), [_uidx(remap[_uname(d)]) for d in dc[ix][1]]) for ix in v_idx]

    # Numbers -> num_identity.
    for i in range(n_eq, len(dc) - n_outs):
        if dc[i][0].tag == "num":
            dc[i] = (func("num_identity", [dc[i][0]]), [])
    if sv_funcs is None:
        return dc
    sv_funcs_dc = [_uidx(ex.val) for ex, _ in dc[len(dc) - n_sv:]] if n_sv else []
    return dc[: len(dc) - n_sv], sv_funcs_dc


def taylor_order_from_tol(tol):
    return int(max(2.0, math.ceil(-math.log(tol) / 2 + 1)))


def dc_to_strings(dc):
    """Canonical textual form of a decomposition (for comparison with the product's)."""
    out = []
    for ex, deps in dc:
        out.append(_ex_str(ex) + "".join(" [dep %d]" % d for d in deps))
    return out


def _ex_str(e):
    if e.tag == "num":
        return "%.17g" % e.val
    if e.tag == "var":
        return e.val
    if e.tag == "par":
        return "p%d" % e.val
    return "%s(%s)" % (e.kind, ", ".join(_ex_str(a) for a in e.args))


# ----------------------------------------------------------------------------------------------
# Flat program + ctypes driver for oracle/taylor_oracle.c.
# ----------------------------------------------------------------------------------------------
class _CProg(ctypes.Structure):
    _fields_ = [
        ("n_eq", ctypes.c_int32),
        ("n_u", ctypes.c_int32),
        ("n_par", ctypes.c_int32),
        ("order", ctypes.c_int32),
        ("n_nodes", ctypes.c_int32),
        ("high_accuracy", ctypes.c_int32),
        ("kind", ctypes.c_void_p),
        ("arg_off", ctypes.c_void_p),
        ("arg_type", ctypes.c_void_p),
        ("arg_idx", ctypes.c_void_p),
        ("arg_val", ctypes.c_void_p),
        ("dep", ctypes.c_void_p),
        ("sv_type", ctypes.c_void_p),
        ("sv_idx", ctypes.c_void_p),
        ("sv_val", ctypes.c_void_p),
        ("dep2", ctypes.c_void_p),
        ("dep3", ctypes.c_void_p),
        ("dep4", ctypes.c_void_p),
    ]


def _operand(e):
    if e.tag == "var":
        return 0, _uidx(e.val), 0.0
    if e.tag == "num":
        return 1, 0, e.val
    if e.tag == "par":
        return 2, e.val, 0.0
    raise ValueError("invalid operand in decomposition: %r" % (e,))


def cpu_tag():
    """Fingerprint of the host CPU's instruction-set flags: libraries built with -march=native on one machine are
    rebuilt on a different one (the repository snapshot, built files included, travels to the GPU box)."""
    import hashlib

    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha256(" ".join(sorted(line.split(":", 1)[1].split())).encode()).hexdigest()[:12]
    except OSError:
        pass
    return "unknown"


def build_oracle_lib(force=False):
    """Compile oracle/taylor_oracle.c into oracle/_build/libtaylor_oracle.so (strict IEEE)."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "libtaylor_oracle.so")
    stamp = so + ".cpu"
    src = os.path.join(_HERE, "taylor_oracle.c")
    same_cpu = os.path.exists(stamp) and open(stamp).read().strip() == cpu_tag()
    if force or not same_cpu or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-march=native", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", so, src, "-lm"]
        )
        with open(stamp, "w") as f:
            f.write(cpu_tag())
    return so


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle_lib())
        _LIB.hy_oracle_scratch_size.restype = ctypes.c_size_t
        _LIB.hy_oracle_scratch_size_e.restype = ctypes.c_size_t
        _LIB.hy_oracle_ensemble_propagate_until.restype = ctypes.c_int64
        _LIB.hy_oracle_max_threads.restype = ctypes.c_int
        _LIB.hy_oracle_inv_kep_E.restype = ctypes.c_double
        _LIB.hy_oracle_inv_kep_E.argtypes = [ctypes.c_double, ctypes.c_double]
        for _n in ("hy_oracle_inv_kep_F", "hy_oracle_inv_kep_DE"):
            getattr(_LIB, _n).restype = ctypes.c_double
            getattr(_LIB, _n).argtypes = [ctypes.c_double] * 3
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleIntegrator:
    """Batch Taylor integrator on the CPU with the reference's semantics (lock-step batch).

    Arrays use the reference layout array[row * batch_size + lane].
    """

    def __init__(self, sys, state, batch_size, tol=None, high_accuracy=False, pars=None, time=None, compact_mode=False):
        self.sys = sys
        self.dc = taylor_decompose_sys(sys)
        self.n_eq = len(sys)
        self.n_u = len(self.dc) - self.n_eq
        self.batch_size = int(batch_size)
        self.tol = np.finfo(np.float64).eps if tol is None else float(tol)
        self.order = taylor_order_from_tol(self.tol)
        self.high_accuracy = bool(high_accuracy)
        self.compact_mode = bool(compact_mode)
        B = self.batch_size
        self._build_program()

        self.state = np.ascontiguousarray(np.array(state, dtype=np.float64).reshape(-1))
        if self.state.size != self.n_eq * B:
            raise ValueError("inconsistent state size")
        self.pars = np.zeros(max(self.n_par, 1) * B) if pars is None else np.ascontiguousarray(np.array(pars, dtype=np.float64).reshape(-1))
        self.time_hi = np.zeros(B) if time is None else np.ascontiguousarray(np.broadcast_to(np.array(time, dtype=np.float64), (B,)).copy())
        self.time_lo = np.zeros(B)
        self.tc = np.zeros(self.n_eq * (self.order + 1) * B)
        self.last_h = np.zeros(B)
        self.step_res = [(OC_SUCCESS, 0.0)] * B
        self.prop_res = None
        self._scratch = np.zeros(_lib().hy_oracle_scratch_size(ctypes.byref(self._prog), B) + 64)

    def _build_program(self):
        kinds, arg_off, at, ai, av, dep, dep2, dep3, dep4 = [], [0], [], [], [], [], [], [], []
        n_par = 0
        for ex, deps in self.dc[self.n_eq : self.n_u]:
            kinds.append(KIND_IDS[ex.kind])
            for a in ex.args:
                t, i, v = _operand(a)
                at.append(t)
                ai.append(i)
                av.append(v)
                if t == 2:
                    n_par = max(n_par, i + 1)
            arg_off.append(len(at))
            dep.append(deps[0] if deps else -1)
            dep2.append(deps[1] if len(deps) > 1 else -1)
            dep3.append(deps[2] if len(deps) > 2 else -1)
            dep4.append(deps[3] if len(deps) > 3 else -1)
        svt, svi, svv = [], [], []
        for ex, _ in self.dc[self.n_u :]:
            t, i, v = _operand(ex)
            svt.append(t)
            svi.append(i)
            svv.append(v)
            if t == 2:
                n_par = max(n_par, i + 1)
        self.n_par = n_par
        i32 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.int32))
        f64 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.float64))
        self._arrs = dict(
            kind=i32(kinds),
            arg_off=i32(arg_off),
            arg_type=i32(at if at else [0]),
            arg_idx=i32(ai if ai else [0]),
            arg_val=f64(av if av else [0.0]),
            dep=i32(dep if dep else [0]),
            dep2=i32(dep2 if dep2 else [0]),
            dep3=i32(dep3 if dep3 else [0]),
            dep4=i32(dep4 if dep4 else [0]),
            sv_type=i32(svt),
            sv_idx=i32(svi),
            sv_val=f64(svv),
        )
        self._prog = _CProg(
            self.n_eq,
            self.n_u,
            self.n_par,
            self.order,
            len(kinds),
            # (bit 0: compensated summation; bit 1: the arithmetic of the reference's compact mode - running sums inside
            # the convolutions, see HY_COMPACT in taylor_oracle.c.)
            int(self.high_accuracy) | (2 if self.compact_mode else 0),
            *[_p(self._arrs[k]) for k in ("kind", "arg_off", "arg_type", "arg_idx", "arg_val", "dep", "sv_type", "sv_idx", "sv_val", "dep2", "dep3", "dep4")]
        )

    def program_ptr(self):
        """Pointer to the C program structure (hy_oracle_program) of this integrator."""
        return ctypes.byref(self._prog)

    # step(max_delta_ts=None, wtc=False); max_delta_ts: signed per-lane limits (default +inf).
    def step(self, max_delta_ts=None, wtc=False, backward=False):
        B = self.batch_size
        if max_delta_ts is None:
            mdt = np.full(B, -np.inf if backward else np.inf)
        else:
            mdt = np.ascontiguousarray(np.array(max_delta_ts, dtype=np.float64))
        oc = np.zeros(B, dtype=np.int64)
        h = np.zeros(B)
        _lib().hy_oracle_step_impl(
            ctypes.byref(self._prog), B, _p(self.state), _p(self.pars), _p(self.time_hi), _p(self.time_lo), _p(mdt),
            _p(self.tc) if wtc else None, _p(oc), _p(h), _p(self._scratch)
        )
        self.last_h = h.copy()
        self.step_res = [(int(oc[i]), float(h[i])) for i in range(B)]
        return self.step_res

    def propagate_until(self, t, max_delta_t=None, max_steps=0):
        B = self.batch_size
        tf = np.ascontiguousarray(np.broadcast_to(np.array(t, dtype=np.float64), (B,)).copy())
        md = np.full(B, np.inf) if max_delta_t is None else np.ascontiguousarray(np.broadcast_to(np.array(max_delta_t, dtype=np.float64), (B,)).copy())
        oc = np.zeros(B, dtype=np.int64)
        mn, mx = np.zeros(B), np.zeros(B)
        ns = np.zeros(B, dtype=np.int64)
        _lib().hy_oracle_propagate_until(
            ctypes.byref(self._prog), B, _p(self.state), _p(self.pars), _p(self.time_hi), _p(self.time_lo), _p(tf),
            _p(md), ctypes.c_int64(max_steps), _p(oc), _p(mn), _p(mx), _p(ns), _p(self._scratch)
        )
        self.prop_res = [(int(oc[i]), float(mn[i]), float(mx[i]), int(ns[i])) for i in range(B)]
        return self.prop_res

    def propagate_for(self, dt, **kw):
        B = self.batch_size
        dts = np.broadcast_to(np.array(dt, dtype=np.float64), (B,))
        tf = np.array([dfloat_add(self.time_hi[i], self.time_lo[i], dts[i], 0.0)[0] for i in range(B)])
        # NOTE: the reference keeps the final times in double-length; the single-length
        # truncation only matters when time_lo != 0 and is irrelevant for the golden vectors.
        return self.propagate_until(tf, **kw)


def dfloat_add(ahi, alo, bhi, blo):
    rh, rl = ctypes.c_double(), ctypes.c_double()
    _lib().hy_oracle_dfloat_add(
        ctypes.c_double(ahi), ctypes.c_double(alo), ctypes.c_double(bhi), ctypes.c_double(blo), ctypes.byref(rh),
        ctypes.byref(rl)
    )
    return rh.value, rl.value


def ensemble_propagate_until(sys, state, n_systems, batch_size, t_final, tol=None, high_accuracy=False, pars=None,
                             max_steps=0, n_threads=0):
    """Propagate n_systems independent ICs (state[row * n_systems + sys]) as n_systems / batch_size
    lock-step batches over host threads. Returns (state, time_hi, time_lo, outcome, min_h, max_h,
    n_steps, total_steps)."""
    tmpl = OracleIntegrator(sys, np.zeros(len(sys) * batch_size), batch_size, tol=tol, high_accuracy=high_accuracy)
    N = int(n_systems)
    assert N % batch_size == 0
    st = np.ascontiguousarray(np.array(state, dtype=np.float64).reshape(-1)).copy()
    pr = np.zeros(max(tmpl.n_par, 1) * N) if pars is None else np.ascontiguousarray(np.array(pars, dtype=np.float64).reshape(-1))
    thi, tlo = np.zeros(N), np.zeros(N)
    oc = np.zeros(N, dtype=np.int64)
    mn, mx = np.zeros(N), np.zeros(N)
    ns = np.zeros(N, dtype=np.int64)
    total = _lib().hy_oracle_ensemble_propagate_until(
        ctypes.byref(tmpl._prog), ctypes.c_int64(N), int(batch_size), _p(st), _p(pr), _p(thi), _p(tlo),
        ctypes.c_double(t_final), ctypes.c_int64(max_steps), _p(oc), _p(mn), _p(mx), _p(ns), int(n_threads)
    )
    return st, thi, tlo, oc, mn, mx, ns, int(total)


def max_threads():
    return int(_lib().hy_oracle_max_threads())


# ----------------------------------------------------------------------------------------------
# Event detection (reference: src/detail/event_detection.cpp, src/taylor_adaptive_batch.cpp:727-1030,
# include/heyoka/events.hpp). Test infrastructure, like the rest of this module.
# ----------------------------------------------------------------------------------------------
DIR_ANY, DIR_POSITIVE, DIR_NEGATIVE = 0, 1, -1


class nt_event:
    """Non-terminal event: callback(ta, time, d_sgn, batch_idx)."""

    def __init__(self, eq, callback, direction=DIR_ANY):
        self.eq, self.callback, self.direction = as_ex(eq), callback, int(direction)


class t_event:
    """Terminal event: callback(ta, d_sgn, batch_idx) -> bool (None: always stop); cooldown < 0 = automatic."""

    def __init__(self, eq, callback=None, direction=DIR_ANY, cooldown=-1.0):
        self.eq, self.callback, self.direction, self.cooldown = as_ex(eq), callback, int(direction), float(cooldown)


def _poly_rescale(a, scal):
    out, cur = [], 1.0
    for c in a:
        out.append(c * cur)
        cur *= scal
    return out


def _poly_rescale_p2(a):
    n = len(a) - 1
    out, cur = [0.0] * (n + 1), 1.0
    for i in range(n + 1):
        out[n - i] = cur * a[n - i]
        cur *= 2.0
    return out


def _poly_translate_1(a):
    n = len(a) - 1
    out = [0.0] * (n + 1)
    for i in range(n + 1):
        for j in range(i + 1):
            out[j] += float(math.comb(i, j)) * a[i]
    return out


def _sgn(x):
    return (0.0 < x) - (x < 0.0)


def _count_sign_changes(a):
    n_sc, last = 0, 0
    for i, c in enumerate(a):
        if i == 0:
            last = _sgn(c)
            continue
        s = _sgn(c)
        if last != 0 and s + last == 0:
            n_sc += 1
        if s != 0:
            last = s
    return n_sc


def _poly_eval(a, x):
    r = a[-1]
    for c in reversed(a[:-1]):
        r = c + r * x
    return r


def _poly_eval_1(a, x):
    n = len(a) - 1
    r = a[n] * n
    for i in range(1, n):
        r = a[n - i] * (n - i) + r * x
    return r


def _fex_check(a, h):
    """Interval Horner enclosure of the polynomial over [0, h]: True if it excludes zero."""
    lo_h, hi_h = (h, 0.0) if h < 0 else (0.0, h)
    lo = hi = a[-1]
    for c in reversed(a[:-1]):
        ps = (lo * lo_h, lo * hi_h, hi * lo_h, hi * hi_h)
        lo, hi = min(ps) + c, max(ps) + c
    return _sgn(lo) == _sgn(hi) and _sgn(lo) != 0


_T748_EPS = 2.0 ** -52
_T748_MAX = 1.7976931348623157e308
_T748_MIN_DIFF = 2.2250738585072014e-308 * 32


def _t748_safe_div(num, den, r):
    """num / den, or r if the quotient would overflow."""
    if abs(den) < 1:
        if abs(den * _T748_MAX) <= abs(num):
            return r
    return num / den


def _t748_secant(a, b, fa, fb):
    tol = _T748_EPS * 5
    c = a - (fa / (fb - fa)) * (b - a)
    if c <= a + abs(a) * tol or c >= b - abs(b) * tol:
        return (a + b) / 2
    return c


def _t748_quadratic(a, b, d, fa, fb, fd, count):
    """`count` Newton steps on the quadratic through (a, fa), (b, fb), (d, fd)."""
    B = _t748_safe_div(fb - fa, b - a, _T748_MAX)
    A = _t748_safe_div(fd - fb, d - b, _T748_MAX)
    A = _t748_safe_div(A - B, d - a, 0.0)
    if A == 0:
        return _t748_secant(a, b, fa, fb)
    c = a if _sgn(A) * _sgn(fa) > 0 else b
    for _ in range(count):
        c -= _t748_safe_div(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a)
    if c <= a or c >= b:
        c = _t748_secant(a, b, fa, fb)
    return c


def _t748_cubic(a, b, d, e, fa, fb, fd, fe):
    """Inverse cubic interpolation through four points with distinct function values."""
    q11 = (d - e) * fd / (fe - fd)
    q21 = (b - d) * fb / (fd - fb)
    q31 = (a - b) * fa / (fb - fa)
    d21 = (b - d) * fd / (fd - fb)
    d31 = (a - b) * fb / (fb - fa)
    q22 = (d21 - q11) * fb / (fe - fb)
    q32 = (d31 - q21) * fa / (fd - fa)
    d32 = (d31 - q21) * fd / (fd - fa)
    q33 = (d32 - q22) * fa / (fe - fa)
    c = q31 + q32 + q33 + a
    if c <= a or c >= b:
        c = _t748_quadratic(a, b, d, fa, fb, fd, 3)
    return c


def _toms748(f, a, b, fa, fb, count):
    """Algorithm 748 (Alefeld, Potra, Shi: "Algorithm 748: enclosing zeros of continuous functions", ACM TOMS 21(3),
    1995, algorithm 4.2) as the reference uses it through Boost.Math's toms748_solve() with eps_tolerance<double>()
    (src/detail/event_detection.cpp:361-363; Boost is a third-party dependency which is not part of the reference tree,
    version not pinned by the reference beyond ">= 1.69": restated from the published algorithm). `count` = budget of
    function evaluations. Returns the final bracket and the budget left."""

    def tol(x, y):
        return abs(x - y) <= 4 * _T748_EPS * min(abs(x), abs(y))

    st = dict(a=a, b=b, fa=fa, fb=fb, d=0.0, fd=0.0)

    def bracket(c):
        a, b, fa, fb = st["a"], st["b"], st["fa"], st["fb"]
        t = _T748_EPS * 2
        if (b - a) < 2 * t * a:
            c = a + (b - a) / 2
        elif c <= a + abs(a) * t:
            c = a + abs(a) * t
        elif c >= b - abs(b) * t:
            c = b - abs(b) * t
        fc = f(c)
        if fc == 0:
            st.update(a=c, fa=0.0, d=0.0, fd=0.0)
            return
        if _sgn(fa) * _sgn(fc) < 0:
            st.update(d=b, fd=fb, b=c, fb=fc)
        else:
            st.update(d=a, fd=fa, a=c, fa=fc)

    def distinct():
        v = (st["fa"], st["fb"], st["fd"], fe)
        return all(abs(v[i] - v[j]) >= _T748_MIN_DIFF for i in range(4) for j in range(i + 1, 4))

    if tol(a, b) or fa == 0 or fb == 0:
        if fa == 0:
            b = a
        elif fb == 0:
            a = b
        return a, b, count
    e = fe = 1e5
    # A secant step, then a quadratic one.
    bracket(_t748_secant(a, b, fa, fb))
    count -= 1
    if count and st["fa"] != 0 and not tol(st["a"], st["b"]):
        c = _t748_quadratic(st["a"], st["b"], st["d"], st["fa"], st["fb"], st["fd"], 2)
        e, fe = st["d"], st["fd"]
        bracket(c)
        count -= 1
    while count and st["fa"] != 0 and not tol(st["a"], st["b"]):
        a0, b0 = st["a"], st["b"]
        if distinct():
            c = _t748_cubic(st["a"], st["b"], st["d"], e, st["fa"], st["fb"], st["fd"], fe)
        else:
            c = _t748_quadratic(st["a"], st["b"], st["d"], st["fa"], st["fb"], st["fd"], 2)
        e, fe = st["d"], st["fd"]
        bracket(c)
        count -= 1
        if count == 0 or st["fa"] == 0 or tol(st["a"], st["b"]):
            break
        if distinct():
            c = _t748_cubic(st["a"], st["b"], st["d"], e, st["fa"], st["fb"], st["fd"], fe)
        else:
            c = _t748_quadratic(st["a"], st["b"], st["d"], st["fa"], st["fb"], st["fd"], 3)
        bracket(c)
        count -= 1
        if count == 0 or st["fa"] == 0 or tol(st["a"], st["b"]):
            break
        # Double-length secant step from the endpoint with the smaller function value.
        a, b, fa, fb = st["a"], st["b"], st["fa"], st["fb"]
        u, fu = (a, fa) if abs(fa) < abs(fb) else (b, fb)
        c = u - 2 * (fu / (fb - fa)) * (b - a)
        if abs(c - u) > (b - a) / 2:
            c = a + (b - a) / 2
        e, fe = st["d"], st["fd"]
        bracket(c)
        count -= 1
        if count == 0 or st["fa"] == 0 or tol(st["a"], st["b"]):
            break
        # Bisection if the three steps did not halve the bracket.
        if (st["b"] - st["a"]) < 0.5 * (b0 - a0):
            continue
        e, fe = st["d"], st["fd"]
        bracket(st["a"] + (st["b"] - st["a"]) / 2)
        count -= 1
    a, b = st["a"], st["b"]
    if st["fa"] == 0:
        b = a
    elif st["fb"] == 0:
        a = b
    return a, b, count


def _bracketed_root(a, lb, ub):
    """Root of the polynomial in [lb, ub) (bracketed_root_find(), src/detail/event_detection.cpp:307-394): TOMS 748 with
    a tolerance of 4 eps and a budget of 53 function evaluations on [lb, prev(ub)]; the result is the midpoint of the
    final bracket; flag -1 = budget exhausted, flag 1 = no sign change at the ends (the reference gets a domain error
    out of Boost through errno): the caller ignores the event in both cases."""
    if math.isfinite(lb) and math.isfinite(ub) and ub > lb:
        ub = float(np.nextafter(ub, lb))
    flb, fub = _poly_eval(a, lb), _poly_eval(a, ub)
    if not (lb < ub) or _sgn(flb) * _sgn(fub) > 0 or not (math.isfinite(flb) and math.isfinite(fub)):
        return 0.0, 1
    ra, rb, left = _toms748(lambda x: _poly_eval(a, x), lb, ub, flb, fub, 53 - 2)
    return ra / 2 + rb / 2, (0 if left > 0 else -1)


def detect_events(poly, h, g_eps, events, is_terminal, cooldowns, order):
    """Per-lane detection for one class of events. poly[e] = Taylor coefficients of event e.
    Returns the list of (idx, root, d_sgn[, abs_der])."""
    out = []
    if not (math.isfinite(h) and math.isfinite(g_eps)) or h == 0:
        return out
    for i, ev in enumerate(events):
        ptr = [float(c) for c in poly[i]]
        if _fex_check(ptr, h):
            continue

        def add(root):
            if not math.isfinite(root):
                return
            if abs(root) >= abs(h):
                root = float(np.nextafter(h, 0.0))
            der = _poly_eval_1(ptr, root)
            if not math.isfinite(der):
                return
            d_sgn = _sgn(der)
            if ev.direction == DIR_ANY or d_sgn == ev.direction:
                out.append((i, root, d_sgn, abs(der)) if is_terminal else (i, root, d_sgn))

        lb_offset = 0.0
        if is_terminal and cooldowns[i] is not None:
            first, second = cooldowns[i]
            lb_offset = ((second - first) if h >= 0 else (second + first)) / abs(h)
        if lb_offset >= 1:
            continue
        wlist = [(0.0, 1.0, _poly_rescale(ptr, h))]
        isol = []
        failed = False
        while wlist:
            lb, ub, tmp = wlist.pop()
            if tmp[0] == 0 and all(math.isfinite(c) for c in tmp[1:]):
                if not (is_terminal and lb < lb_offset):
                    add(lb * h)
            n_sc = _count_sign_changes(_poly_translate_1(list(reversed(tmp))))
            if n_sc == 1:
                isol.append([lb, ub])
            elif n_sc > 1:
                tmp1 = _poly_rescale_p2(tmp)
                tmp2 = _poly_translate_1(tmp1)
                mid = lb / 2 + ub / 2
                if lb_offset < mid:
                    wlist.append((lb, mid, tmp1))
                wlist.append((mid, ub, tmp2))
            if len(wlist) > 250 or len(isol) > order:
                failed = True
                break
        if not isol or failed:
            continue
        tmp1 = _poly_rescale(ptr, h)
        for lb, ub in isol:
            if is_terminal and lb < lb_offset:
                lb = lb_offset
                if not (_poly_eval(tmp1, lb) * _poly_eval(tmp1, ub) < 0):
                    continue
            root, cflag = _bracketed_root(tmp1, lb, ub)
            if cflag == 0:
                add(root * h)
    return out


class OracleEventIntegrator(OracleIntegrator):
    """OracleIntegrator + event detection: step(), propagate_until/for() with the reference's semantics
    (lock-step batch, outcomes: >= 0 continuing terminal event, (success, 0) stopping terminal event)."""

    def __init__(self, sys, state, batch_size, t_events=(), nt_events=(), **kw):
        super().__init__(sys, state, batch_size, **kw)
        self.t_events, self.nt_events = list(t_events), list(nt_events)
        evs = [e.eq for e in self.t_events] + [e.eq for e in self.nt_events]
        dc, ev_u = taylor_decompose_sys(sys, evs)
        assert dc_to_strings(dc)[: self.n_eq] == dc_to_strings(self.dc)[: self.n_eq]
        # Rebuild the program on the decomposition that includes the event equations.
        self._rebuild(dc)
        self._ev_u = np.ascontiguousarray(np.array(ev_u, dtype=np.int32))
        B = self.batch_size
        self._ev_tc = np.zeros(max(len(evs), 1) * (self.order + 1) * B)
        self._mas = np.zeros(B)
        self._scratch = np.zeros(_lib().hy_oracle_scratch_size_e(ctypes.byref(self._prog), B) + 64)
        self.te_cooldowns = [[None] * len(self.t_events) for _ in range(B)]
        _lib().hy_oracle_scratch_size_e.restype = ctypes.c_size_t

    def _rebuild(self, dc):
        st, pars, thi = self.state.copy(), self.pars.copy(), self.time_hi.copy()
        sysv = self.sys
        self.dc = dc
        self.n_u = len(dc) - self.n_eq
        OracleIntegrator._build_program(self)
        self.state, self.time_hi = st, thi
        if pars.size == self.pars.size:
            self.pars = pars
        self.sys = sysv

    def _dense(self, i, h):
        """Dense output of lane i at h from the Taylor coefficients."""
        B, p = self.batch_size, self.order
        tc = self.tc.reshape(self.n_eq, p + 1, B)[:, :, i]
        out = np.empty(self.n_eq)
        for v in range(self.n_eq):
            if self.high_accuracy:
                res, comp, cur_h = tc[v, 0], 0.0, h
                for k in range(1, p + 1):
                    tmp = tc[v, k] * cur_h
                    y = tmp - comp
                    t = res + y
                    comp = (t - res) - y
                    res = t
                    cur_h = cur_h * h
            else:
                res = tc[v, p]
                for k in range(1, p + 1):
                    res = tc[v, p - k] + res * h
            out[v] = res
        return out

    def step(self, max_delta_ts=None, wtc=False, backward=False):
        B, p = self.batch_size, self.order
        if max_delta_ts is None:
            mdt = np.full(B, -np.inf if backward else np.inf)
        else:
            mdt = np.ascontiguousarray(np.array(max_delta_ts, dtype=np.float64))
        h = mdt.copy()
        n_te, n_ev = len(self.t_events), len(self.t_events) + len(self.nt_events)
        _lib().hy_oracle_step_e(
            ctypes.byref(self._prog), B, _p(self.state), _p(self.pars), _p(self.time_hi), _p(h), _p(self.tc),
            _p(self._ev_u), n_ev, _p(self._ev_tc), _p(self._mas), _p(self._scratch))
        ev_tc = self._ev_tc.reshape(max(n_ev, 1), p + 1, B)
        eps = np.finfo(np.float64).eps
        res = []
        st = self.state.reshape(self.n_eq, B)
        for i in range(B):
            mas = self._mas[i]
            if math.isfinite(mas):
                max_r = self.tol if mas < 1 else self.tol * mas
                g_eps = eps * mas if max_r < eps * mas else max_r
            else:
                g_eps = math.inf
            hi = float(h[i])
            d_tes = detect_events(ev_tc[:n_te, :, i], hi, g_eps, self.t_events, True, self.te_cooldowns[i], p)
            d_ntes = detect_events(ev_tc[n_te:n_ev, :, i], hi, g_eps, self.nt_events, False, None, p)
            d_tes.sort(key=lambda e: abs(e[1]))
            d_ntes.sort(key=lambda e: abs(e[1]))
            if d_tes:
                hi = d_tes[0][1]
            st[:, i] = self._dense(i, hi)
            nt = dfloat_add(self.time_hi[i], self.time_lo[i], hi, 0.0)
            self.time_hi[i], self.time_lo[i] = nt
            self.last_h[i] = hi
            if not (math.isfinite(nt[0]) and math.isfinite(nt[1]) and np.all(np.isfinite(st[:, i]))):
                res.append((OC_ERR_NF_STATE, hi))
                continue
            for k, cd in enumerate(self.te_cooldowns[i]):
                if cd is not None:
                    tmp = cd[0] + hi
                    self.te_cooldowns[i][k] = None if abs(tmp) >= cd[1] else (tmp, cd[1])
            for ev in d_ntes:
                if d_tes and not (abs(ev[1]) < abs(hi)):
                    break
                # new_time - last_h + root, in double-length arithmetic.
                t0 = dfloat_add(nt[0], nt[1], -hi, 0.0)
                t_ev = dfloat_add(t0[0], t0[1], ev[1], 0.0)[0]
                self.nt_events[ev[0]].callback(self, t_ev, ev[2], i)
            te_cb_ret = False
            if d_tes:
                idx = d_tes[0][0]
                te = self.t_events[idx]
                cd = te.cooldown if te.cooldown >= 0 else (g_eps / d_tes[0][3] * 10 if math.isfinite(g_eps / d_tes[0][3] * 10) else 0.0)
                self.te_cooldowns[i][idx] = (0.0, cd)
                if te.callback is not None:
                    te_cb_ret = bool(te.callback(self, d_tes[0][2], i))
                res.append((idx if te_cb_ret else (-idx - 1), hi))
            else:
                res.append((OC_TIME_LIMIT if hi == mdt[i] else OC_SUCCESS, hi))
        self.step_res = res
        return res

    def propagate_until(self, t, max_delta_t=None, max_steps=0):
        B = self.batch_size
        tf = np.broadcast_to(np.array(t, dtype=np.float64), (B,)).copy()
        md = np.full(B, np.inf) if max_delta_t is None else np.broadcast_to(np.array(max_delta_t, dtype=np.float64), (B,)).copy()

        def rem_of(i):
            return dfloat_add(tf[i], 0.0, -self.time_hi[i], -self.time_lo[i])

        rem = [rem_of(i) for i in range(B)]
        t_dir = [r[0] > 0 or (r[0] == 0 and r[1] >= 0) for r in rem]
        ts_count, mn, mx = [0] * B, [math.inf] * B, [0.0] * B
        it = 0
        while True:
            lims = np.empty(B)
            for i in range(B):
                r = rem[i]
                if t_dir[i]:
                    lims[i] = r[0] if (r[0] < md[i] or (r[0] == md[i] and r[1] < 0)) else md[i]
                else:
                    lims[i] = r[0] if (-md[i] < r[0] or (-md[i] == r[0] and 0 < r[1])) else -md[i]
            res = self.step(lims)
            n_done, nfs, ste = 0, False, False
            pr = []
            for i, (oc, h) in enumerate(res):
                if oc == OC_ERR_NF_STATE:
                    nfs = True
                else:
                    ts_count[i] += int(h != 0)
                    if oc == OC_SUCCESS:
                        mn[i], mx[i] = min(mn[i], abs(h)), max(mx[i], abs(h))
                    ste = ste or (OC_SUCCESS < oc < 0)
                    if h == rem[i][0]:
                        n_done += 1
                        rem[i] = (0.0, 0.0)
                    else:
                        rem[i] = rem_of(i)
                pr.append((oc, mn[i], mx[i], ts_count[i]))
            self.prop_res = pr
            if nfs:
                return pr
            it += 1
            if n_done == B or ste:
                return pr
            if it == max_steps:
                self.prop_res = [(OC_STEP_LIMIT,) + r[1:] for r in pr]
                return self.prop_res

    def propagate_for(self, dt, **kw):
        B = self.batch_size
        dts = np.broadcast_to(np.array(dt, dtype=np.float64), (B,))
        tf = np.array([dfloat_add(self.time_hi[i], self.time_lo[i], dts[i], 0.0)[0] for i in range(B)])
        return self.propagate_until(tf, **kw)
