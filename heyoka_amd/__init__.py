"""heyoka_amd: MI355X-native batch Taylor integrator with heyoka's interface.

Python host layer over the C ABI (include/heyoka_amd.h). Names, argument meaning and error
behaviour mirror the reference's C++ interface for this path (bluescarni/heyoka v7.12.0:
include/heyoka/expression.hpp, include/heyoka/taylor.hpp:781-1121,
include/heyoka/ensemble_propagate.hpp:222-271, include/heyoka/model/nbody.hpp) in the shape of its
Python bindings (properties `state`, `time`, `step_res`, ...), so that parity tests read like the
reference's own tests. All numerical work happens in generated HIP kernels on the GPU: there is no
CPU fallback in this package.
"""

import ctypes
import enum
import weakref

import numpy as np

from . import _lib
from ._lib import lib, raise_for, check_handle, take_str

__all__ = [
    "cfunc",
    "continuous_output_batch",
    "expression",
    "make_vars",
    "par",
    "time",
    "sin",
    "cos",
    "exp",
    "log",
    "sqrt",
    "pi",
    "pi_constant",
    "tan",
    "tanh",
    "sinh",
    "cosh",
    "asin",
    "acos",
    "atan",
    "asinh",
    "acosh",
    "atanh",
    "erf",
    "sigmoid",
    "pow",
    "sum",
    "prod",
    "model",
    "taylor_outcome",
    "taylor_adaptive_batch",
    "event_direction",
    "nt_event",
    "t_event",
    "taylor_decompose_sys",
    "ensemble_propagate_until_batch",
    "ensemble_propagate_for_batch",
    "device_count",
    "version",
    "native_event_counter",
]

_builtin_sum = sum
_builtin_pow = pow


def version():
    return take_str(lib.hy_version())


_LOG_LEVELS = {"trace": 0, "debug": 1, "info": 2, "warn": 3, "err": 4, "critical": 5, "off": 6}
_log_keep = []


def set_logger_level(level):
    """include/heyoka/logging.hpp:19-24 (set_logger_level_trace() ... set_logger_level_critical()): a name or 0 ... 6."""
    raise_for(lib.hy_set_logger_level(int(_LOG_LEVELS.get(level, level))))


def set_logger_level_trace():
    set_logger_level("trace")


def set_logger_level_debug():
    set_logger_level("debug")


def set_logger_level_info():
    set_logger_level("info")


def set_logger_level_warn():
    set_logger_level("warn")


def set_logger_level_err():
    set_logger_level("err")


def set_logger_level_critical():
    set_logger_level("critical")


def set_log_callback(fn):
    """fn(level, message) receives the log messages instead of stderr (None restores stderr)."""
    if fn is None:
        lib.hy_set_log_callback(None, None)
        del _log_keep[:]
        return
    cb = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)(lambda lvl, msg, _u: fn(int(lvl), msg.decode()))
    _log_keep.append(cb)
    lib.hy_set_log_callback(ctypes.cast(cb, ctypes.c_void_p), None)


def build_id():
    """Build id of the loaded library (checked against the sources of this tree at import, heyoka_amd/_lib.py)."""
    return lib.hy_build_id().decode()


def device_count():
    return int(lib.hy_device_count())


class taylor_outcome(enum.IntEnum):
    """include/heyoka/taylor.hpp:142-155."""

    success = -4294967296 - 1
    step_limit = -4294967296 - 2
    time_limit = -4294967296 - 3
    err_nf_state = -4294967296 - 4
    cb_stop = -4294967296 - 5


def _outcome(v):
    try:
        return taylor_outcome(int(v))
    except ValueError:
        return int(v)


class expression:
    """Symbolic expression (include/heyoka/expression.hpp:73-118)."""

    __slots__ = ("_h",)

    def __init__(self, x=0.0, _handle=None):
        if _handle is not None:
            self._h = _handle
        elif isinstance(x, expression):
            raise TypeError("copy-construct expressions by assignment")
        elif isinstance(x, str):
            self._h = check_handle(lib.hy_expr_var(x.encode()))
        else:
            self._h = check_handle(lib.hy_expr_num(float(x)))

    def __del__(self, _free=lib.hy_expr_free):
        # NOTE: the free function is bound at definition time (module globals may be gone at shutdown).
        h = getattr(self, "_h", None)
        if h:
            _free(h)
            self._h = None

    @staticmethod
    def _wrap(h):
        return expression(_handle=check_handle(h))

    def __repr__(self):
        return take_str(lib.hy_expr_str(self._h))

    def __neg__(self):
        return expression._wrap(lib.hy_expr_neg(self._h))

    def __pos__(self):
        return self

    def __add__(self, o):
        return _bin(lib.hy_expr_add, self, o)

    def __radd__(self, o):
        return _bin(lib.hy_expr_add, o, self)

    def __sub__(self, o):
        return _bin(lib.hy_expr_sub, self, o)

    def __rsub__(self, o):
        return _bin(lib.hy_expr_sub, o, self)

    def __mul__(self, o):
        return _bin(lib.hy_expr_mul, self, o)

    def __rmul__(self, o):
        return _bin(lib.hy_expr_mul, o, self)

    def __truediv__(self, o):
        return _bin(lib.hy_expr_div, self, o)

    def __rtruediv__(self, o):
        return _bin(lib.hy_expr_div, o, self)

    def __pow__(self, o):
        return _bin(lib.hy_expr_pow, self, o)


def _as_ex(x):
    return x if isinstance(x, expression) else expression(x)


def _bin(fn, a, b):
    # NOTE: keep the (possibly temporary) operands alive across the C call.
    a, b = _as_ex(a), _as_ex(b)
    return expression._wrap(fn(a._h, b._h))


def _un(fn, a):
    a = _as_ex(a)
    return expression._wrap(fn(a._h))


def make_vars(*names):
    """make_vars("x", "v") (include/heyoka/expression.hpp)."""
    vs = [expression(n) for n in names]
    return vs[0] if len(vs) == 1 else vs


class _Par:
    """par[i] (include/heyoka/param.hpp)."""

    def __getitem__(self, i):
        return expression._wrap(lib.hy_expr_par(int(i)))


par = _Par()
time = expression._wrap(lib.hy_expr_time())


def sin(e):
    return _un(lib.hy_expr_sin, e)


def cos(e):
    return _un(lib.hy_expr_cos, e)


def exp(e):
    return _un(lib.hy_expr_exp, e)


def log(e):
    return _un(lib.hy_expr_log, e)


def sqrt(e):
    return _un(lib.hy_expr_sqrt, e)


def tan(e):
    return _un(lib.hy_expr_tan, e)


def tanh(e):
    return _un(lib.hy_expr_tanh, e)


def sinh(e):
    return _un(lib.hy_expr_sinh, e)


def cosh(e):
    return _un(lib.hy_expr_cosh, e)


def asin(e):
    return _un(lib.hy_expr_asin, e)


def acos(e):
    return _un(lib.hy_expr_acos, e)


def atan(e):
    return _un(lib.hy_expr_atan, e)


def asinh(e):
    return _un(lib.hy_expr_asinh, e)


def acosh(e):
    return _un(lib.hy_expr_acosh, e)


def atanh(e):
    return _un(lib.hy_expr_atanh, e)


def erf(e):
    return _un(lib.hy_expr_erf, e)


def sigmoid(e):
    return _un(lib.hy_expr_sigmoid, e)


def pow(b, e):  # noqa: A001 - mirrors heyoka::pow
    return _bin(lib.hy_expr_pow, b, e)


def atan2(y, x):
    """atan2(y, x) (src/math/atan2.cpp:763-786)."""
    return _bin(lib.hy_expr_atan2, _as_ex(y), x)


def kepE(e, M):
    """Eccentric anomaly E(e, M), E - e sin E = M (src/math/kepE.cpp:801-809)."""
    return _bin(lib.hy_expr_kepE, _as_ex(e), M)


def kepF(h, k, lam):
    """Eccentric longitude F(h, k, lam), F + h cos F - k sin F = lam (src/math/kepF.cpp:1689-1699). Defined through the
    registry of node rules alone (csrc/builtin_rules.cpp)."""
    h, k, lam = _as_ex(h), _as_ex(k), _as_ex(lam)
    return expression._wrap(lib.hy_expr_kepF(h._h, k._h, lam._h))


def kepDE(s0, c0, DM):
    """Difference of eccentric anomalies DE(s0, c0, DM), DE - c0 sin DE + s0 (1 - cos DE) = DM
    (src/math/kepDE.cpp:113-123). Defined through the registry of node rules alone (csrc/builtin_rules.cpp)."""
    s0, c0, DM = _as_ex(s0), _as_ex(c0), _as_ex(DM)
    return expression._wrap(lib.hy_expr_kepDE(s0._h, c0._h, DM._h))


def pi_constant():
    """The constant pi as the reference has it (heyoka::pi, include/heyoka/math/constants.hpp:117): a function without
    arguments which occupies its own u variable (order 0: the value, 0 beyond, src/math/constants.cpp:258-273). A registered
    node rule (csrc/builtin_rules.cpp). The module attribute `pi` is this expression."""
    return expression._wrap(lib.hy_expr_pi())


pi = pi_constant()


def custom_func(name, *args):
    """f(args) of a function registered with register_node_rule(); not_implemented_error for an unknown name (reference:
    func.hpp:266-267)."""
    exs = [_as_ex(a) for a in args]
    arr = (ctypes.c_void_p * len(exs))(*[e._h for e in exs]) if exs else None
    return expression._wrap(lib.hy_expr_custom(name.encode(), arr, len(exs)))


class _node_rule_desc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("n_args", ctypes.c_uint32), ("n_hidden", ctypes.c_uint32),
                ("decompose", ctypes.c_void_p), ("ctx", ctypes.c_void_p), ("hidden_deps", ctypes.c_void_p),
                ("deps", ctypes.c_void_p), ("n_deps", ctypes.c_uint32), ("hip_source", ctypes.c_char_p)]


_DECOMPOSE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p))
_node_rule_keepalive = []


def register_node_rule(name, n_args, hip_source, hidden=None, hidden_deps=(), deps=()):
    """Register an elementary function with its Taylor rule - the counterpart of deriving from func_base in the reference
    (include/heyoka/func.hpp:94-96, :117-147); see csrc/node_rule.hpp for the contract.

    hidden(self, args, hidden_vars) -> list of expressions: the definitions of the hidden u variables appended behind the
    node (each ONE elementary function of self, args, hidden_vars[j] with j below its own index); hidden_deps[j]: the
    hidden dependencies of definition j (indices into that list); deps: those of the node itself, in the order
    hy_rule_<name>_orderk() reads them. hip_source defines hy_rule_<name>_order0() and hy_rule_<name>_orderk()."""
    n_hidden = len(hidden_deps) if hidden is not None else 0
    if hidden is not None and n_hidden == 0:
        raise ValueError("register_node_rule(): hidden_deps must list one (possibly empty) entry per hidden definition")
    cb = None
    if hidden is not None:
        def tramp(_ctx, self_h, args_p, n, hid_p, out_p):
            try:
                res = hidden(_borrow(self_h), [_borrow(args_p[i]) for i in range(n)], [_borrow(hid_p[j]) for j in range(n_hidden)])
                if len(res) != n_hidden:
                    return 1
                for j, e in enumerate(res):
                    e = _as_ex(e)
                    # The library takes ownership of what it finds in hidden_out: hand over a copy.
                    out_p[j] = _clone_handle(e)
                return 0
            except Exception:  # noqa: BLE001 - reported as a failed decomposition by the library
                return 1

        cb = _DECOMPOSE_FN(tramp)
        _node_rule_keepalive.append(cb)
    hd = (ctypes.c_int32 * (4 * max(n_hidden, 1)))(*([-1] * (4 * max(n_hidden, 1))))
    for j, lst in enumerate(hidden_deps):
        if len(lst) > 4:
            raise ValueError("register_node_rule(): a hidden definition can depend on at most 4 other hidden variables")
        for q, x in enumerate(lst):
            if not 0 <= int(x) < n_hidden:
                raise ValueError("register_node_rule(): an entry of hidden_deps is out of range")
            hd[4 * j + q] = int(x)
    dp = (ctypes.c_uint32 * max(len(deps), 1))(*[int(x) for x in deps]) if deps else None
    d = _node_rule_desc(name.encode(), int(n_args), n_hidden, ctypes.cast(cb, ctypes.c_void_p) if cb else None, None,
                        ctypes.cast(hd, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p) if dp else None, len(deps),
                        hip_source.encode())
    raise_for(lib.hy_node_rule_register(ctypes.byref(d)))


def _borrow(h):
    """A Python expression which shares the value of a handle owned by the library (an independent copy: the handle
    itself is not adopted)."""
    return expression(_handle=_clone_raw(h))


def _clone_raw(h):
    # x + 0 would fold; the product API has no copy entry point: sum([x]) of one term returns the term itself.
    arr = (ctypes.c_void_p * 1)(h)
    return check_handle(lib.hy_expr_sum(arr, 1))


def _clone_handle(e):
    return _clone_raw(e._h)


def relu(x, slope=0.0):
    """relu(x, slope) = x > 0 ? x : slope * x (src/math/relu.cpp:580-590)."""
    x = _as_ex(x)
    return expression._wrap(check_handle(lib.hy_expr_relu(x._h, float(slope))))


def relup(x, slope=0.0):
    """Derivative of relu (src/math/relu.cpp:592-602)."""
    x = _as_ex(x)
    return expression._wrap(check_handle(lib.hy_expr_relup(x._h, float(slope))))


def leaky_relu(slope):
    return lambda x: relu(x, slope)


def leaky_relup(slope):
    return lambda x: relup(x, slope)


def select(cond, t, f):
    """select(c, t, f) = c != 0 ? t : f (src/math/select.cpp:267-270)."""
    c, t, f = _as_ex(cond), _as_ex(t), _as_ex(f)
    return expression._wrap(lib.hy_expr_select(c._h, t._h, f._h))


def _logical(is_and, args):
    exs = [_as_ex(a) for a in args]
    arr = (ctypes.c_void_p * len(exs))(*[e._h for e in exs]) if exs else None
    return expression._wrap(lib.hy_expr_logical(is_and, arr, len(exs)))


def logical_and(args):
    return _logical(1, args)


def logical_or(args):
    return _logical(0, args)


def _rel(op):
    def f(a, b):
        a, b = _as_ex(a), _as_ex(b)
        return expression._wrap(lib.hy_expr_rel(op, a._h, b._h))

    return f


eq, neq, lt, gt, lte, gte = (_rel(i) for i in range(6))


def _handle_array(exs):
    exs = [_as_ex(e) for e in exs]
    arr = (ctypes.c_void_p * max(len(exs), 1))(*[e._h for e in exs])
    return exs, arr


def sum(args):  # noqa: A001 - mirrors heyoka::sum
    exs, arr = _handle_array(args)
    return expression._wrap(lib.hy_expr_sum(arr, len(exs)))


def prod(args):
    exs, arr = _handle_array(args)
    return expression._wrap(lib.hy_expr_prod(arr, len(exs)))


class _Sys:
    """Owned hy_sys handle built from a list of (lhs, rhs) pairs."""

    def __init__(self, sys=None, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        self._h = check_handle(lib.hy_sys_new())
        for lhs, rhs in sys:
            lhs, rhs = _as_ex(lhs), _as_ex(rhs)  # keep temporaries alive across the C call
            raise_for(lib.hy_sys_add(self._h, lhs._h, rhs._h))

    def __del__(self, _free=lib.hy_sys_free):
        h = getattr(self, "_h", None)
        if h:
            _free(h)
            self._h = None

    def __len__(self):
        return int(lib.hy_sys_size(self._h))

    @property
    def vars(self):
        """The state variables (lhs of each equation), in order."""
        n = len(self)
        arr = (ctypes.c_void_p * n)()
        raise_for(lib.hy_sys_get_vars(self._h, arr))
        return [expression(_handle=check_handle(arr[i])) for i in range(n)]


def _to_sys(sys):
    return sys if isinstance(sys, _Sys) else _Sys(sys)


class model:
    """heyoka::model (include/heyoka/model/nbody.hpp:73-78, model/pendulum.hpp)."""

    @staticmethod
    def _nbody_args(masses, Gconst):
        """Masses / G as expressions (numbers or par[i]); returns (array, n, G handle, keep-alive list)."""
        keep = []
        if masses is None:
            arr, n = None, 0
        else:
            ms = [_as_ex(m) for m in masses]
            keep.extend(ms)
            arr = (ctypes.c_void_p * len(ms))(*[m._h for m in ms])
            n = len(ms)
        g = _as_ex(Gconst)
        keep.append(g)
        return arr, n, g._h, keep

    @staticmethod
    def nbody(n, masses=None, Gconst=1.0):
        numeric = not isinstance(Gconst, expression) and (
            masses is None or not any(isinstance(m, expression) for m in masses))
        if numeric:
            if masses is None:
                h = lib.hy_model_nbody(int(n), None, 0, float(Gconst))
            else:
                m = np.ascontiguousarray(np.asarray(masses, dtype=np.float64))
                h = lib.hy_model_nbody(int(n), m.ctypes.data, m.size, float(Gconst))
        else:
            arr, nm, g, _keep = model._nbody_args(masses, Gconst)
            h = lib.hy_model_nbody_ex(int(n), arr, nm, g)
        return _Sys(_handle=check_handle(h))

    @staticmethod
    def nbody_energy(n, masses=None, Gconst=1.0):
        arr, nm, g, _keep = model._nbody_args(masses, Gconst)
        return expression(_handle=check_handle(lib.hy_model_nbody_energy(int(n), arr, nm, g)))

    @staticmethod
    def nbody_potential(n, masses=None, Gconst=1.0):
        arr, nm, g, _keep = model._nbody_args(masses, Gconst)
        return expression(_handle=check_handle(lib.hy_model_nbody_potential(int(n), arr, nm, g)))

    @staticmethod
    def pendulum(gconst=1.0, length=1.0):
        return _Sys(_handle=check_handle(lib.hy_model_pendulum(float(gconst), float(length))))

    @staticmethod
    def pendulum_energy(gconst=1.0, length=1.0):
        return expression(_handle=check_handle(lib.hy_model_pendulum_energy(float(gconst), float(length))))


    # ---- the other point-mass models ----
    @staticmethod
    def _ex_array(values):
        """Sequence of numbers / expressions -> (ctypes array of handles, n, keep-alive list)."""
        exs = [_as_ex(v) for v in (values if values is not None else [])]
        arr = (ctypes.c_void_p * len(exs))(*[e._h for e in exs]) if exs else None
        return arr, len(exs), exs

    @staticmethod
    def np1body(n, masses=None, Gconst=1.0):
        """model::np1body() (src/model/nbody.cpp:236-325): n bodies in the frame of body 0."""
        arr, nm, g, _keep = model._nbody_args(masses, Gconst)
        return _Sys(_handle=check_handle(lib.hy_model_np1body(int(n), arr, nm, g)))

    @staticmethod
    def np1body_energy(n, masses=None, Gconst=1.0):
        arr, nm, g, _keep = model._nbody_args(masses, Gconst)
        return expression(_handle=check_handle(lib.hy_model_np1body_energy(int(n), arr, nm, g)))

    @staticmethod
    def np1body_potential(n, masses=None, Gconst=1.0):
        arr, nm, g, _keep = model._nbody_args(masses, Gconst)
        return expression(_handle=check_handle(lib.hy_model_np1body_potential(int(n), arr, nm, g)))

    @staticmethod
    def cr3bp(mu=1e-3):
        """model::cr3bp() (src/model/cr3bp.cpp): state x, y, z, px, py, pz."""
        m = _as_ex(mu)
        return _Sys(_handle=check_handle(lib.hy_model_cr3bp(m._h)))

    @staticmethod
    def cr3bp_jacobi(mu=1e-3):
        m = _as_ex(mu)
        return expression(_handle=check_handle(lib.hy_model_cr3bp_jacobi(m._h)))

    @staticmethod
    def _fc_call(fn, Gconst, masses, positions, omega=None, with_omega=False, sys_=False):
        g = _as_ex(Gconst)
        ma, nm, k1 = model._ex_array(masses)
        pa, npos, k2 = model._ex_array(positions)
        args = [g._h, ma, nm, pa, npos]
        if with_omega:
            oa, no, k3 = model._ex_array(omega)
            args += [oa, no]
        h = check_handle(fn(*args))
        return _Sys(_handle=h) if sys_ else expression(_handle=h)

    @staticmethod
    def fixed_centres(Gconst=1.0, masses=(), positions=()):
        """model::fixed_centres() (src/model/fixed_centres.cpp): state x, y, z, vx, vy, vz."""
        return model._fc_call(lib.hy_model_fixed_centres, Gconst, masses, positions, sys_=True)

    @staticmethod
    def fixed_centres_energy(Gconst=1.0, masses=(), positions=()):
        return model._fc_call(lib.hy_model_fixed_centres_energy, Gconst, masses, positions)

    @staticmethod
    def fixed_centres_potential(Gconst=1.0, masses=(), positions=()):
        return model._fc_call(lib.hy_model_fixed_centres_potential, Gconst, masses, positions)

    @staticmethod
    def rotating(omega=()):
        """model::rotating() (src/model/rotating.cpp)."""
        oa, no, _k = model._ex_array(omega)
        return _Sys(_handle=check_handle(lib.hy_model_rotating(oa, no)))

    @staticmethod
    def rotating_energy(omega=()):
        oa, no, _k = model._ex_array(omega)
        return expression(_handle=check_handle(lib.hy_model_rotating_energy(oa, no)))

    @staticmethod
    def rotating_potential(omega=()):
        oa, no, _k = model._ex_array(omega)
        return expression(_handle=check_handle(lib.hy_model_rotating_potential(oa, no)))

    @staticmethod
    def mascon(Gconst=1.0, masses=(), positions=(), omega=()):
        """model::mascon() (src/model/mascon.cpp): fixed centres in a uniformly rotating frame."""
        return model._fc_call(lib.hy_model_mascon, Gconst, masses, positions, omega, with_omega=True, sys_=True)

    @staticmethod
    def mascon_energy(Gconst=1.0, masses=(), positions=(), omega=()):
        return model._fc_call(lib.hy_model_mascon_energy, Gconst, masses, positions, omega, with_omega=True)

    @staticmethod
    def mascon_potential(Gconst=1.0, masses=(), positions=(), omega=()):
        return model._fc_call(lib.hy_model_mascon_potential, Gconst, masses, positions, omega, with_omega=True)


def hiprtc_compile(source):
    """Compile a HIP source with hiprtc and the options of the steppers (works without a GPU); returns the code object."""
    n = ctypes.c_size_t(0)
    src = source.encode()
    raise_for(lib.hy_hiprtc_compile(src, None, ctypes.byref(n)))
    buf = ctypes.create_string_buffer(n.value)
    raise_for(lib.hy_hiprtc_compile(src, buf, ctypes.byref(n)))
    return buf.raw[: n.value]


def taylor_decompose_sys(sys):
    """taylor_decompose_sys() (src/taylor_01.cpp:848-1008) -> list of textual entries."""
    s = _to_sys(sys)
    txt = take_str(check_handle(lib.hy_sys_decomposition_str(s._h)))
    return txt.rstrip("\n").split("\n")


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class _DeviceArray:
    """Zero-copy view of one of the integrator's device arrays (for torch.as_tensor / cupy)."""

    def __init__(self, ptr, shape, owner, typestr="<f8"):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape),
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }

    @property
    def ptr(self):
        """Raw device address (int)."""
        return self.__cuda_array_interface__["data"][0]

    @property
    def shape(self):
        return self.__cuda_array_interface__["shape"]


class cfunc:
    """cfunc<double> (reference: include/heyoka/expression.hpp:735-970): compiled evaluation of ``fn(vars)``
    over many input columns - one HIP kernel, one lane per evaluation. Arrays are row-major
    ``inputs[var, eval]`` (the reference's 2D call operator), i.e. the integrator's SoA state layout.

    ``cf(inputs, pars=None, time=None)`` works on host arrays; ``cf.eval_device(out_ptr, in_ptr, ...)`` works
    on device pointers (e.g. ``ta.device_array("state").ptr``) without any host round trip."""

    def __init__(self, fn, vars, device=0, **_ignored):
        fs = [_as_ex(f) for f in fn]
        vs = [_as_ex(v) for v in vars]
        fa = (ctypes.c_void_p * len(fs))(*[f._h for f in fs])
        va = (ctypes.c_void_p * len(vs))(*[v._h for v in vs])
        self._h = check_handle(lib.hy_cfunc_new(fa, len(fs), va, len(vs), int(device)))

    def __del__(self, _free=lib.hy_cfunc_free):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _free(h)

    @property
    def nparams(self):
        return int(lib.hy_cfunc_get_nparams(self._h))

    @property
    def nvars(self):
        return int(lib.hy_cfunc_get_nvars(self._h))

    @property
    def nouts(self):
        return int(lib.hy_cfunc_get_nouts(self._h))

    @property
    def is_time_dependent(self):
        return bool(lib.hy_cfunc_is_time_dependent(self._h))

    @property
    def dc(self):
        return take_str(check_handle(lib.hy_cfunc_decomposition_str(self._h))).rstrip("\n").split("\n")

    @property
    def hip_source(self):
        return take_str(check_handle(lib.hy_cfunc_get_hip_source(self._h)))

    def set_stream(self, stream_ptr):
        raise_for(lib.hy_cfunc_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def __call__(self, inputs, pars=None, time=None):
        x = _f64(inputs)
        single = x.ndim == 1
        nev = 1 if single else x.shape[1]
        out = np.empty(self.nouts if single else (self.nouts, nev))
        p = None if pars is None else _f64(pars)
        t = None if time is None else _f64(time).reshape(-1)
        raise_for(lib.hy_cfunc_eval(self._h, out.ctypes.data, out.size, x.ctypes.data, x.size,
                                    None if p is None else p.ctypes.data, 0 if p is None else p.size,
                                    None if t is None else t.ctypes.data, 0 if t is None else t.size))
        return out

    def eval_device(self, d_out, d_in, nevals, d_pars=None, d_time=None):
        """All pointers are device addresses (ints); asynchronous on the cfunc's stream."""
        raise_for(lib.hy_cfunc_eval_device(self._h, ctypes.c_void_p(d_out), ctypes.c_void_p(d_in),
                                           ctypes.c_void_p(d_pars) if d_pars else None,
                                           ctypes.c_void_p(d_time) if d_time else None, int(nevals)))


class continuous_output_batch:
    """continuous_output_batch<double> (reference: include/heyoka/continuous_output.hpp:151-204): returned
    by propagate_for/until(c_output=True). Coefficients and times live in HBM; ``co(t)`` launches one kernel."""

    def __init__(self, handle):
        self._h = handle
        self._output = None

    def __del__(self, _free=lib.hy_cout_free):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _free(h)

    def __copy__(self):
        return continuous_output_batch(check_handle(lib.hy_cout_clone(self._h)))

    def __deepcopy__(self, memo):
        return self.__copy__()

    @property
    def batch_size(self):
        return int(lib.hy_cout_get_batch_size(self._h))

    @property
    def dim(self):
        return int(lib.hy_cout_get_dim(self._h))

    @property
    def order(self):
        return int(lib.hy_cout_get_order(self._h))

    @property
    def n_steps(self):
        n = ctypes.c_size_t()
        raise_for(lib.hy_cout_get_n_steps(self._h, ctypes.byref(n)))
        return int(n.value)

    @property
    def bounds(self):
        lb, ub = np.empty(self.batch_size), np.empty(self.batch_size)
        raise_for(lib.hy_cout_get_bounds(self._h, lb.ctypes.data, ub.ctypes.data))
        return lb, ub

    @property
    def times(self):
        """(n_steps + 2, batch_size): the last row is the +-inf padding, as in the reference."""
        hi = np.empty((self.n_steps + 2, self.batch_size))
        raise_for(lib.hy_cout_get_times(self._h, hi.ctypes.data, None))
        return hi

    @property
    def tcs(self):
        """(n_steps, dim, order + 1, batch_size)."""
        out = np.empty((self.n_steps, self.dim, self.order + 1, self.batch_size))
        raise_for(lib.hy_cout_get_tcs(self._h, out.ctypes.data))
        return out

    @property
    def output(self):
        return self._output

    def __call__(self, tm):
        """tm: scalar or array of batch_size target times -> (dim, batch_size)."""
        t = _f64(tm).reshape(-1)
        out = np.empty((self.dim, self.batch_size))
        raise_for(lib.hy_cout_eval(self._h, t.ctypes.data, t.size, out.ctypes.data))
        self._output = out
        return out

    def eval_device(self, d_tm_ptr, d_out_ptr):
        """Device pointers (ints): target times [batch_size] -> output [dim * batch_size]; asynchronous."""
        raise_for(lib.hy_cout_eval_device(self._h, ctypes.c_void_p(d_tm_ptr), ctypes.c_void_p(d_out_ptr)))

    def __repr__(self):
        return _lib.take_str(lib.hy_cout_to_string(self._h))


class event_direction(enum.IntEnum):
    """heyoka::event_direction (include/heyoka/events.hpp)."""

    negative = -1
    any = 0
    positive = 1


class native_event_counter:
    """A callback which lives in the library (hy_event_counter_nt / hy_event_counter_t of the C ABI) and only counts its
    invocations - for ensembles in which 10^5 events fire per step and a Python callback per event would be the only thing
    measured. ``.value``: the count; as a terminal callback it always continues."""

    def __init__(self):
        self._c = ctypes.c_uint64(0)

    @property
    def value(self):
        return int(self._c.value)


class nt_event:
    """nt_event_batch<double> (include/heyoka/events.hpp): ``callback(ta, time, d_sgn, batch_idx)``."""

    def __init__(self, eq, callback, direction=event_direction.any):
        if callback is None:
            raise ValueError("Cannot construct a non-terminal event with an empty callback")
        self.eq, self.callback, self.direction = _as_ex(eq), callback, event_direction(direction)


class t_event:
    """t_event_batch<double> (include/heyoka/events.hpp): ``callback(ta, d_sgn, batch_idx) -> bool`` (False stops the
    propagation; no callback: always stop); ``cooldown < 0``: deduced automatically."""

    def __init__(self, eq, callback=None, direction=event_direction.any, cooldown=-1.0):
        self.eq, self.callback, self.direction = _as_ex(eq), callback, event_direction(direction)
        self.cooldown = float(cooldown)


# Integrator objects by handle address: the C callbacks receive the handle of the integrator they run on (which may
# be a copy of the one the events were defined for).
_TAB_REGISTRY = weakref.WeakValueDictionary()


def _enum_arg(name, value, table):
    if isinstance(value, int) and not isinstance(value, bool) and value in table.values():
        return value
    if value not in table:
        raise ValueError("invalid value %r for the keyword argument '%s' (expected one of %s)"
                         % (value, name, sorted(k for k in table if k is not None)))
    return table[value]


class taylor_adaptive_batch:
    """taylor_adaptive_batch<double> (include/heyoka/taylor.hpp:781-1121) on MI355X.

    `batch_size` is the number of systems integrated concurrently (one GPU lane each).
    Arrays are exposed with shape (rows, batch_size), i.e. the reference's flat layout
    array[row * batch_size + lane].
    """

    def __init__(self, sys, state=None, batch_size=None, *, tol=None, high_accuracy=False, compact_mode=False,
                 parallel_mode=False, pars=None, time=None, device=0, t_events=(), nt_events=(), emitter=None,
                 cluster_kernel=None, exact_division=False, events_on_cluster=None, batch_semantics=None, sum_order=None,
                 _handle=None,
                 _events=None, **ignored_llvm_kwargs):
        # MI355X extensions (hy_tab_config, include/heyoka_amd.h): emitter in (None, "unrolled", "cluster", "table",
        # "block"); cluster_kernel in (None, "v5", "v3", "v2", "v1"); exact_division; events_on_cluster (None / False);
        # batch_semantics in (None = "reference", "lockstep", "per_lane").
        # LLVM-only keyword arguments of the reference (opt_level, fast_math, force_avx512,
        # slp_vectorize, code_model, parjit) are accepted and ignored.
        for k in ignored_llvm_kwargs:
            if k not in ("opt_level", "fast_math", "force_avx512", "slp_vectorize", "code_model", "parjit", "mname"):
                raise TypeError("unexpected keyword argument '%s'" % k)
        self._cb_errors = []
        if _handle is not None:
            self._h = _handle
            self._sys = sys
            self._events = _events
            _TAB_REGISTRY[int(self._h)] = self
            return
        self._sys = _to_sys(sys)
        if state is None:
            st = np.zeros(0)
        else:
            st = _f64(state)
            if batch_size is None:
                if st.ndim != 2:
                    raise ValueError("batch_size must be given unless the state is a 2D (n_eq, batch_size) array")
                batch_size = st.shape[1]
            st = st.reshape(-1)
        if batch_size is None:
            raise ValueError("batch_size must be specified")
        cfg = _lib.TabConfig()
        cfg.tol = 0.0 if tol is None else float(tol)
        cfg.high_accuracy = int(bool(high_accuracy))
        cfg.compact_mode = int(bool(compact_mode))
        cfg.parallel_mode = int(bool(parallel_mode))
        keep = []
        if pars is not None:
            p = _f64(pars).reshape(-1)
            keep.append(p)
            cfg.pars = p.ctypes.data
            cfg.n_pars = p.size
        if time is not None:
            t = _f64(time).reshape(-1)
            keep.append(t)
            cfg.time = t.ctypes.data
            cfg.n_time = t.size
        cfg.device = int(device)
        cfg.emitter = _enum_arg("emitter", emitter, {None: 0, "auto": 0, "unrolled": 1, "cluster": 2, "table": 3, "block": 4})
        cfg.cluster_kernel = _enum_arg("cluster_kernel", cluster_kernel, {None: 0, "auto": 0, "v5": 5, "v3": 3, "v2": 2, "v1": 1})
        cfg.exact_division = int(bool(exact_division))
        cfg.events_on_cluster = 1 if events_on_cluster is False else 0
        cfg.sum_order = _enum_arg("sum_order", sum_order, {None: 0, "auto": 0, "pairwise": 1, "running": 2})
        cfg.batch_semantics = _enum_arg("batch_semantics", batch_semantics,
                                        {None: 0, "reference": 0, "lockstep": 1, "per_lane": 2})
        if tol is not None and float(tol) == 0.0:
            cfg.tol = 0.0
        t_events, nt_events = list(t_events), list(nt_events)
        self._events = None
        if t_events or nt_events:
            # C trampolines: look the Python integrator up by handle, record exceptions (re-raised after the C call).
            def make_nt(ev):
                if isinstance(ev.callback, native_event_counter):
                    return ctypes.cast(lib.hy_event_counter_nt, _lib.NT_EVENT_CB)

                def tramp(handle, tm, d_sgn, idx, _user):
                    ta = _TAB_REGISTRY.get(int(handle))
                    try:
                        ev.callback(ta, float(tm), int(d_sgn), int(idx))
                    except BaseException as e:
                        if ta is not None:
                            ta._cb_errors.append(e)
                return _lib.NT_EVENT_CB(tramp)

            def make_t(ev):
                if ev.callback is None:
                    return ctypes.cast(None, _lib.T_EVENT_CB)
                if isinstance(ev.callback, native_event_counter):
                    return ctypes.cast(lib.hy_event_counter_t, _lib.T_EVENT_CB)

                def tramp(handle, d_sgn, idx, _user):
                    ta = _TAB_REGISTRY.get(int(handle))
                    try:
                        return 1 if ev.callback(ta, int(d_sgn), int(idx)) else 0
                    except BaseException as e:
                        if ta is not None:
                            ta._cb_errors.append(e)
                        return 0
                return _lib.T_EVENT_CB(tramp)

            te_arr = (_lib.TEvent * max(len(t_events), 1))()
            nte_arr = (_lib.NtEvent * max(len(nt_events), 1))()
            cbs = []
            for k, ev in enumerate(t_events):
                cb = make_t(ev)
                cbs.append(cb)
                user = ctypes.cast(ctypes.byref(ev.callback._c), ctypes.c_void_p) if isinstance(ev.callback, native_event_counter) else None
                te_arr[k] = _lib.TEvent(ev.eq._h, cb, user, int(ev.direction), ev.cooldown)
            for k, ev in enumerate(nt_events):
                cb = make_nt(ev)
                cbs.append(cb)
                user = ctypes.cast(ctypes.byref(ev.callback._c), ctypes.c_void_p) if isinstance(ev.callback, native_event_counter) else None
                nte_arr[k] = _lib.NtEvent(ev.eq._h, cb, user, int(ev.direction))
            self._events = (t_events, nt_events, cbs)
            self._h = check_handle(
                lib.hy_tab_create_with_events(self._sys._h, st.ctypes.data if st.size else None, st.size,
                                              int(batch_size), ctypes.byref(cfg), te_arr, len(t_events), nte_arr,
                                              len(nt_events)))
        else:
            self._h = check_handle(
                lib.hy_tab_create(self._sys._h, st.ctypes.data if st.size else None, st.size, int(batch_size),
                                  ctypes.byref(cfg))
            )
        _TAB_REGISTRY[int(self._h)] = self

    def _raise_cb_errors(self):
        if self._cb_errors:
            e = self._cb_errors[0]
            self._cb_errors = []
            raise e

    @property
    def with_events(self):
        return bool(lib.hy_tab_with_events(self._h))

    def set_event_timing(self, on=True):
        """Phase timing of the steps with events (one stream synchronisation per phase while it is on)."""
        raise_for(lib.hy_tab_set_event_timing(self._h, 1 if on else 0))

    @property
    def event_stats(self):
        out = np.zeros(8)
        raise_for(lib.hy_tab_get_event_stats(self._h, out.ctypes.data))
        keys = ("steps", "ms_upload", "ms_stepper", "ms_detection", "ms_bookkeeping_flags", "ms_update_records",
                "tc_regeneration_launches", "systems_with_events")
        return dict(zip(keys, [float(x) for x in out]))

    def reset_cooldowns(self, batch_idx=None):
        raise_for(lib.hy_tab_reset_cooldowns(self._h, -1 if batch_idx is None else int(batch_idx)))

    @property
    def te_cooldowns(self):
        """Per lane, per terminal event: None or (elapsed, duration)."""
        n_te = len(self._events[0]) if self._events else 0
        n = self.batch_size
        first, second = np.zeros(max(n * n_te, 1)), np.zeros(max(n * n_te, 1))
        active = np.zeros(max(n * n_te, 1), dtype=np.int32)
        raise_for(lib.hy_tab_get_te_cooldowns(self._h, first.ctypes.data, second.ctypes.data, active.ctypes.data))
        return [[(float(first[i * n_te + e]), float(second[i * n_te + e])) if active[i * n_te + e] else None
                 for e in range(n_te)] for i in range(n)]

    def __del__(self, _free=lib.hy_tab_free):
        h = getattr(self, "_h", None)
        if h:
            _free(h)
            self._h = None

    def __copy__(self):
        return taylor_adaptive_batch(self._sys, _handle=check_handle(lib.hy_tab_copy(self._h)), _events=self._events)

    def __deepcopy__(self, memo):
        return self.__copy__()

    copy = __copy__

    # ---- read-only properties ----
    @property
    def batch_size(self):
        return int(lib.hy_tab_get_batch_size(self._h))

    @property
    def order(self):
        return int(lib.hy_tab_get_order(self._h))

    @property
    def dim(self):
        return int(lib.hy_tab_get_dim(self._h))

    @property
    def n_pars(self):
        return int(lib.hy_tab_get_n_pars(self._h))

    @property
    def n_uvars(self):
        return int(lib.hy_tab_get_n_uvars(self._h))

    @property
    def tol(self):
        return float(lib.hy_tab_get_tol(self._h))

    @property
    def high_accuracy(self):
        return bool(lib.hy_tab_get_high_accuracy(self._h))

    @property
    def compact_mode(self):
        return bool(lib.hy_tab_get_compact_mode(self._h))

    @property
    def event_detection_failures(self):
        """Events ignored in a step because the root isolation or the root finder failed (the reference logs a warning)."""
        return int(lib.hy_tab_get_event_detection_failures(self._h))

    @property
    def compile_seconds(self):
        return float(lib.hy_tab_get_compile_seconds(self._h))

    @property
    def hip_source(self):
        return take_str(lib.hy_tab_get_hip_source(self._h))

    @property
    def internal_program(self):
        """The rewritten internal program the stepper was generated from (list of strings: nodes, then the definitions of
        the state derivatives), or [] when the code was generated from the decomposition itself."""
        txt = take_str(lib.hy_tab_get_internal_program(self._h)).rstrip("\n")
        return txt.split("\n") if txt else []

    @property
    def code_object(self):
        """The gfx950 code object of the stepper module (bytes), e.g. for llvm-objdump."""
        n = ctypes.c_size_t(0)
        raise_for(lib.hy_tab_get_code_object(self._h, None, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        raise_for(lib.hy_tab_get_code_object(self._h, buf, ctypes.byref(n)))
        return buf.raw[: n.value]

    @property
    def hip_source_mode(self):
        """Which code generator produced the kernels ("unrolled", "cluster ...", "table ...")."""
        return take_str(lib.hy_tab_get_codegen_info(self._h))

    @property
    def decomposition(self):
        return take_str(lib.hy_tab_get_decomposition_str(self._h)).rstrip("\n").split("\n")

    # ---- state / time / pars ----
    @property
    def state(self):
        out = np.empty((self.dim, self.batch_size))
        raise_for(lib.hy_tab_get_state(self._h, out.ctypes.data))
        return out

    @state.setter
    def state(self, v):
        a = _f64(v).reshape(-1)
        if a.size != self.dim * self.batch_size:
            raise ValueError("invalid state size")
        raise_for(lib.hy_tab_set_state(self._h, a.ctypes.data))

    def state_data(self):
        """get_state_data(): a WRITABLE (dim, batch_size) view of the host mirror of the state, valid while the integrator
        lives. Like in the reference, what is written through it is what the next step starts from - which makes the
        integrator refresh the mirror after every launch from now on (eager synchronisation); ``ta.state = ...`` and the
        device views do not."""
        p = lib.hy_tab_get_state_data(self._h)
        if not p:
            raise RuntimeError(_lib.last_error())
        n = self.dim * self.batch_size
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(n,))
        return arr.reshape(self.dim, self.batch_size)

    @property
    def pars(self):
        out = np.empty((self.n_pars, self.batch_size))
        if out.size:
            raise_for(lib.hy_tab_get_pars(self._h, out.ctypes.data))
        return out

    @pars.setter
    def pars(self, v):
        a = _f64(v).reshape(-1)
        if a.size != self.n_pars * self.batch_size:
            raise ValueError("invalid pars size")
        if a.size:
            raise_for(lib.hy_tab_set_pars(self._h, a.ctypes.data))

    @property
    def dtime(self):
        hi, lo = np.empty(self.batch_size), np.empty(self.batch_size)
        raise_for(lib.hy_tab_get_dtime(self._h, hi.ctypes.data, lo.ctypes.data))
        return hi, lo

    @dtime.setter
    def dtime(self, v):
        hi, lo = v
        hi, lo = _f64(hi).reshape(-1), _f64(lo).reshape(-1)
        raise_for(lib.hy_tab_set_dtime(self._h, hi.ctypes.data, lo.ctypes.data, hi.size))

    @property
    def time(self):
        return self.dtime[0]

    @time.setter
    def time(self, v):
        t = _f64(v).reshape(-1)
        raise_for(lib.hy_tab_set_time(self._h, t.ctypes.data, t.size))

    def set_time(self, v):
        self.time = v

    def set_dtime(self, hi, lo):
        self.dtime = (hi, lo)

    @property
    def tc(self):
        out = np.empty((self.dim, self.order + 1, self.batch_size))
        raise_for(lib.hy_tab_get_tc(self._h, out.ctypes.data))
        return out

    @property
    def last_h(self):
        out = np.empty(self.batch_size)
        raise_for(lib.hy_tab_get_last_h(self._h, out.ctypes.data))
        return out

    def update_d_output(self, t, rel_time=False):
        tt = _f64(t).reshape(-1)
        out = np.empty((self.dim, self.batch_size))
        raise_for(lib.hy_tab_update_d_output(self._h, tt.ctypes.data, tt.size, int(bool(rel_time)), out.ctypes.data))
        return out

    # ---- stepping ----
    def step(self, max_delta_t=None, write_tc=False):
        if max_delta_t is None:
            rc = lib.hy_tab_step(self._h, int(bool(write_tc)))
            self._raise_cb_errors()
            raise_for(rc)
        else:
            m = _f64(max_delta_t).reshape(-1)
            rc = lib.hy_tab_step_limited(self._h, m.ctypes.data, m.size, int(bool(write_tc)))
            self._raise_cb_errors()
            raise_for(rc)

    def step_backward(self, write_tc=False):
        rc = lib.hy_tab_step_backward(self._h, int(bool(write_tc)))
        self._raise_cb_errors()
        raise_for(rc)

    @property
    def step_res(self):
        n = self.batch_size
        oc = np.empty(n, dtype=np.int64)
        h = np.empty(n)
        raise_for(lib.hy_tab_get_step_res(self._h, oc.ctypes.data, h.ctypes.data))
        return [(_outcome(oc[i]), float(h[i])) for i in range(n)]

    def _callback_descs(self, callback, err):
        """kw::callback: one callback - any callable taking the integrator and returning a truth value, optionally with a
        pre_hook(integrator) method (include/heyoka/step_callback.hpp:46-62) - or a list / tuple of them (a callback set:
        all run at every step, the results are and-ed). Returns (array of hy_step_callback_desc or None, count, keepalive)."""
        if callback is None:
            return None, 0, None
        cbs = list(callback) if isinstance(callback, (list, tuple)) else [callback]
        for c in cbs:
            if c is None or not callable(c):
                raise ValueError("Cannot construct a callback set containing one or more empty callbacks"
                                 if len(cbs) > 1 or isinstance(callback, (list, tuple)) else "the step callback must be callable")
        arr = (_lib.StepCallbackDesc * len(cbs))()
        keep = []

        def make(c):
            def call(_tab, _data):
                try:
                    return 1 if c(self) else 0
                except BaseException as e:  # propagate Python exceptions out of the C frame
                    err.append(e)
                    return -1  # the other members of the set are not run, the propagation stops

            def pre(_tab, _data):
                try:
                    c.pre_hook(self)
                    return 0
                except BaseException as e:
                    err.append(e)
                    return 1  # the propagation is not started (the reference lets the exception out of propagate_*())

            f_call = _lib.STEP_CALLBACK(call)
            f_pre = _lib.STEP_PRE_HOOK(pre) if hasattr(c, "pre_hook") else _lib.STEP_PRE_HOOK()
            keep.extend((f_call, f_pre))
            return f_call, f_pre

        for i, c in enumerate(cbs):
            arr[i].call, arr[i].pre_hook = make(c)
            arr[i].user_data = None
        return arr, len(cbs), keep

    def _propagate(self, fn, t, max_steps, max_delta_t, callback, write_tc, c_output):
        tt = _f64(t).reshape(-1)
        if max_delta_t is None:
            mptr, mn, mkeep = None, 0, None
        else:
            mkeep = _f64(max_delta_t).reshape(-1)
            mptr, mn = mkeep.ctypes.data, mkeep.size
        err = []
        descs, n_cbs, keep = self._callback_descs(callback, err)
        fn_cbs = lib.hy_tab_propagate_until_cbs if fn is lib.hy_tab_propagate_until else lib.hy_tab_propagate_for_cbs
        rc = fn_cbs(self._h, tt.ctypes.data, tt.size, int(max_steps), mptr, mn, descs, n_cbs, int(bool(write_tc)),
                    int(bool(c_output)))
        del keep
        if err:
            raise err[0]
        self._raise_cb_errors()
        raise_for(rc)
        c_out = None
        if c_output:
            h = ctypes.c_void_p()
            raise_for(lib.hy_tab_take_c_output(self._h, ctypes.byref(h)))
            if h.value:
                c_out = continuous_output_batch(h.value)
        # NOTE: like the reference's Python binding: (continuous output or None, callback).
        return c_out, callback

    def propagate_until(self, t, max_steps=0, max_delta_t=None, callback=None, write_tc=False, c_output=False):
        return self._propagate(lib.hy_tab_propagate_until, t, max_steps, max_delta_t, callback, write_tc, c_output)

    def propagate_for(self, delta_t, max_steps=0, max_delta_t=None, callback=None, write_tc=False, c_output=False):
        return self._propagate(lib.hy_tab_propagate_for, delta_t, max_steps, max_delta_t, callback, write_tc, c_output)

    def propagate_grid(self, grid, max_steps=0, max_delta_t=None, callback=None):
        g = _f64(grid)
        if g.ndim == 1:
            g = np.repeat(g[:, None], self.batch_size, axis=1)
        g = np.ascontiguousarray(g)
        n_grid = g.shape[0]
        out = np.empty((n_grid, self.dim, self.batch_size))
        if max_delta_t is None:
            mptr, mn, mkeep = None, 0, None
        else:
            mkeep = _f64(max_delta_t).reshape(-1)
            mptr, mn = mkeep.ctypes.data, mkeep.size
        err = []
        descs, n_cbs, keep = self._callback_descs(callback, err)
        rc = lib.hy_tab_propagate_grid_cbs(self._h, g.ctypes.data, n_grid, int(max_steps), mptr, mn, descs, n_cbs,
                                           out.ctypes.data)
        del keep
        if err:
            raise err[0]
        self._raise_cb_errors()
        raise_for(rc)
        return callback, out

    def propagate_grid_device(self, grid, out_ptr, max_steps=0, max_delta_t=None):
        """propagate_grid() with the samples written to a caller-owned device buffer (``out_ptr``: device address of
        n_grid * dim * batch_size doubles, layout [point, var, lane]); ``grid``: 1-D (shared by all the lanes) or
        (n_grid, batch_size). Nothing of that size is materialised on the host."""
        g = np.ascontiguousarray(_f64(grid))
        scalar = g.ndim == 1
        if max_delta_t is None:
            mptr, mn, mkeep = None, 0, None
        else:
            mkeep = _f64(max_delta_t).reshape(-1)
            mptr, mn = mkeep.ctypes.data, mkeep.size
        raise_for(lib.hy_tab_propagate_grid_device(self._h, g.ctypes.data, g.shape[0], int(scalar), int(max_steps),
                                                   mptr, mn, ctypes.c_void_p(int(out_ptr))))

    @property
    def propagate_res(self):
        n = self.batch_size
        oc = np.empty(n, dtype=np.int64)
        mn, mx = np.empty(n), np.empty(n)
        ns = np.empty(n, dtype=np.uint64)
        raise_for(lib.hy_tab_get_propagate_res(self._h, oc.ctypes.data, mn.ctypes.data, mx.ctypes.data, ns.ctypes.data))
        return [(_outcome(oc[i]), float(mn[i]), float(mx[i]), int(ns[i])) for i in range(n)]

    def propagate_res_arrays(self):
        """(outcome, min_h, max_h, n_steps) as numpy arrays (cheap for 10^6 lanes)."""
        n = self.batch_size
        oc = np.empty(n, dtype=np.int64)
        mn, mx = np.empty(n), np.empty(n)
        ns = np.empty(n, dtype=np.uint64)
        raise_for(lib.hy_tab_get_propagate_res(self._h, oc.ctypes.data, mn.ctypes.data, mx.ctypes.data, ns.ctypes.data))
        return oc, mn, mx, ns

    # ---- device-resident access (MI355X extension) ----
    _BUFS = {"state": 0, "pars": 1, "time_hi": 2, "time_lo": 3, "tc": 4, "n_steps": 5, "outcome": 6, "last_h": 7}

    def device_array(self, which):
        """Zero-copy __cuda_array_interface__ view of a device array: 'state' (dim, N),
        'pars' (n_pars, N), 'time_hi' (N,), 'time_lo' (N,), 'tc' (dim, order + 1, N)."""
        ptr = check_handle(lib.hy_tab_device_ptr(self._h, self._BUFS[which]))
        N = self.batch_size
        shape = {
            "state": (self.dim, N),
            "pars": (self.n_pars, N),
            "time_hi": (N,),
            "time_lo": (N,),
            "tc": (self.dim, self.order + 1, N),
            "n_steps": (N,),
            "outcome": (N,),
            "last_h": (N,),
        }[which]
        return _DeviceArray(ptr, shape, self, "<i8" if which in ("n_steps", "outcome") else "<f8")

    def mark_device_modified(self):
        raise_for(lib.hy_tab_mark_device_modified(self._h))

    def set_stream(self, stream_ptr):
        raise_for(lib.hy_tab_set_stream(self._h, ctypes.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def synchronize(self):
        raise_for(lib.hy_tab_synchronize(self._h))

    @property
    def last_total_steps(self):
        return int(lib.hy_tab_get_last_total_steps(self._h))

    def kernel_ms_history(self, n=64):
        """Durations (ms) of the last n stepper kernels, from HIP events on the launch stream."""
        out = np.empty(int(n))
        k = int(lib.hy_tab_get_kernel_ms_history(self._h, out.ctypes.data, int(n)))
        return out[:k]

    def raw_step(self, d_state, d_pars, d_time, d_h, d_tc, n_systems, d_tape=None):
        """Stepper function-pointer ABI on caller-owned device buffers (integers = device addresses); d_tape: the tape in
        caller-owned memory of raw_tape_size_align() bytes (c_step_f_t)."""
        if d_tape is None:
            raise_for(lib.hy_tab_raw_step(self._h, int(d_state), int(d_pars) if d_pars else None, int(d_time), int(d_h),
                                          int(d_tc) if d_tc else None, int(n_systems)))
        else:
            raise_for(lib.hy_tab_raw_step_tape(self._h, int(d_state), int(d_pars) if d_pars else None, int(d_time), int(d_h),
                                               int(d_tc) if d_tc else None, int(d_tape), int(n_systems)))

    def raw_step_e(self, d_jet, d_state, d_pars, d_time, d_h, d_max_abs_state, n_systems, d_tape=None):
        """step_f_e_t: jets of the state variables and of the event equations, step size, max |x_i|; no state update."""
        if d_tape is None:
            raise_for(lib.hy_tab_raw_step_e(self._h, int(d_jet), int(d_state), int(d_pars) if d_pars else None, int(d_time),
                                            int(d_h), int(d_max_abs_state), int(n_systems)))
        else:
            raise_for(lib.hy_tab_raw_step_e_tape(self._h, int(d_jet), int(d_state), int(d_pars) if d_pars else None, int(d_time),
                                                 int(d_h), int(d_max_abs_state), int(d_tape), int(n_systems)))

    def raw_d_out_f(self, d_out, d_tc, d_h, n_systems):
        """d_out_f_t: the Taylor polynomials d_tc evaluated at d_h."""
        raise_for(lib.hy_tab_raw_d_out_f(self._h, int(d_out), int(d_tc), int(d_h), int(n_systems)))

    def raw_tape_size_align(self, n_systems):
        """(bytes, alignment) of the tape of the compact-mode steppers for n_systems systems (0 bytes: none needed)."""
        sz, al = ctypes.c_size_t(0), ctypes.c_size_t(0)
        raise_for(lib.hy_tab_tape_size_align(self._h, int(n_systems), ctypes.byref(sz), ctypes.byref(al)))
        return int(sz.value), int(al.value)


def _ensemble(fn, ta, t, n_iter, gen, max_steps, n_devices):
    if n_iter <= 0:
        # Reference: an empty result (test/ensemble_propagate.cpp:90-99).
        return []
    err = []

    def _gen(tab_handle, i, _data):
        # Wrap the raw handle (not owned) so that gen can use the integrator interface.
        view = taylor_adaptive_batch(ta._sys, _handle=tab_handle)
        try:
            gen(view, int(i))
            return 0
        except BaseException as e:
            err.append(e)
            return 1
        finally:
            view._h = None  # do not free: owned by the library

    cgen = _lib.ENSEMBLE_GEN(_gen)
    out = (ctypes.c_void_p * n_iter)()
    rc = fn(ta._h, float(t), int(n_iter), ctypes.cast(cgen, ctypes.c_void_p), None, int(max_steps), int(n_devices), out)
    if err:
        raise err[0]
    raise_for(rc)
    return [taylor_adaptive_batch(ta._sys, _handle=out[i]) for i in range(n_iter)]


def ensemble_propagate_until_batch(ta, t, n_iter, gen, max_steps=0, n_devices=0):
    """ensemble_propagate_until_batch() (include/heyoka/ensemble_propagate.hpp:222-237).

    gen(ta_copy, i) modifies the copy in place (the reference's generator returns the copy)."""
    return _ensemble(lib.hy_ensemble_propagate_until_batch, ta, t, n_iter, gen, max_steps, n_devices)


def ensemble_propagate_for_batch(ta, delta_t, n_iter, gen, max_steps=0, n_devices=0):
    """ensemble_propagate_for_batch() (include/heyoka/ensemble_propagate.hpp:239-254)."""
    return _ensemble(lib.hy_ensemble_propagate_for_batch, ta, delta_t, n_iter, gen, max_steps, n_devices)


def ensemble_gather_results(tas, dst_device=0):
    """Everything the propagated copies hold, gathered in one piece (hy_ensemble_gather_results()): a dict with "state"
    (dim, n_total), "time_hi", "time_lo", "outcome" (int64), "n_steps" (uint64), "min_h", "max_h" (n_total each) and
    "used_rccl"."""
    tas = list(tas)
    if not tas:
        return {"state": np.zeros((0, 0)), "used_rccl": False}
    n_total = int(np.sum([t.batch_size for t in tas]))
    dim = tas[0].dim
    out = np.empty((dim + 6, n_total))
    arr = (ctypes.c_void_p * len(tas))(*[t._h for t in tas])
    used = ctypes.c_int(0)
    raise_for(lib.hy_ensemble_gather_results(arr, len(tas), int(dst_device), out.ctypes.data_as(ctypes.c_void_p), out.size, 0,
                                             ctypes.byref(used)))
    return {"state": out[:dim].copy(), "time_hi": out[dim].copy(), "time_lo": out[dim + 1].copy(),
            "outcome": out[dim + 2].copy().view(np.int64), "n_steps": out[dim + 3].copy().view(np.uint64),
            "min_h": out[dim + 4].copy(), "max_h": out[dim + 5].copy(), "used_rccl": bool(used.value)}


def ensemble_gather_states(tas, dst_device=0):
    """The final states of a list of integrators (e.g. the result of ensemble_propagate_*_batch(), each copy on the device
    which propagated it) gathered into one host array (dim, sum of the batch sizes) - the native gather behind the C ABI
    (hy_ensemble_gather_states(): RCCL over xGMI between devices, device-to-device copies on one). Returns (array,
    used_rccl)."""
    tas = list(tas)
    if not tas:
        return np.zeros((0, 0)), False
    n_total = int(np.sum([t.batch_size for t in tas]))
    out = np.empty((tas[0].dim, n_total))
    arr = (ctypes.c_void_p * len(tas))(*[t._h for t in tas])
    used = ctypes.c_int(0)
    raise_for(lib.hy_ensemble_gather_states(arr, len(tas), int(dst_device), out.ctypes.data_as(ctypes.c_void_p), out.size, 0,
                                            ctypes.byref(used)))
    return out, bool(used.value)
