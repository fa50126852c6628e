"""TEST INFRASTRUCTURE (CPU baseline / checker), never imported by the product.

Compiled CPU stand-in for the reference's default-mode SIMD JIT (src/taylor_02.cpp:1339-1418: fully unrolled jet,
pairwise sums; SIMD width from llvm_state.cpp:656; one worker per core over batches, src/ensemble_propagate.cpp:203-219).

The Taylor decomposition built by heyoka_oracle.py is turned into straight-line C, one function per order, every
statement a fixed-width vector operation on a batch of 8 systems (GCC vector extensions -> AVX-512 / 2 x AVX2), in
exactly the operation order of the interpreter oracle/taylor_oracle.c (node_diff() / sv_diff()). The generated
function fills the tape of Taylor coefficients and is plugged into the oracle library through
hy_oracle_set_jet_hook(): step-size selection, state update, double-length time, the propagate_until() loop and
the OpenMP ensemble driver stay the oracle's own code.

Two builds of the same source:
  strict  -O2 -ffp-contract=off : bit-identical to the interpreter (tests/test_oracle_golden.py checks it)
  fast    -O3 -ffp-contract=fast: what bench.py times (the reference's LLVM builder also enables `contract`,
          src/llvm_state.cpp:843-845)
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

import heyoka_oracle as ho

W = 8  # batch width of the generated code

_HERE = os.path.dirname(os.path.abspath(__file__))


def _lit(x):
    x = float(x)
    if x != x:
        return "__builtin_nan(\"\")"
    if x in (float("inf"), float("-inf")):
        return "(-__builtin_inf())" if x < 0 else "__builtin_inf()"
    return "(" + x.hex() + ")"


class _Gen:
    def __init__(self, oi):
        self.oi = oi
        self.n_eq, self.n_u, self.order = oi.n_eq, oi.n_u, oi.order
        self.lines = []

    # Operand accessors -------------------------------------------------------------------------------------------
    def T(self, k, u):
        return "T(%d,%d)" % (k, u)

    def numpar(self, a):
        if a.tag == "num":
            return "B(%s)" % _lit(a.val)
        if a.tag == "par":
            return "P(%d)" % a.val
        raise ValueError(a)

    @staticmethod
    def isvar(a):
        return a.tag == "var"

    @staticmethod
    def uidx(a):
        return ho._uidx(a.val)

    @staticmethod
    def pairwise(terms):
        terms = list(terms)
        while len(terms) != 1:
            nxt = []
            for i in range(0, len(terms), 2):
                if i + 1 == len(terms):
                    nxt.append(terms[i])
                else:
                    nxt.append("(%s + %s)" % (terms[i], terms[i + 1]))
            terms = nxt
        return terms[0]

    def pow_ebs(self, base, e):
        if e == 0:
            return "B(1.)"
        if e == 1:
            return base
        if e % 2 == 0:
            return self.pow_ebs("(%s * %s)" % (base, base), e // 2)
        return "(%s * %s)" % (base, self.pow_ebs("(%s * %s)" % (base, base), (e - 1) // 2))

    def pow_eval(self, out, b, ex, emit):
        """Order-0 evaluation of pow (pow_eval() of taylor_oracle.c)."""
        if np.isfinite(ex) and ex == np.trunc(ex) and abs(ex) <= 16:
            if ex >= 0:
                emit("%s = %s;" % (out, self.pow_ebs(b, int(ex))))
            else:
                emit("%s = B(1.) / %s;" % (out, self.pow_ebs(b, int(-ex))))
            return
        y = 2 * ex
        if np.isfinite(ex) and ex != np.trunc(ex) and y == np.trunc(y) and abs(y) <= 16:
            emit("{ vd t_; const vd b_ = %s; for (int l = 0; l < 8; ++l) t_[l] = sqrt(b_[l]);" % b)
            if y >= 0:
                emit("%s = %s; }" % (out, self.pow_ebs("t_", int(y))))
            else:
                emit("%s = B(1.) / %s; }" % (out, self.pow_ebs("t_", int(-y))))
            return
        emit("{ vd t_; const vd b_ = %s; for (int l = 0; l < 8; ++l) t_[l] = pow(b_[l], %s); %s = t_; }" % (b, _lit(ex), out))

    # One node at one order -----------------------------------------------------------------------------------------
    def node(self, i, k, emit):
        ex, _deps = self.oi.dc[self.n_eq + i]
        u = self.n_eq + i
        out = self.T(k, u)
        a = ex.args
        kind = ex.kind
        iv = self.isvar
        if kind == "num_identity":
            emit("%s = %s;" % (out, self.numpar(a[0]) if k == 0 else "B(0.)"))
        elif kind == "time":
            emit("%s = %s;" % (out, "TM" if k == 0 else ("B(1.)" if k == 1 else "B(0.)")))
        elif kind == "sum":
            terms = [self.T(k, self.uidx(x)) if iv(x) else (self.numpar(x) if k == 0 else "B(0.)") for x in a]
            emit("%s = %s;" % (out, self.pairwise(terms)))
        elif kind == "sub":
            if iv(a[0]) and iv(a[1]):
                emit("%s = %s - %s;" % (out, self.T(k, self.uidx(a[0])), self.T(k, self.uidx(a[1]))))
            elif iv(a[0]):
                x = self.T(k, self.uidx(a[0]))
                emit("%s = %s;" % (out, ("%s - %s" % (x, self.numpar(a[1]))) if k == 0 else x))
            elif iv(a[1]):
                y = self.T(k, self.uidx(a[1]))
                emit("%s = %s;" % (out, ("%s - %s" % (self.numpar(a[0]), y)) if k == 0 else "-" + y))
            else:
                emit("%s = %s;" % (out, ("%s - %s" % (self.numpar(a[0]), self.numpar(a[1]))) if k == 0 else "B(0.)"))
        elif kind == "prod":
            if len(a) != 2:
                raise NotImplementedError("prod with %d arguments" % len(a))
            if iv(a[0]) and iv(a[1]):
                x, y = self.uidx(a[0]), self.uidx(a[1])
                emit("%s = %s;" % (out, self.pairwise(["(%s * %s)" % (self.T(k - j, x), self.T(j, y)) for j in range(k + 1)])))
            elif not iv(a[0]) and not iv(a[1]):
                if k != 0:
                    emit("%s = B(0.);" % out)
                elif a[0].tag == "num" and a[0].val == -1.0:
                    emit("%s = -%s;" % (out, self.numpar(a[1])))
                else:
                    emit("%s = %s * %s;" % (out, self.numpar(a[0]), self.numpar(a[1])))
            else:
                vi = 0 if iv(a[0]) else 1
                x = self.T(k, self.uidx(a[vi]))
                c = a[1 - vi]
                if vi == 1 and c.tag == "num" and c.val == -1.0:
                    emit("%s = -%s;" % (out, x))
                else:
                    emit("%s = %s * %s;" % (out, self.numpar(c), x))
        elif kind == "sum_sq":
            accs = []
            for x in a:
                if not iv(x):
                    if k == 0:
                        accs.append("(%s * %s)" % (self.numpar(x), self.numpar(x)))
                    else:
                        accs.append("B(0.)")
                    continue
                xu = self.uidx(x)
                if k % 2 == 1:
                    nt = (k - 1) // 2 + 1
                    accs.append(self.pairwise(["(%s * %s)" % (self.T(k - j, xu), self.T(j, xu)) for j in range(nt)]))
                else:
                    sq = "(%s * %s)" % (self.T(k // 2, xu), self.T(k // 2, xu))
                    if k > 0:
                        nt = (k - 2) // 2 + 1
                        ps = self.pairwise(["(%s * %s)" % (self.T(k - j, xu), self.T(j, xu)) for j in range(nt)])
                        accs.append("((%s + %s) + %s)" % (ps, ps, sq))
                    else:
                        accs.append(sq)
            if k % 2 == 1:
                # NOTE: the doubled sum is evaluated once (vd temporary): (s + s) with s the pairwise sum.
                emit("{ const vd s_ = %s; %s = s_ + s_; }" % (self.pairwise(accs), out))
            else:
                emit("%s = %s;" % (out, self.pairwise(accs)))
        elif kind == "pow":
            e = a[1]
            if e.tag != "num":
                raise NotImplementedError("pow with a non-numerical exponent")
            exv = float(e.val)
            if not iv(a[0]):
                if k == 0:
                    self.pow_eval(out, self.numpar(a[0]), exv, emit)
                else:
                    emit("%s = B(0.);" % out)
                return
            b = self.uidx(a[0])
            if k == 0:
                self.pow_eval(out, self.T(0, b), exv, emit)
            elif exv == 0.5 or exv == 2.0:
                raise NotImplementedError("sqrt / square special cases")
            else:
                terms = []
                for j in range(k):
                    sf = float(k) * exv - float(j) * (exv + 1.0)
                    terms.append("(B(%s) * (%s * %s))" % (_lit(sf), self.T(k - j, b), self.T(j, u)))
                emit("%s = %s / (B(%s) * %s);" % (out, self.pairwise(terms), _lit(float(k)), self.T(0, b)))
        else:
            raise NotImplementedError("node kind %s" % kind)

    def sv(self, k, emit):
        for i in range(self.n_eq):
            ex, _ = self.oi.dc[self.n_u + i]
            out = self.T(k, i)
            if ex.tag == "var":
                emit("%s = %s / B(%s);" % (out, self.T(k - 1, ho._uidx(ex.val)), _lit(float(k))))
            elif k == 1:
                emit("%s = %s;" % (out, self.numpar(ex)))
            else:
                emit("%s = B(0.);" % out)

    def source(self):
        n_nodes = self.n_u - self.n_eq
        out = []
        out.append("/* Generated by oracle/compiled_baseline.py - test infrastructure, not part of the product. */")
        out.append("#include <math.h>\n#include <stddef.h>")
        out.append("typedef double vd __attribute__((vector_size(64), aligned(8)));")
        out.append("#define NU %d" % self.n_u)
        out.append("#define T(k,u) (*(vd *)(tape + ((size_t)(k) * NU + (u)) * 8))")
        out.append("#define P(i) (*(const vd *)(pars + (size_t)(i) * 8))")
        out.append("#define TM (*(const vd *)(time))")
        out.append("#define B(x) ((vd){(x),(x),(x),(x),(x),(x),(x),(x)})")
        for k in range(self.order + 1):
            body = []
            emit = body.append
            if k >= 1:
                self.sv(k, emit)
            if k < self.order:
                for i in range(n_nodes):
                    self.node(i, k, emit)
            out.append("static void __attribute__((noinline)) order_%d(const double *restrict pars, const double *restrict time, "
                       "double *restrict tape)\n{\n(void)pars; (void)time;\n%s\n}" % (k, "\n".join(body)))
        out.append("void hy_jet_hook(const double *state, const double *pars, const double *time, double *tape)\n{")
        out.append("for (int i = 0; i < %d; ++i) T(0, i) = *(const vd *)(state + (size_t)i * 8);" % self.n_eq)
        for k in range(self.order + 1):
            out.append("order_%d(pars, time, tape);" % k)
        out.append("}")
        return "\n".join(out) + "\n"


def build(oi, fast):
    """Generate + compile the jet function of an OracleIntegrator built with batch size 8.
    Returns (path of the shared library, seconds spent compiling)."""
    import time

    if oi.batch_size != W:
        raise ValueError("the compiled baseline is generated for batches of %d systems" % W)
    src = _Gen(oi).source()
    flags = ["-O3", "-march=native", "-ffp-contract=fast"] if fast else ["-O2", "-march=native", "-ffp-contract=off"]
    tag = hashlib.sha256((src + " ".join(flags) + ho.cpu_tag()).encode()).hexdigest()[:16]
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    c_path = os.path.join(out_dir, "jet_%s.c" % tag)
    so = os.path.join(out_dir, "libjet_%s.so" % tag)
    t0 = time.perf_counter()
    if not os.path.exists(so):
        with open(c_path, "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", *flags, "-shared", "-fPIC", "-o", so, c_path, "-lm"])
    return so, time.perf_counter() - t0


_KEEP = []


def install(oi, fast):
    """Plug the compiled jet function of `oi` into the oracle library (process-wide, batch width 8)."""
    so, secs = build(oi, fast)
    lib = ctypes.CDLL(so)
    _KEEP.append(lib)
    fn = ctypes.cast(lib.hy_jet_hook, ctypes.c_void_p)
    lib_o = ho._lib()
    lib_o.hy_oracle_program_hash.restype = ctypes.c_uint64
    # (Keyed on the content of the program: another oracle program with the same batch width / n_u / order keeps the
    # interpreter.)
    lib_o.hy_oracle_set_jet_hook(fn, ctypes.c_int(W), ctypes.c_uint64(lib_o.hy_oracle_program_hash(oi.program_ptr())))
    return secs


def uninstall():
    ho._lib().hy_oracle_set_jet_hook(ctypes.c_void_p(0), ctypes.c_int(0), ctypes.c_uint64(0))
