// "Table" code generation mode: the GPU analogue of the reference's compact mode
// (taylor_compute_jet_compact_mode(), src/taylor_02.cpp:1194-1260; per-node taylor_c_diff_func
// functions driven by index tables, src/taylor_02.cpp:560-614).
//
// Used for decompositions that are too large to unroll and that do not fit the cluster scheme
// (e.g. model::nbody(64): 18 663 u variables). One system per lane; the code is a fixed set of
// per-node-kind device functions (one per elementary function, the "plugin" layer of the reference,
// include/heyoka/func.hpp:117-147) interpreted over node tables stored in the module; control flow is
// wave-uniform (all lanes process the same node), so table reads are scalar loads. The tape of
// normalised derivatives lives in HBM in one tile per resident wave, laid out
// tile[(u * (order + 1) + k) * 64 + lane]: one 512-byte line per wave access, and all the orders of a u
// variable are contiguous (10.5 KB), so that a convolution walks two short contiguous runs (few pages,
// L2-friendly) instead of striding over the whole allocation. Persistent blocks walk the ensemble with a
// grid stride. Sums are accumulated as running sums in increasing j (the reference's
// compact-mode order, e.g. src/math/prod.cpp:686-698).
//
// This mode is HBM-bound by construction (every convolution term reads two tape entries); it exists for
// coverage (BASELINE config 5), not for speed.
#include <sstream>

#include "hip_emit_detail.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace heyoka_amd
{

namespace
{

const char *table_device_code = R"HIP(
// Node tables (one entry per u variable that is not a state variable).
#define K_SUM 0
#define K_PROD 1
#define K_POW 2
#define K_SUB 3
#define K_DIV 4
#define K_SUM_SQ 5
#define K_SIN 6
#define K_COS 7
#define K_EXP 8
#define K_LOG 9
#define K_TIME 10
#define K_NUM_IDENTITY 11
#define K_TAN 12
#define K_TANH 13
#define K_SINH 14
#define K_COSH 15
#define K_ERF 16
#define K_SIGMOID 17
#define K_ASIN 18
#define K_ACOS 19
#define K_ATAN 20
#define K_ASINH 21
#define K_ACOSH 22
#define K_ATANH 23
#define K_ATAN2 24
#define K_KEPE 25
#define K_RELU 26
#define K_RELUP 27
#define K_SELECT 28
#define K_LAND 29
#define K_LOR 30
#define K_REL_EQ 31
#define K_REL_NEQ 32
#define K_REL_LT 33
#define K_REL_GT 34
#define K_REL_LTE 35
#define K_REL_GTE 36
#define K_CUSTOM 37
#define A_UVAR 0
#define A_NUM 1
#define A_PAR 2

struct hy_tctx {
    double *tape;      // this lane's column of the wave's tape tile
    u64 T;             // total number of resident threads
    const double *pars;
    u64 N, s;
    double t_hi;
};

#if defined(HY_STAGED)
// Staged variant (hip_emit_staged.cpp): one system per workgroup, its whole tape in LDS, u-major with an odd row length
// HY_P >= order + 1 (the lanes of a group hit different banks), one row per u variable whose history is read, a dummy
// row for idle lanes and one cell per u variable which is only read at the current order.
__shared__ double hy_lds_tape[HY_TAPE_DOUBLES];
__device__ __forceinline__ double &hy_tp(const hy_tctx &, unsigned k, unsigned u)
{
    // (hy_row_of[u]: the row of a u variable with a history - state variable i has row i -; the specialised code of the
    // groups carries its offsets in registers, this accessor serves the interpreter and the tail of a step.)
    return hy_lds_tape[hy_row_of[u] + k];
}
#else
__device__ __forceinline__ double &hy_tp(const hy_tctx &c, unsigned k, unsigned u)
{
    return c.tape[((u64)u * (HY_ORDER + 1u) + k) * 64u];
}
#endif

__device__ __forceinline__ double hy_numpar(const hy_tctx &c, unsigned a)
{
    return hy_arg_type[a] == A_NUM ? hy_arg_val[a] : c.pars[(u64)hy_arg_idx[a] * c.N + c.s];
}

__device__ double hy_pow_ebs(double base, unsigned e)
{
    // Exponentiation by squaring (src/math/pow.cpp:136-152): iterative form reproducing the association of
    // the recursive definition, b0 * (b1 * (b2 * ... * base_final)).
    if (e == 0u) return 1.0;
    double pend[6];
    int np = 0;
    while (e > 1u) {
        if (e % 2u == 1u) {
            pend[np++] = base;
            e = (e - 1u) / 2u;
        } else {
            e /= 2u;
        }
        base = base * base;
    }
    double r = base;
    for (int i = np - 1; i >= 0; --i) r = pend[i] * r;
    return r;
}

__device__ double hy_pow_eval(double b, double ex)
{
    // get_pow_eval_algo(), src/math/pow.cpp:292-355.
    if (__builtin_isfinite(ex) && ex == trunc(ex) && fabs(ex) <= 16.0) {
        if (ex >= 0.0) return hy_pow_ebs(b, (unsigned)ex);
        return 1.0 / hy_pow_ebs(b, (unsigned)(-ex));
    }
    if (__builtin_isfinite(ex) && ex != trunc(ex)) {
        const double y = 2.0 * ex;
        if (y == trunc(y) && fabs(y) <= 16.0) {
            const double t = sqrt(b);
            if (y >= 0.0) return hy_pow_ebs(t, (unsigned)y);
            return 1.0 / hy_pow_ebs(t, (unsigned)(-y));
        }
    }
    return hy_pow(b, ex);
}

// ---- one device function per elementary function (the reference's taylor_c_diff_func layer) ----
// pairwise_sum() of the reference (src/detail/llvm_helpers.cpp: adjacent terms are added two by two until one value is
// left; a term without a partner moves up unchanged) for up to 8 terms - the arguments of sum() and sum_sq() are added
// pairwise in the reference's compact mode as well (src/math/sum.cpp:355, src/detail/sum_sq.cpp:371-377), and sums are
// split into chunks of at most 8 arguments by the decomposition (src/detail/udf_split.hpp:49-100). Compile-time loop
// bounds: the 8 slots stay in registers.
__device__ double hy_pairwise8(const double *t, unsigned n)
{
    double v[8];
    bool have[8];
#pragma unroll
    for (unsigned i = 0; i < 8u; ++i) {
        have[i] = i < n;
        v[i] = have[i] ? t[i] : 0.0;
    }
#pragma unroll
    for (unsigned w = 8u; w > 1u; w /= 2u) {
#pragma unroll
        for (unsigned i = 0; i < w / 2u; ++i) {
            const double a = v[2u * i], b = v[2u * i + 1u];
            const bool ha = have[2u * i], hb = have[2u * i + 1u];
            v[i] = hb ? a + b : a;
            have[i] = ha;
        }
    }
    return v[0];
}

__device__ double hy_diff_sum(const hy_tctx &c, unsigned a0, unsigned nargs, unsigned k)
{
    if (nargs <= 8u) {
        double t[8];
#pragma unroll
        for (unsigned j = 0; j < 8u; ++j) {
            const unsigned a = a0 + (j < nargs ? j : 0u);
            t[j] = (hy_arg_type[a] == A_UVAR) ? hy_tp(c, k, hy_arg_idx[a]) : (k == 0u ? hy_numpar(c, a) : 0.0);
        }
        return hy_pairwise8(t, nargs);
    }
    // (More than 8 terms do not come out of the decomposition; kept for hand-built programs: left to right.)
    double acc = 0.0;
    for (unsigned j = 0; j < nargs; ++j) {
        const unsigned a = a0 + j;
        const double v = (hy_arg_type[a] == A_UVAR) ? hy_tp(c, k, hy_arg_idx[a]) : (k == 0u ? hy_numpar(c, a) : 0.0);
        acc = (j == 0u) ? v : acc + v;
    }
    return acc;
}

__device__ double hy_diff_sub(const hy_tctx &c, unsigned a0, unsigned k)
{
    const bool v0 = hy_arg_type[a0] == A_UVAR, v1 = hy_arg_type[a0 + 1u] == A_UVAR;
    if (v0 && v1) return hy_tp(c, k, hy_arg_idx[a0]) - hy_tp(c, k, hy_arg_idx[a0 + 1u]);
    if (v0) return k == 0u ? hy_tp(c, 0, hy_arg_idx[a0]) - hy_numpar(c, a0 + 1u) : hy_tp(c, k, hy_arg_idx[a0]);
    if (v1) return k == 0u ? hy_numpar(c, a0) - hy_tp(c, 0, hy_arg_idx[a0 + 1u]) : -hy_tp(c, k, hy_arg_idx[a0 + 1u]);
    return k == 0u ? hy_numpar(c, a0) - hy_numpar(c, a0 + 1u) : 0.0;
}

__device__ double hy_diff_prod(const hy_tctx &c, unsigned a0, unsigned k)
{
    const bool v0 = hy_arg_type[a0] == A_UVAR, v1 = hy_arg_type[a0 + 1u] == A_UVAR;
    if (v0 && v1) {
        const unsigned b = hy_arg_idx[a0], d = hy_arg_idx[a0 + 1u];
        double acc = 0.0;
        for (unsigned j = 0; j <= k; ++j) acc += hy_tp(c, k - j, b) * hy_tp(c, j, d);
        return acc;
    }
    if (!v0 && !v1) {
        if (k != 0u) return 0.0;
        if (hy_arg_type[a0] == A_NUM && hy_arg_val[a0] == -1.0) return -hy_numpar(c, a0 + 1u);
        return hy_numpar(c, a0) * hy_numpar(c, a0 + 1u);
    }
    const unsigned av = v0 ? a0 : a0 + 1u, an = v0 ? a0 + 1u : a0;
    const double x = hy_tp(c, k, hy_arg_idx[av]);
    if (an == a0 && hy_arg_type[a0] == A_NUM && hy_arg_val[a0] == -1.0) return -x;
    return hy_numpar(c, an) * x;
}

__device__ double hy_diff_div(const hy_tctx &c, unsigned a0, unsigned u, unsigned k)
{
    const bool v0 = hy_arg_type[a0] == A_UVAR, v1 = hy_arg_type[a0 + 1u] == A_UVAR;
    if (v1) {
        const unsigned d = hy_arg_idx[a0 + 1u];
        if (k == 0u) return (v0 ? hy_tp(c, 0, hy_arg_idx[a0]) : hy_numpar(c, a0)) / hy_tp(c, 0, d);
        double acc = 0.0;
        for (unsigned j = 1; j <= k; ++j) acc += hy_tp(c, k - j, u) * hy_tp(c, j, d);
        if (v0) return (hy_tp(c, k, hy_arg_idx[a0]) - acc) / hy_tp(c, 0, d);
        return (-acc) / hy_tp(c, 0, d);
    }
    if (v0) return hy_tp(c, k, hy_arg_idx[a0]) / hy_numpar(c, a0 + 1u);
    return k == 0u ? hy_numpar(c, a0) / hy_numpar(c, a0 + 1u) : 0.0;
}

__device__ double hy_diff_sum_sq(const hy_tctx &c, unsigned a0, unsigned nargs, unsigned k)
{
    // Per-argument running sums (src/detail/sum_sq.cpp:330-345), then the pairwise sum over the arguments (:371-377).
    double terms[8];
    double tot = 0.0;
    for (unsigned q = 0; q < nargs; ++q) {
        const unsigned a = a0 + q;
        double term;
        if (hy_arg_type[a] == A_UVAR) {
            const unsigned b = hy_arg_idx[a];
            double acc = 0.0;
            if (k % 2u == 1u) {
                for (unsigned j = 0; j <= (k - 1u) / 2u; ++j) acc += hy_tp(c, k - j, b) * hy_tp(c, j, b);
                term = acc;
            } else {
                for (unsigned j = 0; k > 0u && j <= (k - 2u) / 2u; ++j) acc += hy_tp(c, k - j, b) * hy_tp(c, j, b);
                const double hv = hy_tp(c, k / 2u, b);
                term = (k > 0u) ? (acc + acc) + hv * hv : hv * hv;
            }
        } else {
            const double v = (k == 0u) ? hy_numpar(c, a) : 0.0;
            term = v * v;
        }
        if (nargs <= 8u) {
            // (A select chain instead of a dynamically indexed store: the slots stay in registers.)
#pragma unroll
            for (unsigned i = 0; i < 8u; ++i) {
                terms[i] = (i == q) ? term : terms[i];
            }
        }
        tot = (q == 0u) ? term : tot + term;
    }
    if (nargs <= 8u) {
        tot = hy_pairwise8(terms, nargs);
    }
    return (k % 2u == 1u) ? tot + tot : tot;
}

__device__ double hy_diff_pow(const hy_tctx &c, unsigned a0, unsigned u, unsigned k)
{
    const double ex = hy_arg_val[a0 + 1u];
    if (hy_arg_type[a0] != A_UVAR) return k == 0u ? hy_pow_eval(hy_numpar(c, a0), ex) : 0.0;
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return hy_pow_eval(hy_tp(c, 0, b), ex);
    if (ex == 0.5) {
        // sqrt (src/math/pow.cpp:432-474).
        const double a_0 = hy_tp(c, 0, u);
        double fac = hy_tp(c, k, b), acc = 0.0;
        const unsigned jmax = (k % 2u == 1u) ? (k - 1u) / 2u : (k - 2u) / 2u;
        for (unsigned j = 1; j <= jmax; ++j) acc += hy_tp(c, k - j, u) * hy_tp(c, j, u);
        if (k % 2u == 0u) {
            const double hv = hy_tp(c, k / 2u, u);
            fac = fac - hv * hv;
        }
        if (jmax >= 1u) fac = fac - (acc + acc);
        return fac / (a_0 + a_0);
    }
    if (ex == 2.0) {
        // square (src/math/pow.cpp:395-430).
        double acc = 0.0;
        if (k % 2u == 1u) {
            for (unsigned j = 0; j <= (k - 1u) / 2u; ++j) acc += hy_tp(c, k - j, b) * hy_tp(c, j, b);
            return acc + acc;
        }
        for (unsigned j = 0; j <= (k - 2u) / 2u; ++j) acc += hy_tp(c, k - j, b) * hy_tp(c, j, b);
        const double hv = hy_tp(c, k / 2u, b);
        return (acc + acc) + hv * hv;
    }
    double acc = 0.0;
    for (unsigned j = 0; j < k; ++j) {
        const double sf = (double)k * ex - (double)j * (ex + 1.0);
        acc += sf * (hy_tp(c, k - j, b) * hy_tp(c, j, u));
    }
    return acc / ((double)k * hy_tp(c, 0, b));
}

__device__ double hy_diff_sincos(const hy_tctx &c, unsigned a0, unsigned dep, unsigned k, bool is_sin)
{
    if (hy_arg_type[a0] != A_UVAR) {
        const double v = hy_numpar(c, a0);
        return k == 0u ? (is_sin ? hy_sin(v) : hy_cos(v)) : 0.0;
    }
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return is_sin ? hy_sin(hy_tp(c, 0, b)) : hy_cos(hy_tp(c, 0, b));
    double acc = 0.0;
    for (unsigned j = 1; j <= k; ++j) acc += (double)j * (hy_tp(c, k - j, dep) * hy_tp(c, j, b));
    return acc / (is_sin ? (double)k : -(double)k);
}

__device__ double hy_diff_exp(const hy_tctx &c, unsigned a0, unsigned u, unsigned k)
{
    if (hy_arg_type[a0] != A_UVAR) return k == 0u ? exp(hy_numpar(c, a0)) : 0.0;
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return exp(hy_tp(c, 0, b));
    double acc = 0.0;
    for (unsigned j = 1; j <= k; ++j) acc += (double)j * (hy_tp(c, k - j, u) * hy_tp(c, j, b));
    return acc / (double)k;
}

__device__ double hy_diff_log(const hy_tctx &c, unsigned a0, unsigned u, unsigned k)
{
    if (hy_arg_type[a0] != A_UVAR) return k == 0u ? log(hy_numpar(c, a0)) : 0.0;
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return log(hy_tp(c, 0, b));
    double ret = (double)k * hy_tp(c, k, b);
    if (k > 1u) {
        double acc = 0.0;
        for (unsigned j = 1; j < k; ++j) acc += (double)j * (hy_tp(c, k - j, b) * hy_tp(c, j, u));
        ret = ret - acc;
    }
    return ret / ((double)k * hy_tp(c, 0, b));
}

__device__ double hy_unary0(unsigned kind, double x)
{
    switch (kind) {
        case K_TAN: return hy_tan(x);
        case K_TANH: return hy_tanh(x);
        case K_SINH: return hy_sinh(x);
        case K_COSH: return hy_cosh(x);
        case K_ERF: return hy_erf(x);
        case K_SIGMOID: return 1.0 / (1.0 + exp(-x));
        case K_ASIN: return hy_asin(x);
        case K_ACOS: return hy_acos(x);
        case K_ATAN: return hy_atan(x);
        case K_ASINH: return hy_asinh(x);
        case K_ACOSH: return hy_acosh(x);
        default: return hy_atanh(x);
    }
}

// tan, tanh, sinh, cosh, erf, sigmoid: k a^[k] = sum_{j=1..k} j X^[k-j] b^[j] (see ssa_emitter::node()).
__device__ double hy_diff_fwd(const hy_tctx &c, unsigned kind, unsigned a0, unsigned u, unsigned dep, unsigned k)
{
    if (hy_arg_type[a0] != A_UVAR) return k == 0u ? hy_unary0(kind, hy_numpar(c, a0)) : 0.0;
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return hy_unary0(kind, hy_tp(c, 0, b));
    double acc = 0.0;
    for (unsigned j = 1; j <= k; ++j) {
        double x = hy_tp(c, k - j, dep);
        if (kind == K_SIGMOID) x = hy_tp(c, k - j, u) - x;
        acc += (double)j * (x * hy_tp(c, j, b));
    }
    acc = acc / (double)k;
    if (kind == K_TAN) return hy_tp(c, k, b) + acc;
    if (kind == K_TANH) return hy_tp(c, k, b) - acc;
    if (kind == K_ERF) return 0x1.20dd750429b6dp+0 * acc;
    return acc;
}

// asin, acos, atan, asinh, acosh, atanh: a^[k] = (k b^[k] -+ sum_{j=1..k-1} j c^[k-j] a^[j]) / (k D).
__device__ double hy_diff_inv(const hy_tctx &c, unsigned kind, unsigned a0, unsigned u, unsigned dep, unsigned k)
{
    if (hy_arg_type[a0] != A_UVAR) return k == 0u ? hy_unary0(kind, hy_numpar(c, a0)) : 0.0;
    const unsigned b = hy_arg_idx[a0];
    if (k == 0u) return hy_unary0(kind, hy_tp(c, 0, b));
    const double c0 = hy_tp(c, 0, dep);
    const double D = (kind == K_ACOS) ? -c0 : (kind == K_ATAN ? (c0 + 1.0) : (kind == K_ATANH ? (1.0 - c0) : c0));
    if (k == 1u) return hy_tp(c, 1, b) / D;
    double acc = 0.0;
    for (unsigned j = 1; j < k; ++j) acc += (double)j * (hy_tp(c, k - j, dep) * hy_tp(c, j, u));
    double ret = (double)k * hy_tp(c, k, b);
    ret = (kind == K_ACOS || kind == K_ATANH) ? (ret + acc) : (ret - acc);
    return ret / ((double)k * D);
}

// atan2(b, c) with d = b^2 + c^2 (see ssa_emitter::node(); src/math/atan2.cpp:113-330).
__device__ double hy_diff_atan2(const hy_tctx &c, unsigned a0, unsigned u, unsigned d, unsigned k)
{
    const bool vy = hy_arg_type[a0] == A_UVAR, vx = hy_arg_type[a0 + 1u] == A_UVAR;
    const unsigned iy = hy_arg_idx[a0], ix = hy_arg_idx[a0 + 1u];
    if (k == 0u) return hy_atan2(vy ? hy_tp(c, 0, iy) : hy_numpar(c, a0), vx ? hy_tp(c, 0, ix) : hy_numpar(c, a0 + 1u));
    if (!vy && !vx) return 0.0;
    const double kf = (double)k;
    double dividend;
    if (vy && vx) {
        dividend = kf * (hy_tp(c, 0, ix) * hy_tp(c, k, iy) - hy_tp(c, 0, iy) * hy_tp(c, k, ix));
    } else if (vy) {
        dividend = kf * (hy_numpar(c, a0 + 1u) * hy_tp(c, k, iy));
    } else {
        dividend = -kf * (hy_numpar(c, a0) * hy_tp(c, k, ix));
    }
    double acc = 0.0;
    for (unsigned j = 1; j < k; ++j) {
        const double t3 = hy_tp(c, k - j, d) * hy_tp(c, j, u);
        if (vy && vx) {
            acc += (double)j * ((hy_tp(c, k - j, ix) * hy_tp(c, j, iy) - hy_tp(c, k - j, iy) * hy_tp(c, j, ix)) - t3);
        } else {
            acc += -(double)j * t3;
        }
    }
    if (k > 1u) dividend += acc;
    return dividend / (kf * hy_tp(c, 0, d));
}

// E = kepE(e, M) with the hidden dependencies dc = e cos E, dd = sin E (src/math/kepE.cpp:140-355).
__device__ double hy_diff_kepE(const hy_tctx &c, unsigned a0, unsigned u, unsigned dc, unsigned dd, unsigned k)
{
    const bool ve = hy_arg_type[a0] == A_UVAR, vm = hy_arg_type[a0 + 1u] == A_UVAR;
    const unsigned ie = hy_arg_idx[a0], im = hy_arg_idx[a0 + 1u];
    if (k == 0u) return hy_kepE(ve ? hy_tp(c, 0, ie) : hy_numpar(c, a0), vm ? hy_tp(c, 0, im) : hy_numpar(c, a0 + 1u));
    if (!ve && !vm) return 0.0;
    const double kf = (double)k;
    double dividend;
    if (ve && vm) {
        dividend = kf * (hy_tp(c, k, ie) * hy_tp(c, 0, dd) + hy_tp(c, k, im));
    } else if (ve) {
        dividend = kf * (hy_tp(c, k, ie) * hy_tp(c, 0, dd));
    } else {
        dividend = kf * hy_tp(c, k, im);
    }
    double acc = 0.0;
    for (unsigned j = 1; j < k; ++j) {
        double t = hy_tp(c, k - j, dc) * hy_tp(c, j, u);
        if (ve) t += hy_tp(c, k - j, dd) * hy_tp(c, j, ie);
        acc += (double)j * t;
    }
    if (k > 1u) dividend += acc;
    return dividend / (kf * (1.0 - hy_tp(c, 0, dc)));
}

// Piecewise functions (see ssa_emitter::node(); src/math/{relu,select,relational,logical}.cpp).
__device__ double hy_diff_piecewise(const hy_tctx &c, unsigned kind, unsigned a0, unsigned nargs, unsigned k)
{
    const auto arg0 = [&](unsigned a) { return hy_arg_type[a] == A_UVAR ? hy_tp(c, 0, hy_arg_idx[a]) : hy_numpar(c, a); };
    const auto argk = [&](unsigned a) {
        return hy_arg_type[a] == A_UVAR ? hy_tp(c, k, hy_arg_idx[a]) : (k == 0u ? hy_numpar(c, a) : 0.0);
    };
    if (kind == K_RELU) {
        const double x = argk(a0);
        return (arg0(a0) > 0.0) ? x : hy_arg_val[a0 + 1u] * x;
    }
    if (kind == K_RELUP) return (k == 0u) ? ((arg0(a0) > 0.0) ? 1.0 : hy_arg_val[a0 + 1u]) : 0.0;
    if (kind == K_SELECT) return (arg0(a0) != 0.0) ? argk(a0 + 1u) : argk(a0 + 2u);
    if (k != 0u) return 0.0;
    bool r;
    if (kind == K_LAND || kind == K_LOR) {
        r = (kind == K_LAND);
        for (unsigned j = 0; j < nargs; ++j) {
            const bool t = arg0(a0 + j) != 0.0;
            r = (kind == K_LAND) ? (r & t) : (r | t);
        }
    } else {
        const double x = arg0(a0), y = arg0(a0 + 1u);
        switch (kind) {
            case K_REL_EQ: r = x == y; break;
            case K_REL_NEQ: r = (x < y) | (x > y); break;
            case K_REL_LT: r = x < y; break;
            case K_REL_GT: r = x > y; break;
            case K_REL_LTE: r = x <= y; break;
            default: r = x >= y; break;
        }
    }
    return r ? 1.0 : 0.0;
}

#if defined(HY_HAS_CUSTOM)
// Functions defined through node rules (node_rule.hpp): generated per module behind this text (see emit_table()).
__device__ double hy_custom_value(const hy_tctx &c, unsigned i, unsigned a0, unsigned nargs, unsigned u, unsigned k);
#endif

// Order-k coefficient of node i (u variable HY_N_EQ + i).
__device__ double hy_node_value(const hy_tctx &c, unsigned i, unsigned k)
{
    {
        const unsigned u = HY_N_EQ + i;
        const unsigned a0 = hy_arg_off[i], nargs = hy_arg_off[i + 1u] - a0;
        double v;
        switch (hy_kind[i]) {
            case K_SUM: v = hy_diff_sum(c, a0, nargs, k); break;
            case K_PROD: v = hy_diff_prod(c, a0, k); break;
            case K_POW: v = hy_diff_pow(c, a0, u, k); break;
            case K_SUB: v = hy_diff_sub(c, a0, k); break;
            case K_DIV: v = hy_diff_div(c, a0, u, k); break;
            case K_SUM_SQ: v = hy_diff_sum_sq(c, a0, nargs, k); break;
            case K_SIN: v = hy_diff_sincos(c, a0, hy_dep[i], k, true); break;
            case K_COS: v = hy_diff_sincos(c, a0, hy_dep[i], k, false); break;
            case K_EXP: v = hy_diff_exp(c, a0, u, k); break;
            case K_LOG: v = hy_diff_log(c, a0, u, k); break;
            case K_TIME: v = (k == 0u) ? c.t_hi : (k == 1u ? 1.0 : 0.0); break;
            case K_TAN: case K_TANH: case K_SINH: case K_COSH: case K_ERF: case K_SIGMOID:
                v = hy_diff_fwd(c, hy_kind[i], a0, u, hy_dep[i], k); break;
            case K_ASIN: case K_ACOS: case K_ATAN: case K_ASINH: case K_ACOSH: case K_ATANH:
                v = hy_diff_inv(c, hy_kind[i], a0, u, hy_dep[i], k); break;
            case K_ATAN2: v = hy_diff_atan2(c, a0, u, hy_dep[i], k); break;
            case K_KEPE: v = hy_diff_kepE(c, a0, u, hy_dep[i], hy_dep2[i], k); break;
            case K_RELU: case K_RELUP: case K_SELECT: case K_LAND: case K_LOR: case K_REL_EQ: case K_REL_NEQ:
            case K_REL_LT: case K_REL_GT: case K_REL_LTE: case K_REL_GTE:
                v = hy_diff_piecewise(c, hy_kind[i], a0, nargs, k); break;
#if defined(HY_HAS_CUSTOM)
            case K_CUSTOM: v = hy_custom_value(c, i, a0, nargs, u, k); break;
#endif
            default: v = (k == 0u) ? hy_numpar(c, a0) : 0.0; break;
        }
        return v;
    }
}

// Order-k coefficient of state variable i (taylor_compute_sv_diff(), src/taylor_02.cpp:245-287).
__device__ double hy_sv_value(const hy_tctx &c, unsigned i, unsigned k)
{
    {
        double v;
        if (hy_sv_type[i] == A_UVAR) {
            v = hy_tp(c, k - 1u, hy_sv_idx[i]) / (double)k;
        } else if (k == 1u) {
            v = hy_sv_type[i] == A_NUM ? hy_sv_val[i] : c.pars[(u64)hy_sv_idx[i] * c.N + c.s];
        } else {
            v = 0.0;
        }
        return v;
    }
}

)HIP";

// The stepper with one system per lane and the tape in HBM.
const char *table_hbm_kernel_code = R"HIP(
// Order-k coefficients of all the u variables that are not state variables / of the state variables.
__device__ void hy_nodes_order(const hy_tctx &c, unsigned k)
{
    for (unsigned i = 0; i < HY_N_NODES; ++i) hy_tp(c, k, HY_N_EQ + i) = hy_node_value(c, i, k);
}
__device__ void hy_sv_order(const hy_tctx &c, unsigned k)
{
    for (unsigned i = 0; i < HY_N_EQ; ++i) hy_tp(c, k, i) = hy_sv_value(c, i, k);
}

// NOTE: latency-bound on the HBM tape: ask for >= 4 waves per SIMD (<= 128 VGPRs).
extern "C" __global__ void __launch_bounds__(256, 4) hy_taylor(const hy_kargs a)
{
    const u64 T = (u64)gridDim.x * 256u;
    const u64 tid = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    hy_tctx c;
    c.tape = a.scratch + (tid >> 6) * ((u64)HY_N_U * (HY_ORDER + 1u) * 64u) + (tid & 63u);
    c.T = T;
    c.pars = a.pars;
    c.N = N;
    for (u64 s = tid; s < N; s += T) {
        c.s = s;
        double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
        hy_df tfin, rem;
        tfin.hi = 0.0; tfin.lo = 0.0; rem.hi = 0.0; rem.lo = 0.0;
        bool t_dir = true;
        double mdt = __builtin_inf();
        double step_lim = 0.0;
        if (a.mode == 1) {
            tfin.hi = (a.tfin_hi != nullptr) ? a.tfin_hi[s] : a.tfin_s_hi;
            tfin.lo = (a.tfin_hi != nullptr) ? a.tfin_lo[s] : a.tfin_s_lo;
            hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
            rem = hy_df_sub(tfin, tcur);
            t_dir = (rem.hi > 0.0) || (rem.hi == 0.0 && rem.lo >= 0.0);
            if (a.lim != nullptr) mdt = a.lim[s];
        } else {
            step_lim = a.lim[s];
        }
        // The order-0 row of the tape doubles as the current state.
        for (unsigned i = 0; i < HY_N_EQ; ++i) hy_tp(c, 0, i) = a.state[(u64)i * N + s];
        u64 n_steps = 0, iter = 0;
        double min_h = __builtin_inf(), max_h = 0.0, last_h = 0.0;
        i64 outcome = HY_OC_SUCCESS;
        for (;;) {
            double lim;
            if (a.mode == 1) {
                hy_df m; m.lo = 0.0;
                // NOTE: selects, not an if/else on the (per-lane) direction: see the note on HY_LIBM1.
                m.hi = t_dir ? mdt : -mdt;
                const bool lt_fwd = hy_df_lt(rem, m), lt_bwd = hy_df_lt(m, rem);
                const bool rem_first = (t_dir & lt_fwd) | (!t_dir & lt_bwd);
                lim = rem_first ? rem.hi : m.hi;
            } else {
                lim = step_lim;
            }
            c.t_hi = t_hi;
            hy_nodes_order(c, 0);
            for (unsigned k = 1; k < HY_ORDER; ++k) {
                hy_sv_order(c, k);
                hy_nodes_order(c, k);
            }
            hy_sv_order(c, HY_ORDER);
#if HY_N_EV > 0
            // The event equations need the order-p coefficients of the u variables as well
            // (src/taylor_02.cpp:1016-1190).
            hy_nodes_order(c, HY_ORDER);
#endif

            // Step size (taylor_determine_h(), compact-mode reduction order, src/taylor_00.cpp:154-167).
            double m0 = fabs(hy_tp(c, 0, 0)), mo = fabs(hy_tp(c, HY_ORDER, 0)), mom1 = fabs(hy_tp(c, HY_ORDER - 1u, 0));
            for (unsigned i = 1; i < HY_N_EQ; ++i) {
                m0 = hy_max(m0, fabs(hy_tp(c, 0, i)));
                mo = hy_max(mo, fabs(hy_tp(c, HY_ORDER, i)));
                mom1 = hy_max(mom1, fabs(hy_tp(c, HY_ORDER - 1u, i)));
            }
#if HY_N_EV > 0
            // The event equations take part in the three norms (taylor_determine_h() iterates up to n_eq + n_sv_funcs,
            // src/taylor_00.cpp:209-219): the step stays inside the convergence radius of their Taylor series too.
            for (unsigned e = 0; e < HY_N_EV; ++e) {
                m0 = hy_max(m0, fabs(hy_tp(c, 0, hy_ev_u[e])));
                mo = hy_max(mo, fabs(hy_tp(c, HY_ORDER, hy_ev_u[e])));
                mom1 = hy_max(mom1, fabs(hy_tp(c, HY_ORDER - 1u, hy_ev_u[e])));
            }
#endif
            const double num_rho = (m0 <= 1.0) ? 1.0 : m0;
            const double rho_o = hy_root(num_rho / mo, 1.0 / (double)HY_ORDER);
            const double rho_om1 = hy_root(num_rho / mom1, 1.0 / (double)(HY_ORDER - 1u));
            const double rho_m = hy_min(rho_o, rho_om1);
            double h = rho_m * HY_RHOFAC;
            h = hy_min(h, fabs(lim));
            h = (lim < 0.0) ? -h : h;

            if (a.tc != nullptr) {
                for (unsigned i = 0; i < HY_N_EQ; ++i)
                    for (unsigned k = 0; k <= HY_ORDER; ++k)
                        a.tc[((u64)i * (HY_ORDER + 1u) + k) * N + s] = hy_tp(c, k, i);
            }

            if (a.mode == 4) {
                // Stepper with events (taylor_add_adaptive_step_with_events(), src/taylor_00.cpp:592-710): jets of the
                // event equations, max |x_i| and the step size; the state is updated later, by the dense-output
                // kernel, at the (possibly truncated) step decided by the event-detection logic.
#if HY_N_EV > 0
                for (unsigned e = 0; e < HY_N_EV; ++e)
                    for (unsigned k = 0; k <= HY_ORDER; ++k)
                        a.ev_tc[((u64)e * (HY_ORDER + 1u) + k) * N + s] = hy_tp(c, k, hy_ev_u[e]);
#endif
                a.max_abs_state[s] = m0;
                last_h = h;
                break;
            }

            bool nf = false;
            for (unsigned i = 0; i < HY_N_EQ; ++i) {
                double res;
#if HY_HIGH_ACCURACY
                res = hy_tp(c, 0, i);
                double comp = 0.0, cur_h = h;
                for (unsigned k = 1; k <= HY_ORDER; ++k) {
                    const double tmp = hy_tp(c, k, i) * cur_h;
                    const double y = tmp - comp;
                    const double t = res + y;
                    comp = (t - res) - y;
                    res = t;
                    cur_h = cur_h * h;
                }
#else
                res = hy_tp(c, HY_ORDER, i);
                for (unsigned k = 1; k <= HY_ORDER; ++k) res = hy_tp(c, HY_ORDER - k, i) + res * h;
#endif
                hy_tp(c, 0, i) = res;
                nf = nf | !hy_finite(res);
            }
            {
                hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
                hy_df hh; hh.hi = h; hh.lo = 0.0;
                const hy_df nt = hy_df_add(tcur, hh);
                t_hi = nt.hi; t_lo = nt.lo;
            }
            last_h = h;
            nf = nf | !(hy_finite(t_hi) & hy_finite(t_lo));
            HY_STEP_TAIL(nf, true)
        }
        if (a.mode == 4) {
            a.last_h[s] = last_h;
            continue;
        }
        for (unsigned i = 0; i < HY_N_EQ; ++i) a.state[(u64)i * N + s] = hy_tp(c, 0, i);
        if (a.mode != 2) {
            a.time_hi[s] = t_hi;
            a.time_lo[s] = t_lo;
        } else {
            const_cast<double *>(a.lim)[s] = last_h;
        }
        a.last_h[s] = last_h;
        a.outcome[s] = outcome;
        if (a.mode == 1) {
            a.min_h[s] = min_h;
            a.max_h[s] = max_h;
            a.n_steps[s] = n_steps;
        }
    }
}
)HIP";

int kind_id(func_kind k)
{
    switch (k) {
        case func_kind::sum:
            return 0;
        case func_kind::prod:
            return 1;
        case func_kind::pow:
            return 2;
        case func_kind::sub:
            return 3;
        case func_kind::div:
            return 4;
        case func_kind::sum_sq:
            return 5;
        case func_kind::sin:
            return 6;
        case func_kind::cos:
            return 7;
        case func_kind::exp:
            return 8;
        case func_kind::log:
            return 9;
        case func_kind::time:
            return 10;
        case func_kind::tan:
            return 12;
        case func_kind::tanh:
            return 13;
        case func_kind::sinh:
            return 14;
        case func_kind::cosh:
            return 15;
        case func_kind::erf:
            return 16;
        case func_kind::sigmoid:
            return 17;
        case func_kind::asin:
            return 18;
        case func_kind::acos:
            return 19;
        case func_kind::atan:
            return 20;
        case func_kind::asinh:
            return 21;
        case func_kind::acosh:
            return 22;
        case func_kind::atanh:
            return 23;
        case func_kind::atan2:
            return 24;
        case func_kind::kepE:
            return 25;
        case func_kind::relu:
            return 26;
        case func_kind::relup:
            return 27;
        case func_kind::select:
            return 28;
        case func_kind::logical_and:
            return 29;
        case func_kind::logical_or:
            return 30;
        case func_kind::rel_eq:
            return 31;
        case func_kind::rel_neq:
            return 32;
        case func_kind::rel_lt:
            return 33;
        case func_kind::rel_gt:
            return 34;
        case func_kind::rel_lte:
            return 35;
        case func_kind::rel_gte:
            return 36;
        case func_kind::custom:
            return 37;
        default:
            return 11;
    }
}

} // namespace

namespace table_detail
{

int kind_id(func_kind k)
{
    return heyoka_amd::kind_id(k);
}

// Everything the rule functions need in front of a stepper kernel: size macros, node tables, the rule functions
// themselves and the dispatcher of the functions defined through node rules. `pre_defs` goes in front of the rule
// functions (variant macros); `stride` is the distance between two orders of a u variable on the tape, in doubles.
void emit_tables_and_rules(std::ostream &src, const taylor_program &p, const emit_options &opts, const std::string &pre_defs,
                           const char *stride)
{
    src << "#define HY_N_EQ " << p.n_eq << "u\n#define HY_N_U " << p.n_u << "u\n#define HY_N_NODES " << p.nodes.size()
        << "u\n#define HY_ORDER " << opts.order << "u\n#define HY_HIGH_ACCURACY " << (opts.high_accuracy ? 1 : 0)
        << "\n#define HY_RHOFAC " << fp_literal(emit_detail::rhofac(opts.order)) << "\n#define HY_N_EV " << p.ev_u.size()
        << "\n";
    src << pre_defs;
    src << "__device__ const unsigned hy_ev_u[] = {";
    for (const auto u : p.ev_u) {
        src << u << ",";
    }
    src << "0};\n";

    std::ostringstream kind, off, at, ai, av, dep, dep2;
    std::size_t n_args = 0;
    off << "0,";
    for (const auto &n : p.nodes) {
        kind << kind_id(n.kind) << ",";
        if (n.kind == func_kind::prod && n.args.size() != 2u) {
            throw std::invalid_argument("The Taylor derivative of a product can be computed only for products of 2 "
                                        "terms");
        }
        if (n.kind == func_kind::pow && n.args[1].type != operand::kind::num) {
            throw std::invalid_argument("An invalid argument type was encountered while trying to build the Taylor "
                                        "derivative of a pow()");
        }
        for (const auto &o : n.args) {
            at << static_cast<int>(o.type) << ",";
            ai << o.idx << ",";
            av << fp_literal(o.type == operand::kind::num ? o.value : 0.) << ",";
            ++n_args;
        }
        off << n_args << ",";
        dep << (n.deps.empty() ? 0u : n.deps[0]) << ",";
        dep2 << (n.deps.size() < 2u ? 0u : n.deps[1]) << ",";
    }
    src << "__device__ const unsigned char hy_kind[] = {" << kind.str() << "0};\n";
    src << "__device__ const unsigned hy_arg_off[] = {" << off.str() << "0};\n";
    src << "__device__ const unsigned char hy_arg_type[] = {" << at.str() << "0};\n";
    src << "__device__ const unsigned hy_arg_idx[] = {" << ai.str() << "0};\n";
    src << "__device__ const double hy_arg_val[] = {" << av.str() << "0.0};\n";
    src << "__device__ const unsigned hy_dep[] = {" << dep.str() << "0};\n";
    src << "__device__ const unsigned hy_dep2[] = {" << dep2.str() << "0};\n";
    src << "__device__ const unsigned char hy_sv_type[] = {";
    for (const auto &d : p.sv_defs) {
        src << static_cast<int>(d.type) << ",";
    }
    src << "0};\n__device__ const unsigned hy_sv_idx[] = {";
    for (const auto &d : p.sv_defs) {
        src << d.idx << ",";
    }
    src << "0};\n__device__ const double hy_sv_val[] = {";
    for (const auto &d : p.sv_defs) {
        src << fp_literal(d.type == operand::kind::num ? d.value : 0.) << ",";
    }
    src << "0.0};\n";
    // Functions defined through node rules: their hidden dependencies (any number), the rule of every node, and the
    // dispatcher which hands the rule its jets as strided views of the tape.
    bool has_custom = false;
    for (const auto &n : p.nodes) {
        has_custom = has_custom || n.kind == func_kind::custom;
    }
    if (!has_custom) {
        src << table_device_code;
        return;
    }
    std::ostringstream doff, dlist, rof;
    std::size_t nd = 0;
    doff << "0,";
    std::vector<std::uint32_t> used;
    for (const auto &n : p.nodes) {
        if (n.kind == func_kind::custom) {
            for (const auto d : n.deps) {
                dlist << d << ",";
                ++nd;
            }
            if (std::find(used.begin(), used.end(), n.rule) == used.end()) {
                used.push_back(n.rule);
            }
        }
        doff << nd << ",";
        rof << n.rule << ",";
    }
    src << "#define HY_HAS_CUSTOM 1\n";
    src << "__device__ const unsigned hy_cdep_off[] = {" << doff.str() << "0};\n";
    src << "__device__ const unsigned hy_cdep[] = {" << dlist.str() << "0};\n";
    src << "__device__ const unsigned hy_rule_of[] = {" << rof.str() << "0};\n";
    src << table_device_code;
    src << "__device__ double hy_custom_value(const hy_tctx &c, unsigned i, unsigned a0, unsigned nargs, unsigned u, "
           "unsigned k)\n{\n";
    src << "double xv[8];\nhy_jet xj[8], hj[8];\n";
    src << "for (unsigned a = 0; a < nargs && a < 8u; ++a) {\n"
           "    if (hy_arg_type[a0 + a] == A_UVAR) {\n"
           "        xj[a].p = &hy_tp(c, 0, hy_arg_idx[a0 + a]); xj[a].s = " << stride << "; xj[a].n = k + 1u; xv[a] = xj[a].p[0];\n"
           "    } else {\n"
           "        xv[a] = hy_numpar(c, a0 + a); xj[a].p = &xv[a]; xj[a].s = 1u; xj[a].n = 1u;\n"
           "    }\n}\n";
    src << "const unsigned d0 = hy_cdep_off[i], ndep = hy_cdep_off[i + 1u] - d0;\n";
    src << "for (unsigned j = 0; j < ndep && j < 8u; ++j) { hj[j].p = &hy_tp(c, 0, hy_cdep[d0 + j]); hj[j].s = " << stride
        << "; hj[j].n = k; }\n";
    src << "hy_jet self; self.p = &hy_tp(c, 0, u); self.s = " << stride << "; self.n = k;\n";
    src << "switch (hy_rule_of[i]) {\n";
    for (const auto id : used) {
        const auto &nm = get_node_rule(id).name;
        src << "case " << id << "u: return (k == 0u) ? hy_rule_" << nm << "_value(";
        for (std::uint32_t q = 0; q < get_node_rule(id).n_args; ++q) {
            src << (q == 0u ? "" : ", ") << "xv[" << q << "]";
        }
        src << ") : hy_rule_" << nm << "_orderk(k, self, xj, hj);\n";
    }
    src << "default: return 0.0;\n}\n}\n";
}

} // namespace table_detail

emitted_module emit_staged(const taylor_program &, const emit_options &, std::string &why_not);

emitted_module emit_table(const taylor_program &p, const emit_options &opts)
{
    // Variant: the staged stepper (hip_emit_staged.cpp: one system per workgroup, the tape in LDS, code specialised per
    // group of nodes) whenever the tape of one system fits in LDS; otherwise one system per lane with the tape in HBM.
    // HEYOKA_AMD_TABLE_LDS=0 forces the latter.
    std::string why_staged = "switched off (HEYOKA_AMD_TABLE_LDS=0)";
    if (opts.dev.table_lds != 0) {
        auto m = emit_staged(p, opts, why_staged);
        if (!m.source.empty()) {
            return m;
        }
    }

    std::ostringstream src;
    src << emit_detail::prelude << emit_detail::rules_source(p);
    emit_detail::emit_dout(src, p, opts);
    table_detail::emit_tables_and_rules(src, p, opts, "", "64u");
    src << table_hbm_kernel_code;

    emitted_module ret;
    ret.source = src.str();
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.mode = emit_mode::table;
    ret.persistent = true;
    ret.tc_optional = true;
    ret.block_size = 256;
    ret.lanes_per_system = 1;
    // One tape tile per resident wave: n_u * (order + 1) rows of 64 doubles.
    ret.scratch_per_wave = static_cast<std::uint64_t>(p.n_u) * (opts.order + 1u) * 64u;
    ret.notes = "table mode: " + std::to_string(p.nodes.size()) + " nodes interpreted from tables, tape in HBM (staged stepper: "
                + why_staged + ")";
    return ret;
}

} // namespace heyoka_amd
