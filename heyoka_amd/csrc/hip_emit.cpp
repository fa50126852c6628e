// HIP source generation. See hip_emit.hpp.
//
// The arithmetic emitted for each node follows the "default mode" formulas of the reference
// (products first, then a pairwise sum; citations next to each rule) so that results agree with
// heyoka's CPU path up to FMA contraction and the <= 1 ulp elementary functions.
#include "hip_emit.hpp"
#include "hip_emit_detail.hpp"
#include "logging.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

namespace heyoka_amd
{

std::string fp_literal(double x)
{
    if (std::isnan(x)) {
        return "__builtin_nan(\"\")";
    }
    if (std::isinf(x)) {
        return x > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
    }
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%a", x);
    std::string s(buf);
    if (x < 0 || (x == 0 && std::signbit(x))) {
        return "(" + s + ")";
    }
    return s;
}

namespace emit_detail
{

const char *const prelude = R"HIP(
typedef unsigned long long u64;
typedef long long i64;

struct hy_kargs {
    double *state;
    const double *pars;
    double *time_hi;
    double *time_lo;
    const double *lim;
    const double *tfin_hi;
    const double *tfin_lo;
    double *last_h;
    i64 *outcome;
    double *min_h;
    double *max_h;
    u64 *n_steps;
    double *tc;
    u64 N;
    u64 max_steps;
    int mode;
    int pad;
    unsigned int *counters;
    double *scratch;
    double tfin_s_hi, tfin_s_lo;
    double *ev_tc;
    double *max_abs_state;
    double *sel_norms;
    const double *tc_thr;
    double *grid_done;
};

#define HY_OC_SUCCESS (-4294967296LL - 1)
#define HY_OC_STEP_LIMIT (-4294967296LL - 2)
#define HY_OC_TIME_LIMIT (-4294967296LL - 3)
#define HY_OC_ERR_NF_STATE (-4294967296LL - 4)

// Double-length (hi, lo) arithmetic: Knuth / Dekker error-free transformations
// (reference: include/heyoka/detail/dfloat.hpp:109-164). No multiplications are involved, hence
// FMA contraction cannot alter these sequences; reassociation is never enabled.
struct hy_df {
    double hi, lo;
};

__device__ __forceinline__ hy_df hy_df_add(hy_df a, hy_df b)
{
    const double x_hi = a.hi + b.hi;
    const double z_hi = x_hi - a.hi;
    const double y_hi = (a.hi - (x_hi - z_hi)) + (b.hi - z_hi);
    const double x_lo = a.lo + b.lo;
    const double z_lo = x_lo - a.lo;
    const double y_lo = (a.lo - (x_lo - z_lo)) + (b.lo - z_lo);
    const double w = y_hi + x_lo;
    const double u = x_hi + w;
    const double v = (x_hi - u) + w;
    const double w2 = v + y_lo;
    hy_df r;
    r.hi = u + w2;
    r.lo = (u - r.hi) + w2;
    return r;
}

__device__ __forceinline__ hy_df hy_df_sub(hy_df a, hy_df b)
{
    hy_df nb;
    nb.hi = -b.hi;
    nb.lo = -b.lo;
    return hy_df_add(a, nb);
}

__device__ __forceinline__ bool hy_df_lt(hy_df x, hy_df y)
{
    // NOTE: non-short-circuit operators on purpose (three compares and two mask operations, no control flow).
    return (x.hi < y.hi) | ((x.hi == y.hi) & (x.lo < y.lo));
}

// End-of-step bookkeeping shared by all the steppers (reference: src/taylor_adaptive_batch.cpp:702-727 for the
// outcome of a step, :1402-1460 for the propagate loop): non-finite state check, outcome, step counters, min/max
// step size, remaining time, step limit. Written with selects and ONE (divergent) loop exit instead of a chain of
// early breaks: see the note on HY_LIBM1 about divergent control flow in kernels under register pressure.
// NF: the non-finite flag; COUNT: which thread reports it in the error counter.
#define HY_STEP_TAIL(NF, COUNT)                                                                                        \
    {                                                                                                                  \
        const bool hy_nf = (NF);                                                                                       \
        outcome = hy_nf ? HY_OC_ERR_NF_STATE : ((h == lim) ? HY_OC_TIME_LIMIT : HY_OC_SUCCESS);                       \
        if (hy_nf & (COUNT)) atomicAdd(a.counters, 1u);                                                                \
        bool hy_done = hy_nf | (a.mode != 1);                                                                          \
        n_steps += (!hy_done & (h != 0.0)) ? 1u : 0u;                                                                  \
        const bool hy_upd = !hy_done & (outcome == HY_OC_SUCCESS);                                                     \
        const double hy_ah = fabs(h);                                                                                  \
        min_h = hy_upd ? hy_min(min_h, hy_ah) : min_h;                                                                 \
        max_h = hy_upd ? hy_max(max_h, hy_ah) : max_h;                                                                 \
        hy_done |= (h == rem.hi);                                                                                      \
        {                                                                                                              \
            hy_df hy_tcur;                                                                                             \
            hy_tcur.hi = t_hi;                                                                                         \
            hy_tcur.lo = t_lo;                                                                                         \
            rem = hy_df_sub(tfin, hy_tcur);                                                                            \
        }                                                                                                              \
        ++iter;                                                                                                        \
        const bool hy_sl = !hy_done & (iter == a.max_steps);                                                           \
        outcome = hy_sl ? HY_OC_STEP_LIMIT : outcome;                                                                  \
        hy_done |= hy_sl;                                                                                              \
        if (hy_done) break;                                                                                            \
    }

// x^c for x >= 0, 0 < c < 1: the (1/p)-th roots of the step-size selector (src/taylor_00.cpp:242-252, llvm.pow in
// the reference). exp(log(x) * c) has the same limits (0 -> 0, +inf -> +inf, nan -> nan) and agrees with pow() to
// a few ulps (the error of log(x) is scaled by c < 1), for ~160 instructions less per step than the general-purpose
// device pow().
__device__ __forceinline__ double hy_root(double x, double c)
{
    return exp(log(x) * c);
}

// Logarithm of the step-size selector: written out (45 VALU instructions) instead of the device library's log() (95:
// double-double arithmetic for < 1 ulp) - the selector takes the logarithms of max(1, |x|_inf), |x^[p]|_inf and
// |x^[p-1]|_inf at every step (no quotients: log(num / m) = log(num) - log(m), with the same limits 0 -> +inf,
// inf -> -inf, inf - inf -> nan), which was a sixth of the serial tail of a step of the cluster kernels (there the three
// arguments sit on three lanes of a quad and share ONE evaluation) and 8 % of the two-body stepper. frexp, m in
// [sqrt(1/2), sqrt(2)), z = (m - 1) / (m + 1) by reciprocal + two Newton steps + one residual correction,
// log m = 2 z + z^3 P(z^2) with the Taylor coefficients 2 / (2 n + 1) up to z^21 (|z| <= 0.1716: the first neglected term
// is 2e-17 relative), e ln 2 added in two pieces. Error < 2 ulp (checked against logl on 2e7 arguments); the roots of
// the selector scale it by 1 / p. 0 -> -inf, +inf -> +inf, nan -> nan like log().
__device__ __forceinline__ double hy_sel_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0x1.6a09e667f3bcdp-1;
    m = m * (lo ? 2.0 : 1.0);
    e -= lo ? 1 : 0;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    double z = num * r;
    z = __builtin_fma(__builtin_fma(-den, z, num), r, z);
    const double w = z * z;
    double p = 0x1.8618618618618p-4;
    p = __builtin_fma(p, w, 0x1.af286bca1af28p-4);
    p = __builtin_fma(p, w, 0x1.e1e1e1e1e1e1ep-4);
    p = __builtin_fma(p, w, 0x1.1111111111111p-3);
    p = __builtin_fma(p, w, 0x1.3b13b13b13b14p-3);
    p = __builtin_fma(p, w, 0x1.745d1745d1746p-3);
    p = __builtin_fma(p, w, 0x1.c71c71c71c71cp-3);
    p = __builtin_fma(p, w, 0x1.2492492492492p-2);
    p = __builtin_fma(p, w, 0x1.999999999999ap-2);
    p = __builtin_fma(p, w, 0x1.5555555555555p-1);
    const double ed = (double)e;
    double res = __builtin_fma(ed, 0x1.abc9e3b39803fp-56, (z * w) * p);
    res = __builtin_fma(2.0, z, res);
    res = __builtin_fma(ed, 0x1.62e42fefa39efp-1, res);
    res = (x == 0.0) ? -__builtin_inf() : res;
    res = (x == __builtin_inf()) ? x : res;
    return res;
}

// Out-of-line calls for the math-library functions with data-dependent control flow in their device implementations
// (the Payne-Hanek branch of sin/cos/tan, the piecewise ranges of erf, ...). Inlined into a straight-line kernel with
// several hundred live registers, those divergent if/else regions are where the register allocator splits long live
// ranges, and with this toolchain (ROCm 7.2) such a split can land in the exec-masked flow block between the two sides
// of the branch: the copy is then executed only by the lanes which took the first side and the other lanes read
// stale registers later on (observed as wild jet addresses in a 3500-statement kernel with tan()). As functions of
// their own, the branches live in a small frame without register pressure and the kernel sees an ordinary call at
// its current exec mask. One call per function instance and step (order 0 only): the cost is not measurable.
#define HY_LIBM1(f)                                                                                                    \
    static __device__ __attribute__((noinline)) double hy_##f(double x)                                                \
    {                                                                                                                  \
        return f(x);                                                                                                   \
    }
HY_LIBM1(sin)
HY_LIBM1(cos)
HY_LIBM1(tan)
HY_LIBM1(tanh)
HY_LIBM1(sinh)
HY_LIBM1(cosh)
HY_LIBM1(erf)
HY_LIBM1(asin)
HY_LIBM1(acos)
HY_LIBM1(atan)
HY_LIBM1(asinh)
HY_LIBM1(acosh)
HY_LIBM1(atanh)
#undef HY_LIBM1
static __device__ __attribute__((noinline)) double hy_pow(double x, double y)
{
    return pow(x, y);
}

static __device__ __attribute__((noinline)) double hy_atan2(double y, double x)
{
    return atan2(y, x);
}

// Inverse of Kepler's equation for the eccentric anomaly: E - e sin E = M (reference: llvm_add_inv_kep_E(),
// src/detail/llvm_helpers_celmec.cpp:181-466). Invalid eccentricities (nan, < 0, >= 1) give nan; M is reduced to
// [0, 2 pi) in double-length arithmetic (llvm_trig_arg_reduce(), :140-177, on top of the Dekker / NTL procedures of
// src/detail/llvm_helpers_dl.cpp:130-281); third-order initial guess in e; Newton-Raphson iterations safeguarded by
// bisection on a bracketing interval, stopped when |f(E)| or the bracket are below 4 eps, nan after 20 iterations
// without convergence. One lane per system: every lane stops at its own convergence (the scalar flavour of the
// reference; its batch flavour keeps iterating all the lanes until the slowest has converged, which moves the
// others by rounding errors at most).
static __device__ __attribute__((noinline)) double hy_kepE(double ecc_in, double M_in)
{
#pragma clang fp contract(off)
    const double qnan = __builtin_nan("");
    const bool ecc_invalid = !(ecc_in >= 0.0) | (ecc_in >= 1.0);
    const double ecc = ecc_invalid ? qnan : ecc_in;

    // ---- M mod 2 pi: x - y * floor(x / y) in double-length arithmetic.
    const double y_hi = 0x1.921fb54442d18p+2, y_lo = 0x1.1a62633145c07p-52;
    const double twopi_prev = 0x1.921fb54442d17p+2; // the double preceding 2 pi
    double M;
    {
        // x / y (div2()).
        const double c = M_in / y_hi;
        const double u = c * y_hi, uu = fma(c, y_hi, -u);
        double cc = M_in - u;
        cc = cc - uu;
        cc = cc + 0.0;
        cc = cc - c * y_lo;
        cc = cc / y_hi;
        const double q_hi = c + cc, q_lo = (c - q_hi) + cc;
        // floor().
        const double fhi = floor(q_hi);
        const double flo = (fhi == q_hi) ? floor(q_lo) : 0.0;
        const double fl_hi = fhi + flo, fl_lo = (fhi - fl_hi) + flo;
        // y * floor (mul2()).
        const double pc = y_hi * fl_hi;
        double pcc = fma(y_hi, fl_hi, -pc);
        pcc = (y_hi * fl_lo + y_lo * fl_hi) + pcc;
        const double p_hi = pc + pcc, p_lo = (pc - p_hi) + pcc;
        // x - y * floor.
        hy_df xx, yy;
        xx.hi = M_in;
        xx.lo = 0.0;
        yy.hi = -p_hi;
        yy.lo = -p_lo;
        M = hy_df_add(xx, yy).hi;
        M = (M < 0.0) ? 0.0 : M;
        M = (twopi_prev < M) ? twopi_prev : M;
    }

    // ---- Initial guess: E = M + e sin M + e^2 sin M cos M + e^3 sin M (3/2 cos^2 M - 1/2).
    double sE = sin(M), cE = cos(M);
    double E;
    {
        const double e_sin = ecc * sE, e_cos = ecc * cE, e2 = ecc * ecc, cos2 = cE * cE;
        const double ig1 = (M + e_sin) + e_sin * e_cos;
        const double ig2 = (e2 * e_sin) * (1.5 * cos2 - 0.5);
        E = ig1 + ig2;
    }
    double lb = 0.0, ub = twopi_prev;
    E = (E < lb) ? lb : E;
    E = (ub < E) ? ub : E;
    sE = sin(E);
    cE = cos(E);
    double fE = (E - M) - ecc * sE;

    const double tol = 4.0 * 0x1p-52;
    bool not_converged = false;
    unsigned it = 0;
    for (;;) {
        // Bracket update from the sign of f(E) (0 for nan: the bracket collapses and the loop ends).
        const int sgn = (0.0 < fE) - (fE < 0.0);
        const double n_ub = (sgn >= 0) ? E : ub, n_lb = (sgn <= 0) ? E : lb;
        ub = n_ub;
        lb = n_lb;
        not_converged = (fabs(fE) > tol) & ((ub - lb) > tol);
        if (!(it < 20u) | !not_converged) {
            break;
        }
        // Newton-Raphson step, replaced by a bisection when it leaves the bracket.
        double nE = E - fE / (1.0 - ecc * cE);
        nE = (nE > ub) ? 0.5 * (E + ub) : nE;
        nE = (nE < lb) ? 0.5 * (E + lb) : nE;
        E = nE;
        sE = sin(E);
        cE = cos(E);
        fE = (E - M) - ecc * sE;
        ++it;
    }
    return (it == 20u && not_converged) ? qnan : E;
}

// max(a, b) = (a < b) ? b : a and min(a, b) = (b < a) ? b : a
// (reference: src/detail/llvm_helpers_cmp.cpp:315-329; NaN handling is part of the semantics).
__device__ __forceinline__ double hy_max(double a, double b)
{
    return (a < b) ? b : a;
}
__device__ __forceinline__ double hy_min(double a, double b)
{
    return (b < a) ? b : a;
}

__device__ __forceinline__ bool hy_finite(double x)
{
    return __builtin_isfinite(x);
}
)HIP";

// Scaling + safety factor of the step-size selector, folded on the host in double precision
// (reference: taylor_determine_h_rhofac(), src/taylor_00.cpp:84-94).
double rhofac(std::uint32_t order)
{
    const double m7_10 = -7. / 10.;
    const double e2 = std::exp(1.) * std::exp(1.);
    return std::exp(m7_10 / static_cast<double>(order - 1u)) / e2;
}

// Emit the dense-output kernel (reference: taylor_add_d_out_function(), src/taylor_01.cpp:1015-1185).
void emit_dout(std::ostringstream &os, const taylor_program &p, const emit_options &opts)
{
    const auto n_eq = p.n_eq;
    const auto order = opts.order;
    os << "extern \"C\" __global__ void __launch_bounds__(256) hy_dout(double *out, const double *tc, const double "
          "*hs, u64 N)\n{\n";
    os << "const u64 s = (u64)blockIdx.x * 256u + threadIdx.x;\nif (s >= N) return;\n";
    os << "const double h = hs[s];\n";
    os << "for (unsigned i = 0; i < " << n_eq << "u; ++i) {\n";
    os << "const double *c = tc + ((u64)i * " << (order + 1u) << "u) * N + s;\n";
    if (opts.high_accuracy) {
        // Compensated summation (reference: src/taylor_01.cpp:1074-1134).
        os << "double res = c[0], comp = 0.0, cur_h = h;\n";
        os << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
        os << "const double tmp = c[(u64)k * N] * cur_h;\nconst double y = tmp - comp;\nconst double t = res + y;\n";
        os << "comp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
    } else {
        os << "double res = c[(u64)" << order << "u * N];\n";
        os << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
        os << "res = c[(u64)(" << order << "u - k) * N] + res * h;\n}\n";
    }
    os << "out[(u64)i * N + s] = res;\n}\n}\n";
}

// One system per lane, fully unrolled SSA code (the GPU analogue of the reference's default mode,
// taylor_compute_jet() src/taylor_02.cpp:1339-1418).
// One stepper kernel. reg_jets = false: the jets of the state variables go through the tc buffer (always
// needed, also serves kw::write_tc). reg_jets = true: they stay in SSA values (registers); state variables
// defined by other state variables (x' = v) are re-derived in the final evaluation instead of being stored.
// stream_tc (with reg_jets): the Taylor coefficients are additionally streamed to a.tc as they are produced (stores
// only, nothing is read back): the write_tc / continuous-output / propagate_grid variant of a register-resident stepper.
// The u variables whose coefficient HISTORY is read (operands of convolutions and of recurrences, functions with a
// recurrence on themselves): what a lane of the straight-line stepper keeps through the orders.
// (folded: with ssa_emitter::fold_scaled / fold_zeros a history which is a scaled copy of another one - prod(number, u) -, or
// its coefficients beyond order 0 - u + number, u - number -, is not kept: the mark moves to u.)
std::vector<char> history_operands(const taylor_program &p, bool folded = false)
{
    std::vector<char> hist(p.n_u, 0);
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto &nd = p.nodes[i];
        const auto &a = nd.args;
        const auto mark = [&](const operand &o) {
            if (o.type == operand::kind::uvar) {
                hist[o.idx] = 1;
            }
        };
        switch (nd.kind) {
            case func_kind::prod:
                if (a.size() == 2u && a[0].type == operand::kind::uvar && a[1].type == operand::kind::uvar) {
                    mark(a[0]);
                    mark(a[1]);
                }
                break;
            case func_kind::sum_sq:
                for (const auto &o : a) {
                    mark(o);
                }
                break;
            case func_kind::pow:
            case func_kind::exp:
            case func_kind::log:
            case func_kind::sin:
            case func_kind::cos:
                if (a[0].type == operand::kind::uvar) {
                    mark(a[0]);
                    // (The square is a convolution of its argument with itself: no recurrence on its own coefficients.)
                    const bool square = nd.kind == func_kind::pow && a.size() == 2u && a[1].type == operand::kind::num && a[1].value == 2.;
                    if (!(folded && square)) {
                        hist[p.n_eq + i] = 1;
                    }
                }
                break;
            case func_kind::div:
                if (a[1].type == operand::kind::uvar) {
                    mark(a[1]);
                    hist[p.n_eq + i] = 1;
                }
                break;
            default:
                break;
        }
    }
    for (std::uint32_t i = static_cast<std::uint32_t>(p.nodes.size()); folded && i-- > 0u;) {
        const auto &nd = p.nodes[i];
        const auto &a = nd.args;
        if (hist[p.n_eq + i] == 0) {
            continue;
        }
        std::uint32_t n_var = 0, parent = 0;
        for (const auto &o : a) {
            if (o.type == operand::kind::uvar) {
                ++n_var;
                parent = o.idx;
            }
        }
        const bool copy = n_var == 1u
                          && ((nd.kind == func_kind::prod && a.size() == 2u) || nd.kind == func_kind::sum
                              || (nd.kind == func_kind::sub && a[0].type == operand::kind::uvar));
        if (copy) {
            hist[p.n_eq + i] = 0;
            hist[parent] = 1;
        }
    }
    return hist;
}

// (waves: amdgpu_waves_per_eu of the kernel, 0 = the compiler's choice; n_derived: the number of state variables whose
// coefficient histories are not kept because they are re-derived at the end of the step - see "The other way round" below.)
std::string emit_unrolled_kernel(const taylor_program &p, const emit_options &opts, const std::string &kname,
                                 bool reg_jets, std::uint64_t &n_stmt, bool stream_tc = false, int waves = 0,
                                 std::uint32_t *n_derived = nullptr)
{
    const auto n_eq = p.n_eq;
    const auto order = opts.order;
    const auto bs = opts.block_size;

    ssa_emitter e(p, order);
    auto &os = e.os;
    e.enable_pow_rcp(!opts.exact_division);
    e.running_sums = opts.sum_order != 1;
    // Divisions by the (constant) order: one multiplication by RN(1 / k), within 1 ulp of the quotient (like the pair
    // kernels), unless kw::exact_division asks for the correctly rounded 3-operation sequence - 120 of them per step of
    // the two-body problem.
    e.recip_div = !opts.exact_division;
    e.fold_zeros = opts.dev.unrolled_trim;
    e.fold_scaled = opts.dev.unrolled_trim && !opts.exact_division && opts.sum_order == 0;
    e.merge_sum_sq = opts.sum_order == 0 && !opts.exact_division && opts.dev.unrolled_merge_ssq;

    os << "extern \"C\" __global__ void __launch_bounds__(" << bs << ") ";
    if (waves > 0) {
        os << "__attribute__((amdgpu_waves_per_eu(" << waves << ", " << waves << "))) ";
    }
    os << kname << "(const hy_kargs a)\n{\n";
    os << "const u64 s = (u64)blockIdx.x * " << bs << "u + threadIdx.x;\n";
    os << "if (s >= a.N) return;\n";
    os << "const u64 N = a.N;\n";
    os << "double t_hi = a.time_hi[s], t_lo = a.time_lo[s];\n";
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        os << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
    }
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        os << "double x" << i << " = a.state[(u64)" << i << "u * N + s];\n";
        if (reg_jets) {
            os << "double x" << i << "n = 0.0;\n";
        }
    }
    if (stream_tc) {
        os << "double *jet = a.tc + s;\n";
    } else if (!reg_jets) {
        os << "double *jet = a.tc + s;\n";
    }
    os << R"HIP(
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
u64 n_steps = 0, iter = 0;
double min_h = __builtin_inf(), max_h = 0.0, last_h = 0.0;
i64 outcome = HY_OC_SUCCESS;
for (;;) {
// Time limit for this step (reference: src/taylor_adaptive_batch.cpp:1378-1387).
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
)HIP";
    if (stream_tc || !reg_jets) {
        // NOTE: the (order + 1) * n_eq store addresses are invariants of the step loop: left alone, the compiler hoists
        // all of them into registers (2 per address). Laundering the base pointer once per step makes them per-step values.
        // (Round 6: also in the kernel which keeps the jets of the state variables in memory - cr3bp: 526 -> 0 spilled
        // registers together with the folded histories, see ssa_emitter::fold_scaled.)
        os << "asm volatile(\"\" : \"+v\"(jet));\n";
        // (The same for the scalar halves of the addresses, index * N: hoisted, they are 2 SGPRs each - 651 spilled SGPRs in the
        // stepper of two massive bodies.)
        os << "u64 hy_Ns = N;\n#if defined(HY_HOST_EMU)\nasm volatile(\"\" : \"+r\"(hy_Ns));\n#else\nasm volatile(\"\" : "
              "\"+s\"(hy_Ns));\n#endif\n";
    }

    // ---- Jet of normalised derivatives. ----
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        e.val(i, 0) = "x" + std::to_string(i);
    }
    const auto store_sv = [&](std::uint32_t i, std::uint32_t k) {
        if (reg_jets && !stream_tc) {
            return;
        }
        os << "jet[(u64)" << (static_cast<std::uint64_t>(i) * (order + 1u) + k) << "u * hy_Ns] = " << e.val(i, k)
           << ";\n";
    };
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        store_sv(i, 0);
    }
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        e.node(i, 0);
    }
    // x' = v, v' = u with a v which no elementary function reads (register-resident jets, not under kw::exact_division):
    // x^[k] = u^[k-2] RN(1 / (k (k - 1))) - one multiplication instead of the two through v^[k-1], whose only other use is
    // the final evaluation (where it is re-derived from x, see below; the compiler drops the unused ones).
    std::vector<std::int64_t> second_order(n_eq, -1);
    if (reg_jets && opts.dev.unrolled_derive && !opts.exact_division) {
        std::vector<char> read_by_node(n_eq, 0);
        for (const auto &nd : p.nodes) {
            for (const auto &o : nd.args) {
                if (o.type == operand::kind::uvar && o.idx < n_eq) {
                    read_by_node[o.idx] = 1;
                }
            }
        }
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            const auto &d = p.sv_defs[i];
            if (d.type == operand::kind::uvar && d.idx < n_eq && read_by_node[d.idx] == 0) {
                const auto &dd = p.sv_defs[d.idx];
                if (dd.type == operand::kind::uvar && dd.idx >= n_eq) {
                    second_order[i] = dd.idx;
                }
            }
        }
    }
    const auto emit_sv = [&](std::uint32_t i, std::uint32_t k) {
        if (k >= 2u && second_order[i] >= 0) {
            const auto &src = e.val(static_cast<std::uint32_t>(second_order[i]), k - 2u);
            e.val(i, k) = ssa_emitter::is_zero_lit(src)
                              ? std::string("0.0")
                              : e.def(ssa_emitter::mul(src, fp_literal(1. / (static_cast<double>(k) * static_cast<double>(k - 1u)))));
        } else {
            e.sv(i, k);
        }
        store_sv(i, k);
    };
    for (std::uint32_t k = 1; k < order; ++k) {
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            emit_sv(i, k);
        }
        for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
            e.node(i, k);
        }
    }
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        emit_sv(i, order);
    }

    // Integrators with events: every step is a step with events (mode 4), which needs the order-p coefficients of the u
    // variables (src/taylor_02.cpp:1016-1190) - and the event equations take part in the norms of the step-size
    // selector (taylor_determine_h() iterates up to n_eq + n_sv_funcs, src/taylor_00.cpp:209-219), so that the step
    // stays inside the convergence radius of their Taylor series as well.
    const bool ev_norms = !p.ev_u.empty() && !reg_jets;
    if (ev_norms) {
        for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
            e.node(i, order);
        }
    }

    // ---- Step size (reference: taylor_determine_h(), src/taylor_00.cpp:102-273). ----
    const auto max_abs = [&](std::uint32_t k) {
        std::vector<std::string> v;
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            v.push_back(e.def("fabs(" + e.val(i, k) + ")"));
        }
        if (ev_norms) {
            for (const auto u : p.ev_u) {
                v.push_back(e.def("fabs(" + e.val(u, k) + ")"));
            }
        }
        while (v.size() != 1u) {
            std::vector<std::string> nv;
            for (std::size_t i = 0; i < v.size(); i += 2u) {
                if (i + 1u == v.size()) {
                    nv.push_back(v[i]);
                } else {
                    nv.push_back(e.def("hy_max(" + v[i] + ", " + v[i + 1u] + ")"));
                }
            }
            v.swap(nv);
        }
        return v[0];
    };
    const auto m0 = max_abs(0), mo = max_abs(order), mom1 = max_abs(order - 1u);
    os << "const double num_rho = (" << m0 << " <= 1.0) ? 1.0 : " << m0 << ";\n";
    // rho = exp(log(num / m) / order) (the (1/p)-th roots of src/taylor_00.cpp:242-252): the minimum of the two estimates
    // is taken on the exponents (exp is monotone and keeps nans), the quotients become differences of logarithms.
    os << "const double lg_num = hy_sel_log(num_rho);\n";
    os << "const double lr_o = (lg_num - hy_sel_log(" << mo << ")) * " << fp_literal(1. / static_cast<double>(order)) << ";\n";
    os << "const double lr_om1 = (lg_num - hy_sel_log(" << mom1 << ")) * " << fp_literal(1. / static_cast<double>(order - 1u))
       << ";\n";
    os << "const double rho_m = exp(hy_min(lr_o, lr_om1));\n";
    os << "double h = rho_m * " << fp_literal(rhofac(order)) << ";\n";
    os << "h = hy_min(h, fabs(lim));\n";
    os << "h = (lim < 0.0) ? -h : h;\n";

    if (!p.ev_u.empty() && !reg_jets) {
        // ---- Stepper with events (mode 4; taylor_add_adaptive_step_with_events(), src/taylor_00.cpp:592-710): the
        // order-p coefficients of the u variables (src/taylor_02.cpp:1016-1190), the jets of the event equations,
        // max |x_i| and the step size; the state is updated later by the dense-output kernel.
        os << "if (a.mode == 4) {\n";
        for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
            for (std::uint32_t k = 0; k <= order; ++k) {
                os << "a.ev_tc[(u64)" << (ev * (order + 1u) + k) << "u * N + s] = " << e.val(p.ev_u[ev], k) << ";\n";
            }
        }
        os << "a.max_abs_state[s] = " << m0 << ";\nlast_h = h;\nbreak;\n}\n";
    }

    if (reg_jets) {
        // ---- State update straight from the SSA coefficients. ----
        // Coefficient k of state variable i: re-derived from the defining state variable when x' = v.
        // The other way round for a variable v which only DEFINES another one (x' = v, no elementary function reads v) when
        // the coefficients of x are live anyway (they are the coefficients of a u variable with a history: x^[k] = d^[k]
        // where the partner of the difference d has vanishing derivatives - a test particle): v^[k] = (k + 1) x^[k+1], one
        // multiplication at the end of the step instead of a coefficient history through the orders (two-body problem:
        // 3 of 8 histories). Within 1.5 ulp of the coefficient (x^[k+1] = RN(v^[k] RN(1 / (k + 1)))): not under
        // kw::exact_division.
        std::set<std::string> hist_names;
        std::vector<std::int64_t> defines(n_eq, -1);
        if (opts.dev.unrolled_derive && !opts.exact_division) {
            std::vector<char> read_by_node(n_eq, 0);
            for (const auto &nd : p.nodes) {
                for (const auto &o : nd.args) {
                    if (o.type == operand::kind::uvar && o.idx < n_eq) {
                        read_by_node[o.idx] = 1;
                    }
                }
            }
            // (Names of the coefficients which are kept anyway: the histories.)
            const auto hist = history_operands(p);
            for (std::uint32_t u = 0; u < p.n_u; ++u) {
                for (std::uint32_t k = 0; hist[u] != 0 && k <= order; ++k) {
                    hist_names.insert(e.val(u, k));
                }
            }
            for (std::uint32_t j = 0; j < n_eq; ++j) {
                const auto &d = p.sv_defs[j];
                if (d.type == operand::kind::uvar && d.idx < n_eq && read_by_node[d.idx] == 0) {
                    defines[d.idx] = (defines[d.idx] == -1) ? static_cast<std::int64_t>(j) : -2;
                }
            }
        }
        std::vector<std::uint32_t> derived_count(n_eq, 0);
        // (v is re-derived from x when x is the one variable it defines and the coefficients of x are those of a history -
        // checked on order 2, the first one which is not a plain copy of another state variable.)
        const auto derive_v = [&](std::uint32_t v) {
            return order >= 3u && defines[v] >= 0 && hist_names.count(e.val(static_cast<std::uint32_t>(defines[v]), 2u)) != 0u;
        };
        std::function<std::string(std::uint32_t, std::uint32_t)> coef = [&](std::uint32_t i, std::uint32_t k) {
            const auto &d = p.sv_defs[i];
            if (k > 0u && d.type == operand::kind::uvar && d.idx < n_eq) {
                if (derive_v(d.idx)) {
                    // (Never back from the re-derived v.)
                    return e.val(i, k);
                }
                return e.div_const(coef(d.idx, k - 1u), k);
            }
            if (k > 0u && k < order && derive_v(i)) {
                const auto &xk = e.val(static_cast<std::uint32_t>(defines[i]), k + 1u);
                if (!ssa_emitter::is_zero_lit(xk)) {
                    ++derived_count[i];
                    return e.def(ssa_emitter::mul(fp_literal(static_cast<double>(k + 1u)), xk));
                }
            }
            return e.val(i, k);
        };
        // A pair x' = v with a re-derived v, plain Horner evaluation: sum_k (k + 1) x^[k+1] h^k is the DERIVATIVE of the series
        // of x - both come out of one pass (res' = res' h + res, res = res h + x^[k]: two FMAs per coefficient, like the two
        // chains) without the multiplications by k + 1; the term v^[p] h^p, which the series of x knows nothing of, is added at
        // the end (h^p by squaring, once per step).
        std::vector<char> paired(n_eq, 0);
        std::string h_to_p;
        for (std::uint32_t v = 0; v < n_eq && !opts.high_accuracy && opts.dev.unrolled_derive; ++v) {
            if (!derive_v(v)) {
                continue;
            }
            const auto x = static_cast<std::uint32_t>(defines[v]);
            bool ok = true;
            for (std::uint32_t k = 2; k <= order; ++k) {
                ok = ok && !ssa_emitter::is_zero_lit(e.val(x, k)) && !e.val(x, k).empty();
            }
            if (!ok) {
                continue;
            }
            if (h_to_p.empty()) {
                h_to_p = e.pow_ebs("h", order);
            }
            // (x^[1] = v^[0] / 1 = the current v; x^[0] = the current x.)
            os << "{\ndouble res = " << e.val(x, order) << ", der = " << e.val(x, order) << ";\n";
            os << "res = " << e.val(x, order - 1u) << " + res * h;\n";
            for (std::uint32_t k = order - 1u; k-- > 0u;) {
                os << "der = res + der * h;\nres = " << coef(x, k) << " + res * h;\n";
            }
            os << "x" << x << "n = res;\nx" << v << "n = " << e.val(v, order) << " * " << h_to_p << " + der;\n}\n";
            paired[x] = paired[v] = 1;
            derived_count[v] = order - 1u;
        }
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            if (paired[i] != 0) {
                continue;
            }
            if (opts.high_accuracy) {
                os << "{\ndouble res = " << coef(i, 0) << ", comp = 0.0, cur_h = h;\n";
                for (std::uint32_t k = 1; k <= order; ++k) {
                    const auto c = coef(i, k);
                    os << "{\nconst double tmp = " << c << " * cur_h;\nconst double y = tmp - comp;\n";
                    os << "const double t = res + y;\ncomp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
                }
                os << "x" << i << "n = res;\n}\n";
            } else {
                // NOTE: coefficients are named before the block so that the chain is a pure Horner recursion.
                std::vector<std::string> cs;
                for (std::uint32_t k = 0; k <= order; ++k) {
                    cs.push_back(coef(i, k));
                }
                // Leading coefficients which are the literal +0 (derivatives which vanish identically): the steps
                // res = 0 + res * h over them yield +0 for a finite h and a NaN otherwise from the first one on - ONE of them
                // is all of them (the compiler cannot drop any: no fast-math flags).
                std::uint32_t top = order;
                if (opts.dev.unrolled_trim) {
                    while (top >= 1u && ssa_emitter::is_zero_lit(cs[top]) && ssa_emitter::is_zero_lit(cs[top - 1u])) {
                        --top;
                    }
                }
                os << "{\ndouble res = " << cs[top] << ";\n";
                for (std::uint32_t k = order - top + 1u; k <= order; ++k) {
                    os << "res = " << cs[order - k] << " + res * h;\n";
                }
                os << "x" << i << "n = res;\n}\n";
            }
        }
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            os << "x" << i << " = x" << i << "n;\n";
        }
        if (n_derived != nullptr) {
            *n_derived = static_cast<std::uint32_t>(
                std::count_if(derived_count.begin(), derived_count.end(), [&](std::uint32_t c) { return c + 1u == order; }));
        }
    } else {
        // ---- State update: reload the coefficients from the jet buffer. ----
        // NOTE: the memory clobber prevents the compiler from forwarding the stored coefficients (which
        // would keep (order + 1) * n_eq values live in registers across the whole step).
        os << "asm volatile(\"\" ::: \"memory\");\n";
        if (opts.high_accuracy) {
            // Compensated summation (reference: taylor_run_ceval(), src/taylor_00.cpp:355-460).
            for (std::uint32_t i = 0; i < n_eq; ++i) {
                os << "{\nconst double *c = jet + (u64)" << (static_cast<std::uint64_t>(i) * (order + 1u))
                   << "u * hy_Ns;\n";
                os << "double res = c[0], comp = 0.0, cur_h = h;\n";
                os << "#pragma unroll\nfor (unsigned k = 1; k <= " << order << "u; ++k) {\n";
                os << "const double tmp = c[(u64)k * hy_Ns] * cur_h;\nconst double y = tmp - comp;\nconst double t = res + "
                      "y;\n";
                os << "comp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
                os << "x" << i << " = res;\n}\n";
            }
        } else {
            // Horner (reference: taylor_run_multihorner(), src/taylor_00.cpp:279-351).
            for (std::uint32_t i = 0; i < n_eq; ++i) {
                os << "{\nconst double *c = jet + (u64)" << (static_cast<std::uint64_t>(i) * (order + 1u))
                   << "u * hy_Ns;\n";
                os << "double res = c[(u64)" << order << "u * hy_Ns];\n";
                os << "#pragma unroll\nfor (unsigned k = 1; k <= " << order << "u; ++k) {\n";
                os << "res = c[(u64)(" << order << "u - k) * hy_Ns] + res * h;\n}\n";
                os << "x" << i << " = res;\n}\n";
            }
        }

    }

    // ---- Bookkeeping (reference: src/taylor_adaptive_batch.cpp:702-727, :1402-1460). ----
    os << R"HIP(
{
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df nt = hy_df_add(tcur, hh);
    t_hi = nt.hi; t_lo = nt.lo;
}
last_h = h;
bool nf = !(hy_finite(t_hi) & hy_finite(t_lo));
)HIP";
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        os << "nf = nf | !hy_finite(x" << i << ");\n";
    }
    os << R"HIP(
HY_STEP_TAIL(nf, true)
}
)HIP";
    if (!p.ev_u.empty() && !reg_jets) {
        os << "if (a.mode == 4) {\na.last_h[s] = last_h;\nreturn;\n}\n";
    }
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        os << "a.state[(u64)" << i << "u * N + s] = x" << i << ";\n";
    }
    os << R"HIP(
if (a.mode != 2) {
    a.time_hi[s] = t_hi;
    a.time_lo[s] = t_lo;
} else {
    // Raw stepper ABI: the step taken is returned through the (in/out) limits array.
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
)HIP";

    n_stmt += e.n_stmt;
    return os.str();
}

// Estimate (in doubles) of what a lane must keep alive in register-jet mode: the jets of the state variables
// that are neither constant nor defined by another state variable, plus the history of the operands of the
// nonlinear nodes.
std::uint64_t reg_jet_estimate(const taylor_program &p, std::uint32_t order, bool folded = false)
{
    std::uint64_t n = 0;
    for (const auto &d : p.sv_defs) {
        if (d.type == operand::kind::uvar && d.idx >= p.n_eq) {
            n += order + 1u;
        }
    }
    for (const auto h : history_operands(p, folded)) {
        n += (h != 0) ? order : 0u;
    }
    return n;
}

emitted_module emit_unrolled(const taylor_program &p, const emit_options &opts)
{
    std::ostringstream src;
    src << prelude << rules_source(p);
    emit_dout(src, p, opts);

    emitted_module ret;
    // Register-resident jets when they fit comfortably in the 512 VGPR+AGPR of a lane.
    // NOTE: the stepper with events needs the Taylor coefficients in memory (event detection, dense output).
    // (Round 6: the estimate counts what the generator really keeps when it folds scaled copies and u + number into their
    // parents - default arithmetic only -, and the limit went from 200 to 270 doubles: two massive bodies 226, 1.24e9 -> 8.7e9
    // system-steps/s, cr3bp 265, 3.0e9 -> 6.6e9, both with 10 ... 16 spilled registers of 512 - against jets which travel
    // through HBM at every step; profiles/r06_model_rates.log. HEYOKA_AMD_REG_JETS_MAX overrides the limit.)
    const bool folds = opts.dev.unrolled_trim && !opts.exact_division && opts.sum_order == 0;
    const bool reg_jets
        = p.ev_u.empty()
          && reg_jet_estimate(p, opts.order, folds)
                 <= static_cast<std::uint64_t>(opts.dev.reg_jets_max >= 0 ? opts.dev.reg_jets_max : (folds ? 270 : 200));
    if (reg_jets) {
        // Two wavefronts per SIMD (256 registers per lane) when what a lane keeps through the orders leaves room for the
        // working set: measured on the two-body problem with a test particle, where the histories of the velocities are
        // re-derived (5 histories instead of 8: 1.26e10 -> 1.38e10 system-steps/s with 50 spilled registers; with all 8
        // histories the same attribute costs 2.3x, profiles/r05_ab_two_body_one_vs_two_wavefronts.log). Only there: a
        // decomposition which keeps that little WITHOUT the re-derivation is compiled as before (not measured).
        // HEYOKA_AMD_UNROLLED_WAVES=n forces n.
        int waves = opts.dev.unrolled_waves;
        if (waves == 0) {
            std::uint64_t n_tmp = 0;
            std::uint32_t n_derived = 0;
            emit_unrolled_kernel(p, opts, "hy_taylor", true, n_tmp, false, 0, &n_derived);
            const auto est = reg_jet_estimate(p, opts.order);
            detail::log_message(log_level::debug, "straight-line stepper: " + std::to_string(est) + " doubles kept per lane (estimate), the histories of "
                                                      + std::to_string(n_derived) + " state variables re-derived at the end of the step");
            if (n_derived > 0u && est >= static_cast<std::uint64_t>(n_derived) * (opts.order + 1u)
                && est - static_cast<std::uint64_t>(n_derived) * (opts.order + 1u) <= 126u) {
                waves = 2;
            }
        }
        src << emit_unrolled_kernel(p, opts, "hy_taylor", true, ret.n_statements, false, waves);
        // NOTE: the variant serving write_tc streams the coefficients out of the same register-resident code (round 1
        // went through the jets-in-memory kernel: 512 registers, 790 spilled dwords for the two-body problem).
        src << emit_unrolled_kernel(p, opts, "hy_taylor_tc", true, ret.n_statements, true, waves);
        ret.tc_kernel_name = "hy_taylor_tc";
        ret.tc_optional = true;
        ret.notes = "register-resident state jets (+ variant streaming the Taylor coefficients out)";
        if (waves == 2 && opts.dev.unrolled_waves == 0) {
            ret.notes += ", two wavefronts per SIMD (coefficient histories of the variables which only define x' = v re-derived)";
        }
    } else {
        src << emit_unrolled_kernel(p, opts, "hy_taylor", false, ret.n_statements, false, opts.dev.unrolled_waves);
    }
    ret.source = src.str();
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.block_size = opts.block_size;
    ret.lanes_per_system = 1;
    ret.mode = emit_mode::unrolled;
    return ret;
}

} // namespace emit_detail

// Jets of the event equations from the jets of the state variables (see hip_emit.hpp). The part of the decomposition the
// event equations depend on is run order by order, one system per lane, with the coefficients of the state variables
// loaded from a.tc instead of being produced by the recursion x^[k+1] = f^[k] / (k + 1) - the elementary-function rules
// (and hence the coefficients) are the ones of the stepper with events of the one-system-per-lane kernels. The norms of
// the step-size selector are those of the state variables (a.sel_norms, written by the cluster stepper in mode 4)
// extended to the event equations (taylor_determine_h() iterates up to n_eq + n_sv_funcs, src/taylor_00.cpp:209-219).
emitted_module emit_event_jets(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    using emit_detail::ssa_emitter;
    emitted_module ret;
    const auto n_eq = p.n_eq;
    const auto order = opts.order;
    if (p.ev_u.empty()) {
        why_not = "no event equations";
        return ret;
    }
    // Nodes needed by the event equations (transitive closure over the arguments and the hidden dependencies).
    std::vector<char> need(p.n_u, 0);
    for (const auto u : p.ev_u) {
        need[u] = 1;
    }
    for (std::uint32_t u = p.n_u; u-- > n_eq;) {
        if (need[u] == 0) {
            continue;
        }
        const auto &n = p.nodes[u - n_eq];
        for (const auto &o : n.args) {
            if (o.type == operand::kind::uvar) {
                need[o.idx] = 1;
            }
        }
        for (const auto d : n.deps) {
            need[d] = 1;
        }
    }
    std::uint32_t n_nodes = 0, n_sv = 0;
    for (std::uint32_t u = 0; u < p.n_u; ++u) {
        if (need[u] != 0) {
            (u < n_eq ? n_sv : n_nodes) += 1u;
        }
    }
    // NOTE: straight-line code with the histories in registers, like the unrolled stepper.
    const std::uint32_t max_nodes = 40;
    if (n_nodes > max_nodes && !opts.ev_helpers_only) {
        why_not = "the event equations depend on " + std::to_string(n_nodes) + " nodes of the decomposition (limit: "
                  + std::to_string(max_nodes) + ")";
        return ret;
    }

    // Compact Taylor coefficients (emit_options::compact_tc): the stepper wrote only the order-0 row of the state
    // variables defined by another state variable; their higher orders are parent^[k-1] / k - a multiplication by
    // RN(1 / k) like the state recursion of the pair kernels, the correctly rounded quotient under kw::exact_division.
    const auto derived = [&](std::uint32_t i) {
        return opts.compact_tc && p.sv_defs[i].type == operand::kind::uvar && p.sv_defs[i].idx < n_eq;
    };
    // NOTE: through hy_mul_nc() - a product which is never contracted into an FMA with its consumer: the value must be
    // the ROUNDED product the stepper would have stored, whatever comes next (x_1 - x_2, the Horner step).
    const auto derived_coeff = [&](const std::string &parent, std::uint32_t k) {
        return opts.exact_division ? ("(" + parent + " / " + fp_literal(static_cast<double>(k)) + ")")
                                   : ("hy_mul_nc(" + parent + ", " + fp_literal(1. / static_cast<double>(k)) + ")");
    };

    ssa_emitter e(p, order);
    auto &os = e.os;
    // (The same node rules and addition order as the one-system-per-lane stepper with events.)
    e.running_sums = opts.sum_order != 1;
    if (opts.compact_tc) {
        // (An empty asm statement on the product: with -ffp-contract=fast the backend fuses whatever it can reach, pragmas
        // or not; a value which went through an asm operand is opaque to it.)
        os << "__device__ __forceinline__ double hy_mul_nc(double x, double y)\n{\n    double t = x * y;\n"
              "    asm(\"\" : \"+v\"(t));\n    return t;\n}\n";
        // hy_dout_c: dense output (state update of a step with events) from the compact coefficients; hy_tc_expand: fills
        // in the rows the stepper left out, for whoever reads the full array afterwards (get_tc(), update_d_output(),
        // continuous output, propagate_grid()).
        os << "__constant__ double hy_rk_c[" << (order + 1u) << "] = {0.0";
        for (std::uint32_t k = 1; k <= order; ++k) {
            os << "," << fp_literal(opts.exact_division ? static_cast<double>(k) : 1. / static_cast<double>(k));
        }
        os << "};\n__constant__ int hy_tc_parent[" << n_eq << "] = {";
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            os << (derived(i) ? static_cast<long long>(p.sv_defs[i].idx) : -1ll) << ",";
        }
        os << "};\n";
        const std::string coeff
            = opts.exact_division ? "(cp[(u64)(k - 1u) * N] / hy_rk_c[k])" : "hy_mul_nc(cp[(u64)(k - 1u) * N], hy_rk_c[k])";
        // (hfull != nullptr: only the lanes whose step was truncated - hs[s] != hfull[s] - are updated: the stepper which
        // evaluates the event equations itself has already taken the full step everywhere.)
        os << "struct hy_doutc_args { double *out; const double *tc; const double *hs; u64 N; const double *hfull; };\n";
        // hy_dout_c: ONE pass over the coefficients of every stored variable v updates its own sum and the sum of the
        // variable x it defines (x' = v: x^[k] = v^[k-1] / k) - the rows of v are read once, with the operations and the
        // operation order of hy_dout on the full set.
        os << "__constant__ int hy_tc_child[" << n_eq << "] = {";
        for (std::uint32_t j = 0; j < n_eq; ++j) {
            long long child = -1;
            for (std::uint32_t i = 0; i < n_eq; ++i) {
                if (derived(i) && p.sv_defs[i].idx == j) {
                    child = static_cast<long long>(i);
                }
            }
            os << child << ",";
        }
        os << "};\n";
        const std::string xk = opts.exact_division ? "(ck / hy_rk_c[k + 1u])" : "hy_mul_nc(ck, hy_rk_c[k + 1u])";
        os << "extern \"C\" __global__ void __launch_bounds__(256) hy_dout_c(const hy_doutc_args a)\n{\n";
        os << "const u64 N = a.N;\nconst u64 s = (u64)blockIdx.x * 256u + threadIdx.x;\nif (s >= N) return;\n";
        os << "const double h = a.hs[s];\n";
        os << "if (a.hfull != nullptr && h == a.hfull[s]) return;\n";
        os << "for (unsigned j = 0; j < " << n_eq << "u; ++j) {\n";
        os << "if (hy_tc_parent[j] >= 0) continue;\n";
        os << "const int ch = hy_tc_child[j];\n";
        os << "const double *c = a.tc + ((u64)j * " << (order + 1u) << "u) * N + s;\n";
        os << "const double x0 = (ch >= 0) ? a.tc[((u64)(unsigned)(ch < 0 ? 0 : ch) * " << (order + 1u) << "u) * N + s] : 0.0;\n";
        if (opts.high_accuracy) {
            // (Ascending orders: res = c[0]; res += c[k] * h^k with the compensation of hy_dout.)
            os << "double ck = c[0];\n";
            os << "double rv = ck, cv = 0.0, rx = x0, cx = 0.0, cur_h = h;\n";
            os << "for (unsigned k = 0; k < " << order << "u; ++k) {\n";
            os << "{\nconst double tmp = " << xk << " * cur_h;\nconst double y = tmp - cx;\nconst double t = rx + y;\n";
            os << "cx = (t - rx) - y;\nrx = t;\n}\n";
            os << "ck = c[(u64)(k + 1u) * N];\n";
            os << "{\nconst double tmp = ck * cur_h;\nconst double y = tmp - cv;\nconst double t = rv + y;\n";
            os << "cv = (t - rv) - y;\nrv = t;\n}\n";
            os << "cur_h = cur_h * h;\n}\n";
        } else {
            // (Descending orders: res = c[p]; res = c[k] + res * h.)
            os << "double ck = c[(u64)" << order << "u * N];\ndouble rv = ck, rx = 0.0;\n";
            os << "for (unsigned k = " << order - 1u << "u;; --k) {\n";
            os << "ck = c[(u64)k * N];\nrv = ck + rv * h;\n";
            os << "rx = (k == " << order - 1u << "u) ? " << xk << " : (" << xk << " + rx * h);\n";
            os << "if (k == 0u) break;\n}\n";
            os << "rx = x0 + rx * h;\n";
        }
        os << "a.out[(u64)j * N + s] = rv;\n";
        os << "if (ch >= 0) a.out[(u64)(unsigned)ch * N + s] = rx;\n}\n}\n";
        os << "extern \"C\" __global__ void __launch_bounds__(256) hy_tc_expand(const hy_doutc_args a)\n{\n";
        os << "const u64 N = a.N;\nconst u64 s = (u64)blockIdx.x * 256u + threadIdx.x;\nif (s >= N) return;\n";
        os << "double *tc = a.out;\n";
        os << "for (unsigned i = 0; i < " << n_eq << "u; ++i) {\n";
        os << "const int par = hy_tc_parent[i];\nif (par < 0) continue;\n";
        os << "double *c = tc + ((u64)i * " << (order + 1u) << "u) * N + s;\n";
        os << "const double *cp = tc + ((u64)(unsigned)par * " << (order + 1u) << "u) * N + s;\n";
        os << "for (unsigned k = 1; k <= " << order << "u; ++k) c[(u64)k * N] = " << coeff << ";\n}\n}\n";
    }
    if (opts.ev_helpers_only) {
        std::ostringstream src;
        src << emit_detail::prelude << os.str();
        ret.source = src.str();
        ret.kernel_name = "hy_dout_c";
        ret.block_size = 256;
        ret.lanes_per_system = 1;
        ret.mode = emit_mode::unrolled;
        ret.notes = std::to_string(p.ev_u.size()) + " event equation(s) evaluated by the stepper";
        return ret;
    }
    os << "extern \"C\" __global__ void __launch_bounds__(256) hy_ev_jets(const hy_kargs a)\n{\n";
    os << "const u64 N = a.N;\nconst u64 s = (u64)blockIdx.x * 256u + threadIdx.x;\nif (s >= N) return;\n";
    std::vector<char> par_used(p.n_par, 0);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (need[u] != 0) {
            for (const auto &o : p.nodes[u - n_eq].args) {
                if (o.type == operand::kind::par) {
                    par_used[o.idx] = 1;
                }
            }
        }
    }
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        if (par_used[i] != 0) {
            os << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
        }
    }
    os << "const double *const jet = a.tc + s;\n";
    // (The order-0 rule of func_kind::time reads the time coordinate by this name: an event equation like time - 0.5.)
    os << "const double t_hi = a.time_hi[s];\n(void)t_hi;\n";
    for (std::uint32_t k = 0; k <= order; ++k) {
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            if (need[i] != 0) {
                if (derived(i) && k > 0u) {
                    // (Compact Taylor coefficients: x^[k] = v^[k-1] / k with the arithmetic of the stepper.)
                    e.val(i, k) = e.def(derived_coeff("jet[(u64)"
                                                          + std::to_string(static_cast<std::uint64_t>(p.sv_defs[i].idx) * (order + 1u)
                                                                           + (k - 1u))
                                                          + "u * N]",
                                                      k));
                } else {
                    e.val(i, k)
                        = e.def("jet[(u64)" + std::to_string(static_cast<std::uint64_t>(i) * (order + 1u) + k) + "u * N]");
                }
            }
        }
        for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
            if (need[u] != 0) {
                e.node(u - n_eq, k);
            }
        }
        for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
            os << "a.ev_tc[(u64)" << (ev * (order + 1u) + k) << "u * N + s] = " << e.val(p.ev_u[ev], k) << ";\n";
        }
    }
    // Norms: state variables first, then the event equations, combined like the pairwise maximum of the stepper with
    // events (the maximum of non-NaN values does not depend on the order of the comparisons).
    const auto ext = [&](const std::string &base, std::uint32_t k) {
        auto m = e.def(base);
        for (const auto u : p.ev_u) {
            m = e.def("hy_max(" + m + ", fabs(" + e.val(u, k) + "))");
        }
        return m;
    };
    const auto m0 = ext("a.sel_norms[s]", 0), mo = ext("a.sel_norms[N + s]", order), mom1 = ext("a.sel_norms[2u * N + s]", order - 1u);
    os << "const double num_rho = (" << m0 << " <= 1.0) ? 1.0 : " << m0 << ";\n";
    os << "const double rho_o = hy_root(num_rho / " << mo << ", " << fp_literal(1. / static_cast<double>(order)) << ");\n";
    os << "const double rho_om1 = hy_root(num_rho / " << mom1 << ", " << fp_literal(1. / static_cast<double>(order - 1u))
       << ");\n";
    os << "const double rho_m = hy_min(rho_o, rho_om1);\n";
    os << "double h = rho_m * " << fp_literal(emit_detail::rhofac(order)) << ";\n";
    os << "const double lim = a.lim[s];\nh = hy_min(h, fabs(lim));\nh = (lim < 0.0) ? -h : h;\n";
    os << "a.last_h[s] = h;\na.max_abs_state[s] = " << m0 << ";\n}\n";

    std::ostringstream src;
    src << emit_detail::prelude << emit_detail::rules_source(p) << os.str();
    ret.source = src.str();
    ret.kernel_name = "hy_ev_jets";
    ret.block_size = 256;
    ret.lanes_per_system = 1;
    ret.mode = emit_mode::unrolled;
    ret.n_statements = e.n_stmt;
    ret.notes = "jets of " + std::to_string(p.ev_u.size()) + " event equation(s) from the jets of " + std::to_string(n_sv)
                + " state variable(s), " + std::to_string(n_nodes) + " nodes";
    return ret;
}

bool emit_event_jets_inline(const taylor_program &p, const emit_options &opts,
                            const std::function<std::string(std::uint32_t, std::uint32_t)> &sv,
                            const std::function<std::string(std::uint32_t, std::uint32_t, const std::string &)> &ev_store,
                            std::string &out, std::vector<std::vector<std::string>> &ev_coeffs, std::string &why_not,
                            ev_lane_hooks *lanes, const std::vector<char> *skip_events)
{
    using emit_detail::ssa_emitter;
    const auto n_eq = p.n_eq;
    const auto order = opts.order;
    if (p.ev_u.empty()) {
        why_not = "no event equations";
        return false;
    }
    // Nodes needed by the event equations (transitive closure over the arguments and the hidden dependencies), and how
    // often each of them is read (by a needed node or as an event equation).
    std::vector<char> need(p.n_u, 0);
    std::vector<std::uint32_t> uses(p.n_u, 0);
    const auto skipped_ev = [&](std::size_t ev) { return skip_events != nullptr && (*skip_events)[ev] != 0; };
    for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
        if (!skipped_ev(ev)) {
            need[p.ev_u[ev]] = 1;
            ++uses[p.ev_u[ev]];
        }
    }
    std::uint32_t n_nodes = 0;
    for (std::uint32_t u = p.n_u; u-- > n_eq;) {
        if (need[u] == 0) {
            continue;
        }
        ++n_nodes;
        const auto &n = p.nodes[u - n_eq];
        if (n.kind == func_kind::custom) {
            why_not = "an event equation depends on a function defined through a node rule";
            return false;
        }
        for (const auto &o : n.args) {
            if (o.type == operand::kind::uvar) {
                need[o.idx] = 1;
                ++uses[o.idx];
            } else if (o.type == operand::kind::par) {
                why_not = "an event equation depends on a runtime parameter";
                return false;
            }
        }
        for (const auto d : n.deps) {
            need[d] = 1;
            ++uses[d];
        }
    }

    // Sums of isomorphic terms (ev_lane_hooks): which needed nodes are evaluated by lane c of the system for term c. A sum
    // written with binary operators arrives as a chain, sum(sum(t0, t1), t2): a later sum which adds more terms of the same
    // shape to the result of an earlier one extends its group.
    struct lane_group {
        std::vector<std::uint32_t> roots;               // root of term c
        std::vector<std::uint32_t> nodes0;              // nodes of term 0 (ascending)
        std::vector<std::vector<std::uint32_t>> leaves; // [c][p]: state variable at leaf position p of term c
        std::string sig;
        // sum node -> (argument position, term) pairs it collects
        std::map<std::uint32_t, std::vector<std::pair<std::uint32_t, std::uint32_t>>> collect;
        std::uint32_t last_sum = 0;
    };
    std::vector<lane_group> groups;
    std::vector<int> owner(p.n_u, -1);   // group of a node which belongs to a term (roots included)
    std::vector<char> skipped(p.n_u, 0); // nodes of the terms c >= 1: never emitted
    if (lanes != nullptr) {
        lanes->leaf_vars.clear();
        lanes->leaf_class.clear();
        std::vector<std::vector<std::uint32_t>> claimed; // nodes of all the terms accepted so far
        const auto linear_node = [&](const dc_node &n) {
            return n.kind == func_kind::sum || n.kind == func_kind::sub
                   || (n.kind == func_kind::prod
                       && std::any_of(n.args.begin(), n.args.end(), [](const auto &o) { return o.type != operand::kind::uvar; }));
        };
        for (std::uint32_t S = n_eq; S < p.n_u; ++S) {
            const auto &ns = p.nodes[S - n_eq];
            const auto m = static_cast<std::uint32_t>(ns.args.size());
            if (need[S] == 0 || ns.kind != func_kind::sum || m < 2u) {
                continue;
            }
            bool ok = true;
            int ext = -1; // the group whose running sum is one of the arguments
            std::vector<std::uint32_t> cand_pos;
            for (std::uint32_t a = 0; ok && a < m; ++a) {
                const auto &o = ns.args[a];
                ok = ok && o.type == operand::kind::uvar && o.idx >= n_eq && uses[o.idx] == 1u;
                if (!ok) {
                    break;
                }
                int gs = -1;
                for (std::size_t gi = 0; gi < groups.size(); ++gi) {
                    if (groups[gi].last_sum == o.idx) {
                        gs = static_cast<int>(gi);
                    }
                }
                if (gs >= 0) {
                    ok = ok && ext < 0;
                    ext = gs;
                } else {
                    ok = ok && owner[o.idx] < 0;
                    cand_pos.push_back(a);
                }
                for (std::uint32_t b = a + 1u; b < m; ++b) {
                    ok = ok && ns.args[a].idx != ns.args[b].idx;
                }
            }
            const auto n_old = ext >= 0 ? static_cast<std::uint32_t>(groups[static_cast<std::size_t>(ext)].roots.size()) : 0u;
            ok = ok && !cand_pos.empty() && (ext >= 0 || cand_pos.size() >= 2u) && n_old + cand_pos.size() <= lanes->max_terms;
            if (!ok) {
                continue;
            }
            std::vector<std::string> sigs;
            std::vector<std::vector<std::uint32_t>> term_nodes, term_leaves;
            for (std::size_t ci = 0; ok && ci < cand_pos.size(); ++ci) {
                const auto root = ns.args[cand_pos[ci]].idx;
                std::vector<std::uint32_t> nodes, leaves;
                std::map<std::uint32_t, std::uint32_t> refs; // references to the nodes of the term from inside it
                const std::function<std::string(std::uint32_t)> sig = [&](std::uint32_t u) -> std::string {
                    if (u < n_eq) {
                        auto it = std::find(leaves.begin(), leaves.end(), u);
                        if (it == leaves.end()) {
                            leaves.push_back(u);
                            it = leaves.end() - 1;
                        }
                        return "s" + std::to_string(lanes->sv_class(u)) + ":" + std::to_string(it - leaves.begin());
                    }
                    ++refs[u];
                    if (const auto it = std::find(nodes.begin(), nodes.end(), u); it != nodes.end()) {
                        return "r" + std::to_string(it - nodes.begin());
                    }
                    const auto &n = p.nodes[u - n_eq];
                    if (!n.deps.empty() || n.kind == func_kind::time || owner[u] >= 0) {
                        ok = false;
                        return "";
                    }
                    nodes.push_back(u);
                    std::string r = std::string(func_kind_name(n.kind)) + "[";
                    for (const auto &o : n.args) {
                        if (o.type == operand::kind::par) {
                            ok = false;
                        }
                        r += (o.type == operand::kind::num) ? ("n" + fp_literal(o.value)) : sig(o.idx);
                        r += ",";
                    }
                    return r + "]";
                };
                sigs.push_back(sig(root));
                // Nothing outside the term reads its nodes (the reference of the sum to the root is the first call of sig()).
                for (const auto u : nodes) {
                    ok = ok && uses[u] == refs[u];
                }
                // (Terms are disjoint.)
                for (const auto *lst : {&claimed, &term_nodes}) {
                    for (const auto &tn : *lst) {
                        for (const auto u : nodes) {
                            ok = ok && std::find(tn.begin(), tn.end(), u) == tn.end();
                        }
                    }
                }
                const auto &ref_sig = ext >= 0 ? groups[static_cast<std::size_t>(ext)].sig : sigs[0];
                ok = ok && sigs.back() == ref_sig && !leaves.empty();
                std::sort(nodes.begin(), nodes.end());
                term_nodes.push_back(std::move(nodes));
                term_leaves.push_back(std::move(leaves));
            }
            // A term without a convolution is not worth the lane broadcasts.
            bool has_conv = ext >= 0;
            if (ok && ext < 0) {
                for (const auto u : term_nodes[0]) {
                    has_conv = has_conv || !linear_node(p.nodes[u - n_eq]);
                }
            }
            if (!ok || !has_conv) {
                continue;
            }
            if (ext < 0) {
                groups.emplace_back();
                ext = static_cast<int>(groups.size()) - 1;
                groups.back().sig = sigs[0];
                groups.back().nodes0 = term_nodes[0];
            }
            auto &g = groups[static_cast<std::size_t>(ext)];
            for (std::size_t ci = 0; ci < cand_pos.size(); ++ci) {
                const auto c = static_cast<std::uint32_t>(g.roots.size());
                for (const auto u : term_nodes[ci]) {
                    owner[u] = ext;
                    skipped[u] = c >= 1u ? 1 : 0;
                }
                g.roots.push_back(ns.args[cand_pos[ci]].idx);
                g.leaves.push_back(term_leaves[ci]);
                g.collect[S].emplace_back(cand_pos[ci], c);
                claimed.push_back(term_nodes[ci]);
            }
            g.last_sum = S;
        }
    }

    // NOTE: every lane of a system runs these statements (the wavefront of the one-lane-per-pair kernel holds FOUR systems
    // where hy_ev_jets holds 64: a convolution costs 16 times the lane-time here). What the stepper saves is the
    // dense-output pass over the Taylor coefficients and the launch of hy_ev_jets, ~1.1 ms per 1 048 576 outer-SS systems; one
    // nonlinear node (a convolution per order: ~230 multiply-adds) costs ~0.3 ms on that scale. The budget: three nonlinear
    // nodes, any number of linear ones (coordinates, differences, sums, multiples, the time: practically free); the terms
    // of a sum which the lanes evaluate side by side count once (a squared distance AND a radial velocity fit).
    std::uint32_t n_nonlin = 0;
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (need[u] == 0 || skipped[u] != 0) {
            continue;
        }
        const auto &n = p.nodes[u - n_eq];
        bool linear = n.kind == func_kind::sum || n.kind == func_kind::sub || n.kind == func_kind::time;
        if (n.kind == func_kind::prod) {
            linear = std::any_of(n.args.begin(), n.args.end(), [](const auto &o) { return o.type != operand::kind::uvar; });
        }
        n_nonlin += linear ? 0u : (n.kind == func_kind::sum_sq ? static_cast<std::uint32_t>((n.args.size() + 1u) / 2u) : 1u);
    }
    std::uint32_t max_nonlin = 3, max_nodes = 40;
    if (opts.dev.ev_inline_max_nonlinear >= 0) {
        max_nonlin = static_cast<std::uint32_t>(opts.dev.ev_inline_max_nonlinear);
        max_nodes = std::max(max_nodes, 12u * max_nonlin);
    }
    if (n_nonlin > max_nonlin || n_nodes > max_nodes) {
        why_not = "the event equations depend on " + std::to_string(n_nonlin) + " nonlinear / " + std::to_string(n_nodes)
                  + " nodes of the decomposition (budget inside the stepper: " + std::to_string(max_nonlin) + " / "
                  + std::to_string(max_nodes) + ")";
        return false;
    }

    // Leaf positions of the groups, globally numbered; state variables which are (also) read outside the terms.
    std::vector<std::uint32_t> leaf_base(groups.size(), 0);
    for (std::size_t gi = 0; gi < groups.size(); ++gi) {
        leaf_base[gi] = static_cast<std::uint32_t>(lanes->leaf_vars.size());
        const auto &g = groups[gi];
        for (std::size_t pp = 0; pp < g.leaves[0].size(); ++pp) {
            std::vector<std::uint32_t> vars;
            for (const auto &lv : g.leaves) {
                vars.push_back(lv[pp]);
            }
            lanes->leaf_class.push_back(lanes->sv_class(vars[0]));
            lanes->leaf_vars.push_back(std::move(vars));
        }
    }
    std::vector<char> scalar_use(n_eq, 0);
    for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
        if (p.ev_u[ev] < n_eq && !skipped_ev(ev)) {
            scalar_use[p.ev_u[ev]] = 1;
        }
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (need[u] != 0 && owner[u] < 0) {
            for (const auto &o : p.nodes[u - n_eq].args) {
                if (o.type == operand::kind::uvar && o.idx < n_eq) {
                    scalar_use[o.idx] = 1;
                }
            }
        }
    }

    ssa_emitter e(p, order);
    // (Names of their own: the statements are pasted into the body of another generator.)
    e.counter = 1000000000ull;
    // (The node rules and addition order of hy_ev_jets.)
    e.running_sums = opts.sum_order != 1;
    // lane_vals[position][k]: coefficient k of the state variable this lane holds at a leaf position.
    std::vector<std::vector<std::string>> lane_vals(lanes != nullptr ? lanes->leaf_vars.size() : 0u,
                                                    std::vector<std::string>(order + 1u));
    // (What this lane computed for its term, by group and order: read by every sum of the chain which collects terms.)
    std::vector<std::vector<std::string>> own_val(groups.size(), std::vector<std::string>(order + 1u));
    const auto swap_leaves = [&](const lane_group &g, std::uint32_t base, std::uint32_t k) {
        for (std::size_t pp = 0; pp < g.leaves[0].size(); ++pp) {
            for (std::uint32_t j = 0; j <= k; ++j) {
                std::swap(e.val(g.leaves[0][pp], j), lane_vals[base + pp][j]);
            }
        }
    };
    for (std::uint32_t k = 0; k <= order; ++k) {
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            if (need[i] != 0 && scalar_use[i] != 0) {
                e.val(i, k) = e.def(sv(i, k));
            }
        }
        for (std::size_t pp = 0; pp < lane_vals.size(); ++pp) {
            lane_vals[pp][k] = e.def(lanes->sv_lane(static_cast<std::uint32_t>(pp), k, lanes->leaf_class[pp]));
        }
        for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
            if (need[u] == 0 || skipped[u] != 0) {
                continue;
            }
            if (owner[u] >= 0) {
                // A node of term 0 of a group: evaluated by every lane on ITS term's state variables.
                const auto gi = static_cast<std::size_t>(owner[u]);
                swap_leaves(groups[gi], leaf_base[gi], k);
                e.node(u - n_eq, k);
                swap_leaves(groups[gi], leaf_base[gi], k);
                continue;
            }
            for (auto &g : groups) {
                if (const auto it = g.collect.find(u); it != g.collect.end()) {
                    // The terms in the order of the arguments: the value of term c is what lane c computed.
                    if (own_val[&g - groups.data()][k].empty()) {
                        own_val[&g - groups.data()][k] = e.val(g.roots[0], k);
                    }
                    for (const auto &[pos, c] : it->second) {
                        (void)pos;
                        e.val(g.roots[c], k) = e.def(lanes->lane_bcast(own_val[&g - groups.data()][k], c));
                    }
                }
            }
            e.node(u - n_eq, k);
        }
        for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
            if (!skipped_ev(ev)) {
                e.os << ev_store(static_cast<std::uint32_t>(ev), k, e.val(p.ev_u[ev], k));
            }
        }
    }
    ev_coeffs.clear();
    for (std::size_t ev = 0; ev < p.ev_u.size(); ++ev) {
        std::vector<std::string> c;
        for (std::uint32_t k = 0; !skipped_ev(ev) && k <= order; ++k) {
            c.push_back(e.val(p.ev_u[ev], k));
        }
        ev_coeffs.push_back(std::move(c));
    }
    out = e.os.str();
    return true;
}

bool match_pair_distance_event(const taylor_program &p, std::uint32_t u, pair_distance_event &out)
{
    const auto n_eq = p.n_eq;
    if (u < n_eq) {
        return false;
    }
    // Flatten the sums / differences at the top: squares and numeric constants.
    std::vector<std::uint32_t> sq_args; // the u variable whose square a term is
    std::vector<bool> sq_neg;           // ... and whether the square is subtracted
    double c = 0;
    bool ok = true;
    const std::function<void(std::uint32_t, bool)> term = [&](std::uint32_t v, bool negated) {
        if (!ok || v < n_eq) {
            ok = false;
            return;
        }
        const auto &n = p.nodes[v - n_eq];
        const auto arg = [&](const operand &o, bool neg) {
            if (o.type == operand::kind::num) {
                c += neg ? -o.value : o.value;
            } else if (o.type == operand::kind::uvar) {
                term(o.idx, neg);
            } else {
                ok = false;
            }
        };
        if (n.kind == func_kind::sum) {
            for (const auto &o : n.args) {
                arg(o, negated);
            }
        } else if (n.kind == func_kind::sub && n.args.size() == 2u) {
            arg(n.args[0], negated);
            arg(n.args[1], !negated);
        } else if (n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::num
                   && n.args[0].value == -1. && n.args[1].type == operand::kind::uvar) {
            // (-x is written as -1 * x.)
            term(n.args[1].idx, !negated);
        } else if (n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::uvar
                   && n.args[1].type == operand::kind::uvar && n.args[0].idx == n.args[1].idx) {
            sq_args.push_back(n.args[0].idx);
            sq_neg.push_back(negated);
        } else if (n.kind == func_kind::pow && n.args.size() == 2u && n.args[0].type == operand::kind::uvar
                   && n.args[1].type == operand::kind::num && n.args[1].value == 2.) {
            sq_args.push_back(n.args[0].idx);
            sq_neg.push_back(negated);
        } else if (n.kind == func_kind::sum_sq) {
            for (const auto &o : n.args) {
                if (o.type == operand::kind::uvar) {
                    sq_args.push_back(o.idx);
                    sq_neg.push_back(negated);
                } else {
                    ok = false;
                }
            }
        } else {
            ok = false;
        }
    };
    term(u, false);
    if (!ok || sq_args.size() != 3u || sq_neg[0] != sq_neg[1] || sq_neg[0] != sq_neg[2]) {
        return false;
    }
    out.sign = sq_neg[0] ? -1. : 1.;
    for (std::size_t i = 0; i < 3u; ++i) {
        const auto d = sq_args[i];
        if (d < n_eq) {
            return false;
        }
        const auto &n = p.nodes[d - n_eq];
        if (n.kind != func_kind::sub || n.args.size() != 2u || n.args[0].type != operand::kind::uvar
            || n.args[1].type != operand::kind::uvar || n.args[0].idx >= n_eq || n.args[1].idx >= n_eq) {
            return false;
        }
        out.diffs[i] = {n.args[0].idx, n.args[1].idx};
    }
    out.c = c;
    return true;
}

emitted_module emit_cluster_or_empty(const taylor_program &, const emit_options &, std::string &why_not);
emitted_module emit_cluster_multi_or_empty(const taylor_program &, const emit_options &, std::string &why_not);
emitted_module emit_table(const taylor_program &, const emit_options &);
emitted_module emit_block(const taylor_program &, const emit_options &, std::string &why_not);
bool add_state_aliases(const taylor_program &, taylor_program &);
bool pad_clusters(const taylor_program &, std::uint32_t, taylor_program &);
bool insert_unit_scalings(const taylor_program &, taylor_program &);
bool privatise_cluster_inputs(const taylor_program &, taylor_program &);
bool linearise_accelerations(const taylor_program &, taylor_program &);
bool externalise_scalings(const taylor_program &, taylor_program &);

std::string program_to_string(const taylor_program &p)
{
    std::ostringstream oss;
    const auto op = [&](const operand &o) {
        char buf[64];
        switch (o.type) {
            case operand::kind::uvar:
                oss << "u_" << o.idx;
                break;
            case operand::kind::num:
                std::snprintf(buf, sizeof(buf), "%.17g", o.value);
                oss << buf;
                break;
            default:
                oss << "p" << o.idx;
        }
    };
    for (const auto &n : p.nodes) {
        oss << func_kind_name(n.kind) << '(';
        for (std::size_t a = 0; a < n.args.size(); ++a) {
            if (a != 0u) {
                oss << ", ";
            }
            op(n.args[a]);
        }
        oss << ')';
        for (const auto d : n.deps) {
            oss << " [dep " << d << "]";
        }
        oss << '\n';
    }
    for (const auto &d : p.sv_defs) {
        op(d);
        oss << '\n';
    }
    return oss.str();
}

dev_switches dev_switches::from_env()
{
    dev_switches d;
    const auto off = [](const char *name) {
        const char *e = std::getenv(name);
        return e != nullptr && std::atoi(e) == 0;
    };
    const auto set = [](const char *name) { return std::getenv(name) != nullptr; };
    const auto num = [](const char *name, int dflt) {
        const char *e = std::getenv(name);
        return e != nullptr ? std::atoi(e) : dflt;
    };
    const auto str = [](const char *name) {
        const char *e = std::getenv(name);
        return e != nullptr ? std::string(e) : std::string{};
    };
    d.v5_events = !off("HEYOKA_AMD_V5_EVENTS");
    d.compact_tc = !off("HEYOKA_AMD_COMPACT_TC");
    d.events_in_stepper = !set("HEYOKA_AMD_NO_EVENTS_IN_STEPPER");
    d.pair_events = !set("HEYOKA_AMD_NO_PAIR_EVENTS");
    d.refill = !set("HEYOKA_AMD_NO_REFILL");
    d.block_v2 = !off("HEYOKA_AMD_BLOCK_V2");
    d.state_aliases = !set("HEYOKA_AMD_NO_STATE_ALIASES");
    d.cluster_v1 = set("HEYOKA_AMD_CLUSTER_V1");
    d.multi_class = !(std::getenv("HEYOKA_AMD_MULTI_CLASS") != nullptr && std::string(std::getenv("HEYOKA_AMD_MULTI_CLASS")) == "0");
    d.linearise = !set("HEYOKA_AMD_NO_LINEARISED_SUMS");
    d.table_lds = num("HEYOKA_AMD_TABLE_LDS", -1);
    d.ev_inline_max_nonlinear = num("HEYOKA_AMD_EV_INLINE_MAX_NONLINEAR", -1);
    d.v5_prio = num("HEYOKA_AMD_V5_PRIO", 2);
    d.unrolled_waves = num("HEYOKA_AMD_UNROLLED_WAVES", 0);
    d.unrolled_trim = !off("HEYOKA_AMD_UNROLLED_TRIM");
    d.unrolled_merge_ssq = !off("HEYOKA_AMD_UNROLLED_MERGE_SSQ");
    d.unrolled_derive = !off("HEYOKA_AMD_UNROLLED_DERIVE");
    d.reg_jets_max = num("HEYOKA_AMD_REG_JETS_MAX", -1);
    d.v5_opts = str("HEYOKA_AMD_V5_OPTS");
    d.v5_pad = str("HEYOKA_AMD_V5_PAD");
    d.block_opts = str("HEYOKA_AMD_BLOCK_OPTS");
    return d;
}

emitted_module emit_hip_module(const taylor_program &prog, const emit_options &opts)
{
    if (opts.order < 2u) {
        throw std::invalid_argument("The Taylor order must be at least 2");
    }
    switch (opts.mode) {
        case emit_mode::unrolled:
            return emit_detail::emit_unrolled(prog, opts);
        case emit_mode::cluster: {
            std::string why;
            auto m = emit_cluster_or_empty(prog, opts, why);
            // Pair-interaction systems whose accelerations are not plain sums over the partners (equal or repeated masses:
            // sum / sub / negation trees, reactions as glue nodes of their own) miss the one-lane-per-pair kernel only
            // because of that: retry on the internal program with the accelerations flattened (linearise_accelerations()).
            // (Generations of the wave-cluster kernels, best first: one lane per pair, lane pairs, pipelined, first one.)
            const auto kernel_rank = [](const emitted_module &em) {
                if (em.source.empty()) {
                    return 0;
                }
                for (const auto &[tag, r] : {std::pair{"cluster mode v5", 4}, std::pair{"cluster mode v3", 3}, std::pair{"cluster mode v2", 2}}) {
                    if (em.notes.find(tag) != std::string::npos) {
                        return r;
                    }
                }
                return 1;
            };
            if (opts.dev.linearise && !opts.dev.cluster_v1 && opts.cluster_kernel != 1 && kernel_rank(m) < 4) {
                taylor_program lin;
                if (linearise_accelerations(prog, lin)) {
                    std::string w2;
                    auto m2 = emit_cluster_or_empty(lin, opts, w2);
                    // (Unit masses next to other masses: the elided unit factors of the scaled powers, like below.)
                    taylor_program lin_scaled;
                    if (m2.source.empty() && w2.rfind("clusters are not isomorphic", 0) == 0 && insert_unit_scalings(lin, lin_scaled)) {
                        m2 = emit_cluster_or_empty(lin_scaled, opts, w2);
                        lin = std::move(lin_scaled);
                    }
                    if (kernel_rank(m2) > kernel_rank(m)) {
                        m2.notes += "; accelerations rewritten as flat sums of scaled pair products in the internal program "
                                    "(re-associated additions: equal to the decomposition to rounding)";
                        m2.internal_program = program_to_string(lin);
                        return m2;
                    }
                }
            }
            if (m.source.empty() && why.rfind("a state variable is a history operand", 0) == 0
                && opts.dev.state_aliases) {
                // Retry with alias u variables for those state variables (see add_state_aliases()).
                taylor_program aliased;
                if (add_state_aliases(prog, aliased)) {
                    auto o2 = opts;
                    std::string why2;
                    auto m2 = emit_cluster_or_empty(aliased, o2, why2);
                    if (m2.source.empty() && why2.rfind("more than 64 clusters", 0) == 0) {
                        std::string why_b;
                        m2 = emit_block(aliased, o2, why_b);
                    }
                    if (!m2.source.empty()) {
                        m2.notes += "; state variables in history-operand position aliased by u variables";
                        m2.internal_program = program_to_string(aliased);
                        return m2;
                    }
                    why += "; with state-variable aliases: " + why2;
                }
            }
            if (m.source.empty() && why.rfind("clusters are not isomorphic", 0) == 0) {
                // Clusters which are sub-shapes of the largest one (test particles next to massive bodies): pad them
                // in the internal program (see pad_clusters()).
                taylor_program padded;
                if (pad_clusters(prog, opts.order, padded)) {
                    std::string why2;
                    auto m2 = emit_cluster_or_empty(padded, opts, why2);
                    if (!m2.source.empty()) {
                        m2.notes += "; clusters of " + std::to_string(padded.n_u - prog.n_u)
                                    + " missing members padded to the shape of the largest one";
                        m2.internal_program = program_to_string(padded);
                        return m2;
                    }
                    why += "; with padded clusters: " + why2;
                }
                // Clusters which differ by an elided unit factor (unit masses next to other masses / test particles).
                taylor_program scaled_p;
                if (insert_unit_scalings(prog, scaled_p)) {
                    std::string why3;
                    auto m3 = emit_cluster_or_empty(scaled_p, opts, why3);
                    const taylor_program *base = &scaled_p;
                    taylor_program padded2;
                    if (m3.source.empty() && why3.rfind("clusters are not isomorphic", 0) == 0
                        && pad_clusters(scaled_p, opts.order, padded2)) {
                        m3 = emit_cluster_or_empty(padded2, opts, why3);
                        base = &padded2;
                    }
                    if (!m3.source.empty()) {
                        m3.notes += "; " + std::to_string(base->n_u - prog.n_u)
                                    + " members added to the internal program (unit scalings of elided factors, padding)";
                        m3.internal_program = program_to_string(*base);
                        return m3;
                    }
                    why += "; with unit scalings: " + why3;
                }
            }
            // Clusters which share coordinate differences or have a negation where the others have a difference (fixed
            // centres / mascons with repeated or zero coordinates): private copies per cluster in the internal program
            // (see privatise_cluster_inputs()), then - if needed - the unit scalings and the padding on top of it; wave-cluster
            // kernels, or block mode beyond 64 clusters.
            const auto try_private_inputs = [&](emitted_module &res) {
                taylor_program priv;
                if (!privatise_cluster_inputs(prog, priv)) {
                    return false;
                }
                std::vector<taylor_program> variants;
                variants.push_back(priv);
                taylor_program tmp;
                if (insert_unit_scalings(priv, tmp)) {
                    variants.push_back(tmp);
                }
                std::string why_p;
                for (const auto &v : variants) {
                    std::string w;
                    auto mp = emit_cluster_or_empty(v, opts, w);
                    const taylor_program *base = &v;
                    taylor_program padded;
                    if (mp.source.empty() && w.rfind("clusters are not isomorphic", 0) == 0
                        && pad_clusters(v, opts.order, padded)) {
                        mp = emit_cluster_or_empty(padded, opts, w);
                        base = &padded;
                    }
                    if (mp.source.empty() && w.rfind("more than 64 clusters", 0) == 0) {
                        std::string wb;
                        mp = emit_block(*base, opts, wb);
                        w += wb.empty() ? "" : ("; block mode: " + wb);
                    }
                    if (!mp.source.empty()) {
                        mp.notes += "; internal program with private coordinate differences per cluster ("
                                    + std::to_string(static_cast<long long>(base->n_u) - static_cast<long long>(prog.n_u))
                                    + " members more than the decomposition)";
                        mp.internal_program = program_to_string(*base);
                        res = std::move(mp);
                        return true;
                    }
                    why_p = w;
                }
                why += "; with private cluster inputs: " + why_p;
                return false;
            };
            if (m.source.empty() && why.find("clusters are not isomorphic") != std::string::npos) {
                emitted_module mp;
                if (try_private_inputs(mp)) {
                    return mp;
                }
            }
            if (m.source.empty() && why.rfind("more than 64 clusters", 0) == 0) {
                // Too many clusters for one wavefront: one system per workgroup.
                std::string why_b;
                // Measured on an MI355X: block mode beats the table-driven one-lane-per-system kernels by 5x
                // (nbody(12), 66 clusters) to 44x (nbody(64)) - it is used whenever it is applicable.
                auto b = emit_block(prog, opts, why_b);
                // (Distinct masses: the scalings of the pair products sit inside the clusters, which keeps the decomposition off
                // the v2 cluster phase - rolled order loop, rows in registers, index-pair convolutions. externalise_scalings()
                // gives the internal program the clusters of the equal-mass system, the scalings become glue nodes.)
                if (opts.dev.linearise && b.notes.find("v2 cluster phase") == std::string::npos && !opts.exact_division) {
                    taylor_program ext;
                    if (externalise_scalings(prog, ext)) {
                        std::string wl;
                        auto o2 = opts;
                        o2.block_no_absorb = true;
                        auto b2 = emit_block(ext, o2, wl);
                        if (!b2.source.empty() && b2.notes.find("v2 cluster phase") != std::string::npos) {
                            b2.notes += "; scalings of the pair products moved out of the clusters in the internal program (c (d "
                                        "r^-3) for d (c r^-3): equal to the decomposition to rounding)";
                            b2.internal_program = program_to_string(ext);
                            return b2;
                        }
                    }
                }
                if (!b.source.empty()) {
                    return b;
                }
                why += why_b.empty() ? "; block mode: too few clusters" : ("; block mode: " + why_b);
                if (why_b.find("clusters are not isomorphic") != std::string::npos) {
                    emitted_module mp;
                    if (try_private_inputs(mp)) {
                        return mp;
                    }
                }
            }
            // (Decompositions small enough for straight-line code on ONE lane stay there: a handful of clusters spread
            // over two or four lanes with an LDS exchange is not faster than the registers of a single lane.)
            if (m.source.empty() && opts.dev.multi_class && !opts.event_stepper && prog.nodes.size() > opts.unroll_max_nodes) {
                // Clusters of several shapes (point-mass pairs next to oblateness / drag / relativistic terms, mixed
                // models) or at several dependency levels: one section of straight-line code per CLASS of clusters, cluster
                // i of a class on lane i of the system (emit_cluster_v1() on a multi-class plan); with alias u variables
                // for state variables in history position where needed.
                std::string why_m;
                auto mm = emit_cluster_multi_or_empty(prog, opts, why_m);
                if (mm.source.empty() && why_m.rfind("a state variable is a history operand", 0) == 0 && opts.dev.state_aliases) {
                    taylor_program aliased;
                    if (add_state_aliases(prog, aliased)) {
                        std::string why_a;
                        mm = emit_cluster_multi_or_empty(aliased, opts, why_a);
                        if (!mm.source.empty()) {
                            mm.notes += "; state variables in history-operand position aliased by u variables";
                            mm.internal_program = program_to_string(aliased);
                        }
                        why_m += "; with state-variable aliases: " + why_a;
                    }
                }
                if (!mm.source.empty()) {
                    mm.notes += " (single-class planners: " + why + ")";
                    return mm;
                }
                why += "; multi-class plan: " + why_m;
            }
            if (m.source.empty()) {
                // Not applicable to this DAG: fall back to the generic one-system-per-lane code, unrolled
                // for small decompositions (registers), table-driven for large ones (bounded code size, the
                // reason compact mode exists in the reference).
                auto u = (prog.nodes.size() > opts.unroll_max_nodes) ? emit_table(prog, opts) : emit_detail::emit_unrolled(prog, opts);
                u.notes += (u.notes.empty() ? "" : "; ") + ("cluster mode not applicable: " + why);
                return u;
            }
            return m;
        }
        case emit_mode::table:
            return emit_table(prog, opts);
        case emit_mode::block: {
            std::string why;
            auto b = emit_block(prog, opts, why);
            if (b.source.empty()) {
                throw std::invalid_argument("Block mode is not applicable to this system: " + why);
            }
            return b;
        }
        default:
            throw not_implemented_error("The requested code generation mode is not implemented yet");
    }
}

} // namespace heyoka_amd
