// kepF and kepDE, defined through the registry of node rules and nothing else (node_rule.hpp): no generator, planner
// or decomposition pass knows about them.
//
//   F = kepF(h, k, lam):    F + h cos F - k sin F = lam            (eccentric longitude; reference: src/math/kepF.cpp)
//   DE = kepDE(s0, c0, DM): DE - c0 sin DE + s0 (1 - cos DE) = DM  (difference of eccentric anomalies; reference:
//                                                                    src/math/kepDE.cpp, which stops at evaluation and
//                                                                    gradient - the Taylor rule below is derived here)
// Decompositions (the order of the reference's kepF_impl::taylor_decompose(), src/math/kepF.cpp:110-156):
//   a = kepF(h, k, lam) | e = sin(a) | f = cos(a) | c = h * e | d = k * f,    a reads (c, d, e, f), e <-> f
//   a = kepDE(s0, c0, DM) | e = sin(a) | f = cos(a) | c = c0 * f | d = s0 * e, a reads (c, d, e, f), e <-> f
// Rules, k >= 1 (differentiate the defining equation, collect order k; src/math/kepF.cpp:633-711 for kepF):
//   kepF:  a^[k] = (k (k_^[k] e^[0] - h^[k] f^[0] + lam^[k])
//                   + sum_{j=1..k-1} j (a^[j] (c^[k-j] + d^[k-j]) + k_^[j] e^[k-j] - h^[j] f^[k-j])) / (k (1 - c^[0] - d^[0]))
//   kepDE: a^[k] = (k (DM^[k] - s0^[k] + c0^[k] e^[0] + s0^[k] f^[0])
//                   + sum_{j=1..k-1} j (c0^[j] e^[k-j] + s0^[j] f^[k-j] - a^[j] (d^[k-j] - c^[k-j]))) / (k (1 - c^[0] + d^[0]))
// (numerical / parameter arguments have no coefficients beyond order 0: their terms vanish, which reproduces the
// reference's specialised overloads). Order 0: Newton-Raphson safeguarded by bisection on [-1, 2 pi + 1), argument
// reduced to [0, 2 pi) in double-length arithmetic, tolerance 4 eps on |f| and on the bracket, nan for invalid
// (h^2 + k^2 >= 1 or nan) arguments and after 20 iterations without convergence, result folded back into [0, 2 pi)
// (llvm_add_inv_kep_F() / llvm_add_inv_kep_DE(), src/detail/llvm_helpers_celmec.cpp:540-856, :857-1170).
#include "node_rule.hpp"

namespace heyoka_amd
{

namespace
{

const char *const kep_common = R"HIP(
#ifndef HY_RULE_KEP_COMMON
#define HY_RULE_KEP_COMMON
// x mod 2 pi in [0, 2 pi): x - y * floor(x / y) in double-length arithmetic (llvm_trig_arg_reduce(),
// src/detail/llvm_helpers_celmec.cpp:140-177).
static __device__ double hy_rule_mod_2pi(double x)
{
#pragma clang fp contract(off)
    const double y_hi = 0x1.921fb54442d18p+2, y_lo = 0x1.1a62633145c07p-52;
    const double below_2pi = 0x1.921fb54442d17p+2;
    const double c = x / y_hi;
    const double u = c * y_hi, uu = fma(c, y_hi, -u);
    double cc = x - u;
    cc = cc - uu;
    cc = cc + 0.0;
    cc = cc - c * y_lo;
    cc = cc / y_hi;
    const double q_hi = c + cc, q_lo = (c - q_hi) + cc;
    const double fhi = floor(q_hi);
    const double flo = (fhi == q_hi) ? floor(q_lo) : 0.0;
    const double fl_hi = fhi + flo, fl_lo = (fhi - fl_hi) + flo;
    const double pc = y_hi * fl_hi;
    double pcc = fma(y_hi, fl_hi, -pc);
    pcc = (y_hi * fl_lo + y_lo * fl_hi) + pcc;
    const double p_hi = pc + pcc, p_lo = (pc - p_hi) + pcc;
    hy_df xx, yy;
    xx.hi = x;
    xx.lo = 0.0;
    yy.hi = -p_hi;
    yy.lo = -p_lo;
    double r = hy_df_add(xx, yy).hi;
    r = (r < 0.0) ? 0.0 : r;
    r = (below_2pi < r) ? below_2pi : r;
    return r;
}
// Root of f(X) = X - T + p (1 - cos X) + q cos X - r sin X on [-1, 2 pi + 1) from the initial guess X
//   kepF:  p = 0,  q = h, r = k,  T = lam;      kepDE: p = s0, q = 0, r = c0, T = DM
// (f' = 1 + (p - q) sin X - r cos X > 0 for valid arguments).
static __device__ __attribute__((noinline)) double hy_rule_kep_newton(double X, double T, double p, double q, double r)
{
#pragma clang fp contract(off)
    const double twopi = 0x1.921fb54442d18p+2;
    double lb = -1.0, ub = 0x1.d21fb54442d17p+2; // the double preceding 2 pi + 1
    X = (X < lb) ? lb : X;
    X = (ub < X) ? ub : X;
    double sX = sin(X), cX = cos(X);
    double fX = (((X - T) + p * (1.0 - cX)) + q * cX) - r * sX;
    const double tol = 4.0 * 0x1p-52;
    bool not_converged = false;
    unsigned it = 0;
    for (;;) {
        const int sgn = (0.0 < fX) - (fX < 0.0);
        const double n_ub = (sgn >= 0) ? X : ub, n_lb = (sgn <= 0) ? X : lb;
        ub = n_ub;
        lb = n_lb;
        not_converged = (fabs(fX) > tol) & ((ub - lb) > tol);
        if (!(it < 20u) | !not_converged) {
            break;
        }
        double nX = X - fX / ((1.0 + (p - q) * sX) - r * cX);
        nX = (nX > ub) ? 0.5 * (X + ub) : nX;
        nX = (nX < lb) ? 0.5 * (X + lb) : nX;
        X = nX;
        sX = sin(X);
        cX = cos(X);
        fX = (((X - T) + p * (1.0 - cX)) + q * cX) - r * sX;
        ++it;
    }
    double ret = (it == 20u && not_converged) ? __builtin_nan("") : X;
    ret = (ret < 0.0) ? twopi + ret : ret;
    ret = (ret >= twopi) ? ret - twopi : ret;
    return ret;
}
#endif
)HIP";

const char *const kepF_src = R"HIP(
static __device__ double hy_rule_kepF_order0(const double *x)
{
#pragma clang fp contract(off)
    const double h2 = x[0] * x[0], k2 = x[1] * x[1];
    const bool invalid = !(h2 + k2 < 1.0);
    const double h = invalid ? __builtin_nan("") : x[0], k = invalid ? __builtin_nan("") : x[1];
    const double L = hy_rule_mod_2pi(x[2]);
    // Initial guess: L + k sL - h cL + (k^2 - h^2) cL sL + h k (sL^2 - cL^2) + 1/2 (k sL - h cL) (2 (k cL + h sL)^2 -
    // (k sL - h cL)^2).
    const double sL = sin(L), cL = cos(L);
    const double u = k * sL - h * cL, v = k * cL + h * sL;
    const double g1 = L + u, g2 = (k2 - h2) * (cL * sL), g3 = (h * k) * (sL * sL - cL * cL);
    const double g4 = (0.5 * u) * ((v * v + v * v) - u * u);
    return hy_rule_kep_newton((g1 + g2) + (g3 + g4), L, 0.0, h, k);
}
static __device__ __forceinline__ double hy_rule_kepF_orderk(unsigned n, const hy_jet &a, const hy_jet *x, const hy_jet *hd)
{
    // x = (h, k, lam), hd = (c = h sin a, d = k cos a, e = sin a, f = cos a).
    const double nf = (double)n;
    const double divisor = nf * ((1.0 - hy_jc(hd[0], 0)) - hy_jc(hd[1], 0));
    double dividend = nf * ((hy_jc(x[1], n) * hy_jc(hd[2], 0) - hy_jc(x[0], n) * hy_jc(hd[3], 0)) + hy_jc(x[2], n));
    double acc = 0.0;
    for (unsigned j = 1; j < n; ++j) {
        const double t1 = hy_jc(a, j) * (hy_jc(hd[0], n - j) + hy_jc(hd[1], n - j));
        const double t2 = hy_jc(x[1], j) * hy_jc(hd[2], n - j) - hy_jc(x[0], j) * hy_jc(hd[3], n - j);
        acc = acc + (double)j * (t1 + t2);
    }
    if (n > 1u) dividend = dividend + acc;
    return dividend / divisor;
}
)HIP";

const char *const kepDE_src = R"HIP(
static __device__ double hy_rule_kepDE_order0(const double *x)
{
#pragma clang fp contract(off)
    const double s2 = x[0] * x[0], c2 = x[1] * x[1];
    const bool invalid = !(s2 + c2 < 1.0);
    const double s0 = invalid ? __builtin_nan("") : x[0], c0 = invalid ? __builtin_nan("") : x[1];
    const double DM = hy_rule_mod_2pi(x[2]);
    // Initial guess (kep3's propagate_lagrangian): with A = c0 cos DM - s0 sin DM, B = c0 sin DM + s0 cos DM, C = B - s0:
    // DM + C + A C + 1/2 C (2 A^2 - C B).
    const double sM = sin(DM), cM = cos(DM);
    const double A = c0 * cM - s0 * sM, B = c0 * sM + s0 * cM;
    const double C = B - s0;
    const double g1 = DM + C, g2 = A * C, g3 = (0.5 * C) * ((A * A + A * A) - C * B);
    return hy_rule_kep_newton((g1 + g2) + g3, DM, s0, 0.0, c0);
}
static __device__ __forceinline__ double hy_rule_kepDE_orderk(unsigned n, const hy_jet &a, const hy_jet *x, const hy_jet *hd)
{
    // x = (s0, c0, DM), hd = (c = c0 cos a, d = s0 sin a, e = sin a, f = cos a).
    const double nf = (double)n;
    const double divisor = nf * ((1.0 - hy_jc(hd[0], 0)) + hy_jc(hd[1], 0));
    double dividend = nf * (((hy_jc(x[2], n) - hy_jc(x[0], n)) + hy_jc(x[1], n) * hy_jc(hd[2], 0)) + hy_jc(x[0], n) * hy_jc(hd[3], 0));
    double acc = 0.0;
    for (unsigned j = 1; j < n; ++j) {
        const double t1 = hy_jc(x[1], j) * hy_jc(hd[2], n - j) + hy_jc(x[0], j) * hy_jc(hd[3], n - j);
        const double t2 = hy_jc(a, j) * (hy_jc(hd[1], n - j) - hy_jc(hd[0], n - j));
        acc = acc + (double)j * (t1 - t2);
    }
    if (n > 1u) dividend = dividend + acc;
    return dividend / divisor;
}
)HIP";

// A product node as it stands: operator*() folds 0 * x and 1 * x into a number / the bare variable, and a hidden definition
// must be ONE function of leaves (kepF(0.0, k, lam), kepF(1.0, k, lam) ... used to break the decomposition).
expression unfolded_prod(const expression &a, const expression &b)
{
    return detail::make_func(func_kind::prod, {a, b});
}

bool both_zero(const std::vector<expression> &a)
{
    return a[0].is_number() && a[1].is_number() && a[0].num() == 0. && a[1].num() == 0.;
}

// Registration at load time (a function-local static: whoever asks first triggers it).
bool register_builtin_rules()
{
    {
        node_rule r;
        r.name = "kepF";
        r.n_args = 3;
        r.decompose = [](const expression &self, const std::vector<expression> &args,
                         const std::function<expression(std::uint32_t)> &hidden) {
            return std::vector<hidden_def>{{sin(self), {1u}}, {cos(self), {0u}}, {unfolded_prod(args[0], hidden(0)), {}}, {unfolded_prod(args[1], hidden(1)), {}}};
        };
        r.deps = {2u, 3u, 0u, 1u};
        r.hip_source = std::string(kep_common) + kepF_src;
        // kepF(0, 0, lam) = lam (src/math/kepF.cpp:1689-1699).
        r.fold = [](const std::vector<expression> &a, expression &out) {
            if (both_zero(a)) {
                out = a[2];
                return true;
            }
            return false;
        };
        register_node_rule(std::move(r));
    }
    {
        node_rule r;
        r.name = "kepDE";
        r.n_args = 3;
        r.decompose = [](const expression &self, const std::vector<expression> &args,
                         const std::function<expression(std::uint32_t)> &hidden) {
            return std::vector<hidden_def>{{sin(self), {1u}}, {cos(self), {0u}}, {unfolded_prod(args[1], hidden(1)), {}}, {unfolded_prod(args[0], hidden(0)), {}}};
        };
        r.deps = {2u, 3u, 0u, 1u};
        r.hip_source = std::string(kep_common) + kepDE_src;
        // kepDE(0, 0, DM) = DM (src/math/kepDE.cpp:113-123).
        r.fold = [](const std::vector<expression> &a, expression &out) {
            if (both_zero(a)) {
                out = a[2];
                return true;
            }
            return false;
        };
        register_node_rule(std::move(r));
    }
    {
        // A constant: no arguments, no hidden dependencies (src/math/constants.cpp:258-273).
        node_rule r;
        r.name = "pi";
        r.n_args = 0;
        r.decompose = [](const expression &, const std::vector<expression> &, const std::function<expression(std::uint32_t)> &) {
            return std::vector<hidden_def>{};
        };
        r.hip_source = R"HIP(
static __device__ __forceinline__ double hy_rule_pi_order0(const double *)
{
    return 0x1.921fb54442d18p+1;
}
static __device__ __forceinline__ double hy_rule_pi_orderk(unsigned, const hy_jet &, const hy_jet *, const hy_jet *)
{
    return 0.0;
}
)HIP";
        register_node_rule(std::move(r));
    }
    return true;
}

} // namespace

void ensure_builtin_rules()
{
    // (register_node_rule() calls this first, and the registration of the built-ins goes through register_node_rule():
    // the nested call on the registering thread returns at once.)
    thread_local bool busy = false;
    if (busy) {
        return;
    }
    busy = true;
    try {
        static const bool done = register_builtin_rules();
        (void)done;
    } catch (...) {
        busy = false;
        throw;
    }
    busy = false;
}

expression kepF(expression h, expression k, expression lam)
{
    ensure_builtin_rules();
    return custom_func("kepF", {std::move(h), std::move(k), std::move(lam)});
}

expression kepDE(expression s0, expression c0, expression DM)
{
    ensure_builtin_rules();
    return custom_func("kepDE", {std::move(s0), std::move(c0), std::move(DM)});
}

expression pi_constant()
{
    ensure_builtin_rules();
    return custom_func("pi", {});
}

} // namespace heyoka_amd
