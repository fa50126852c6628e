// Event detection kernel. See event_detection.hpp.
#include "event_detection.hpp"

#include <algorithm>
#include <cmath>
#include <sstream>

#include "hip_emit.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd::detail
{

std::uint32_t ed_max_detected(std::uint32_t order, std::uint32_t n_te, std::uint32_t n_nte)
{
    // An event equation is a polynomial of degree `order` over the step: at most `order` isolating intervals are
    // accepted (more is a failure of the isolation, src/detail/event_detection.cpp:2082), plus possibly a root exactly
    // at the beginning of the step.
    const auto n = std::max<std::uint32_t>(std::max(n_te, n_nte), 1u);
    return n * (order + 1u);
}

std::size_t ed_work_list_bytes_per_slot(std::uint32_t order)
{
    return static_cast<std::size_t>(ed_work_list_cap + 2u) * (order + 3u) * sizeof(double);
}

std::string make_event_detection_source(std::uint32_t order, std::uint32_t max_detected)
{
    std::ostringstream src;
    src << emit_detail::prelude;
    src << "#define HY_ORDER " << order << "u\n#define HY_P " << (order + 1u) << "u\n#define HY_MAXD " << max_detected
        << "u\n#define HY_WL_CAP " << ed_work_list_cap << "u\n";
    // Binomial coefficients for the translation by 1 (exact in double precision up to the orders in use).
    src << "__device__ const double hy_bc[" << (order + 1u) * (order + 1u) << "] = {";
    for (std::uint32_t i = 0; i <= order; ++i) {
        double c = 1;
        for (std::uint32_t j = 0; j <= order; ++j) {
            if (j > i) {
                src << "0.0,";
            } else {
                if (j > 0u) {
                    c = c * static_cast<double>(i - j + 1u) / static_cast<double>(j);
                    c = std::round(c);
                }
                src << fp_literal(c) << ",";
            }
        }
    }
    src << "};\n";
    src << R"HIP(
struct hy_ed_args {
    const double *ev_tc;
    const double *h;
    const double *g_eps;
    const int *dirs;
    const double *cd_first;
    const double *cd_second;
    const int *cd_active;
    double *out;
    unsigned *counts;
    unsigned *flags;
    u64 N;
    unsigned n_te, n_nte;
    const double *mas;
    double *g_eps_out;
    double tol;
    double *wl;
    u64 wl_slots;
    const double *maybe;
};

struct hy_ep_args {
    const double *h;
    const double *ed_out;
    const unsigned *counts;
    double *dout_h;
    const double *g_eps;
    const double *state;
    double *time_hi, *time_lo;
    const double *lim;
    double *cd_first, *cd_second;
    int *cd_active;
    i64 *outcome;
    double *last_h;
    double *rec;
    u64 *cursor;
    const double *upd;
    u64 N;
    unsigned n_te, n_nte, dim, n_cd, n_oc, pad;
    int native;
    u64 *ev_counts;
    const double *te_cd;
};

__device__ __forceinline__ int hy_sgn(double x)
{
    return (0.0 < x) - (x < 0.0);
}

// ret[i] = a[i] * scal^i (poly_rescale(), src/detail/event_detection.cpp:171-192).
__device__ void hy_poly_rescale(double *ret, const double *a, double scal)
{
    double cur = 1.0;
    for (unsigned i = 0; i < HY_P; ++i) {
        ret[i] = a[i] * cur;
        cur *= scal;
    }
}

// ret[i] = a[i] / 2^i, scaled by 2^n (poly_rescale_p2(), :197-221).
__device__ void hy_poly_rescale_p2(double *ret, const double *a)
{
    double cur = 1.0;
    for (unsigned i = 0; i < HY_P; ++i) {
        ret[HY_ORDER - i] = cur * a[HY_ORDER - i];
        cur *= 2.0;
    }
}

// ret(x) = a(x + 1) (add_poly_translator_1(), :413-507).
__device__ void hy_poly_translate_1(double *ret, const double *a)
{
    for (unsigned j = 0; j < HY_P; ++j) ret[j] = 0.0;
    for (unsigned i = 0; i < HY_P; ++i) {
        for (unsigned j = 0; j <= i; ++j) ret[j] += hy_bc[i * HY_P + j] * a[i];
    }
}

// Sign changes in the coefficients, zeros skipped (llvm_add_csc(), src/detail/llvm_helpers_ed.cpp:55-210).
__device__ unsigned hy_count_sc(const double *a)
{
    unsigned n_sc = 0;
    int last = hy_sgn(a[0]);
    for (unsigned i = 1; i < HY_P; ++i) {
        const int s = hy_sgn(a[i]);
        if (last != 0 && s + last == 0) ++n_sc;
        if (s != 0) last = s;
    }
    return n_sc;
}

__device__ double hy_poly_eval(const double *a, double x)
{
    double r = a[HY_ORDER];
    for (unsigned i = 1; i <= HY_ORDER; ++i) r = a[HY_ORDER - i] + r * x;
    return r;
}

// First derivative (poly_eval_1(), :249-264).
__device__ double hy_poly_eval_1(const double *a, double x)
{
    double r = a[HY_ORDER] * (double)HY_ORDER;
    for (unsigned i = 1; i < HY_ORDER; ++i) r = a[HY_ORDER - i] * (double)(HY_ORDER - i) + r * x;
    return r;
}

// Fast exclusion check: interval Horner enclosure over [0, h] (llvm_add_fex_check(), :704-816).
__device__ bool hy_fex_check(const double *a, double h)
{
    const double lo_h = (h < 0.0) ? h : 0.0, hi_h = (h < 0.0) ? 0.0 : h;
    double lo = a[HY_ORDER], hi = a[HY_ORDER];
    for (unsigned i = 1; i <= HY_ORDER; ++i) {
        const double p0 = lo * lo_h, p1 = lo * hi_h, p2 = hi * lo_h, p3 = hi * hi_h;
        const double mn = fmin(fmin(p0, p1), fmin(p2, p3)), mx = fmax(fmax(p0, p1), fmax(p2, p3));
        lo = mn + a[HY_ORDER - i];
        hi = mx + a[HY_ORDER - i];
    }
    return hy_sgn(lo) == hy_sgn(hi) && hy_sgn(lo) != 0;
}

// Algorithm 748 (Alefeld, Potra, Shi, ACM TOMS 21(3), 1995, algorithm 4.2): the reference calls Boost.Math's
// toms748_solve() with eps_tolerance<double>() and a budget of 53 function evaluations (bracketed_root_find(),
// :307-394). Boost is not part of the reference tree: restated from the published algorithm - a secant step, a
// quadratic step, then rounds of (inverse cubic or quadratic, the same again, double-length secant, bisection if the
// bracket did not halve).
#define HY_T748_EPS 0x1p-52
#define HY_T748_MAX 1.7976931348623157e308
#define HY_T748_MIN_DIFF (2.2250738585072014e-308 * 32)

__device__ inline double hy_t748_safe_div(double num, double den, double r)
{
    if (fabs(den) < 1.0) {
        if (fabs(den * HY_T748_MAX) <= fabs(num)) return r;
    }
    return num / den;
}

__device__ inline double hy_t748_secant(double a, double b, double fa, double fb)
{
    const double tol = HY_T748_EPS * 5;
    const double c = a - (fa / (fb - fa)) * (b - a);
    if (c <= a + fabs(a) * tol || c >= b - fabs(b) * tol) return (a + b) / 2;
    return c;
}

__device__ double hy_t748_quadratic(double a, double b, double d, double fa, double fb, double fd, int count)
{
    const double B = hy_t748_safe_div(fb - fa, b - a, HY_T748_MAX);
    double A = hy_t748_safe_div(fd - fb, d - b, HY_T748_MAX);
    A = hy_t748_safe_div(A - B, d - a, 0.0);
    if (A == 0.0) return hy_t748_secant(a, b, fa, fb);
    double c = (hy_sgn(A) * hy_sgn(fa) > 0) ? a : b;
    for (int i = 0; i < count; ++i) {
        c -= hy_t748_safe_div(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a);
    }
    if (c <= a || c >= b) c = hy_t748_secant(a, b, fa, fb);
    return c;
}

__device__ double hy_t748_cubic(double a, double b, double d, double e, double fa, double fb, double fd, double fe)
{
    const double q11 = (d - e) * fd / (fe - fd);
    const double q21 = (b - d) * fb / (fd - fb);
    const double q31 = (a - b) * fa / (fb - fa);
    const double d21 = (b - d) * fd / (fd - fb);
    const double d31 = (a - b) * fb / (fb - fa);
    const double q22 = (d21 - q11) * fb / (fe - fb);
    const double q32 = (d31 - q21) * fa / (fd - fa);
    const double d32 = (d31 - q21) * fd / (fd - fa);
    const double q33 = (d32 - q22) * fa / (fe - fa);
    double c = q31 + q32 + q33 + a;
    if (c <= a || c >= b) c = hy_t748_quadratic(a, b, d, fa, fb, fd, 3);
    return c;
}

struct hy_t748_state {
    double a, b, fa, fb, d, fd;
};

__device__ inline bool hy_t748_tol(double x, double y)
{
    return fabs(x - y) <= 4 * HY_T748_EPS * fmin(fabs(x), fabs(y));
}

__device__ void hy_t748_bracket(const double *poly, hy_t748_state &s, double c)
{
    const double t = HY_T748_EPS * 2;
    if ((s.b - s.a) < 2 * t * s.a) {
        c = s.a + (s.b - s.a) / 2;
    } else if (c <= s.a + fabs(s.a) * t) {
        c = s.a + fabs(s.a) * t;
    } else if (c >= s.b - fabs(s.b) * t) {
        c = s.b - fabs(s.b) * t;
    }
    const double fc = hy_poly_eval(poly, c);
    if (fc == 0.0) {
        s.a = c;
        s.fa = 0.0;
        s.d = 0.0;
        s.fd = 0.0;
        return;
    }
    if (hy_sgn(s.fa) * hy_sgn(fc) < 0) {
        s.d = s.b;
        s.fd = s.fb;
        s.b = c;
        s.fb = fc;
    } else {
        s.d = s.a;
        s.fd = s.fa;
        s.a = c;
        s.fa = fc;
    }
}

__device__ inline bool hy_t748_distinct(const hy_t748_state &s, double fe)
{
    return !(fabs(s.fa - s.fb) < HY_T748_MIN_DIFF || fabs(s.fa - s.fd) < HY_T748_MIN_DIFF || fabs(s.fa - fe) < HY_T748_MIN_DIFF
             || fabs(s.fb - s.fd) < HY_T748_MIN_DIFF || fabs(s.fb - fe) < HY_T748_MIN_DIFF
             || fabs(s.fd - fe) < HY_T748_MIN_DIFF);
}

// Root of the polynomial in [lb, ub): the midpoint of the final bracket of TOMS 748 on [lb, prev(ub)]. *flag: 0 = ok,
// -1 = budget of function evaluations exhausted, 1 = no sign change at the ends (Boost's domain error); the caller
// ignores the event unless 0, like the reference (:1460-1490).
__device__ double hy_bracketed_root(const double *poly, double lb, double ub, int *flag)
{
    if (hy_finite(lb) && hy_finite(ub) && ub > lb) ub = nextafter(ub, lb);
    hy_t748_state s;
    s.a = lb;
    s.b = ub;
    s.fa = hy_poly_eval(poly, lb);
    s.fb = hy_poly_eval(poly, ub);
    s.d = 0.0;
    s.fd = 0.0;
    if (!(lb < ub) || hy_sgn(s.fa) * hy_sgn(s.fb) > 0 || !hy_finite(s.fa) || !hy_finite(s.fb)) {
        *flag = 1;
        return 0.0;
    }
    int count = 53 - 2;
    if (!(hy_t748_tol(s.a, s.b) || s.fa == 0.0 || s.fb == 0.0)) {
        double e = 1e5, fe = 1e5;
        hy_t748_bracket(poly, s, hy_t748_secant(s.a, s.b, s.fa, s.fb));
        --count;
        if (count != 0 && s.fa != 0.0 && !hy_t748_tol(s.a, s.b)) {
            const double c = hy_t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 2);
            e = s.d;
            fe = s.fd;
            hy_t748_bracket(poly, s, c);
            --count;
        }
        while (count != 0 && s.fa != 0.0 && !hy_t748_tol(s.a, s.b)) {
            const double a0 = s.a, b0 = s.b;
            double c = hy_t748_distinct(s, fe) ? hy_t748_cubic(s.a, s.b, s.d, e, s.fa, s.fb, s.fd, fe)
                                               : hy_t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 2);
            e = s.d;
            fe = s.fd;
            hy_t748_bracket(poly, s, c);
            if (--count == 0 || s.fa == 0.0 || hy_t748_tol(s.a, s.b)) break;
            c = hy_t748_distinct(s, fe) ? hy_t748_cubic(s.a, s.b, s.d, e, s.fa, s.fb, s.fd, fe)
                                        : hy_t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 3);
            hy_t748_bracket(poly, s, c);
            if (--count == 0 || s.fa == 0.0 || hy_t748_tol(s.a, s.b)) break;
            // Double-length secant step from the end with the smaller function value.
            const bool at_a = fabs(s.fa) < fabs(s.fb);
            const double u = at_a ? s.a : s.b, fu = at_a ? s.fa : s.fb;
            c = u - 2 * (fu / (s.fb - s.fa)) * (s.b - s.a);
            if (fabs(c - u) > (s.b - s.a) / 2) c = s.a + (s.b - s.a) / 2;
            e = s.d;
            fe = s.fd;
            hy_t748_bracket(poly, s, c);
            if (--count == 0 || s.fa == 0.0 || hy_t748_tol(s.a, s.b)) break;
            // Bisection if the three steps did not halve the bracket.
            if ((s.b - s.a) < 0.5 * (b0 - a0)) continue;
            e = s.d;
            fe = s.fd;
            hy_t748_bracket(poly, s, s.a + (s.b - s.a) / 2);
            --count;
        }
    }
    if (s.fa == 0.0) {
        s.b = s.a;
    } else if (s.fb == 0.0) {
        s.a = s.b;
    }
    *flag = (count > 0) ? 0 : -1;
    return s.a / 2 + s.b / 2;
}

// Detection for one lane (ed_data_batch<T>::detect_events(), :1733-2173). The working list of the root isolation
// (interval + rescaled polynomial per entry, at most 250 entries like the reference, :2082) lives in global memory,
// one column per resident thread ("slot"): entry e, word k of slot s at wl[(e * (HY_P + 2) + k) * S + s], words
// 0 .. order = coefficients, then lb, ub.
__device__ void hy_detect_lane(const hy_ed_args &a, const u64 j, const u64 slot)
{
    const u64 N = a.N;
    const u64 S = a.wl_slots;
    double *const wl = a.wl + slot;
    a.counts[j] = 0u;
    a.counts[N + j] = 0u;
    if (a.maybe != nullptr && a.maybe[j] == 0.0) return;
    const double h = a.h[j];
    double g_eps;
    if (a.mas != nullptr) {
        // Maximum error on the Taylor series of the event equations (src/taylor_adaptive_batch.cpp:744-767).
        const double mas = a.mas[j];
        const double eps = 0x1p-52;
        if (hy_finite(mas)) {
            const double max_r_size = (mas < 1.0) ? a.tol : (a.tol * mas);
            g_eps = (max_r_size < eps * mas) ? (eps * mas) : max_r_size;
        } else {
            g_eps = __builtin_inf();
        }
        a.g_eps_out[j] = g_eps;
    } else {
        g_eps = a.g_eps[j];
    }
    if (!hy_finite(h) || !hy_finite(g_eps) || h == 0.0) return;

    double ptr[HY_P], tmp[HY_P], tmp1[HY_P], tmp2[HY_P];
    double isol_lb[HY_P], isol_ub[HY_P];

    const unsigned n_ev = a.n_te + a.n_nte;
    for (unsigned e = 0; e < n_ev; ++e) {
        const bool terminal = e < a.n_te;
        const unsigned cls = terminal ? 0u : 1u;
        const unsigned idx = terminal ? e : (e - a.n_te);
        const double *cf = a.ev_tc + (u64)e * HY_P * N + j;
        for (unsigned k = 0; k < HY_P; ++k) ptr[k] = cf[(u64)k * N];
        if (hy_fex_check(ptr, h)) continue;
        const int dir = a.dirs[e];
        double *out = a.out + (((u64)cls * N + j) * HY_MAXD) * 4u;

        auto add_event = [&](double root) {
            if (!hy_finite(root)) return;
            if (fabs(root) >= fabs(h)) root = nextafter(h, 0.0);
            const double der = hy_poly_eval_1(ptr, root);
            if (!hy_finite(der)) return;
            const int d_sgn = hy_sgn(der);
            if (dir != 0 && d_sgn != dir) return;
            const unsigned c = a.counts[(u64)cls * N + j];
            if (c >= HY_MAXD) {
                atomicAdd(a.flags + 1, 1u);
                return;
            }
            out[c * 4u + 0u] = (double)idx;
            out[c * 4u + 1u] = root;
            out[c * 4u + 2u] = (double)d_sgn;
            out[c * 4u + 3u] = fabs(der);
            a.counts[(u64)cls * N + j] = c + 1u;
        };

        double lb_offset = 0.0;
        if (terminal && a.cd_active[(u64)idx * N + j] != 0) {
            const double first = a.cd_first[(u64)idx * N + j], second = a.cd_second[(u64)idx * N + j];
            lb_offset = ((h >= 0.0) ? (second - first) : (second + first)) / fabs(h);
        }
        if (lb_offset >= 1.0) continue;

        // Working list of (lb, ub, polynomial rescaled to [0, 1]).
        // (The first entry - the whole step - never goes through memory.)
        unsigned n_wl = 0, n_isol = 0;
        hy_poly_rescale(tmp, ptr, h);
        double lb = 0.0, ub = 1.0;
        bool failed = false, first = true;
        while (first || n_wl != 0u) {
            if (!first) {
                --n_wl;
                const double *top = wl + (u64)n_wl * (HY_P + 2u) * S;
                lb = top[(u64)HY_P * S];
                ub = top[(u64)(HY_P + 1u) * S];
                for (unsigned k = 0; k < HY_P; ++k) tmp[k] = top[(u64)k * S];
            }
            first = false;
            // A root exactly at the beginning of the interval.
            if (tmp[0] == 0.0) {
                bool fin = true;
                for (unsigned k = 1; k < HY_P; ++k) fin = fin && hy_finite(tmp[k]);
                if (fin && !(terminal && lb < lb_offset)) add_event(lb * h);
            }
            // Reverse, translate by 1, count the sign changes.
            for (unsigned k = 0; k < HY_P; ++k) tmp1[k] = tmp[HY_ORDER - k];
            hy_poly_translate_1(tmp2, tmp1);
            const unsigned n_sc = hy_count_sc(tmp2);
            if (n_sc == 1u) {
                if (n_isol >= HY_P) {
                    failed = true;
                    break;
                }
                isol_lb[n_isol] = lb;
                isol_ub[n_isol] = ub;
                ++n_isol;
            } else if (n_sc > 1u) {
                // Bisection: [lb, mid] and [mid, ub].
                hy_poly_rescale_p2(tmp1, tmp);
                hy_poly_translate_1(tmp2, tmp1);
                const double mid = lb / 2 + ub / 2;
                if (lb_offset < mid) {
                    double *dst = wl + (u64)n_wl * (HY_P + 2u) * S;
                    for (unsigned k = 0; k < HY_P; ++k) dst[(u64)k * S] = tmp1[k];
                    dst[(u64)HY_P * S] = lb;
                    dst[(u64)(HY_P + 1u) * S] = mid;
                    ++n_wl;
                }
                double *dst = wl + (u64)n_wl * (HY_P + 2u) * S;
                for (unsigned k = 0; k < HY_P; ++k) dst[(u64)k * S] = tmp2[k];
                dst[(u64)HY_P * S] = mid;
                dst[(u64)(HY_P + 1u) * S] = ub;
                ++n_wl;
            }
            // (The reference's limits, :2082.)
            if (n_wl > HY_WL_CAP || n_isol > HY_ORDER) {
                failed = true;
                break;
            }
        }
        if (failed) {
            atomicAdd(a.flags, 1u);
            continue;
        }
        if (n_isol == 0u) continue;
        hy_poly_rescale(tmp1, ptr, h);
        for (unsigned q = 0; q < n_isol; ++q) {
            double rlb = isol_lb[q];
            const double rub = isol_ub[q];
            if (terminal && rlb < lb_offset) {
                rlb = lb_offset;
                if (!(hy_poly_eval(tmp1, rlb) * hy_poly_eval(tmp1, rub) < 0.0)) continue;
            }
            int rflag = 0;
            const double root = hy_bracketed_root(tmp1, rlb, rub, &rflag);
            if (rflag == 0) {
                add_event(root * h);
            } else {
                // (The reference logs a warning and ignores the event.)
                atomicAdd(a.flags + 2, 1u);
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(64) hy_detect_events(const hy_ed_args a)
{
    const u64 slot = (u64)blockIdx.x * 64u + threadIdx.x;
    for (u64 j = slot; j < a.N; j += (u64)gridDim.x * 64u) hy_detect_lane(a, j, slot);
}

// Step sizes of the state update: the step is truncated at the first terminal event of the lane - the one with the
// smallest |root|, the earliest detected among equals like the stable sort of the reference
// (src/taylor_adaptive_batch.cpp:771-781). Also: the size of the record buffer hy_ev_post needs.
extern "C" __global__ void __launch_bounds__(256) hy_ev_pre(const hy_ep_args a)
{
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    if (j >= N) return;
    const unsigned c_te = a.counts[j], c_nte = a.counts[N + j];
    double h = a.h[j];
    if (c_te != 0u) {
        const double *r = a.ed_out + (j * HY_MAXD) * 4u;
        double best = r[1];
        for (unsigned c = 1; c < c_te; ++c) {
            const double root = r[c * 4u + 1u];
            if (fabs(root) < fabs(best)) best = root;
        }
        h = best;
    }
    a.dout_h[j] = h;
    if (c_te + c_nte != 0u) atomicAdd(a.cursor, (u64)(8u + 4u * (c_te + c_nte)));
}

// After the state update (:783-835 and the parts of :837-1030 which do not depend on callbacks).
extern "C" __global__ void __launch_bounds__(256) hy_ev_post(const hy_ep_args a)
{
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    if (j >= N) return;
    const double h = a.dout_h[j];
    hy_df tcur; tcur.hi = a.time_hi[j]; tcur.lo = a.time_lo[j];
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df nt = hy_df_add(tcur, hh);
    a.time_hi[j] = nt.hi;
    a.time_lo[j] = nt.lo;
    a.last_h[j] = h;
    bool nf = !(hy_finite(nt.hi) && hy_finite(nt.lo));
    for (unsigned v = 0; v < a.dim; ++v) nf = nf | !hy_finite(a.state[(u64)v * N + j]);
    if (nf) {
        a.outcome[j] = HY_OC_ERR_NF_STATE;
        return;
    }
    // Cooldowns (:822-835).
    for (unsigned e = 0; e < a.n_te; ++e) {
        const u64 p = (u64)e * N + j;
        if (a.cd_active[p] != 0) {
            const double tmp = a.cd_first[p] + h;
            if (fabs(tmp) >= a.cd_second[p]) {
                a.cd_active[p] = 0;
            } else {
                a.cd_first[p] = tmp;
            }
        }
    }
    // (The outcome of a lane with a terminal event is set by the host once its callback has run.)
    a.outcome[j] = (h == a.lim[j]) ? HY_OC_TIME_LIMIT : HY_OC_SUCCESS;
    const unsigned c_te = a.counts[j], c_nte = a.counts[N + j];
    if (c_te + c_nte == 0u) return;
    // (Library-side counting callbacks only: hy_ev_native applies the events, no records.)
    if (a.native != 0) return;
    const u64 off = atomicAdd(a.cursor + 1, (u64)(8u + 4u * (c_te + c_nte)));
    double *r = a.rec + off;
    r[0] = (double)j;
    r[1] = (double)c_te;
    r[2] = (double)c_nte;
    r[3] = a.g_eps[j];
    r[4] = h;
    r[5] = nt.hi;
    r[6] = nt.lo;
    r[7] = 0.0;
    r += 8;
    for (unsigned cls = 0; cls < 2u; ++cls) {
        const unsigned cnt = cls == 0u ? c_te : c_nte;
        const double *src = a.ed_out + (((u64)cls * N + j) * HY_MAXD) * 4u;
        for (unsigned c = 0; c < cnt * 4u; ++c) r[c] = src[c];
        r += cnt * 4u;
    }
}

// Every callback is the library's counting callback (hy_ep_args::native): what the host loop of step_impl() does for a
// lane with events (src/taylor_adaptive_batch.cpp:837-1030), without the round trip - the non-terminal events which
// trigger before the first terminal event are counted, the first terminal event (smallest |root|, the first one detected
// among equals: the stable sort of src/detail/event_detection.cpp:771-781) gets its cooldown, its count and the outcome
// "continuing" (the counting callback returns true). Runs behind hy_ev_post (times, non-finite check, cooldown ageing).
// The counts are summed over the wavefront before they touch memory: 10^5 lanes with events per step on two addresses
// would serialise otherwise.
extern "C" __global__ void __launch_bounds__(256) hy_ev_native(const hy_ep_args a)
{
    const u64 N = a.N;
    const u64 j0 = (u64)blockIdx.x * 256u + threadIdx.x;
    const bool in = j0 < N;
    const u64 j = in ? j0 : (N - 1u);
    const unsigned c_te = in ? a.counts[j] : 0u, c_nte = in ? a.counts[N + j] : 0u;
    const bool act = in && (c_te + c_nte != 0u) && (a.outcome[j] != HY_OC_ERR_NF_STATE);
    const double h = a.dout_h[j];
    const double *te = a.ed_out + ((u64)j * HY_MAXD) * 4u;
    const double *nte = a.ed_out + (((u64)N + j) * HY_MAXD) * 4u;
    unsigned first = 0;
    for (unsigned c = 1; act && c < c_te; ++c) {
        if (fabs(te[c * 4u + 1u]) < fabs(te[first * 4u + 1u])) first = c;
    }
    const unsigned te_idx = (act && c_te != 0u) ? (unsigned)te[first * 4u] : 0xffffffffu;
    if (te_idx != 0xffffffffu) {
        double cd = a.te_cd[te_idx];
        if (!(cd >= 0.0)) {
            // taylor_deduce_cooldown(), src/detail/event_detection.cpp:519-550.
            cd = a.g_eps[j] / te[first * 4u + 3u] * 10.0;
            if (!hy_finite(cd)) cd = 0.0;
        }
        const u64 p = (u64)te_idx * N + j;
        a.cd_first[p] = 0.0;
        a.cd_second[p] = cd;
        a.cd_active[p] = 1;
        a.outcome[j] = (i64)te_idx;
    }
    // Counts per event, one atomic per wavefront and event.
    for (unsigned e = 0; e < a.n_te + a.n_nte; ++e) {
        unsigned mine = 0;
        if (e < a.n_te) {
            mine = (te_idx == e) ? 1u : 0u;
        } else {
            for (unsigned c = 0; act && c < c_nte; ++c) {
                if ((unsigned)nte[c * 4u] == e - a.n_te && (c_te == 0u || fabs(nte[c * 4u + 1u]) < fabs(h))) ++mine;
            }
        }
        for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
        if ((threadIdx.x & 63u) == 0u && mine != 0u) atomicAdd(a.ev_counts + e, (u64)mine);
    }
    {
        // (Lanes with events, for the statistics: ballot + population count, one atomic per wavefront.)
        const u64 m = __builtin_amdgcn_ballot_w64(act);
        if ((threadIdx.x & 63u) == 0u && m != 0ull) atomicAdd(a.cursor + 2, (u64)__builtin_popcountll(m));
    }
}

extern "C" __global__ void __launch_bounds__(256) hy_ev_scatter(const hy_ep_args a)
{
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j < a.n_cd) {
        const double *u = a.upd + j * 3u;
        const u64 p = (u64)u[0];
        a.cd_first[p] = u[1];
        a.cd_second[p] = u[2];
        a.cd_active[p] = 1;
    } else if (j < (u64)a.n_cd + a.n_oc) {
        const double *u = a.upd + (u64)a.n_cd * 3u + (j - a.n_cd) * 2u;
        a.outcome[(u64)u[0]] = (i64)u[1];
    }
}

// Copies of up to four arrays of doubles in one launch (the snapshot of state / times / parameters in front of a step whose
// Taylor coefficients are stored on demand). One launch instead of three or four asynchronous copies (0.12 ms for
// 300 MB).
struct hy_copy_args {
    double *dst[4];
    const double *src[4];
    u64 n[4];
};
extern "C" __global__ void __launch_bounds__(256) hy_copy_arrays(const hy_copy_args a)
{
    const u64 stride = (u64)gridDim.x * 256u;
    for (unsigned q = 0; q < 4u; ++q) {
        double *d = a.dst[q];
        const double *s = a.src[q];
        const u64 n = a.n[q];
        for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += stride) d[i] = s[i];
    }
}
)HIP";
    return src.str();
}

} // namespace heyoka_amd::detail
