// Event detection kernel. See event_detection.hpp.
#include "event_detection.hpp"

#include <cmath>
#include <sstream>

#include "hip_emit.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd::detail
{

std::string make_event_detection_source(std::uint32_t order)
{
    std::ostringstream src;
    src << emit_detail::prelude;
    src << "#define HY_ORDER " << order << "u\n#define HY_P " << (order + 1u) << "u\n#define HY_MAXD "
        << max_detected_per_lane << "u\n#define HY_WL_CAP 64u\n";
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

// Root inside a bracket with a sign change. The reference uses TOMS 748 with an eps tolerance and returns the
// midpoint of the final bracket (bracketed_root_find(), :307-394); here: bisection down to adjacent doubles.
__device__ double hy_bracketed_root(const double *a, double lb, double ub)
{
    if (hy_finite(lb) && hy_finite(ub) && ub > lb) ub = nextafter(ub, lb);
    double flb = hy_poly_eval(a, lb);
    const double fub = hy_poly_eval(a, ub);
    if (flb == 0.0) return lb;
    if (fub == 0.0) return ub;
    for (int it = 0; it < 200; ++it) {
        const double mid = lb / 2 + ub / 2;
        if (mid <= lb || mid >= ub) break;
        const double fm = hy_poly_eval(a, mid);
        if (fm == 0.0) return mid;
        if ((fm < 0.0) == (flb < 0.0)) {
            lb = mid;
            flb = fm;
        } else {
            ub = mid;
        }
    }
    return lb / 2 + ub / 2;
}

// One lane per thread (ed_data_batch<T>::detect_events(), :1733-2173).
extern "C" __global__ void __launch_bounds__(64) hy_detect_events(const hy_ed_args a)
{
    const u64 j = (u64)blockIdx.x * 64u + threadIdx.x;
    const u64 N = a.N;
    if (j >= N) return;
    a.counts[j] = 0u;
    a.counts[N + j] = 0u;
    const double h = a.h[j], g_eps = a.g_eps[j];
    if (!hy_finite(h) || !hy_finite(g_eps) || h == 0.0) return;

    double ptr[HY_P], tmp[HY_P], tmp1[HY_P], tmp2[HY_P];
    double wl_poly[HY_WL_CAP * HY_P], wl_lb[HY_WL_CAP], wl_ub[HY_WL_CAP];
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
                atomicAdd(a.flags, 1u);
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
        unsigned n_wl = 1, n_isol = 0;
        hy_poly_rescale(wl_poly, ptr, h);
        wl_lb[0] = 0.0;
        wl_ub[0] = 1.0;
        bool failed = false;
        while (n_wl != 0u) {
            --n_wl;
            const double lb = wl_lb[n_wl], ub = wl_ub[n_wl];
            for (unsigned k = 0; k < HY_P; ++k) tmp[k] = wl_poly[n_wl * HY_P + k];
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
                if (n_wl + 2u > HY_WL_CAP) {
                    failed = true;
                    break;
                }
                if (lb_offset < mid) {
                    for (unsigned k = 0; k < HY_P; ++k) wl_poly[n_wl * HY_P + k] = tmp1[k];
                    wl_lb[n_wl] = lb;
                    wl_ub[n_wl] = mid;
                    ++n_wl;
                }
                for (unsigned k = 0; k < HY_P; ++k) wl_poly[n_wl * HY_P + k] = tmp2[k];
                wl_lb[n_wl] = mid;
                wl_ub[n_wl] = ub;
                ++n_wl;
            }
            if (n_isol > HY_ORDER) {
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
            double lb = isol_lb[q];
            const double ub = isol_ub[q];
            if (terminal && lb < lb_offset) {
                lb = lb_offset;
                if (!(hy_poly_eval(tmp1, lb) * hy_poly_eval(tmp1, ub) < 0.0)) continue;
            }
            add_event(hy_bracketed_root(tmp1, lb, ub) * h);
        }
    }
}
)HIP";
    return src.str();
}

} // namespace heyoka_amd::detail
