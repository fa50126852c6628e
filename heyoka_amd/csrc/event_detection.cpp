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
    const double *mas;
    double *g_eps_out;
    double tol;
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
)HIP";
    return src.str();
}

} // namespace heyoka_amd::detail
