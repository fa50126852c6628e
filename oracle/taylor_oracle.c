/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * CPU restatement (the "oracle") of the numerical hot path of heyoka's
 * taylor_adaptive_batch<double>: Taylor-coefficient recursion, Jorba step-size selector,
 * Horner / compensated state update, double-length time and the step()/propagate_until()
 * bookkeeping. Plain C, strict IEEE double (compile with -ffp-contract=off), one flat
 * "program" (the Taylor decomposition exported by oracle/heyoka_oracle.py) interpreted over a
 * tape of normalised derivatives laid out tape[(order * n_u + u) * B + lane].
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 *
 * Reference (bluescarni/heyoka v7.12.0) citations, file:line relative to /root/reference:
 *  - jet ordering:        src/taylor_02.cpp:1339-1418 (taylor_compute_jet, default mode)
 *  - state-variable rule: src/taylor_02.cpp:245-287   (taylor_compute_sv_diff)
 *  - prod:                src/math/prod.cpp:314-395
 *  - sum / sub / div:     src/math/sum.cpp:185-235, src/detail/sub.cpp:60-124, src/detail/div.cpp:62-160
 *  - sum_sq:              src/detail/sum_sq.cpp:100-245
 *  - pow:                 src/math/pow.cpp:136-152 (ebs), :292-355 (eval algo), :395-550 (diff)
 *  - sin / cos:           src/math/sin.cpp:152-192, src/math/cos.cpp:152-185
 *  - exp / log:           src/math/exp.cpp:84-120, src/math/log.cpp
 *  - time / num_identity: src/math/time.cpp:81-101, src/detail/num_identity.cpp
 *  - pairwise sum:        src/detail/llvm_helpers_algo.cpp:271-308
 *  - step size:           src/taylor_00.cpp:84-94, :102-273; min/max src/detail/llvm_helpers_cmp.cpp:315-329
 *  - Horner / ceval:      src/taylor_00.cpp:279-351, :355-460
 *  - TC layout:           src/taylor_00.cpp:467-584
 *  - dfloat:              include/heyoka/detail/dfloat.hpp:109-164
 *  - step_impl:           src/taylor_adaptive_batch.cpp:632-727
 *  - propagate_until:     src/taylor_adaptive_batch.cpp:1137-1534
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Function kinds: must match oracle/heyoka_oracle.py KIND_IDS. */
enum {
    K_SUM = 0,
    K_PROD = 1,
    K_POW = 2,
    K_SUB = 3,
    K_DIV = 4,
    K_SUM_SQ = 5,
    K_SIN = 6,
    K_COS = 7,
    K_EXP = 8,
    K_LOG = 9,
    K_TIME = 10,
    K_NUM_IDENTITY = 11,
    K_TAN = 12,
    K_TANH = 13,
    K_SINH = 14,
    K_COSH = 15,
    K_ERF = 16,
    K_SIGMOID = 17,
    K_ASIN = 18,
    K_ACOS = 19,
    K_ATAN = 20,
    K_ASINH = 21,
    K_ACOSH = 22,
    K_ATANH = 23,
    K_ATAN2 = 24,
    K_KEPE = 25,
    K_RELU = 26,
    K_RELUP = 27,
    K_SELECT = 28,
    K_LAND = 29,
    K_LOR = 30,
    K_REL_EQ = 31,
    K_REL_NEQ = 32,
    K_REL_LT = 33,
    K_REL_GT = 34,
    K_REL_LTE = 35,
    K_REL_GTE = 36,
    K_KEPF = 37,
    K_KEPDE = 38,
    K_PI = 39
};

enum { A_UVAR = 0, A_NUM = 1, A_PAR = 2 };

/* taylor_outcome values (reference: include/heyoka/taylor.hpp:142-155). */
#define OC_SUCCESS (-4294967296LL - 1)
#define OC_STEP_LIMIT (-4294967296LL - 2)
#define OC_TIME_LIMIT (-4294967296LL - 3)
#define OC_ERR_NF_STATE (-4294967296LL - 4)
#define OC_CB_STOP (-4294967296LL - 5)

typedef struct {
    int32_t n_eq, n_u, n_par, order, n_nodes, high_accuracy;
    const int32_t *kind;     /* [n_nodes] */
    const int32_t *arg_off;  /* [n_nodes + 1] */
    const int32_t *arg_type; /* [n_args] */
    const int32_t *arg_idx;  /* [n_args] */
    const double *arg_val;   /* [n_args] */
    const int32_t *dep;      /* [n_nodes], -1 if none */
    const int32_t *sv_type;  /* [n_eq] */
    const int32_t *sv_idx;   /* [n_eq] */
    const double *sv_val;    /* [n_eq] */
    const int32_t *dep2;     /* [n_nodes], second hidden dependency (kepE), -1 if none */
    const int32_t *dep3;     /* [n_nodes], third / fourth hidden dependencies (kepF, kepDE), -1 if none */
    const int32_t *dep4;
} hy_oracle_program;

/* ---- helpers ---- */

/* Arithmetic flavour (bit 1 of hy_oracle_program::high_accuracy; bit 0 is the compensated-summation flag): the
 * reference's default mode forms the products of a convolution first and adds them with pairwise_sum()
 * (src/math/prod.cpp:386-395); its COMPACT mode (kw::compact_mode, src/taylor_02.cpp:1194-1260) runs the same terms
 * through a rolled loop with a running sum which starts from 0, acc = acc + term (src/math/prod.cpp:686-698,
 * src/math/pow.cpp:905-925, src/math/sin.cpp:314-329, src/detail/sum_sq.cpp:330-345, ...). The terms themselves, the
 * pairwise sum over the ARGUMENTS of sum / sum_sq (src/math/sum.cpp:355, src/detail/sum_sq.cpp:371-377) and everything
 * outside the convolutions are the same in both modes. */
#define HY_COMPACT(p) (((p)->high_accuracy & 2) != 0)
#define HY_HA(p) (((p)->high_accuracy & 1) != 0)

/* In-place pairwise sum of n vectors of B lanes: terms[j*B + l]. Result in terms[0..B). */
static void pairwise_sum(double *terms, int n, int B)
{
    while (n != 1) {
        int m = 0;
        for (int i = 0; i < n; i += 2) {
            double *dst = terms + (size_t)m * B;
            const double *a = terms + (size_t)i * B;
            if (i + 1 == n) {
                if (dst != a) {
                    for (int l = 0; l < B; ++l) dst[l] = a[l];
                }
            } else {
                const double *b = terms + (size_t)(i + 1) * B;
                for (int l = 0; l < B; ++l) dst[l] = a[l] + b[l];
            }
            ++m;
        }
        n = m;
    }
}

/* Sum of the n terms of a convolution (terms[j * B + l], j in loop order), result in terms[0 .. B): pairwise (default mode)
 * or the running sum from 0 of the reference's compact mode. */
static void conv_sum(const hy_oracle_program *p, double *terms, int n, int B)
{
    if (!HY_COMPACT(p)) {
        pairwise_sum(terms, n, B);
        return;
    }
    for (int l = 0; l < B; ++l) {
        double acc = 0.;
        for (int j = 0; j < n; ++j) {
            acc = acc + terms[(size_t)j * B + l];
        }
        terms[l] = acc;
    }
}

static double pow_ebs(double base, uint32_t e)
{
    if (e == 0u) return 1.;
    if (e == 1u) return base;
    if (e % 2u == 0u) return pow_ebs(base * base, e / 2u);
    return base * pow_ebs(base * base, (e - 1u) / 2u);
}

/* pow evaluation at order 0, selected by the exponent class. */
static double pow_eval(double b, double ex)
{
    if (isfinite(ex) && ex == trunc(ex) && fabs(ex) <= 16.) {
        if (ex >= 0) return pow_ebs(b, (uint32_t)ex);
        return 1. / pow_ebs(b, (uint32_t)(-ex));
    }
    if (isfinite(ex) && ex != trunc(ex)) {
        const double y = 2 * ex;
        if (y == trunc(y) && fabs(y) <= 16.) {
            const double t = sqrt(b);
            if (y >= 0) return pow_ebs(t, (uint32_t)y);
            return 1. / pow_ebs(t, (uint32_t)(-y));
        }
    }
    return pow(b, ex);
}

/* Order-0 value of the unary functions beyond the N-body set. */
static double unary0(int kind, double x)
{
    switch (kind) {
        case K_TAN: return tan(x);
        case K_TANH: return tanh(x);
        case K_SINH: return sinh(x);
        case K_COSH: return cosh(x);
        case K_ERF: return erf(x);
        case K_SIGMOID: return 1. / (1. + exp(-x));
        case K_ASIN: return asin(x);
        case K_ACOS: return acos(x);
        case K_ATAN: return atan(x);
        case K_ASINH: return asinh(x);
        case K_ACOSH: return acosh(x);
        default: return atanh(x);
    }
}

/* Inverse of Kepler's equation E - e sin E = M (llvm_add_inv_kep_E(), src/detail/llvm_helpers_celmec.cpp:181-466, scalar
 * flavour: every lane iterates until its own convergence). Double-length reduction of M to [0, 2 pi)
 * (llvm_trig_arg_reduce(), :140-177; div2 / floor / mul2 / add of src/detail/llvm_helpers_dl.cpp:57-281), third-order
 * initial guess, Newton-Raphson safeguarded by bisection, absolute tolerance 4 eps on f(E) and on the bracket, 20
 * iterations at most (then nan). */
static double inv_kep_E(double ecc_in, double M_in)
{
    const double ecc = (!(ecc_in >= 0.) || ecc_in >= 1.) ? NAN : ecc_in;

    const double y_hi = 6.283185307179586, y_lo = 2.4492935982947064e-16;
    const double twopi_prev = nextafter(y_hi, 0.);
    /* x / y. */
    const double c = M_in / y_hi;
    const double u = c * y_hi, uu = fma(c, y_hi, -u);
    double cc = M_in - u;
    cc = cc - uu;
    cc = cc + 0.;
    cc = cc - c * y_lo;
    cc = cc / y_hi;
    const double q_hi = c + cc, q_lo = (c - q_hi) + cc;
    /* floor. */
    const double fhi = floor(q_hi);
    const double flo = (fhi == q_hi) ? floor(q_lo) : 0.;
    const double fl_hi = fhi + flo, fl_lo = (fhi - fl_hi) + flo;
    /* y * floor. */
    const double pc = y_hi * fl_hi;
    double pcc = fma(y_hi, fl_hi, -pc);
    pcc = (y_hi * fl_lo + y_lo * fl_hi) + pcc;
    const double p_hi = pc + pcc, p_lo = (pc - p_hi) + pcc;
    /* x - y * floor (llvm_dl_add(), :57-92, with the second operand negated). */
    double M;
    {
        const double x_hi = M_in, x_lo = 0., yh = -p_hi, yl = -p_lo;
        const double S = x_hi + yh, T = x_lo + yl;
        double e = S - x_hi, f = T - x_lo;
        double t1 = S - e;
        t1 = x_hi - t1;
        double s_ = yh - e;
        s_ = s_ + t1;
        t1 = T - f;
        t1 = x_lo - t1;
        double t = yl - f;
        t = t + t1;
        s_ = s_ + T;
        const double H = S + s_;
        double h = S - H;
        h = h + s_;
        h = h + t;
        M = H + h;
    }
    M = (M < 0.) ? 0. : M;
    M = (twopi_prev < M) ? twopi_prev : M;

    double sE = sin(M), cE = cos(M);
    const double e_sin = ecc * sE, e_cos = ecc * cE, e2 = ecc * ecc, cos2 = cE * cE;
    double E = ((M + e_sin) + e_sin * e_cos) + (e2 * e_sin) * (1.5 * cos2 - 0.5);
    double lb = 0., ub = twopi_prev;
    E = (E < lb) ? lb : E;
    E = (ub < E) ? ub : E;
    sE = sin(E);
    cE = cos(E);
    double fE = (E - M) - ecc * sE;
    const double tol = 4. * 2.220446049250313e-16;
    int it = 0, not_converged = 0;
    for (;;) {
        const int sgn = (0. < fE) - (fE < 0.);
        const double n_ub = (sgn >= 0) ? E : ub, n_lb = (sgn <= 0) ? E : lb;
        ub = n_ub;
        lb = n_lb;
        not_converged = (fabs(fE) > tol) && ((ub - lb) > tol);
        if (!(it < 20) || !not_converged) break;
        double nE = E - fE / (1. - ecc * cE);
        nE = (nE > ub) ? 0.5 * (E + ub) : nE;
        nE = (nE < lb) ? 0.5 * (E + lb) : nE;
        E = nE;
        sE = sin(E);
        cE = cos(E);
        fE = (E - M) - ecc * sE;
        ++it;
    }
    return (it == 20 && not_converged) ? NAN : E;
}

/* Exported for the tests (known answers of the Kepler solver). */
double hy_oracle_inv_kep_E(double ecc, double M)
{
    return inv_kep_E(ecc, M);
}

/* x mod 2 pi in [0, 2 pi) (llvm_trig_arg_reduce(), src/detail/llvm_helpers_celmec.cpp:140-177): the reduction used by
 * inv_kep_E() above, as a function of its own for the two solvers below. */
static double reduce_2pi(double x)
{
    const double y_hi = 6.283185307179586, y_lo = 2.4492935982947064e-16;
    const double twopi_prev = nextafter(y_hi, 0.);
    const double c = x / y_hi;
    const double u = c * y_hi, uu = fma(c, y_hi, -u);
    double cc = x - u;
    cc = cc - uu;
    cc = cc + 0.;
    cc = cc - c * y_lo;
    cc = cc / y_hi;
    const double q_hi = c + cc, q_lo = (c - q_hi) + cc;
    const double fhi = floor(q_hi);
    const double flo = (fhi == q_hi) ? floor(q_lo) : 0.;
    const double fl_hi = fhi + flo, fl_lo = (fhi - fl_hi) + flo;
    const double pc = y_hi * fl_hi;
    double pcc = fma(y_hi, fl_hi, -pc);
    pcc = (y_hi * fl_lo + y_lo * fl_hi) + pcc;
    const double p_hi = pc + pcc, p_lo = (pc - p_hi) + pcc;
    const double x_hi = x, x_lo = 0., yh = -p_hi, yl = -p_lo;
    const double S = x_hi + yh, T = x_lo + yl;
    double e = S - x_hi, f = T - x_lo;
    double t1 = S - e;
    t1 = x_hi - t1;
    double s_ = yh - e;
    s_ = s_ + t1;
    t1 = T - f;
    t1 = x_lo - t1;
    double t = yl - f;
    t = t + t1;
    s_ = s_ + T;
    const double H = S + s_;
    double h = S - H;
    h = h + s_;
    h = h + t;
    double r = H + h;
    r = (r < 0.) ? 0. : r;
    r = (twopi_prev < r) ? twopi_prev : r;
    return r;
}

/* Safeguarded Newton-Raphson iteration shared by the eccentric-longitude and delta-eccentric-anomaly solvers
 * (llvm_add_inv_kep_F() / llvm_add_inv_kep_DE(), src/detail/llvm_helpers_celmec.cpp:540-856, :857-1170): bracket
 * [-1, 2 pi + 1), absolute tolerance 4 eps on f and on the bracket, 20 iterations at most (then nan), result folded into
 * [0, 2 pi). which = 0: f(F) = F - lam + h cos F - k sin F (p = h, q = k); which = 1: f(DE) = DE - DM + s0 (1 - cos DE)
 * - c0 sin DE (p = s0, q = c0). */
static double kep_newton(int which, double X, double T, double p, double q)
{
    const double twopi = 6.283185307179586;
    double lb = -1., ub = nextafter(twopi + 1., 0.);
    X = (X < lb) ? lb : X;
    X = (ub < X) ? ub : X;
    double sX = sin(X), cX = cos(X);
#define KEP_F() (which == 0 ? (((X - T) + p * cX) - q * sX) : (((X - T) + p * (1. - cX)) - q * sX))
#define KEP_DF() (which == 0 ? ((1. - p * sX) - q * cX) : ((1. + p * sX) - q * cX))
    double fX = KEP_F();
    const double tol = 4. * 2.220446049250313e-16;
    int it = 0, not_converged = 0;
    for (;;) {
        const int sgn = (0. < fX) - (fX < 0.);
        const double n_ub = (sgn >= 0) ? X : ub, n_lb = (sgn <= 0) ? X : lb;
        ub = n_ub;
        lb = n_lb;
        not_converged = (fabs(fX) > tol) && ((ub - lb) > tol);
        if (!(it < 20) || !not_converged) break;
        double nX = X - fX / KEP_DF();
        nX = (nX > ub) ? 0.5 * (X + ub) : nX;
        nX = (nX < lb) ? 0.5 * (X + lb) : nX;
        X = nX;
        sX = sin(X);
        cX = cos(X);
        fX = KEP_F();
        ++it;
    }
#undef KEP_F
#undef KEP_DF
    double ret = (it == 20 && not_converged) ? NAN : X;
    ret = (ret < 0.) ? twopi + ret : ret;
    ret = (ret >= twopi) ? ret - twopi : ret;
    return ret;
}

static double inv_kep_F(double h_in, double k_in, double lam_in)
{
    const double h2 = h_in * h_in, k2 = k_in * k_in;
    const int invalid = !(h2 + k2 < 1.);
    const double h = invalid ? NAN : h_in, k = invalid ? NAN : k_in;
    const double L = reduce_2pi(lam_in);
    const double sL = sin(L), cL = cos(L);
    const double ksL_m_hcL = k * sL - h * cL, kcL_p_hsL = k * cL + h * sL;
    const double ig1 = L + ksL_m_hcL;
    const double ig2 = (k2 - h2) * (cL * sL);
    const double ig3 = (h * k) * (sL * sL - cL * cL);
    const double ig4 = (0.5 * ksL_m_hcL) * ((kcL_p_hsL * kcL_p_hsL + kcL_p_hsL * kcL_p_hsL) - ksL_m_hcL * ksL_m_hcL);
    return kep_newton(0, (ig1 + ig2) + (ig3 + ig4), L, h, k);
}

static double inv_kep_DE(double s0_in, double c0_in, double DM_in)
{
    const double s2 = s0_in * s0_in, c2 = c0_in * c0_in;
    const int invalid = !(s2 + c2 < 1.);
    const double s0 = invalid ? NAN : s0_in, c0 = invalid ? NAN : c0_in;
    const double DM = reduce_2pi(DM_in);
    const double sM = sin(DM), cM = cos(DM);
    const double A = c0 * cM - s0 * sM, Bv = c0 * sM + s0 * cM;
    const double C = Bv - s0;
    const double ig1 = DM + C, ig2 = A * C, ig3 = (0.5 * C) * ((A * A + A * A) - C * Bv);
    return kep_newton(1, (ig1 + ig2) + ig3, DM, s0, c0);
}

double hy_oracle_inv_kep_F(double h, double k, double lam)
{
    return inv_kep_F(h, k, lam);
}

double hy_oracle_inv_kep_DE(double s0, double c0, double DM)
{
    return inv_kep_DE(s0, c0, DM);
}

#define TAPE(k, u) (tape + ((size_t)(k) * n_u + (size_t)(u)) * B)

/* Value of a num/par argument, per lane. */
static inline double numpar(const hy_oracle_program *p, int a, const double *pars, int B, int l)
{
    return p->arg_type[a] == A_NUM ? p->arg_val[a] : pars[(size_t)p->arg_idx[a] * B + l];
}

/* Order-k normalised derivative of node i (u variable n_eq + i). */
static void node_diff(const hy_oracle_program *p, int i, int k, double *tape, const double *pars, const double *time,
                      int B, double *scratch)
{
    const int n_u = p->n_u;
    const int u = p->n_eq + i;
    const int a0 = p->arg_off[i];
    const int nargs = p->arg_off[i + 1] - a0;
    const int32_t *at = p->arg_type + a0;
    const int32_t *ai = p->arg_idx + a0;
    double *out = TAPE(k, u);

    switch (p->kind[i]) {
        case K_NUM_IDENTITY:
            for (int l = 0; l < B; ++l) out[l] = k == 0 ? numpar(p, a0, pars, B, l) : 0.;
            break;
        case K_TIME:
            for (int l = 0; l < B; ++l) out[l] = k == 0 ? time[l] : (k == 1 ? 1. : 0.);
            break;
        case K_PI:
            /* A constant (src/math/constants.cpp:258-273): the value at order 0, zero beyond. */
            for (int l = 0; l < B; ++l) out[l] = k == 0 ? 0x1.921fb54442d18p+1 : 0.;
            break;
        case K_SUM: {
            for (int j = 0; j < nargs; ++j) {
                double *t = scratch + (size_t)j * B;
                if (at[j] == A_UVAR) {
                    const double *v = TAPE(k, ai[j]);
                    for (int l = 0; l < B; ++l) t[l] = v[l];
                } else {
                    for (int l = 0; l < B; ++l) t[l] = k == 0 ? numpar(p, a0 + j, pars, B, l) : 0.;
                }
            }
            pairwise_sum(scratch, nargs, B);
            for (int l = 0; l < B; ++l) out[l] = scratch[l];
            break;
        }
        case K_SUB: {
            if (at[0] == A_UVAR && at[1] == A_UVAR) {
                const double *x = TAPE(k, ai[0]), *y = TAPE(k, ai[1]);
                for (int l = 0; l < B; ++l) out[l] = x[l] - y[l];
            } else if (at[0] == A_UVAR) {
                const double *x = TAPE(k, ai[0]);
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? x[l] - numpar(p, a0 + 1, pars, B, l) : x[l];
            } else if (at[1] == A_UVAR) {
                const double *y = TAPE(k, ai[1]);
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? numpar(p, a0, pars, B, l) - y[l] : -y[l];
            } else {
                for (int l = 0; l < B; ++l)
                    out[l] = k == 0 ? numpar(p, a0, pars, B, l) - numpar(p, a0 + 1, pars, B, l) : 0.;
            }
            break;
        }
        case K_PROD: {
            if (at[0] == A_UVAR && at[1] == A_UVAR) {
                for (int j = 0; j <= k; ++j) {
                    const double *x = TAPE(k - j, ai[0]), *y = TAPE(j, ai[1]);
                    double *t = scratch + (size_t)j * B;
                    for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                }
                conv_sum(p, scratch, k + 1, B);
                for (int l = 0; l < B; ++l) out[l] = scratch[l];
            } else if (at[0] != A_UVAR && at[1] != A_UVAR) {
                const int neg = (at[0] == A_NUM && p->arg_val[a0] == -1.);
                for (int l = 0; l < B; ++l) {
                    if (k != 0) {
                        out[l] = 0.;
                    } else if (neg) {
                        out[l] = -numpar(p, a0 + 1, pars, B, l);
                    } else {
                        out[l] = numpar(p, a0, pars, B, l) * numpar(p, a0 + 1, pars, B, l);
                    }
                }
            } else {
                /* numpar * var (either position). */
                const int vi = at[0] == A_UVAR ? 0 : 1;
                const int ni = 1 - vi;
                const double *x = TAPE(k, ai[vi]);
                if (ni == 0 && at[0] == A_NUM && p->arg_val[a0] == -1.) {
                    for (int l = 0; l < B; ++l) out[l] = -x[l];
                } else {
                    for (int l = 0; l < B; ++l) out[l] = numpar(p, a0 + ni, pars, B, l) * x[l];
                }
            }
            break;
        }
        case K_DIV: {
            if (at[1] == A_UVAR) {
                const double *d0 = TAPE(0, ai[1]);
                if (k == 0) {
                    for (int l = 0; l < B; ++l) {
                        const double num = at[0] == A_UVAR ? TAPE(0, ai[0])[l] : numpar(p, a0, pars, B, l);
                        out[l] = num / d0[l];
                    }
                } else {
                    for (int j = 1; j <= k; ++j) {
                        const double *x = TAPE(k - j, u), *y = TAPE(j, ai[1]);
                        double *t = scratch + (size_t)(j - 1) * B;
                        for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                    }
                    conv_sum(p, scratch, k, B);
                    if (at[0] == A_UVAR) {
                        const double *nv = TAPE(k, ai[0]);
                        for (int l = 0; l < B; ++l) out[l] = (nv[l] - scratch[l]) / d0[l];
                    } else {
                        for (int l = 0; l < B; ++l) out[l] = (-scratch[l]) / d0[l];
                    }
                }
            } else if (at[0] == A_UVAR) {
                const double *x = TAPE(k, ai[0]);
                for (int l = 0; l < B; ++l) out[l] = x[l] / numpar(p, a0 + 1, pars, B, l);
            } else {
                for (int l = 0; l < B; ++l)
                    out[l] = k == 0 ? numpar(p, a0, pars, B, l) / numpar(p, a0 + 1, pars, B, l) : 0.;
            }
            break;
        }
        case K_SUM_SQ: {
            /* Per-argument partial results in acc[arg*B + l], located after the term scratch. */
            double *acc = scratch + (size_t)(p->order + 2) * B;
            if (k % 2 == 1) {
                for (int a = 0; a < nargs; ++a) {
                    double *dst = acc + (size_t)a * B;
                    if (at[a] == A_UVAR) {
                        const int nt = (k - 1) / 2 + 1;
                        for (int j = 0; j < nt; ++j) {
                            const double *x = TAPE(k - j, ai[a]), *y = TAPE(j, ai[a]);
                            double *t = scratch + (size_t)j * B;
                            for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                        }
                        conv_sum(p, scratch, nt, B);
                        for (int l = 0; l < B; ++l) dst[l] = scratch[l];
                    } else {
                        for (int l = 0; l < B; ++l) dst[l] = 0.;
                    }
                }
                pairwise_sum(acc, nargs, B);
                for (int l = 0; l < B; ++l) out[l] = acc[l] + acc[l];
            } else {
                for (int a = 0; a < nargs; ++a) {
                    double *dst = acc + (size_t)a * B;
                    if (at[a] == A_UVAR) {
                        const double *h = TAPE(k / 2, ai[a]);
                        for (int l = 0; l < B; ++l) dst[l] = h[l] * h[l];
                        if (k > 0) {
                            const int nt = (k - 2) / 2 + 1;
                            for (int j = 0; j < nt; ++j) {
                                const double *x = TAPE(k - j, ai[a]), *y = TAPE(j, ai[a]);
                                double *t = scratch + (size_t)j * B;
                                for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                            }
                            conv_sum(p, scratch, nt, B);
                            for (int l = 0; l < B; ++l) dst[l] = (scratch[l] + scratch[l]) + dst[l];
                        }
                    } else {
                        for (int l = 0; l < B; ++l) {
                            if (k == 0) {
                                const double v = numpar(p, a0 + a, pars, B, l);
                                dst[l] = v * v;
                            } else {
                                /* 2 * 0 + 0. */
                                dst[l] = 0.;
                            }
                        }
                    }
                }
                pairwise_sum(acc, nargs, B);
                for (int l = 0; l < B; ++l) out[l] = acc[l];
            }
            break;
        }
        case K_POW: {
            /* The exponent is always a number here (pow_to_explog handles the rest). */
            const double ex = p->arg_val[a0 + 1];
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? pow_eval(numpar(p, a0, pars, B, l), ex) : 0.;
                break;
            }
            const int b = ai[0];
            if (k == 0) {
                const double *x = TAPE(0, b);
                for (int l = 0; l < B; ++l) out[l] = pow_eval(x[l], ex);
            } else if (ex == .5) {
                /* sqrt special case. */
                const double *a_0 = TAPE(0, u), *bn = TAPE(k, b);
                int nt = 0;
                const int jmax = (k % 2 == 1) ? (k - 1) / 2 : (k - 2) / 2;
                for (int j = 1; j <= jmax; ++j, ++nt) {
                    const double *x = TAPE(k - j, u), *y = TAPE(j, u);
                    double *t = scratch + (size_t)nt * B;
                    for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                }
                double *fac = scratch + (size_t)(p->order + 2) * B;
                for (int l = 0; l < B; ++l) fac[l] = bn[l];
                if (k % 2 == 0) {
                    const double *hh = TAPE(k / 2, u);
                    for (int l = 0; l < B; ++l) fac[l] = fac[l] - hh[l] * hh[l];
                }
                if (nt > 0) {
                    conv_sum(p, scratch, nt, B);
                    for (int l = 0; l < B; ++l) fac[l] = fac[l] - (scratch[l] + scratch[l]);
                }
                for (int l = 0; l < B; ++l) out[l] = fac[l] / (a_0[l] + a_0[l]);
            } else if (ex == 2.) {
                /* square special case. */
                if (k % 2 == 1) {
                    const int nt = (k - 1) / 2 + 1;
                    for (int j = 0; j < nt; ++j) {
                        const double *x = TAPE(k - j, b), *y = TAPE(j, b);
                        double *t = scratch + (size_t)j * B;
                        for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                    }
                    conv_sum(p, scratch, nt, B);
                    for (int l = 0; l < B; ++l) out[l] = scratch[l] + scratch[l];
                } else {
                    const double *hh = TAPE(k / 2, b);
                    const int nt = (k - 2) / 2 + 1;
                    for (int j = 0; j < nt; ++j) {
                        const double *x = TAPE(k - j, b), *y = TAPE(j, b);
                        double *t = scratch + (size_t)j * B;
                        for (int l = 0; l < B; ++l) t[l] = x[l] * y[l];
                    }
                    conv_sum(p, scratch, nt, B);
                    for (int l = 0; l < B; ++l) out[l] = (scratch[l] + scratch[l]) + hh[l] * hh[l];
                }
            } else {
                const double *b0 = TAPE(0, b);
                for (int j = 0; j < k; ++j) {
                    const double *x = TAPE(k - j, b), *y = TAPE(j, u);
                    const double sf = (double)k * ex - (double)j * (ex + 1.);
                    double *t = scratch + (size_t)j * B;
                    for (int l = 0; l < B; ++l) t[l] = sf * (x[l] * y[l]);
                }
                conv_sum(p, scratch, k, B);
                for (int l = 0; l < B; ++l) out[l] = scratch[l] / ((double)k * b0[l]);
            }
            break;
        }
        case K_SIN:
        case K_COS: {
            const int is_sin = p->kind[i] == K_SIN;
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) {
                    const double v = numpar(p, a0, pars, B, l);
                    out[l] = k == 0 ? (is_sin ? sin(v) : cos(v)) : 0.;
                }
                break;
            }
            const int b = ai[0];
            if (k == 0) {
                const double *x = TAPE(0, b);
                for (int l = 0; l < B; ++l) out[l] = is_sin ? sin(x[l]) : cos(x[l]);
            } else {
                const int d = p->dep[i];
                for (int j = 1; j <= k; ++j) {
                    const double *x = TAPE(k - j, d), *y = TAPE(j, b);
                    double *t = scratch + (size_t)(j - 1) * B;
                    for (int l = 0; l < B; ++l) t[l] = (double)j * (x[l] * y[l]);
                }
                conv_sum(p, scratch, k, B);
                const double dv = is_sin ? (double)k : -(double)k;
                for (int l = 0; l < B; ++l) out[l] = scratch[l] / dv;
            }
            break;
        }
        case K_EXP: {
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? exp(numpar(p, a0, pars, B, l)) : 0.;
                break;
            }
            const int b = ai[0];
            if (k == 0) {
                const double *x = TAPE(0, b);
                for (int l = 0; l < B; ++l) out[l] = exp(x[l]);
            } else {
                for (int j = 1; j <= k; ++j) {
                    const double *x = TAPE(k - j, u), *y = TAPE(j, b);
                    double *t = scratch + (size_t)(j - 1) * B;
                    for (int l = 0; l < B; ++l) t[l] = (double)j * (x[l] * y[l]);
                }
                conv_sum(p, scratch, k, B);
                for (int l = 0; l < B; ++l) out[l] = scratch[l] / (double)k;
            }
            break;
        }
        case K_LOG: {
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? log(numpar(p, a0, pars, B, l)) : 0.;
                break;
            }
            const int b = ai[0];
            const double *b0 = TAPE(0, b);
            if (k == 0) {
                for (int l = 0; l < B; ++l) out[l] = log(b0[l]);
            } else {
                const double *bn = TAPE(k, b);
                double *ret = scratch + (size_t)(p->order + 2) * B;
                for (int l = 0; l < B; ++l) ret[l] = (double)k * bn[l];
                if (k > 1) {
                    for (int j = 1; j < k; ++j) {
                        const double *x = TAPE(k - j, b), *y = TAPE(j, u);
                        double *t = scratch + (size_t)(j - 1) * B;
                        for (int l = 0; l < B; ++l) t[l] = (double)j * (x[l] * y[l]);
                    }
                    conv_sum(p, scratch, k - 1, B);
                    for (int l = 0; l < B; ++l) ret[l] = ret[l] - scratch[l];
                }
                for (int l = 0; l < B; ++l) out[l] = ret[l] / ((double)k * b0[l]);
            }
            break;
        }
        case K_TAN:
        case K_TANH:
        case K_SINH:
        case K_COSH:
        case K_ERF:
        case K_SIGMOID: {
            /* Forward rules (src/math/tan.cpp:105-131, tanh.cpp, sinh.cpp, cosh.cpp, erf.cpp, sigmoid.cpp:150-172):
             * k a^[k] = pairwise_sum_j(j * (X^[k-j] * b^[j])), X = hidden dependency (a - a^2 for the sigmoid). */
            const int kind = p->kind[i];
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? unary0(kind, numpar(p, a0, pars, B, l)) : 0.;
                break;
            }
            const int b = ai[0];
            if (k == 0) {
                const double *x = TAPE(0, b);
                for (int l = 0; l < B; ++l) out[l] = unary0(kind, x[l]);
                break;
            }
            const int d = p->dep[i];
            for (int j = 1; j <= k; ++j) {
                const double *x = TAPE(k - j, d), *y = TAPE(j, b), *self = TAPE(k - j, u);
                double *t = scratch + (size_t)(j - 1) * B;
                for (int l = 0; l < B; ++l) {
                    const double xv = kind == K_SIGMOID ? (self[l] - x[l]) : x[l];
                    t[l] = (double)j * (xv * y[l]);
                }
            }
            conv_sum(p, scratch, k, B);
            {
                const double *bk = TAPE(k, b);
                for (int l = 0; l < B; ++l) {
                    const double acc = scratch[l] / (double)k;
                    out[l] = kind == K_TAN ? (bk[l] + acc)
                                           : (kind == K_TANH ? (bk[l] - acc)
                                                             : (kind == K_ERF ? 1.1283791670955126 * acc : acc));
                }
            }
            break;
        }
        case K_ASIN:
        case K_ACOS:
        case K_ATAN:
        case K_ASINH:
        case K_ACOSH:
        case K_ATANH: {
            /* Inverse rules (src/math/asin.cpp:140-178, acos.cpp, atan.cpp:125-164, atanh.cpp, asinh.cpp, acosh.cpp):
             * a^[k] = (k b^[k] -+ pairwise_sum_{j<k}(j * (c^[k-j] * a^[j]))) / (k D). */
            const int kind = p->kind[i];
            if (at[0] != A_UVAR) {
                for (int l = 0; l < B; ++l) out[l] = k == 0 ? unary0(kind, numpar(p, a0, pars, B, l)) : 0.;
                break;
            }
            const int b = ai[0];
            if (k == 0) {
                const double *x = TAPE(0, b);
                for (int l = 0; l < B; ++l) out[l] = unary0(kind, x[l]);
                break;
            }
            const int d = p->dep[i];
            const double *c0 = TAPE(0, d), *bk = TAPE(k, b);
            double *D = scratch + (size_t)(p->order + 2) * B;
            for (int l = 0; l < B; ++l) {
                D[l] = kind == K_ACOS ? -c0[l] : (kind == K_ATAN ? (c0[l] + 1.) : (kind == K_ATANH ? (1. - c0[l]) : c0[l]));
            }
            if (k == 1) {
                for (int l = 0; l < B; ++l) out[l] = bk[l] / D[l];
                break;
            }
            for (int j = 1; j < k; ++j) {
                const double *x = TAPE(k - j, d), *y = TAPE(j, u);
                double *t = scratch + (size_t)(j - 1) * B;
                for (int l = 0; l < B; ++l) t[l] = (double)j * (x[l] * y[l]);
            }
            conv_sum(p, scratch, k - 1, B);
            for (int l = 0; l < B; ++l) {
                double ret = (double)k * bk[l];
                ret = (kind == K_ACOS || kind == K_ATANH) ? (ret + scratch[l]) : (ret - scratch[l]);
                out[l] = ret / ((double)k * D[l]);
            }
            break;
        }
        case K_ATAN2: {
            /* a = atan2(b, c), d = b^2 + c^2 (src/math/atan2.cpp:113-330):
             * a^[k] = (k (c^[0] b^[k] - b^[0] c^[k]) + pairwise_sum_{j=1..k-1} j (c^[k-j] b^[j] - b^[k-j] c^[j] - d^[k-j] a^[j]))
             *         / (k d^[0]); constant arguments drop their terms. */
            const int vy = at[0] == A_UVAR, vx = at[1] == A_UVAR;
            if (k == 0) {
                for (int l = 0; l < B; ++l) {
                    const double y = vy ? TAPE(0, ai[0])[l] : numpar(p, a0, pars, B, l);
                    const double x = vx ? TAPE(0, ai[1])[l] : numpar(p, a0 + 1, pars, B, l);
                    out[l] = atan2(y, x);
                }
                break;
            }
            if (!vy && !vx) {
                for (int l = 0; l < B; ++l) out[l] = 0.;
                break;
            }
            const int d = p->dep[i];
            const double n = (double)k;
            double *dividend = scratch + (size_t)(p->order + 2) * B;
            for (int l = 0; l < B; ++l) {
                if (vy && vx) {
                    double t = TAPE(0, ai[1])[l] * TAPE(k, ai[0])[l];
                    t = t - TAPE(0, ai[0])[l] * TAPE(k, ai[1])[l];
                    dividend[l] = n * t;
                } else if (vy) {
                    dividend[l] = n * (numpar(p, a0 + 1, pars, B, l) * TAPE(k, ai[0])[l]);
                } else {
                    dividend[l] = -n * (numpar(p, a0, pars, B, l) * TAPE(k, ai[1])[l]);
                }
            }
            if (k > 1) {
                for (int j = 1; j < k; ++j) {
                    double *t = scratch + (size_t)(j - 1) * B;
                    const double *dnj = TAPE(k - j, d), *aj = TAPE(j, u);
                    for (int l = 0; l < B; ++l) {
                        const double t3 = dnj[l] * aj[l];
                        if (vy && vx) {
                            const double t1 = TAPE(k - j, ai[1])[l] * TAPE(j, ai[0])[l];
                            const double t2 = TAPE(k - j, ai[0])[l] * TAPE(j, ai[1])[l];
                            t[l] = (double)j * ((t1 - t2) - t3);
                        } else {
                            t[l] = -(double)j * t3;
                        }
                    }
                }
                conv_sum(p, scratch, k - 1, B);
                for (int l = 0; l < B; ++l) dividend[l] = dividend[l] + scratch[l];
            }
            for (int l = 0; l < B; ++l) out[l] = dividend[l] / (n * TAPE(0, d)[l]);
            break;
        }
        case K_KEPE: {
            /* a = E(e, M), c = e cos(a) (dep), d = sin(a) (dep2) (src/math/kepE.cpp:140-355):
             * a^[k] = (k (e^[k] d^[0] + M^[k]) + pairwise_sum_{j=1..k-1} j (c^[k-j] a^[j] + d^[k-j] e^[j])) / (k (1 - c^[0])). */
            const int ve = at[0] == A_UVAR, vm = at[1] == A_UVAR;
            if (k == 0) {
                for (int l = 0; l < B; ++l) {
                    const double e = ve ? TAPE(0, ai[0])[l] : numpar(p, a0, pars, B, l);
                    const double M = vm ? TAPE(0, ai[1])[l] : numpar(p, a0 + 1, pars, B, l);
                    out[l] = inv_kep_E(e, M);
                }
                break;
            }
            if (!ve && !vm) {
                for (int l = 0; l < B; ++l) out[l] = 0.;
                break;
            }
            const int c = p->dep[i], d = p->dep2[i];
            const double n = (double)k;
            double *dividend = scratch + (size_t)(p->order + 2) * B;
            for (int l = 0; l < B; ++l) {
                if (ve && vm) {
                    double t = TAPE(k, ai[0])[l] * TAPE(0, d)[l];
                    t = t + TAPE(k, ai[1])[l];
                    dividend[l] = n * t;
                } else if (ve) {
                    dividend[l] = n * (TAPE(k, ai[0])[l] * TAPE(0, d)[l]);
                } else {
                    dividend[l] = n * TAPE(k, ai[1])[l];
                }
            }
            if (k > 1) {
                for (int j = 1; j < k; ++j) {
                    double *t = scratch + (size_t)(j - 1) * B;
                    const double *cnj = TAPE(k - j, c), *aj = TAPE(j, u);
                    for (int l = 0; l < B; ++l) {
                        if (ve) {
                            double tmp = TAPE(k - j, d)[l] * TAPE(j, ai[0])[l];
                            tmp = cnj[l] * aj[l] + tmp;
                            t[l] = (double)j * tmp;
                        } else {
                            t[l] = (double)j * (cnj[l] * aj[l]);
                        }
                    }
                }
                conv_sum(p, scratch, k - 1, B);
                for (int l = 0; l < B; ++l) dividend[l] = dividend[l] + scratch[l];
            }
            for (int l = 0; l < B; ++l) out[l] = dividend[l] / (n * (1. - TAPE(0, c)[l]));
            break;
        }
        case K_KEPF:
        case K_KEPDE: {
            /* a = kepF(h, k, lam) / kepDE(s0, c0, DM) with the hidden dependencies c (dep), d (dep2), e = sin a (dep3),
             * f = cos a (dep4): c = h e, d = k f (src/math/kepF.cpp:110-156) resp. c = c0 f, d = s0 e. A numerical or
             * parameter argument has no coefficients beyond order 0. kepF (src/math/kepF.cpp:633-711):
             *   a^[n] = (n (k^[n] e^[0] - h^[n] f^[0] + lam^[n]) + sum_{j=1..n-1} j (a^[j] (c^[n-j] + d^[n-j]) + k^[j] e^[n-j]
             *           - h^[j] f^[n-j])) / (n (1 - c^[0] - d^[0]));
             * kepDE (same derivation from DE - c0 sin DE + s0 (1 - cos DE) = DM):
             *   a^[n] = (n (DM^[n] - s0^[n] + c0^[n] e^[0] + s0^[n] f^[0]) + sum_{j=1..n-1} j (c0^[j] e^[n-j] + s0^[j] f^[n-j]
             *           - a^[j] (d^[n-j] - c^[n-j]))) / (n (1 - c^[0] + d^[0])). */
            const int isF = p->kind[i] == K_KEPF;
            const int v0 = at[0] == A_UVAR, v1 = at[1] == A_UVAR, v2 = at[2] == A_UVAR;
            if (k == 0) {
                for (int l = 0; l < B; ++l) {
                    const double x0 = v0 ? TAPE(0, ai[0])[l] : numpar(p, a0, pars, B, l);
                    const double x1 = v1 ? TAPE(0, ai[1])[l] : numpar(p, a0 + 1, pars, B, l);
                    const double x2 = v2 ? TAPE(0, ai[2])[l] : numpar(p, a0 + 2, pars, B, l);
                    out[l] = isF ? inv_kep_F(x0, x1, x2) : inv_kep_DE(x0, x1, x2);
                }
                break;
            }
            const int c = p->dep[i], d = p->dep2[i], e = p->dep3[i], f = p->dep4[i];
            const double n = (double)k;
#define XC(arg, ord, l) ((at[arg] == A_UVAR) ? TAPE(ord, ai[arg])[l] : 0.)
            double *dividend = scratch + (size_t)(p->order + 2) * B;
            for (int l = 0; l < B; ++l) {
                if (isF) {
                    dividend[l] = n * ((XC(1, k, l) * TAPE(0, e)[l] - XC(0, k, l) * TAPE(0, f)[l]) + XC(2, k, l));
                } else {
                    dividend[l] = n * (((XC(2, k, l) - XC(0, k, l)) + XC(1, k, l) * TAPE(0, e)[l]) + XC(0, k, l) * TAPE(0, f)[l]);
                }
            }
            if (k > 1) {
                for (int j = 1; j < k; ++j) {
                    double *t = scratch + (size_t)(j - 1) * B;
                    for (int l = 0; l < B; ++l) {
                        const double aj = TAPE(j, u)[l];
                        const double cnj = TAPE(k - j, c)[l], dnj = TAPE(k - j, d)[l], enj = TAPE(k - j, e)[l], fnj = TAPE(k - j, f)[l];
                        if (isF) {
                            const double t1 = aj * (cnj + dnj);
                            const double t2 = XC(1, j, l) * enj - XC(0, j, l) * fnj;
                            t[l] = (double)j * (t1 + t2);
                        } else {
                            const double t1 = XC(1, j, l) * enj + XC(0, j, l) * fnj;
                            const double t2 = aj * (dnj - cnj);
                            t[l] = (double)j * (t1 - t2);
                        }
                    }
                }
                conv_sum(p, scratch, k - 1, B);
                for (int l = 0; l < B; ++l) dividend[l] = dividend[l] + scratch[l];
            }
#undef XC
            for (int l = 0; l < B; ++l) {
                const double den = isF ? ((1. - TAPE(0, c)[l]) - TAPE(0, d)[l]) : ((1. - TAPE(0, c)[l]) + TAPE(0, d)[l]);
                out[l] = dividend[l] / (n * den);
            }
            break;
        }
        case K_RELU:
        case K_RELUP:
        case K_SELECT:
        case K_LAND:
        case K_LOR:
        case K_REL_EQ:
        case K_REL_NEQ:
        case K_REL_LT:
        case K_REL_GT:
        case K_REL_LTE:
        case K_REL_GTE: {
            /* Piecewise functions (src/math/relu.cpp:144-178, :392-424; select.cpp:88-131; relational.cpp:197-238;
             * logical.cpp:93-127): the branch is chosen by the order-0 values; truth values are 1 / 0 at order 0 and 0
             * beyond; relu / select pick the order-k coefficient of the chosen branch. */
            const int kind = p->kind[i];
            for (int l = 0; l < B; ++l) {
#define ARG0(j) (at[j] == A_UVAR ? TAPE(0, ai[j])[l] : numpar(p, a0 + (j), pars, B, l))
#define ARGK(j) (at[j] == A_UVAR ? TAPE(k, ai[j])[l] : (k == 0 ? numpar(p, a0 + (j), pars, B, l) : 0.))
                double r;
                if (kind == K_RELU) {
                    const double x = ARGK(0);
                    r = (ARG0(0) > 0.) ? x : p->arg_val[a0 + 1] * x;
                } else if (kind == K_RELUP) {
                    r = (k == 0) ? ((ARG0(0) > 0.) ? 1. : p->arg_val[a0 + 1]) : 0.;
                } else if (kind == K_SELECT) {
                    r = (ARG0(0) != 0.) ? ARGK(1) : ARGK(2);
                } else if (k != 0) {
                    r = 0.;
                } else if (kind == K_LAND || kind == K_LOR) {
                    int t = (kind == K_LAND);
                    for (int j = 0; j < nargs; ++j) {
                        const int nz = ARG0(j) != 0.;
                        t = (kind == K_LAND) ? (t && nz) : (t || nz);
                    }
                    r = t ? 1. : 0.;
                } else {
                    const double x = ARG0(0), y = ARG0(1);
                    int t;
                    switch (kind) {
                        case K_REL_EQ: t = x == y; break;
                        case K_REL_NEQ: t = (x < y) || (x > y); break; /* ordered: false with a nan operand */
                        case K_REL_LT: t = x < y; break;
                        case K_REL_GT: t = x > y; break;
                        case K_REL_LTE: t = x <= y; break;
                        default: t = x >= y; break;
                    }
                    r = t ? 1. : 0.;
                }
                out[l] = r;
#undef ARG0
#undef ARGK
            }
            break;
        }
        default:
            for (int l = 0; l < B; ++l) out[l] = NAN;
    }
}

/* Order-k coefficients of the state variables. */
static void sv_diff(const hy_oracle_program *p, int k, double *tape, const double *pars, int B)
{
    const int n_u = p->n_u;
    for (int i = 0; i < p->n_eq; ++i) {
        double *out = TAPE(k, i);
        if (p->sv_type[i] == A_UVAR) {
            const double *x = TAPE(k - 1, p->sv_idx[i]);
            for (int l = 0; l < B; ++l) out[l] = x[l] / (double)k;
        } else {
            for (int l = 0; l < B; ++l) {
                if (k == 1) {
                    out[l] = p->sv_type[i] == A_NUM ? p->sv_val[i] : pars[(size_t)p->sv_idx[i] * B + l];
                } else {
                    out[l] = 0.;
                }
            }
        }
    }
}

/* Number of doubles of scratch needed by hy_oracle_step() for batch size B. */
size_t hy_oracle_scratch_size(const hy_oracle_program *p, int B)
{
    int max_args = 2;
    for (int i = 0; i < p->n_nodes; ++i) {
        const int n = p->arg_off[i + 1] - p->arg_off[i];
        if (n > max_args) max_args = n;
    }
    const size_t tape = ((size_t)p->n_u * (size_t)p->order + (size_t)p->n_eq) * (size_t)B;
    const size_t terms = (size_t)(p->order + 2 + max_args + 2) * (size_t)B;
    /* + reduction area of the step-size selector (n_eq + 3) * B and the step_impl() copy of the limits. */
    return tape + terms + 8u * (size_t)B + ((size_t)p->n_eq + 4u) * (size_t)B;
}

/* Same for hy_oracle_step_e() (full order for every u variable). */
size_t hy_oracle_scratch_size_e(const hy_oracle_program *p, int B)
{
    /* (+ n_u rows: the event equations take part in the reductions of the step-size selector.) */
    return hy_oracle_scratch_size(p, B) + (2u * (size_t)p->n_u + 8u) * (size_t)B;
}

static double rhofac(int order)
{
    /* exp(-7/10 / (order - 1)) / (e * e), folded in double precision. */
    const double m7_10 = -7. / 10.;
    const double e2 = exp(1.) * exp(1.);
    return exp(m7_10 / (double)(order - 1)) / e2;
}

/*
 * One step of the batch integrator for B lanes in lock-step (the JIT'd `step` function of the
 * reference, default mode). h_inout: in = signed max step, out = step taken.
 * tc (nullable): tc[(var * (order + 1) + k) * B + lane].
 */
/* Optional compiled replacement of the interpreter for the computation of the jet (oracle/compiled_baseline.py): a
 * generated, fully unrolled, vectorised function which fills the tape of one batch of hook_B systems with exactly
 * the operations of node_diff() / sv_diff(). Process-wide; used by the CPU baseline of bench.py and checked against
 * the interpreter bit by bit in tests/test_oracle_golden.py. */
typedef void (*hy_jet_hook_t)(const double *state, const double *pars, const double *time, double *tape);
static hy_jet_hook_t g_jet_hook = NULL;
static int g_hook_B = 0;
static uint64_t g_hook_hash = 0;

/* Content hash of a program (FNV-1a over sizes, node kinds, arguments, hidden dependencies and the definitions of the
 * state variables): the compiled jet function is only used for the program it was generated from, not for any program
 * which happens to share (batch width, n_u, order) with it. */
uint64_t hy_oracle_program_hash(const hy_oracle_program *p)
{
    uint64_t h = 1469598103934665603ull;
#define HY_FNV(ptr, nbytes)                                                                                           \
    do {                                                                                                              \
        const unsigned char *q_ = (const unsigned char *)(ptr);                                                       \
        for (size_t i_ = 0; i_ < (size_t)(nbytes); ++i_) {                                                            \
            h ^= q_[i_];                                                                                              \
            h *= 1099511628211ull;                                                                                    \
        }                                                                                                             \
    } while (0)
    HY_FNV(&p->n_eq, sizeof(int32_t) * 6);
    const int32_t n_args = p->arg_off[p->n_nodes];
    HY_FNV(p->kind, sizeof(int32_t) * (size_t)p->n_nodes);
    HY_FNV(p->arg_off, sizeof(int32_t) * (size_t)(p->n_nodes + 1));
    HY_FNV(p->arg_type, sizeof(int32_t) * (size_t)n_args);
    HY_FNV(p->arg_idx, sizeof(int32_t) * (size_t)n_args);
    HY_FNV(p->arg_val, sizeof(double) * (size_t)n_args);
    HY_FNV(p->dep, sizeof(int32_t) * (size_t)p->n_nodes);
    HY_FNV(p->dep2, sizeof(int32_t) * (size_t)p->n_nodes);
    HY_FNV(p->sv_type, sizeof(int32_t) * (size_t)p->n_eq);
    HY_FNV(p->sv_idx, sizeof(int32_t) * (size_t)p->n_eq);
    HY_FNV(p->sv_val, sizeof(double) * (size_t)p->n_eq);
#undef HY_FNV
    return h;
}

static volatile uint64_t g_hook_gen = 0;
void hy_oracle_set_jet_hook(hy_jet_hook_t f, int B, uint64_t program_hash)
{
    g_jet_hook = f;
    g_hook_B = B;
    g_hook_hash = program_hash;
    ++g_hook_gen;
}

/* Does the installed hook belong to program p? (The hash is computed once per thread, program object and installation.) */
static int jet_hook_matches(const hy_oracle_program *p, int B)
{
    static _Thread_local const hy_oracle_program *tl_p = NULL;
    static _Thread_local uint64_t tl_gen = 0;
    static _Thread_local int32_t tl_nu = 0, tl_nn = 0;
    static _Thread_local int tl_match = 0;
    if (g_jet_hook == NULL || B != g_hook_B) {
        return 0;
    }
    if (p != tl_p || tl_gen != g_hook_gen || tl_nu != p->n_u || tl_nn != p->n_nodes) {
        tl_p = p;
        tl_gen = g_hook_gen;
        tl_nu = p->n_u;
        tl_nn = p->n_nodes;
        tl_match = hy_oracle_program_hash(p) == g_hook_hash;
    }
    return tl_match;
}

static void step_core(const hy_oracle_program *p, int B, double *state, const double *pars, const double *time,
                      double *h_inout, double *tc, double *scratch_mem, int with_events, const int32_t *ev_u, int n_ev,
                      double *ev_tc, double *max_abs_state)
{
    const int n_eq = p->n_eq, n_u = p->n_u, order = p->order;
    double *tape = scratch_mem;
    /* NOTE: with events the order-p coefficients of all the u variables are needed (src/taylor_02.cpp:1016-1190). */
    const size_t tape_rows = with_events ? (size_t)n_u * (size_t)(order + 1) : ((size_t)n_u * (size_t)order + (size_t)n_eq);
    double *scratch = tape + tape_rows * (size_t)B;

    if (!with_events && jet_hook_matches(p, B)) {
        g_jet_hook(state, pars, time, tape);
    } else {
        for (int i = 0; i < n_eq; ++i) {
            memcpy(TAPE(0, i), state + (size_t)i * B, sizeof(double) * (size_t)B);
        }
        for (int i = 0; i < p->n_nodes; ++i) node_diff(p, i, 0, tape, pars, time, B, scratch);
        for (int k = 1; k < order; ++k) {
            sv_diff(p, k, tape, pars, B);
            for (int i = 0; i < p->n_nodes; ++i) node_diff(p, i, k, tape, pars, time, B, scratch);
        }
        sv_diff(p, order, tape, pars, B);
        if (with_events) {
            for (int i = 0; i < p->n_nodes; ++i) node_diff(p, i, order, tape, pars, time, B, scratch);
        }
    }

    /* Step size: pairwise max reduction over the variables (default mode). */
    /* Per-step temporaries live at the end of the caller-provided scratch (no allocation in the hot path). */
    /* With events the event equations take part in the three norms: taylor_determine_h() iterates up to
     * n_eq + n_sv_funcs (src/taylor_00.cpp:209-219), state variables first, then the event equations in order. */
    const int n_red = n_eq + (with_events ? n_ev : 0);
    double *mx = scratch_mem + (with_events ? hy_oracle_scratch_size_e(p, B) : hy_oracle_scratch_size(p, B))
                 - ((size_t)n_eq + (with_events ? (size_t)n_u : 0u) + 4u) * (size_t)B;
    double *red = mx + (size_t)3 * B;
    const int ks[3] = {0, order, order - 1};
    for (int q = 0; q < 3; ++q) {
        for (int i = 0; i < n_red; ++i) {
            const double *x = TAPE(ks[q], i < n_eq ? i : ev_u[i - n_eq]);
            for (int l = 0; l < B; ++l) red[(size_t)i * B + l] = fabs(x[l]);
        }
        int n = n_red;
        while (n != 1) {
            int mcount = 0;
            for (int i = 0; i < n; i += 2) {
                double *dst = red + (size_t)mcount * B;
                const double *a = red + (size_t)i * B;
                if (i + 1 == n) {
                    if (dst != a)
                        for (int l = 0; l < B; ++l) dst[l] = a[l];
                } else {
                    const double *b = red + (size_t)(i + 1) * B;
                    /* max(a, b) = (a < b) ? b : a */
                    for (int l = 0; l < B; ++l) dst[l] = (a[l] < b[l]) ? b[l] : a[l];
                }
                ++mcount;
            }
            n = mcount;
        }
        for (int l = 0; l < B; ++l) mx[(size_t)q * B + l] = red[l];
    }

    const double rf = rhofac(order);
    const double inv_o = 1. / (double)order, inv_om1 = 1. / (double)(order - 1);
    double *h = scratch; /* B */
    for (int l = 0; l < B; ++l) {
        const double m0 = mx[l], mo = mx[(size_t)B + l], mom1 = mx[(size_t)2 * B + l];
        const double num_rho = (m0 <= 1.) ? 1. : m0;
        const double rho_o = pow(num_rho / mo, inv_o);
        const double rho_om1 = pow(num_rho / mom1, inv_om1);
        /* min(a, b) = (b < a) ? b : a */
        const double rho_m = (rho_om1 < rho_o) ? rho_om1 : rho_o;
        double hh = rho_m * rf;
        const double max_h = h_inout[l];
        const double amh = fabs(max_h);
        hh = (amh < hh) ? amh : hh;
        if (max_h < 0.) hh = -1. * hh;
        else hh = 1. * hh;
        h[l] = hh;
    }

    if (with_events) {
        /* step_e: no state update (the caller truncates the step at the first terminal event). */
        for (int l = 0; l < B; ++l) max_abs_state[l] = mx[l];
        for (int e = 0; e < n_ev; ++e) {
            for (int k = 0; k <= order; ++k) {
                memcpy(ev_tc + ((size_t)e * (size_t)(order + 1) + (size_t)k) * (size_t)B, TAPE(k, ev_u[e]),
                       sizeof(double) * (size_t)B);
            }
        }
    } else if (HY_HA(p)) {
        double *cur_h = scratch + (size_t)B;
        double *comp = scratch + (size_t)2 * B;
        for (int i = 0; i < n_eq; ++i) {
            const double *c0 = TAPE(0, i);
            double *r = state + (size_t)i * B;
            for (int l = 0; l < B; ++l) {
                r[l] = c0[l];
                comp[l] = 0.;
                cur_h[l] = h[l];
            }
            for (int k = 1; k <= order; ++k) {
                const double *ck = TAPE(k, i);
                for (int l = 0; l < B; ++l) {
                    const double tmp = ck[l] * cur_h[l];
                    const double y = tmp - comp[l];
                    const double t = r[l] + y;
                    comp[l] = (t - r[l]) - y;
                    r[l] = t;
                    cur_h[l] = cur_h[l] * h[l];
                }
            }
        }
    } else {
        for (int i = 0; i < n_eq; ++i) {
            double *r = state + (size_t)i * B;
            const double *cp = TAPE(order, i);
            for (int l = 0; l < B; ++l) r[l] = cp[l];
            for (int k = 1; k <= order; ++k) {
                const double *ck = TAPE(order - k, i);
                for (int l = 0; l < B; ++l) r[l] = ck[l] + r[l] * h[l];
            }
        }
    }

    if (tc != NULL) {
        for (int i = 0; i < n_eq; ++i) {
            for (int k = 0; k <= order; ++k) {
                memcpy(tc + ((size_t)i * (size_t)(order + 1) + (size_t)k) * (size_t)B, TAPE(k, i),
                       sizeof(double) * (size_t)B);
            }
        }
    }

    for (int l = 0; l < B; ++l) h_inout[l] = h[l];
}

void hy_oracle_step(const hy_oracle_program *p, int B, double *state, const double *pars, const double *time,
                    double *h_inout, double *tc, double *scratch_mem)
{
    step_core(p, B, state, pars, time, h_inout, tc, scratch_mem, 0, NULL, 0, NULL, NULL);
}

/*
 * The stepper with events of the reference (taylor_add_adaptive_step_with_events(), src/taylor_00.cpp:592-710):
 * jets of the state variables (tc) and of the event equations (ev_tc[(event * (order + 1) + k) * B + lane], u variables
 * ev_u), step size and max |x_i|; the state is NOT updated.
 */
void hy_oracle_step_e(const hy_oracle_program *p, int B, const double *state, const double *pars, const double *time,
                      double *h_inout, double *tc, const int32_t *ev_u, int n_ev, double *ev_tc, double *max_abs_state,
                      double *scratch_mem)
{
    step_core(p, B, (double *)state, pars, time, h_inout, tc, scratch_mem, 1, ev_u, n_ev, ev_tc, max_abs_state);
}

/* ---- double-length arithmetic ---- */
typedef struct {
    double hi, lo;
} dfloat;

static inline void eft_add_knuth(double a, double b, double *x, double *y)
{
    *x = a + b;
    const double z = *x - a;
    *y = (a - (*x - z)) + (b - z);
}

static inline void eft_add_dekker(double a, double b, double *x, double *y)
{
    *x = a + b;
    *y = (a - *x) + b;
}

static inline dfloat df_add(dfloat a, dfloat b)
{
    double x_hi, y_hi, x_lo, y_lo, u, v;
    eft_add_knuth(a.hi, b.hi, &x_hi, &y_hi);
    eft_add_knuth(a.lo, b.lo, &x_lo, &y_lo);
    eft_add_dekker(x_hi, y_hi + x_lo, &u, &v);
    double u2, v2;
    eft_add_dekker(u, v + y_lo, &u2, &v2);
    dfloat r = {u2, v2};
    return r;
}

static inline dfloat df_sub(dfloat a, dfloat b)
{
    dfloat nb = {-b.hi, -b.lo};
    return df_add(a, nb);
}

static inline int df_lt(dfloat x, dfloat y)
{
    return (x.hi < y.hi) || (x.hi == y.hi && x.lo < y.lo);
}

/* Exposed for the dfloat parity tests. */
void hy_oracle_dfloat_add(double ahi, double alo, double bhi, double blo, double *rhi, double *rlo)
{
    dfloat a = {ahi, alo}, b = {bhi, blo};
    dfloat r = df_add(a, b);
    *rhi = r.hi;
    *rlo = r.lo;
}

/*
 * step_impl() for one batch (reference: src/taylor_adaptive_batch.cpp:632-727).
 * outcome[l], h_out[l] = m_step_res.
 */
void hy_oracle_step_impl(const hy_oracle_program *p, int B, double *state, const double *pars, double *time_hi,
                         double *time_lo, const double *max_delta_ts, double *tc, int64_t *outcome, double *h_out,
                         double *scratch_mem)
{
    double *dts = scratch_mem + hy_oracle_scratch_size(p, B) - (size_t)B;
    memcpy(dts, max_delta_ts, sizeof(double) * (size_t)B);

    hy_oracle_step(p, B, state, pars, time_hi, dts, tc, scratch_mem);

    for (int l = 0; l < B; ++l) {
        const double h = dts[l];
        dfloat t = {time_hi[l], time_lo[l]};
        dfloat hh = {h, 0.};
        const dfloat nt = df_add(t, hh);
        time_hi[l] = nt.hi;
        time_lo[l] = nt.lo;
        h_out[l] = h;
        int nf = !(isfinite(nt.hi) && isfinite(nt.lo));
        for (int i = 0; i < p->n_eq && !nf; ++i) {
            if (!isfinite(state[(size_t)i * B + l])) nf = 1;
        }
        if (nf) {
            outcome[l] = OC_ERR_NF_STATE;
        } else {
            outcome[l] = (h == max_delta_ts[l]) ? OC_TIME_LIMIT : OC_SUCCESS;
        }
    }
}

/*
 * propagate_until() for one batch of B lanes in lock-step, with the reference's semantics
 * (reference: src/taylor_adaptive_batch.cpp:1137-1534, no callback / no continuous output).
 * t_final[l] single-length final times; max_delta_t[l] > 0 (or +inf); max_steps 0 = unlimited.
 * Outputs: outcome, min_h, max_h, n_steps per lane.
 */
void hy_oracle_propagate_until(const hy_oracle_program *p, int B, double *state, const double *pars, double *time_hi,
                               double *time_lo, const double *t_final, const double *max_delta_t, int64_t max_steps,
                               int64_t *outcome, double *min_h, double *max_h, int64_t *n_steps, double *scratch_mem)
{
    dfloat *rem = (dfloat *)malloc(sizeof(dfloat) * (size_t)B);
    int *t_dir = (int *)malloc(sizeof(int) * (size_t)B);
    double *cur_dt = (double *)malloc(sizeof(double) * (size_t)B);
    double *h_out = (double *)malloc(sizeof(double) * (size_t)B);
    int64_t *oc = (int64_t *)malloc(sizeof(int64_t) * (size_t)B);

    int64_t iter = 0;
    for (int l = 0; l < B; ++l) {
        n_steps[l] = 0;
        min_h[l] = INFINITY;
        max_h[l] = 0;
        dfloat tf = {t_final[l], 0.}, t = {time_hi[l], time_lo[l]};
        rem[l] = df_sub(tf, t);
        t_dir[l] = (rem[l].hi > 0.) || (rem[l].hi == 0. && rem[l].lo >= 0.);
    }

    while (1) {
        for (int l = 0; l < B; ++l) {
            dfloat lim;
            if (t_dir[l]) {
                dfloat m = {max_delta_t[l], 0.};
                lim = df_lt(rem[l], m) ? rem[l] : m; /* std::min(m, rem) */
            } else {
                dfloat m = {-max_delta_t[l], 0.};
                lim = df_lt(m, rem[l]) ? rem[l] : m; /* std::max(m, rem) */
            }
            cur_dt[l] = lim.hi;
        }

        hy_oracle_step_impl(p, B, state, pars, time_hi, time_lo, cur_dt, NULL, oc, h_out, scratch_mem);

        int n_done = 0, nfs = 0;
        for (int l = 0; l < B; ++l) {
            const double h = h_out[l];
            if (oc[l] == OC_ERR_NF_STATE) {
                nfs = 1;
            } else {
                n_steps[l] += (h != 0);
                if (oc[l] == OC_SUCCESS) {
                    const double ah = fabs(h);
                    min_h[l] = (ah < min_h[l]) ? ah : min_h[l];
                    max_h[l] = (max_h[l] < ah) ? ah : max_h[l];
                }
                const int cur_done = (h == rem[l].hi);
                n_done += cur_done;
                if (cur_done) {
                    rem[l].hi = 0.;
                    rem[l].lo = 0.;
                } else {
                    dfloat tf = {t_final[l], 0.}, t = {time_hi[l], time_lo[l]};
                    rem[l] = df_sub(tf, t);
                }
            }
            outcome[l] = oc[l];
        }

        if (nfs) break;
        ++iter;
        if (n_done == B) break;
        if (iter == max_steps) {
            for (int l = 0; l < B; ++l) outcome[l] = OC_STEP_LIMIT;
            break;
        }
    }

    free(rem);
    free(t_dir);
    free(cur_dt);
    free(h_out);
    free(oc);
}

/*
 * Ensemble driver: N systems organised as N / B batches (N % B == 0), arrays laid out
 * [row * N + sys]; every batch is gathered into a private [row * B + lane] buffer, propagated
 * with the reference's lock-step semantics, and scattered back (mirrors the TBB parallel_for
 * over batch integrators of src/ensemble_propagate.cpp:193-222). Returns the total number of
 * steps taken (sum over lanes).
 */
int64_t hy_oracle_ensemble_propagate_until(const hy_oracle_program *p, int64_t N, int B, double *state,
                                           const double *pars, double *time_hi, double *time_lo, double t_final,
                                           int64_t max_steps, int64_t *outcome, double *min_h, double *max_h,
                                           int64_t *n_steps, int n_threads)
{
    const int64_t n_batches = N / B;
    int64_t total = 0;
    const size_t ssz = hy_oracle_scratch_size(p, B);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel reduction(+ : total)
#endif
    {
        double *scratch = (double *)malloc(sizeof(double) * ssz);
        double *st = (double *)malloc(sizeof(double) * (size_t)p->n_eq * (size_t)B);
        double *pr = (double *)malloc(sizeof(double) * (size_t)(p->n_par > 0 ? p->n_par : 1) * (size_t)B);
        double *tf = (double *)malloc(sizeof(double) * (size_t)B);
        double *md = (double *)malloc(sizeof(double) * (size_t)B);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t b = 0; b < n_batches; ++b) {
            const int64_t off = b * B;
            for (int i = 0; i < p->n_eq; ++i)
                for (int l = 0; l < B; ++l) st[(size_t)i * B + l] = state[(size_t)i * N + off + l];
            for (int i = 0; i < p->n_par; ++i)
                for (int l = 0; l < B; ++l) pr[(size_t)i * B + l] = pars[(size_t)i * N + off + l];
            for (int l = 0; l < B; ++l) {
                tf[l] = t_final;
                md[l] = INFINITY;
            }
            hy_oracle_propagate_until(p, B, st, pr, time_hi + off, time_lo + off, tf, md, max_steps, outcome + off,
                                      min_h + off, max_h + off, n_steps + off, scratch);
            for (int i = 0; i < p->n_eq; ++i)
                for (int l = 0; l < B; ++l) state[(size_t)i * N + off + l] = st[(size_t)i * B + l];
            for (int l = 0; l < B; ++l) total += n_steps[off + l];
        }
        free(scratch);
        free(st);
        free(pr);
        free(tf);
        free(md);
    }
    return total;
}

int hy_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
