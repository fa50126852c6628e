/*
 * heyoka_amd C ABI: the drop-in boundary of the MI355X batch Taylor integrator.
 *
 * Every entry point replaces (and is named after) a piece of the public C++ interface of
 * bluescarni/heyoka v7.12.0 on the taylor_adaptive_batch<double> / ensemble_propagate_* path; the
 * reference interface each one stands for is cited as file:line relative to the reference tree.
 * Plain pointers and sizes only: no C++ or torch types cross this boundary.
 *
 * Conventions
 *  - All per-system arrays use the reference's batch layout array[row * batch_size + lane]
 *    (src/taylor_adaptive_batch.cpp:679, src/taylor_00.cpp:574-575, :835).
 *  - Functions returning int return 0 on success, otherwise one of the HY_ERR_* codes; the message
 *    (identical to the exception message of the reference where one exists) is available from
 *    hy_last_error() on the calling thread.
 *  - Functions returning a handle return NULL on error.
 *  - Pointers named d_* are device (HBM) pointers; everything else is host memory.
 */
#ifndef HEYOKA_AMD_H
#define HEYOKA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HY_OK 0
#define HY_ERR_INVALID_ARGUMENT 1 /* std::invalid_argument in the reference */
#define HY_ERR_OVERFLOW 2         /* std::overflow_error */
#define HY_ERR_NOT_IMPLEMENTED 3  /* heyoka::not_implemented_error (include/heyoka/exceptions.hpp) */
#define HY_ERR_RUNTIME 4          /* std::runtime_error / HIP / hiprtc failures */

/* taylor_outcome (include/heyoka/taylor.hpp:142-155). */
#define HY_OUTCOME_SUCCESS (-4294967296LL - 1)
#define HY_OUTCOME_STEP_LIMIT (-4294967296LL - 2)
#define HY_OUTCOME_TIME_LIMIT (-4294967296LL - 3)
#define HY_OUTCOME_ERR_NF_STATE (-4294967296LL - 4)
#define HY_OUTCOME_CB_STOP (-4294967296LL - 5)

const char *hy_last_error(void);
/* HY_ERR_* code of the last failed call on the calling thread (for functions returning handles). */
int hy_last_error_code(void);
void hy_free_str(char *);
/* Library/toolchain information: "heyoka_amd <version>; gfx950; hiprtc <ver>". Caller frees. */
char *hy_version(void);
/* Build id of the library: the first 16 hex digits of the SHA-256 over its sources (the .cpp and .hpp files of heyoka_amd/csrc in sorted
 * order, then this header). The Python host layer recomputes it from the tree at import and refuses a library which does
 * not match (a stale prebuilt .so); bench.py prints it next to the sha of the generated kernel. Static string. */
const char *hy_build_id(void);
/* Stage logger (include/heyoka/logging.hpp:19-24: set_logger_level_trace() ... _critical(); src/logging.cpp:20-48): level
 * 0 trace, 1 debug, 2 info, 3 warn (default), 4 err, 5 critical, 6 off. Logged at construction: the size of the Taylor
 * decomposition (debug), the generator the planner chose with the reasons the others did not apply (info; warn when a large
 * decomposition falls to a generic stepper), and the decomposition / code generation / hiprtc times (trace). Messages go to
 * stderr unless a callback is installed. */
typedef void (*hy_log_callback_t)(int level, const char *msg, void *user);
int hy_set_logger_level(int level);
int hy_get_logger_level(void);
void hy_set_log_callback(hy_log_callback_t cb, void *user);
/* Number of visible HIP devices (0 without a GPU). */
int hy_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Expressions (include/heyoka/expression.hpp:73-118; operators src/expression_ops.cpp:34-91).
 * Handles are owned by the caller and released with hy_expr_free(). Functions never consume
 * their arguments.
 * --------------------------------------------------------------------------------------------- */
typedef struct hy_expr_s *hy_expr;

hy_expr hy_expr_var(const char *name);    /* expression{variable{name}}   expression.hpp:73 */
hy_expr hy_expr_num(double value);        /* expression{number{value}}    number.hpp */
hy_expr hy_expr_par(uint32_t index);      /* par[index]                   param.hpp */
hy_expr hy_expr_time(void);               /* heyoka::time                 math/time.hpp */
hy_expr hy_expr_neg(hy_expr);             /* operator-(e)                 expression_ops.cpp:45-52 */
hy_expr hy_expr_add(hy_expr, hy_expr);    /* operator+                    expression_ops.cpp:55-62 */
hy_expr hy_expr_sub(hy_expr, hy_expr);    /* operator-                    expression_ops.cpp:65-72 */
hy_expr hy_expr_mul(hy_expr, hy_expr);    /* operator*                    expression_ops.cpp:75-82 */
hy_expr hy_expr_div(hy_expr, hy_expr);    /* operator/                    expression_ops.cpp:85-91 */
hy_expr hy_expr_pow(hy_expr, hy_expr);    /* pow()                        src/math/pow.cpp:1066 */
hy_expr hy_expr_sqrt(hy_expr);            /* sqrt()                       src/math/sqrt.cpp:16 */
hy_expr hy_expr_sin(hy_expr);             /* sin()                        src/math/sin.cpp:406 */
hy_expr hy_expr_cos(hy_expr);             /* cos()                        src/math/cos.cpp:406 */
hy_expr hy_expr_exp(hy_expr);             /* exp()                        src/math/exp.cpp */
hy_expr hy_expr_log(hy_expr);             /* log()                        src/math/log.cpp */
/* Elementary functions beyond the N-body set (include/heyoka/math/{tan,tanh,sinh,cosh,asin,acos,atan,asinh,
 * acosh,atanh,erf,sigmoid}.hpp; Taylor rules src/math/<name>.cpp). */
hy_expr hy_expr_tan(hy_expr);
hy_expr hy_expr_tanh(hy_expr);
hy_expr hy_expr_sinh(hy_expr);
hy_expr hy_expr_cosh(hy_expr);
hy_expr hy_expr_asin(hy_expr);
hy_expr hy_expr_acos(hy_expr);
hy_expr hy_expr_atan(hy_expr);
hy_expr hy_expr_asinh(hy_expr);
hy_expr hy_expr_acosh(hy_expr);
hy_expr hy_expr_atanh(hy_expr);
hy_expr hy_expr_erf(hy_expr);
hy_expr hy_expr_sigmoid(hy_expr);
hy_expr hy_expr_atan2(hy_expr y, hy_expr x); /* atan2(y, x)                  src/math/atan2.cpp:763 */
hy_expr hy_expr_kepE(hy_expr e, hy_expr M);  /* eccentric anomaly E(e, M)    src/math/kepE.cpp:801 */
/* Piecewise functions: relu / relup with the slope of the leaky variants (src/math/relu.cpp:580-602), select(c, t, f)
 * (src/math/select.cpp:267), logical_and (is_and != 0) / logical_or (src/math/logical.cpp:314-338), comparisons
 * op = 0..5 -> eq, neq, lt, gt, lte, gte (src/math/relational.cpp:343-354). */
hy_expr hy_expr_relu(hy_expr x, double slope);
hy_expr hy_expr_relup(hy_expr x, double slope);
hy_expr hy_expr_select(hy_expr cond, hy_expr t, hy_expr f);
hy_expr hy_expr_logical(int is_and, const hy_expr *args, size_t n);
hy_expr hy_expr_rel(int op, hy_expr a, hy_expr b);
/* ---- Node rules: the per-function extension seam (reference: func_base, include/heyoka/func.hpp:94-96, :117-147; a
 * function without a Taylor rule raises not_implemented_error, func.hpp:266-267 -> HY_ERR_NOT_IMPLEMENTED here).
 * A rule is the counterpart of a func_base subclass (heyoka_amd/csrc/node_rule.hpp):
 *   - decompose(): fills hidden_out[0 .. n_hidden) with the definitions of the hidden u variables appended behind the node
 *     (what a taylor_decompose() override appends by hand); each is ONE elementary function of `self` (the node), its
 *     arguments and hidden_vars[j], j < own index. Returns 0 on success. The callee owns nothing: every handle it
 *     creates and stores in hidden_out is released by the library.
 *   - hidden_deps[4 * j .. 4 * j + 4): the hidden dependencies of hidden definition j (indices into the hidden
 *     definitions, -1 = none), e.g. sin(self) <-> cos(self);
 *   - deps[0 .. n_deps): the hidden dependencies of the node itself, in the order hy_rule_<name>_orderk() reads them;
 *   - hip_source: HIP device code defining
 *         double hy_rule_<name>_order0(const double *x);
 *         double hy_rule_<name>_orderk(unsigned k, const hy_jet &a, const hy_jet *x, const hy_jet *h);
 *     (the taylor_diff() / taylor_c_diff_func() pair of the reference: one text serves the straight-line generator, the
 *     interpreted steppers and compiled functions). */
typedef int (*hy_node_rule_decompose_fn)(void *ctx, hy_expr self, const hy_expr *args, uint32_t n_args,
                                         const hy_expr *hidden_vars, hy_expr *hidden_out);
typedef struct hy_node_rule_desc {
    const char *name;
    uint32_t n_args;
    uint32_t n_hidden;
    hy_node_rule_decompose_fn decompose; /* NULL iff n_hidden == 0 */
    void *ctx;
    const int32_t *hidden_deps; /* 4 * n_hidden entries, may be NULL if no hidden definition has dependencies */
    const uint32_t *deps;
    uint32_t n_deps;
    const char *hip_source;
} hy_node_rule_desc;
int hy_node_rule_register(const hy_node_rule_desc *); /* HY_OK / HY_ERR_INVALID_ARGUMENT (duplicate, malformed) */
/* f(args) of a registered rule; NULL + HY_ERR_NOT_IMPLEMENTED for an unknown name. */
hy_expr hy_expr_custom(const char *name, const hy_expr *args, size_t n);
/* Defined through the registry alone (csrc/builtin_rules.cpp): kepF(h, k, lam), F + h cos F - k sin F = lam
 * (src/math/kepF.cpp:1689), and kepDE(s0, c0, DM), DE - c0 sin DE + s0 (1 - cos DE) = DM (src/math/kepDE.cpp:113). */
hy_expr hy_expr_kepF(hy_expr h, hy_expr k, hy_expr lam);
hy_expr hy_expr_kepDE(hy_expr s0, hy_expr c0, hy_expr DM);
/* heyoka::pi: a constant which is a function without arguments with its own u variable (include/heyoka/math/constants.hpp:117,
 * src/math/constants.cpp:258-273); a registered rule as well. */
hy_expr hy_expr_pi(void);
hy_expr hy_expr_sum(const hy_expr *, size_t n);  /* sum(vector)           src/math/sum.cpp:548 */
hy_expr hy_expr_prod(const hy_expr *, size_t n); /* prod(vector)          src/math/prod.cpp:913 */
void hy_expr_free(hy_expr);
char *hy_expr_str(hy_expr); /* caller frees with hy_free_str() */

/* ------------------------------------------------------------------------------------------------
 * ODE systems: std::vector<std::pair<expression, expression>> (prime(x) = rhs).
 * --------------------------------------------------------------------------------------------- */
typedef struct hy_sys_s *hy_sys;

hy_sys hy_sys_new(void);
int hy_sys_add(hy_sys, hy_expr lhs, hy_expr rhs); /* prime(lhs) = rhs */
size_t hy_sys_size(hy_sys);
void hy_sys_free(hy_sys);
/* model::nbody(n, kw::masses, kw::Gconst) (include/heyoka/model/nbody.hpp:73-78, src/model/nbody.cpp:53-174).
 * masses == NULL -> all masses equal to 1 (n_masses ignored). */
hy_sys hy_model_nbody(uint32_t n, const double *masses, size_t n_masses, double Gconst);
/* model::pendulum(kw::gconst, kw::length) (src/model/pendulum.cpp:23-28). */
hy_sys hy_model_pendulum(double gconst, double length);
/* General form of model::nbody(): masses and Gconst as expressions (numbers or par[i]; runtime parameters
 * select the non-grouped branch src/model/nbody.cpp:131-150). masses == NULL -> all 1; Gconst == NULL -> 1. */
hy_sys hy_model_nbody_ex(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
/* model::nbody_energy() / nbody_potential() (include/heyoka/model/nbody.hpp:80-92, src/model/nbody.cpp:176-235),
 * model::pendulum_energy() (src/model/pendulum.cpp:31-36): expressions of the state variables
 * x_i, y_i, z_i, vx_i, vy_i, vz_i (resp. x, v), ready for hy_cfunc_new(). */
hy_expr hy_model_nbody_energy(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
hy_expr hy_model_nbody_potential(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
hy_expr hy_model_pendulum_energy(double gconst, double length);
/* The other point-mass models. Expression arguments are numbers or par[i]; a NULL
 * Gconst / mu selects the reference's default (1 resp. 1e-3), an empty omega a non-rotating frame.
 *   model::np1body / np1body_energy / np1body_potential  include/heyoka/model/nbody.hpp:94-120, src/model/nbody.cpp:236-445
 *   model::cr3bp / cr3bp_jacobi                          include/heyoka/model/cr3bp.hpp:30-64, src/model/cr3bp.cpp
 *   model::fixed_centres{,_energy,_potential}            include/heyoka/model/fixed_centres.hpp, src/model/fixed_centres.cpp
 *   model::rotating{,_energy,_potential}                 include/heyoka/model/rotating.hpp, src/model/rotating.cpp
 *   model::mascon{,_energy,_potential}                   include/heyoka/model/mascon.hpp, src/model/mascon.cpp
 * State variables: x_i, y_i, z_i, vx_i, vy_i, vz_i (i = 1..n-1) for np1body; x, y, z, px, py, pz for cr3bp;
 * x, y, z, vx, vy, vz for the others. positions holds 3 entries per mass. */
hy_sys hy_model_np1body(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
hy_expr hy_model_np1body_energy(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
hy_expr hy_model_np1body_potential(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst);
hy_sys hy_model_cr3bp(hy_expr mu);
hy_expr hy_model_cr3bp_jacobi(hy_expr mu);
hy_sys hy_model_fixed_centres(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                              size_t n_positions);
hy_expr hy_model_fixed_centres_energy(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                                      size_t n_positions);
hy_expr hy_model_fixed_centres_potential(hy_expr Gconst, const hy_expr *masses, size_t n_masses,
                                         const hy_expr *positions, size_t n_positions);
hy_sys hy_model_rotating(const hy_expr *omega, size_t n_omega);
hy_expr hy_model_rotating_energy(const hy_expr *omega, size_t n_omega);
hy_expr hy_model_rotating_potential(const hy_expr *omega, size_t n_omega);
hy_sys hy_model_mascon(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                       size_t n_positions, const hy_expr *omega, size_t n_omega);
hy_expr hy_model_mascon_energy(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                               size_t n_positions, const hy_expr *omega, size_t n_omega);
hy_expr hy_model_mascon_potential(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                                  size_t n_positions, const hy_expr *omega, size_t n_omega);
/* The state variables of a system, in order (the lhs of each equation); out[hy_sys_size()] receives new
 * handles owned by the caller. */
int hy_sys_get_vars(hy_sys, hy_expr *out);
/* taylor_decompose_sys() (src/taylor_01.cpp:848-1008): one line per entry of the decomposition,
 * "u_i = ..." textual form with hidden dependencies. Caller frees with hy_free_str(). */
char *hy_sys_decomposition_str(hy_sys);

/* ------------------------------------------------------------------------------------------------
 * taylor_adaptive_batch<double> (include/heyoka/taylor.hpp:781-1121).
 * --------------------------------------------------------------------------------------------- */
typedef struct hy_tab_s *hy_tab;

/* Keyword arguments of the constructor (taylor.hpp:814-821, :905-941). */
typedef struct {
    double tol;            /* kw::tol; 0 -> default (machine epsilon) */
    int high_accuracy;     /* kw::high_accuracy */
    int compact_mode;      /* kw::compact_mode (accepted; code generation mode is chosen internally) */
    int parallel_mode;     /* kw::parallel_mode (validated like the reference, otherwise ignored) */
    const double *pars;    /* kw::pars, n_pars values (NULL -> zeros) */
    size_t n_pars;
    const double *time;    /* kw::time: NULL -> 0; n_time == 1 -> scalar splat; else batch_size values */
    size_t n_time;
    int device;            /* HIP device ordinal (MI355X extension) */
    /* MI355X extensions; 0 = default everywhere (a zero-initialised structure behaves like the reference). */
    int emitter;           /* code generator: 0 automatic, 1 unrolled, 2 wave-cluster, 3 table (the compact-mode analogue), 4 block */
    int cluster_kernel;    /* wave-cluster generator: 0 automatic, 5 one lane per pair, 3 lane pairs, 2 pipelined, 1 first generation */
    int exact_division;    /* != 0: correctly rounded quotients in the recurrences of the pair kernels (default: reciprocal forms, within 1 ulp) */
    int events_on_cluster; /* 0 automatic; 1: integrators with events always use the one-system-per-lane steppers */
    int sum_order;         /* additions inside the convolutions of the unrolled generator: 0 automatic (= 2), 1 the reference's
                            * default mode (products first, pairwise sum), 2 its compact mode (running sums: one FMA per term) */
    int batch_semantics;   /* propagate_for/until when a lane goes non-finite or max_steps is hit: 0 the reference's batch-wide
                            * outcomes (src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467, :1516), 1 always the lock-step loop,
                            * 2 per-lane outcomes of the device-resident path (no snapshot, fully asynchronous) */
} hy_tab_config;

/* Constructor (taylor.hpp:905-941 -> finalise_ctor_impl(), src/taylor_adaptive_batch.cpp:78-427).
 * state: n_state = n_eq * batch_size values, or n_state == 0 for a zero-initialised state.
 * Performs decomposition, HIP code generation and hiprtc compilation; does not need a GPU. */
hy_tab hy_tab_create(hy_sys sys, const double *state, size_t n_state, uint32_t batch_size, const hy_tab_config *cfg);
/* Event detection (kw::t_events / kw::nt_events, include/heyoka/events.hpp:52-325; detection
 * src/detail/event_detection.cpp:1733-2173; event branch of step_impl() src/taylor_adaptive_batch.cpp:727-1030).
 * direction: -1 negative, 0 any, +1 positive (event_direction). Callbacks run on the host, one lane at a time:
 *   non-terminal: cb(integrator, trigger time, sign of d(eq)/dt, batch index, user)
 *   terminal:     cb(...) returns non-zero to continue, zero to stop the propagation; cb == NULL always stops.
 *   cooldown < 0 -> deduced automatically (taylor_deduce_cooldown()).
 * Outcomes follow the reference: step outcome = event index (continuing) or -index - 1 (stopping). */
typedef void (*hy_nt_event_cb)(hy_tab, double time, int d_sgn, uint32_t batch_idx, void *user);
typedef int (*hy_t_event_cb)(hy_tab, int d_sgn, uint32_t batch_idx, void *user);
typedef struct {
    hy_expr eq;
    hy_nt_event_cb cb;
    void *user;
    int direction;
} hy_nt_event;
typedef struct {
    hy_expr eq;
    hy_t_event_cb cb;
    void *user;
    int direction;
    double cooldown;
} hy_t_event;
hy_tab hy_tab_create_with_events(hy_sys sys, const double *state, size_t n_state, uint32_t batch_size,
                                 const hy_tab_config *cfg, const hy_t_event *t_events, size_t n_t_events,
                                 const hy_nt_event *nt_events, size_t n_nt_events);
int hy_tab_with_events(hy_tab);
/* Accounting of the steps with events (bench.py's events leg). out8 = {steps with events, ms upload / buffers, ms stepper
 * (+ event jets), ms detection kernel, ms bookkeeping kernel + flags to the host, ms state update + records, regeneration
 * launches of the Taylor coefficients, systems which reported events}; the five phase times accumulate only while the
 * timing is on (hy_tab_set_event_timing(): one stream synchronisation per phase). */
int hy_tab_set_event_timing(hy_tab, int on);
int hy_tab_get_event_stats(hy_tab, double *out8);
/* Ready-made callbacks which count their invocations in the uint64_t `user` points to (the terminal one continues).
 * When EVERY event of an integrator has one of them as its callback, the step applies the events on the device (counts per
 * event, cooldown and "continuing" outcome of the first terminal event of a lane: what the host loop of
 * src/taylor_adaptive_batch.cpp:837-1030 does) and adds the counts to the counters once per step: no per-event host work. */
void hy_event_counter_nt(hy_tab, double time, int d_sgn, uint32_t batch_idx, void *user);
int hy_event_counter_t(hy_tab, int d_sgn, uint32_t batch_idx, void *user);
/* reset_cooldowns(): batch_idx < 0 -> all the lanes. */
int hy_tab_reset_cooldowns(hy_tab, int64_t batch_idx);
/* get_te_cooldowns(): [batch_size * n_t_events] arrays indexed [lane * n_t_events + event]; active != 0 where a
 * cooldown is in progress, (first, second) = (elapsed, duration). */
int hy_tab_get_te_cooldowns(hy_tab, double *first, double *second, int *active);
hy_tab hy_tab_copy(hy_tab);  /* copy constructor (taylor.hpp:943) */
void hy_tab_free(hy_tab);

/* Getters (taylor.hpp:951-1029). */
uint32_t hy_tab_get_batch_size(hy_tab);
uint32_t hy_tab_get_order(hy_tab);
uint32_t hy_tab_get_dim(hy_tab);
uint32_t hy_tab_get_n_pars(hy_tab);   /* number of runtime parameters per system */
uint32_t hy_tab_get_n_uvars(hy_tab);  /* size of the decomposition minus n_eq */
double hy_tab_get_tol(hy_tab);
int hy_tab_get_high_accuracy(hy_tab);
int hy_tab_get_compact_mode(hy_tab);
/* No reference counterpart: number of (event, lane, step) triples in which the device-side event detection gave up - root
 * isolation beyond 250 working intervals or more isolating intervals than the order, root finder out of iterations: the
 * cases in which the reference's detect_events(), src/detail/event_detection.cpp:2082-2090, :2150-2165, logs a warning and
 * ignores the event for the step. */
unsigned long long hy_tab_get_event_detection_failures(hy_tab);
double hy_tab_get_compile_seconds(hy_tab);
char *hy_tab_get_hip_source(hy_tab);  /* generated HIP module (cf. llvm_state::get_ir()); caller frees */
char *hy_tab_get_decomposition_str(hy_tab);
/* No reference counterpart: the rewritten INTERNAL program the stepper was generated from when the planner of the
 * wave-cluster kernels changed it (state-variable aliases, padded clusters, restored unit scalings; the decomposition the
 * user sees is never touched) - one node per line in the format of hy_tab_get_decomposition_str(), then the definitions
 * of the state derivatives; an empty string otherwise. For tests: the rewrites must not change the jets. Caller frees. */
char *hy_tab_get_internal_program(hy_tab);
/* Description of the code generation mode chosen for this system ("unrolled", "cluster ...", "table ..."). Caller frees. */
char *hy_tab_get_codegen_info(hy_tab);
/* The gfx950 code object of the stepper module (cf. llvm_state::get_object_code(), include/heyoka/llvm_state.hpp) and a
 * plain hiprtc compilation of a HIP source with the options of the steppers (for offline inspection: disassembly, the
 * code-generation checks of tests/test_codegen_hazards.py). Two-call convention: out == NULL -> *size receives the size. */
int hy_tab_get_code_object(hy_tab, void *out, size_t *size);
int hy_hiprtc_compile(const char *source, void *out, size_t *size);

int hy_tab_get_state(hy_tab, double *out);                 /* get_state()        n_eq * batch_size */
int hy_tab_set_state(hy_tab, const double *in);            /* the values of get_state_data()[...] = in[...]; by value: the
                                                            * integrator stays lazily synchronised */
int hy_tab_get_pars(hy_tab, double *out);                  /* get_pars() */
int hy_tab_set_pars(hy_tab, const double *in);             /* like hy_tab_set_state() for get_pars_data() */
/* get_state_data() / get_pars_data() (include/heyoka/taylor.hpp:984-990): MUTABLE pointers to the host mirrors, valid for
 * the lifetime of the integrator. Reference call sites write through them between steps (ta.get_state_data()[i] = ...), so
 * handing one out switches the integrator to eager synchronisation: the mirrors are refreshed after every launch and
 * uploaded before the next one (a download of the whole state per step: use the setters above, or the device views of
 * hy_tab_device_ptr(), where that matters). NULL + hy_last_error() on failure. */
double *hy_tab_get_state_data(hy_tab);
double *hy_tab_get_pars_data(hy_tab);
int hy_tab_get_dtime(hy_tab, double *hi, double *lo);      /* get_dtime()        batch_size each; lo may be NULL */
int hy_tab_set_time(hy_tab, const double *t, size_t n);    /* set_time(): n == 1 scalar, else batch_size */
int hy_tab_set_dtime(hy_tab, const double *hi, const double *lo, size_t n); /* set_dtime() */
int hy_tab_get_tc(hy_tab, double *out);                    /* get_tc()           n_eq * (order + 1) * batch_size: the
                                                            * coefficients of the last step taken with write_tc (zeros
                                                            * before the first one), like the reference's m_tc */
int hy_tab_get_last_h(hy_tab, double *out);                /* get_last_h() */
/* update_d_output(t, rel_time) (src/taylor_adaptive_batch.cpp:2251-2327): n == 1 scalar, else batch_size. */
int hy_tab_update_d_output(hy_tab, const double *t, size_t n, int rel_time, double *out);

/* step() / step_backward() / step(max_delta_ts) (taylor.hpp:1001-1003; step_impl()
 * src/taylor_adaptive_batch.cpp:632-727). */
int hy_tab_step(hy_tab, int write_tc);
int hy_tab_step_backward(hy_tab, int write_tc);
int hy_tab_step_limited(hy_tab, const double *max_delta_ts, size_t n, int write_tc);
int hy_tab_get_step_res(hy_tab, int64_t *outcome, double *h); /* get_step_res() */

/* Step callback: bool(taylor_adaptive_batch &) (include/heyoka/step_callback.hpp:59-62).
 * Return non-zero to continue, 0 to stop (outcome cb_stop). */
typedef int (*hy_step_callback)(hy_tab, void *user_data);

/* propagate_until() / propagate_for() (taylor.hpp:1064-1107; propagate_until_impl()
 * src/taylor_adaptive_batch.cpp:1137-1534, propagate_for_impl() :1082-1118).
 *  ts / n_ts ............ final times (durations for propagate_for): n_ts == 1 scalar, else batch_size
 *  max_steps ............ kw::max_steps (0 = unlimited)
 *  max_delta_ts / n_mdt . kw::max_delta_t: n_mdt == 0 none, 1 scalar, else batch_size
 *  cb ................... kw::callback (NULL = none). Without a callback the whole propagation runs
 *                         device-resident in one kernel launch; with a callback the reference's
 *                         lock-step loop is used (one launch per step, callback on the host).
 *  write_tc, c_output ... kw::write_tc, kw::c_output. With c_output != 0 the lock-step loop is used as well and
 *                         the continuous output object is retrieved with hy_tab_take_c_output(). */
int hy_tab_propagate_until(hy_tab, const double *ts, size_t n_ts, uint64_t max_steps, const double *max_delta_ts,
                           size_t n_mdt, hy_step_callback cb, void *cb_data, int write_tc, int c_output);
int hy_tab_propagate_for(hy_tab, const double *dts, size_t n_dts, uint64_t max_steps, const double *max_delta_ts,
                         size_t n_mdt, hy_step_callback cb, void *cb_data, int write_tc, int c_output);
/* The full step-callback protocol (include/heyoka/step_callback.hpp:46-62, :139-185): a callback may provide a
 * pre_hook(), run once after the validation of the arguments and before the first step
 * (src/taylor_adaptive_batch.cpp:1356-1365, :1782-1791; it must not move the time coordinate), and kw::callback accepts a
 * range of callbacks, which form a set: every member runs at every step and the results are and-ed
 * (src/step_callback.cpp:108-127); a NULL `call` inside a set of more than one member is an error
 * ("Cannot construct a callback set containing one or more empty callbacks"), n_cbs == 0 means no callback. */
/* A pre-hook returns 0, or non-zero to abort the propagation before its first step (how an exception thrown by
 * pre_hook() travels through the C boundary; the reference lets it propagate out of propagate_*()); a `call` member
 * may return a NEGATIVE value for the same purpose: the remaining members of the set are not run and the propagation
 * stops like after a `false`. */
typedef int (*hy_step_pre_hook)(hy_tab, void *user_data);
typedef struct {
    hy_step_callback call;
    hy_step_pre_hook pre_hook; /* NULL: the default no-op */
    void *user_data;
} hy_step_callback_desc;
int hy_tab_propagate_until_cbs(hy_tab, const double *ts, size_t n_ts, uint64_t max_steps, const double *max_delta_ts,
                               size_t n_mdt, const hy_step_callback_desc *cbs, size_t n_cbs, int write_tc, int c_output);
int hy_tab_propagate_for_cbs(hy_tab, const double *dts, size_t n_dts, uint64_t max_steps, const double *max_delta_ts,
                             size_t n_mdt, const hy_step_callback_desc *cbs, size_t n_cbs, int write_tc, int c_output);
int hy_tab_propagate_grid_cbs(hy_tab, const double *grid, size_t n_grid, uint64_t max_steps, const double *max_delta_ts,
                              size_t n_mdt, const hy_step_callback_desc *cbs, size_t n_cbs, double *out);
/* ------------------------------------------------------------------------------------------------
 * continuous_output_batch<double> (include/heyoka/continuous_output.hpp:151-204,
 * src/continuous_output.cpp:602-1236): the optional<continuous_output_batch> slot of the tuple returned
 * by propagate_for/until(). Coefficients and times stay in HBM; evaluation is one kernel launch. */
typedef struct hy_cout_s *hy_cout;
/* Moves the continuous output recorded by the last propagate_for/until(c_output != 0) into *out
 * (*out = NULL if no step was taken, like the empty optional of the reference). Free with hy_cout_free(). */
int hy_tab_take_c_output(hy_tab, hy_cout *out);
void hy_cout_free(hy_cout);
/* Copy (the device data is shared, the output buffers are not). */
hy_cout hy_cout_clone(hy_cout);
/* operator()(T) (n_tm == 1), operator()(const std::vector<T> &) (n_tm == batch_size; any other size is
 * an error). out receives get_output(): n_eq * batch_size values, out[var * batch_size + lane]. */
int hy_cout_eval(hy_cout, const double *tm, size_t n_tm, double *out);
/* MI355X extension: target times d_tm[batch_size] and d_out[n_eq * batch_size] in device memory;
 * asynchronous on the integrator's stream. */
int hy_cout_eval_device(hy_cout, const double *d_tm, double *d_out);
uint32_t hy_cout_get_batch_size(hy_cout);
uint32_t hy_cout_get_dim(hy_cout);
uint32_t hy_cout_get_order(hy_cout);
/* get_n_steps() (padding excluded). */
int hy_cout_get_n_steps(hy_cout, size_t *n_steps);
/* get_bounds(): lb[batch_size], ub[batch_size]. */
int hy_cout_get_bounds(hy_cout, double *lb, double *ub);
/* get_times(): (n_steps + 2) * batch_size values (last row = +-inf padding); lo may be NULL. */
int hy_cout_get_times(hy_cout, double *hi, double *lo);
/* get_tcs(): n_steps * n_eq * (order + 1) * batch_size values, downloaded from the device. */
int hy_cout_get_tcs(hy_cout, double *out);
/* operator<<: malloc'ed string, free with hy_free_str(). */
char *hy_cout_to_string(hy_cout);

/* propagate_grid() (taylor.hpp:1113-1119; propagate_grid_impl() src/taylor_adaptive_batch.cpp:1546-2055).
 * grid: n_grid * batch_size values laid out grid[point * batch_size + lane];
 * out: n_grid * n_eq * batch_size values (NaN where not reached). */
int hy_tab_propagate_grid(hy_tab, const double *grid, size_t n_grid, uint64_t max_steps, const double *max_delta_ts,
                          size_t n_mdt, hy_step_callback cb, void *cb_data, double *out);
/* MI355X extension of propagate_grid() for ensemble-scale batches: the n_grid * n_eq * batch_size samples are
 * written to a caller-owned device buffer d_out (same layout as above) instead of a host vector; no callback.
 * scalar_grid != 0: grid holds n_grid values used by every lane (the splat of ensemble_propagate_grid_batch(),
 * src/ensemble_propagate.cpp:266-273), otherwise n_grid * batch_size values. */
int hy_tab_propagate_grid_device(hy_tab, const double *grid, size_t n_grid, int scalar_grid, uint64_t max_steps,
                                 const double *max_delta_ts, size_t n_mdt, double *d_out);
/* get_propagate_res(): (outcome, min |h|, max |h|, n_steps) per lane. */
int hy_tab_get_propagate_res(hy_tab, int64_t *outcome, double *min_h, double *max_h, uint64_t *n_steps);

/* ------------------------------------------------------------------------------------------------
 * Device-resident access (MI355X extension; the reference has no device boundary).
 * --------------------------------------------------------------------------------------------- */
#define HY_BUF_STATE 0   /* double[n_eq * batch_size] */
#define HY_BUF_PARS 1    /* double[n_pars * batch_size] */
#define HY_BUF_TIME_HI 2 /* double[batch_size] */
#define HY_BUF_TIME_LO 3 /* double[batch_size] */
#define HY_BUF_TC 4      /* double[n_eq * (order + 1) * batch_size] */
#define HY_BUF_N_STEPS 5 /* uint64[batch_size]: step counters of the last propagate_*() */
#define HY_BUF_OUTCOME 6 /* int64[batch_size]: taylor_outcome of the last step()/propagate_*() */
#define HY_BUF_LAST_H 7  /* double[batch_size] */
/* Device pointer of one of the SoA arrays of the integrator (uploads pending host changes first). */
void *hy_tab_device_ptr(hy_tab, int which);
/* Tell the integrator that the caller wrote to the device arrays. */
int hy_tab_mark_device_modified(hy_tab);
/* Run all subsequent kernels/copies on the given hipStream_t (NULL = default stream). */
int hy_tab_set_stream(hy_tab, void *hip_stream);
int hy_tab_synchronize(hy_tab);
/* Sum over lanes of the step counters of the last propagate_*() call. */
uint64_t hy_tab_get_last_total_steps(hy_tab);
/* Durations (ms) of the last (at most) n stepper-kernel launches, oldest first, measured with HIP
 * events recorded on the launch stream immediately around each launch. Returns the count written. */
size_t hy_tab_get_kernel_ms_history(hy_tab, double *out, size_t n);

/* Stepper function-pointer ABI of the reference
 * (`void step(T *state, const T *pars, const T *time, T *h, T *tc)`,
 *  include/heyoka/detail/ta_jit_data.hpp:35-44, looked up at src/taylor_adaptive_batch.cpp:2391-2393)
 * on caller-owned device buffers holding n_systems systems:
 *   d_state rw [n_eq * n], d_pars [n_pars * n] (may be NULL if no parameters), d_time [n] (hi part),
 *   d_h in: signed max step (+-inf allowed), out: step taken [n], d_tc nullable [n_eq * (order+1) * n].
 * No error channel for numerical failures (non-finite results are the caller's to detect). */
int hy_tab_raw_step(hy_tab, double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                    uint64_t n_systems);
/* The other pointer types of the reference's stepper ABI (include/heyoka/detail/ta_jit_data.hpp:34-43), same argument
 * order, on caller-owned device buffers of n_systems systems:
 *   step_f_e_t   (jet, state, pars, time, h, max_abs_state): the stepper with events of an integrator constructed with
 *                events (taylor_add_adaptive_step_with_events(), src/taylor_00.cpp:592-710) - d_jet receives the Taylor
 *                coefficients of the state variables, [n_eq * (order + 1) * n], followed by those of the event equations,
 *                [(n_t_events + n_nt_events) * (order + 1) * n] (terminal events first), d_h in: signed max step, out: the
 *                step size of the selector, d_max_abs_state [n]: max |x_i| over the state; the state is NOT updated;
 *   d_out_f_t    (d_out, tc, h): dense output (taylor_add_d_out_function(), src/taylor_01.cpp:1015-1185) - d_out [n_eq * n]
 *                = the Taylor polynomials d_tc [n_eq * (order + 1) * n] evaluated at d_h [n] (compensated in high-accuracy
 *                mode, Horner otherwise);
 *   c_step_f_t / c_step_f_e_t (... , void *tape): the same two steppers with the tape in caller-owned device memory of
 *                hy_tab_tape_size_align() bytes (0: this stepper keeps its coefficients on chip and ignores the argument;
 *                src/taylor_02.cpp:1194-1260 for the reference's tape). */
int hy_tab_raw_step_e(hy_tab, double *d_jet, const double *d_state, const double *d_pars, const double *d_time, double *d_h,
                      double *d_max_abs_state, uint64_t n_systems);
int hy_tab_raw_d_out_f(hy_tab, double *d_out, const double *d_tc, const double *d_h, uint64_t n_systems);
int hy_tab_raw_step_tape(hy_tab, double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                         void *d_tape, uint64_t n_systems);
int hy_tab_raw_step_e_tape(hy_tab, double *d_jet, const double *d_state, const double *d_pars, const double *d_time,
                           double *d_h, double *d_max_abs_state, void *d_tape, uint64_t n_systems);
int hy_tab_tape_size_align(hy_tab, uint64_t n_systems, size_t *size, size_t *align);

/* Build-time check: hiprtc-compiles for gfx950 the auxiliary kernels that are otherwise compiled at first use on a
 * GPU (continuous output, propagate_grid post-step, event detection) for the given order / dimension; does not need
 * a GPU. Returns HY_OK or an error code (hy_last_error()). */
int hy_compile_aux_kernels(uint32_t order, uint32_t dim, int high_accuracy);

/* ------------------------------------------------------------------------------------------------
 * cfunc<double> (include/heyoka/expression.hpp:735-970, src/cfunc_class.cpp, function_decompose()
 * src/expression_cfunc.cpp:723-900): compiled evaluation of fn(vars) over many input columns.
 * One HIP kernel, one lane per evaluation, on SoA arrays in[var * nevals + eval] (the row-major 2D
 * layout of the reference's multi-evaluation call operator, which is also the integrator's state
 * layout: the device-side invariant monitor of an ensemble is hy_cfunc_eval_device() on
 * hy_tab_device_ptr(HY_BUF_STATE)). */
typedef struct hy_cfunc_s *hy_cfunc;
hy_cfunc hy_cfunc_new(const hy_expr *fn, size_t n_fn, const hy_expr *vars, size_t n_vars, int device);
void hy_cfunc_free(hy_cfunc);
uint32_t hy_cfunc_get_nparams(hy_cfunc);
uint32_t hy_cfunc_get_nvars(hy_cfunc);
uint32_t hy_cfunc_get_nouts(hy_cfunc);
int hy_cfunc_is_time_dependent(hy_cfunc);
/* get_dc() in textual form / generated HIP source; caller frees with hy_free_str(). */
char *hy_cfunc_decomposition_str(hy_cfunc);
char *hy_cfunc_get_hip_source(hy_cfunc);
int hy_cfunc_set_stream(hy_cfunc, void *hip_stream);
/* Call operator on host arrays: out_size = nouts * nevals, in_size = nvars * nevals, pars
 * (nparams * nevals; NULL if none), time (nevals; NULL if none). */
int hy_cfunc_eval(hy_cfunc, double *out, size_t out_size, const double *in, size_t in_size, const double *pars,
                  size_t pars_size, const double *time, size_t time_size);
/* Same on device-resident arrays, asynchronous on the stream. */
int hy_cfunc_eval_device(hy_cfunc, double *d_out, const double *d_in, const double *d_pars, const double *d_time,
                         uint64_t nevals);

/* ------------------------------------------------------------------------------------------------
 * ensemble_propagate_until_batch() (include/heyoka/ensemble_propagate.hpp:222-237,
 * src/ensemble_propagate.cpp:193-222): n_iter independent propagations of copies of `ta`.
 *  gen(ta_copy, iteration, user_data): sets the initial conditions of the copy (on the host),
 *      returns 0 on success. Invoked serially on the calling thread.
 *  Iterations are distributed round-robin over `n_devices` HIP devices (0 = all visible), one host
 *  thread per device; n_devices = -k: k host threads spread round-robin over the visible devices (more
 *  workers than devices). The propagated copies are returned in out[0..n_iter) (caller frees each with
 *  hy_tab_free()); every copy lives on the device which propagated it and its getters fetch from there. */
typedef int (*hy_ensemble_gen)(hy_tab ta_copy, size_t iteration, void *user_data);
int hy_ensemble_propagate_until_batch(hy_tab ta, double t, size_t n_iter, hy_ensemble_gen gen, void *gen_data,
                                      uint64_t max_steps, int n_devices, hy_tab *out);
int hy_ensemble_propagate_for_batch(hy_tab ta, double delta_t, size_t n_iter, hy_ensemble_gen gen, void *gen_data,
                                    uint64_t max_steps, int n_devices, hy_tab *out);
/* The final states of n integrators (e.g. the copies returned above, each on the device which propagated it) gathered
 * into ONE buffer: out[row * n_total + offset_i + lane], n_total = sum of the batch sizes, offset_i = sum of those of the
 * integrators before i; out is host memory (out_is_device = 0) or memory of device dst_device. Between devices the
 * blocks travel through RCCL (librccl loaded at run time: ncclSend / ncclRecv over xGMI inside one group; set
 * HEYOKA_AMD_GATHER_RCCL=1 to take that route on a single device too, =0 to force plain device-to-device copies, which are
 * also the fallback when librccl is absent). *used_rccl (may be NULL) reports the route taken. The reference returns the
 * iterations in host memory of one process (src/ensemble_propagate.cpp:193-297): this is how a caller of the drop-in gets
 * the 8-GPU ensemble in one place. */
int hy_ensemble_gather_states(const hy_tab *tabs, size_t n, int dst_device, double *out, size_t out_doubles,
                              int out_is_device, int *used_rccl);
/* The same gather with everything the reference's returned integrators hold (src/ensemble_propagate.cpp:193-297: m_state,
 * m_time_hi / m_time_lo, m_prop_res): out receives (dim + 6) rows of n_total 8-byte words, out[row * n_total + offset_i +
 * lane] - the dim state rows, then time_hi, time_lo (doubles), the outcome of the last propagation (int64_t), its number
 * of steps (uint64_t), min |h| and max |h| (doubles). Each integrator sends ONE packed block; the RCCL communicators are
 * created once per list of devices and kept. */
#define HY_GATHER_RESULT_ROWS 6
int hy_ensemble_gather_results(const hy_tab *tabs, size_t n, int dst_device, double *out, size_t out_words, int out_is_device,
                               int *used_rccl);

#ifdef __cplusplus
}
#endif

#endif
