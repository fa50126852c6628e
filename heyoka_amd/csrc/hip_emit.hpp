// HIP source generation for the batch Taylor stepper (gfx950 / CDNA4).
//
// Replaces the LLVM IR emission of the reference (taylor_add_adaptive_step(),
// src/taylor_00.cpp:712-865; taylor_compute_jet(), src/taylor_02.cpp:1307-1419; the per-node
// taylor_diff() rules under src/math/*.cpp) with generation of a HIP source module that is
// compiled at run time with hiprtc. One system per lane ("unrolled" mode) or one system per group
// of L lanes with the jets of the nonlinear sub-DAGs resident in registers ("cluster" mode).
#pragma once

#include <cstdint>
#include <array>
#include <functional>
#include <string>
#include <vector>

#include "decompose.hpp"

namespace heyoka_amd
{

// Kernel argument block. Must match the struct emitted in the generated source.
struct hy_kargs {
    double *state;            // [n_eq * N] rw
    const double *pars;       // [n_par * N]
    double *time_hi;          // [N] rw
    double *time_lo;          // [N] rw
    const double *lim;        // step mode: signed max step [N]; propagate mode: max_delta_t > 0 [N] or null
    const double *tfin_hi;    // propagate mode: final times (double-length) [N]
    const double *tfin_lo;    // [N]
    double *last_h;           // [N] out
    long long *outcome;       // [N] out (taylor_outcome)
    double *min_h;            // [N] out (propagate mode)
    double *max_h;            // [N] out (propagate mode)
    unsigned long long *n_steps; // [N] out (propagate mode)
    double *tc;               // [n_eq * (order + 1) * N] jet scratch == Taylor coefficients output
    unsigned long long N;     // number of systems
    unsigned long long max_steps; // propagate mode: 0 = unlimited
    int mode;                 // 0 = single step, 1 = propagate_until, 2 = raw step, 4 = step with events
    // Stepper with events which evaluates the event equations itself (emitted_module::events_in_stepper): bit 0 = every
    // workgroup stores its Taylor coefficients (otherwise only those in which an event is possible), bit 1 = store NOTHING
    // but the coefficients (the launch which regenerates them from the snapshot of the state before the step). 0 elsewhere.
    int pad;
    unsigned int *counters;   // [16] device counters: [0] lanes with non-finite state, [1] work-queue head
    double *scratch;          // cluster mode: jet scratch, scratch_per_wave doubles per resident wave
    double tfin_s_hi, tfin_s_lo; // propagate mode, scalar final time (used when tfin_hi == nullptr)
    // mode 4 (stepper with events, no state update): Taylor coefficients of the event equations
    // [(event * (order + 1) + k) * N + system] and max_i |x_i| per system.
    double *ev_tc;
    double *max_abs_state;
    // mode 4 on the wave-cluster steppers (jets of the state variables to tc, no state update): the three norms of the
    // step-size selector over the state variables, [3 * N] = max |x^[0]|, max |x^[p]|, max |x^[p-1]| per system. The
    // jets of the event equations and the final step size are computed from tc by hy_ev_jets (emit_event_jets()).
    // (With the event equations inside the stepper the first N entries carry its verdict per system for the detection
    // kernel instead: 0.0 = no event possible in this step.)
    double *sel_norms;
    // Taylor coefficients on demand (hy_kargs::pad bit 2, emitted_module::tc_by_threshold / grid_multi_step): a time per
    // system - the next grid point of propagate_grid() -; a step stores its coefficients only if it reaches that time, and in
    // propagate mode (mode 1) the system leaves the step loop after that step: one launch takes every system from one grid
    // point to the next instead of one step further.
    const double *tc_thr;
    // ... and, out, per system: 1.0 if its last step was clamped to the remaining time (h == rem.hi: the system has reached
    // its final time - the post-step kernel of propagate_grid() then treats it as done), 0.0 otherwise.
    double *grid_done;
};

enum class emit_mode { unrolled, cluster, table, block };

// Developer switches of the generators: experiments and the test matrix. ALL of them are read from the environment in ONE
// place, dev_switches::from_env() (hip_emit.cpp), when an integrator is constructed; the generators only see this struct.
//   HEYOKA_AMD_V5_EVENTS=0            stepper with events: not on the one-lane-per-pair kernel (lane-pair kernel instead)
//   HEYOKA_AMD_COMPACT_TC=0           stepper with events: full instead of compact Taylor coefficients
//   HEYOKA_AMD_NO_EVENTS_IN_STEPPER   event equations in a kernel of their own (hy_ev_jets) instead of inside the stepper
//   HEYOKA_AMD_NO_PAIR_EVENTS         close-encounter events through the generic statements, not on the lanes of their pairs
//   HEYOKA_AMD_NO_REFILL              one-lane-per-pair kernel: whole groups of systems from the queue, no per-system refill
//   HEYOKA_AMD_BLOCK_V2=0             block mode: generic cluster phase
//   HEYOKA_AMD_BLOCK_OPTS             block mode, v2 cluster phase: items switched one by one (see hip_emit_block.cpp)
//   HEYOKA_AMD_NO_STATE_ALIASES       planner: no alias u variables for state variables in history position
//   HEYOKA_AMD_CLUSTER_V1             first-generation cluster generator
//   HEYOKA_AMD_MULTI_CLASS=0          planner: no multi-class plans (clusters of several shapes / levels)
//   HEYOKA_AMD_TABLE_LDS=0|1          table stepper: tape in HBM / in LDS (default: by size)
//   HEYOKA_AMD_EV_INLINE_MAX_NONLINEAR  budget of nonlinear nodes of the event equations inside the stepper
//   HEYOKA_AMD_V5_PRIO, HEYOKA_AMD_V5_OPTS, HEYOKA_AMD_V5_PAD   one-lane-per-pair kernel: issue priorities, round-5 items
//                                     switched off one by one (A/B harness), sensitivity padding (see hip_emit_cluster2.cpp)
// (HEYOKA_AMD_EMIT_MODE, HEYOKA_AMD_ONE_LANE, HEYOKA_AMD_PAIR_SPLIT, HEYOKA_AMD_EVENTS_ON_CLUSTER map onto kwargs of the
// integrator - taylor_adaptive_batch.cpp -; HEYOKA_AMD_HIPRTC_FLAGS, HEYOKA_AMD_SCRATCH_GIB, HEYOKA_AMD_GATHER_RCCL and
// HEYOKA_AMD_EVENTS_TIMING belong to the runtime, not to code generation.)
struct dev_switches {
    bool v5_events = true, compact_tc = true, events_in_stepper = true, pair_events = true, refill = true, block_v2 = true,
         state_aliases = true, cluster_v1 = false, linearise = true, multi_class = true;
    int table_lds = -1, ev_inline_max_nonlinear = -1, v5_prio = 2;
    // HEYOKA_AMD_UNROLLED_WAVES=n: the straight-line stepper is compiled for n wavefronts per SIMD (amdgpu_waves_per_eu: the
    // register allocator gets 512 / n registers per lane and spills the rest - the two-wavefront experiment of round 5).
    int unrolled_waves = 0;
    // Straight-line stepper with register-resident jets (see emit_unrolled_kernel(); A/B: profiles/r06_two_body_unrolled_ab.log):
    // literal-zero coefficients folded in the generator and the Horner steps over leading zeros collapsed into one
    // (HEYOKA_AMD_UNROLLED_TRIM=0: off), one running sum for the sum of squares (HEYOKA_AMD_UNROLLED_MERGE_SSQ=0: off), the
    // coefficients of a variable which only defines x' = v re-derived from those of x (HEYOKA_AMD_UNROLLED_DERIVE=0: off).
    bool unrolled_trim = true, unrolled_merge_ssq = true, unrolled_derive = true;
    // Register-resident jets of the state variables up to this estimate of the doubles a lane keeps (reg_jet_estimate()).
    // (-1: automatic - 270 with the folded histories of the default arithmetic, 200 otherwise.)
    int reg_jets_max = -1;
    std::string v5_opts, v5_pad;
    // HEYOKA_AMD_BLOCK_OPTS: comma-separated items of the v2 cluster phase of block mode switched one by one (A/B harness,
    // timing experiments - see hip_emit_block.cpp).
    std::string block_opts;
    static dev_switches from_env();
};

struct emit_options {
    dev_switches dev;
    std::uint32_t order = 20;
    bool high_accuracy = false;
    emit_mode mode = emit_mode::unrolled;
    std::uint32_t block_size = 256;
    // Number of systems integrated at the same time (0: unknown); steers latency- vs throughput-oriented variants.
    std::uint64_t batch_size = 0;
    // Wave-cluster steppers: emit the stepper of an integrator with events - every launch is a mode-4 step (jets of the
    // state variables to a.tc, selector norms to a.sel_norms, no state update; emitted_module::cluster_mode4).
    bool event_stepper = false;
    // Wave-cluster generator (kw::cluster_kernel / hy_tab_config::cluster_kernel): 0 = automatic - the first one which
    // applies out of 5 (one lane per pair cluster, two wavefronts per SIMD), 3 (lane pairs), 2 (one lane per cluster,
    // pipelined orders), 1 (first generation) -, otherwise the search starts at the given generator.
    int cluster_kernel = 0;
    // Correctly rounded quotients in the recurrences of the pair kernels (division by the order, quotient of the pow
    // recurrence) instead of the reciprocal forms (within 1 ulp of them): kw::exact_division.
    bool exact_division = false;
    // Order of the additions inside the convolutions of the straight-line (unrolled) generator: 0 automatic = 2; 1 = the
    // reference's default mode (products first, pairwise sum: src/math/prod.cpp:386-395), 2 = the reference's compact mode
    // (running sum from 0: src/math/prod.cpp:686-698; one FMA per term). kw::sum_order. The table stepper always uses the
    // running sums, the cluster kernels their own FMA chains.
    int sum_order = 0;
    // Decompositions the wave-cluster / block planners cannot shape: straight-line code up to this many nodes, the
    // table-driven steppers beyond (kw::compact_mode lowers it: short compile times).
    std::uint32_t unroll_max_nodes = 150;
    // Block mode: linear nodes fed by one cluster stay glue nodes (plan_limits::absorb_linear off) - the internal program whose
    // scalings were moved out of the clusters for the v2 cluster phase (externalise_scalings()).
    bool block_no_absorb = false;
    // emit_event_jets(): the stepper it accompanies leaves out the Taylor coefficients of order >= 1 of the state variables
    // defined by another state variable (emitted_module::compact_tc): read them as parent^[k-1] / k.
    bool compact_tc = false;
    // Stepper with events on the one-lane-per-pair kernel: the decomposition of the system WITH the event equations
    // (prog.ev_u). When set, the stepper evaluates the event equations itself from
    // the jets of the state variables in LDS, extends the norms of the step-size selector to them, takes the final step
    // size and updates the state (emitted_module::events_in_stepper): hy_ev_jets and the dense-output pass over the Taylor
    // coefficients drop out of a step.
    const taylor_program *ev_prog = nullptr;
    // Number of TERMINAL events among the event equations of ev_prog (they come first). Without terminal events no step is
    // ever truncated, so nothing behind the stepper reads the Taylor coefficients of the state variables: they are stored
    // on request only (hy_kargs::pad bit 0; a later get_tc() regenerates them from the snapshot of the step).
    std::uint32_t n_t_events = 0;
    // emit_event_jets(): only the kernels which serve the compact Taylor coefficients (hy_dout_c, hy_tc_expand) - the
    // stepper evaluates the event equations itself (emitted_module::events_in_stepper).
    bool ev_helpers_only = false;
};

struct emitted_module {
    std::string source;
    std::string kernel_name; // step / propagate kernel
    std::string dout_name;   // dense-output kernel
    std::string tc_kernel_name; // optional variant of the stepper that writes the Taylor coefficients
    bool tc_optional = false;   // the main kernel works with a.tc == nullptr
    std::uint32_t block_size = 256;
    std::uint32_t lanes_per_system = 1;
    std::uint32_t n_clusters = 0; // cluster / block modes
    std::uint32_t lds_bytes = 0;
    emit_mode mode = emit_mode::unrolled;
    // Statistics (logged like the reference logs decomposition sizes).
    std::uint64_t n_statements = 0;
    std::string notes;
    // Cluster mode: doubles of jet scratch needed per resident wave.
    std::uint64_t scratch_per_wave = 0;
    bool persistent = false;
    // Extra hiprtc options of the module (space separated), on top of the common ones.
    std::string compile_flags;
    // The stepper implements mode 4 in its cluster form (see hy_kargs::sel_norms).
    bool cluster_mode4 = false;
    // Mode 4 writes a COMPACT set of Taylor coefficients: for a state variable x defined by another state variable
    // (x' = v) only the order-0 row; x^[k] = v^[k-1] / k is left to the consumers (hy_ev_jets, hy_dout_c, hy_tc_expand in
    // the module of emit_event_jets()): half of the 6 KB per system of an N-body ensemble never travel.
    bool compact_tc = false;
    // Mode 4 also evaluates the event equations, takes the final step size and updates the state (emit_options::ev_prog).
    bool events_in_stepper = false;
    // Single-step launches (mode 0) understand hy_kargs::pad bit 2: a.tfin_hi[] holds a time per system, and the Taylor
    // coefficients of a step are stored only if the step reaches it (or has length zero). The lock-step loop of
    // propagate_grid() passes the next grid time of every system: 6 GB of coefficients per sweep of 1 048 576 outer Solar
    // Systems otherwise, for dense output which a handful of steps need.
    bool tc_by_threshold = false;
    // Propagate-mode launches (mode 1) understand hy_kargs::pad bit 2 as well: see hy_kargs::tc_thr.
    bool grid_multi_step = false;
    // When the code was generated from a rewritten INTERNAL program (state-variable aliases, padded clusters, restored unit
    // scalings - the user-visible decomposition is never touched): its text, one node per line in the format of the
    // decomposition strings, then the definitions of the state derivatives. Lets the tests run the oracle's interpreter
    // on the rewritten program and check that the rewrites do not change a single bit of the jets.
    std::string internal_program;
};

// Textual form of a flattened program (see emitted_module::internal_program).
std::string program_to_string(const taylor_program &);

emitted_module emit_hip_module(const taylor_program &prog, const emit_options &opts);

// Companion of the wave-cluster steppers for integrators with events: kernel hy_ev_jets, one system per lane - the jets
// of the event equations from the jets of the state variables in a.tc (only the part of the decomposition the event
// equations depend on), the norms of the step-size selector extended to the event equations and the final step size.
// prog is the decomposition of the system *with* the event equations (prog.ev_u). Returns a module with an empty source
// (and the reason in why_not) when the event equations need too much of the decomposition.
emitted_module emit_event_jets(const taylor_program &prog, const emit_options &opts, std::string &why_not);

// The same computation as straight-line statements for use INSIDE a stepper (the one-lane-per-pair kernel in mode 4): sv(i, k)
// gives the expression of the order-k coefficient of state variable i (every use of a coefficient goes through ONE
// definition: one LDS read), ev_store(event, k, value) the statement which publishes a coefficient of an event equation.
// Appends the statements to `out` and returns, per event equation, the names of its coefficients by order (for the norms
// of the step-size selector and the exclusion test). Returns false (and the reason) if the event equations depend on
// too much of the decomposition or on functions defined through node rules.
// Lanes of a system as a vector unit for ISOMORPHIC terms of a sum (optional): the event equations of N-body problems are
// sums of a few terms of one shape - (x_1 - x_2)^2 + (y_1 - y_2)^2 + (z_1 - z_2)^2, x vx + y vy + z vz - whose convolutions
// dominate the cost. With these hooks the terms of such a sum are evaluated ONCE, term c by lane c of the system (leaf
// position p of the shared shape reads the state variable leaf_vars[p][c]), and the sum collects them with lane broadcasts
// in the order of its arguments - the same operations on the same operands as the term-by-term evaluation, bit for bit.
struct ev_lane_hooks {
    // Access class of a state variable (variables of one class are read by the same expression up to an offset).
    std::function<int(std::uint32_t)> sv_class;
    // Coefficient k of the state variable which THIS lane holds at leaf position p (class cls).
    std::function<std::string(std::uint32_t p, std::uint32_t k, int cls)> sv_lane;
    // Value of v in lane c of the system.
    std::function<std::string(const std::string &v, std::uint32_t c)> lane_bcast;
    std::uint32_t max_terms = 4;
    // Out: per leaf position the state variable of term c (c < number of terms of the sum, <= max_terms), and its class.
    std::vector<std::vector<std::uint32_t>> leaf_vars;
    std::vector<int> leaf_class;
};

// skip_events (optional, one flag per event equation): event equations which the caller evaluates by other means - their
// entry of ev_coeffs stays empty and the nodes only they depend on are not generated.
bool emit_event_jets_inline(const taylor_program &prog, const emit_options &opts,
                            const std::function<std::string(std::uint32_t, std::uint32_t)> &sv,
                            const std::function<std::string(std::uint32_t, std::uint32_t, const std::string &)> &ev_store,
                            std::string &out, std::vector<std::vector<std::string>> &ev_coeffs, std::string &why_not,
                            ev_lane_hooks *lanes = nullptr, const std::vector<char> *skip_events = nullptr);

// Is the event equation u (a u variable of prog) a squared distance plus a constant,
//     +-((a_0 - b_0)^2 + (a_1 - b_1)^2 + (a_2 - b_2)^2) + c,        a_i, b_i state variables,
// written with products, pow(., 2) or sum_sq and sums / differences in any nesting? The close-encounter events of an
// N-body problem have this form - and the pair kernels hold the Taylor coefficients of exactly these squared distances.
struct pair_distance_event {
    std::array<std::pair<std::uint32_t, std::uint32_t>, 3> diffs; // (a_i, b_i), state variable indices
    double c = 0;
    double sign = 1; // -1: the squares are subtracted, c - |r_i - r_j|^2
};
bool match_pair_distance_event(const taylor_program &prog, std::uint32_t u, pair_distance_event &out);

// Format a double as a C++17 hexadecimal floating-point literal (exact round trip).
std::string fp_literal(double);

} // namespace heyoka_amd
