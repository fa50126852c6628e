// Drop-in counterpart of heyoka::taylor_adaptive_batch<double> running on MI355X.
//
// Reference interface: include/heyoka/taylor.hpp:781-1121 (class), :142-155 (taylor_outcome),
// src/taylor_adaptive_batch.cpp:78-427 (construction), :632-727 (step), :1082-1534 (propagate_for/until),
// :2251-2327 (update_d_output). "batch_size" is the number of systems integrated concurrently: the
// SIMD width of the reference becomes the number of GPU lanes here (10^6 and beyond).
#pragma once

#include <cmath>
#include <cstddef>
#include <array>
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <limits>
#include <locale>
#include <ostream>
#include <sstream>
#include <memory>
#include <optional>
#include <stdexcept>
#include <span>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "continuous_output.hpp"
#include "decompose.hpp"
#include "dfloat.hpp"
#include "event_detection.hpp"
#include "expression.hpp"
#include "kw.hpp"
#include "step_callback.hpp"

namespace heyoka_amd
{

// Reference: include/heyoka/taylor.hpp:142-155.
enum class taylor_outcome : std::int64_t {
    success = -4294967296ll - 1,
    step_limit = -4294967296ll - 2,
    time_limit = -4294967296ll - 3,
    err_nf_state = -4294967296ll - 4,
    cb_stop = -4294967296ll - 5
};

// Human-readable outcome ("taylor_outcome::success", "taylor_outcome::terminal_event_2 (stopping)", ...): the output
// format of the reference's operator<< (src/taylor_stream_ops.cpp:244-266), which its tutorials print.
inline std::ostream &operator<<(std::ostream &os, taylor_outcome oc)
{
    switch (oc) {
        case taylor_outcome::success:
            return os << "taylor_outcome::success";
        case taylor_outcome::step_limit:
            return os << "taylor_outcome::step_limit";
        case taylor_outcome::time_limit:
            return os << "taylor_outcome::time_limit";
        case taylor_outcome::err_nf_state:
            return os << "taylor_outcome::err_nf_state";
        case taylor_outcome::cb_stop:
            return os << "taylor_outcome::cb_stop";
        default:
            break;
    }
    const auto v = static_cast<std::int64_t>(oc);
    if (v >= 0) {
        return os << "taylor_outcome::terminal_event_" << v << " (continuing)";
    }
    if (oc > taylor_outcome::success) {
        return os << "taylor_outcome::terminal_event_" << (-v - 1) << " (stopping)";
    }
    return os << "taylor_outcome::??";
}

template <typename T>
class taylor_adaptive_batch;

// Type-erased step callback bool(taylor_adaptive_batch<T> &) with the optional pre_hook() member, and sets of them
// (reference: include/heyoka/step_callback.hpp:46-62, :139-185) - step_callback.hpp.
template <typename T>
using step_callback_batch = detail::step_cb_wrap<taylor_adaptive_batch<T>>;
template <typename T>
using step_callback_batch_set = detail::step_cb_set<taylor_adaptive_batch<T>>;

namespace detail
{

// Non-template implementation of the fp64 batch integrator.
class tab_core
{
public:
    using sys_t = std::vector<std::pair<expression, expression>>;

    struct config {
        std::optional<double> tol;
        bool high_accuracy = false;
        bool compact_mode = false;
        bool parallel_mode = false;
        std::vector<double> pars;
        std::vector<double> time; // empty -> zeros; size 1 -> splat; else size == batch_size.
        bool time_is_scalar = false;
        int device = 0;
        // Event detection (reference: kw::t_events / kw::nt_events, include/heyoka/taylor.hpp:814-821).
        std::vector<core_t_event> t_events;
        std::vector<core_nt_event> nt_events;
        // ---- MI355X extensions (kw::emitter, kw::cluster_kernel, kw::exact_division, kw::events_on_cluster,
        // kw::batch_semantics; hy_tab_config carries the same fields). ----
        // Code generator: 0 automatic, 1 unrolled, 2 wave-cluster, 3 table, 4 block.
        int emitter = 0;
        // Wave-cluster generator: 0 automatic, or 5 / 3 / 2 / 1 (see emit_options::cluster_kernel).
        int cluster_kernel = 0;
        bool exact_division = false;
        // Order of the additions inside the convolutions of the unrolled generator: 0 automatic (running sums), 1 the
        // reference's default mode (pairwise), 2 its compact mode (running sums) - emit_options::sum_order.
        int sum_order = 0;
        // Events next to a system which qualifies for a wave-cluster stepper: 0 automatic (cluster stepper + hy_ev_jets when
        // the event equations are small), 1 always the one-system-per-lane steppers with events.
        int events_on_cluster = 0;
        // Outcomes of propagate_for / propagate_until when a lane produces a non-finite state or max_steps is hit:
        //   0 (default) the reference's batch-wide results (src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467, :1516),
        //     obtained on the device-resident path: a step-limited batch reports step_limit in every lane, and a batch
        //     with a non-finite lane is rolled back and re-run through the lock-step loop;
        //   1 always the lock-step loop (one launch per iteration of the batch);
        //   2 per-lane results of the device-resident path (every lane stops on its own), no snapshot, no host sync.
        int batch_semantics = 0;
    };

    // Default construction: an empty object which can only be destroyed, assigned to or copied (the reference's
    // default constructor leaves the integrator in an invalid state, src/taylor_adaptive_batch.cpp:430).
    tab_core() noexcept;
    tab_core(sys_t sys, std::vector<double> state, std::uint32_t batch_size, config cfg);
    tab_core(const tab_core &);
    tab_core(tab_core &&) noexcept;
    tab_core &operator=(const tab_core &);
    tab_core &operator=(tab_core &&) noexcept;
    ~tab_core();

    // ---- getters ----
    [[nodiscard]] const taylor_dc_t &get_decomposition() const;
    [[nodiscard]] const taylor_program &get_program() const;
    [[nodiscard]] std::uint32_t get_batch_size() const;
    [[nodiscard]] std::uint32_t get_order() const;
    [[nodiscard]] double get_tol() const;
    [[nodiscard]] bool get_high_accuracy() const;
    [[nodiscard]] bool get_compact_mode() const;
    // Number of events which the device-side detection ignored in a step because the root isolation or the root finder
    // failed (the reference logs a warning, src/detail/event_detection.cpp:2082-2090).
    [[nodiscard]] std::uint64_t get_event_detection_failures() const;
    [[nodiscard]] std::uint32_t get_dim() const;
    [[nodiscard]] const sys_t &get_sys() const;
    [[nodiscard]] int get_device() const;
    [[nodiscard]] const std::string &get_hip_source() const;
    // Text of the rewritten internal program the stepper was generated from (empty: the decomposition itself).
    [[nodiscard]] const std::string &get_internal_program() const;
    // The gfx950 code object of the stepper module (the counterpart of llvm_state::get_object_code()).
    [[nodiscard]] const std::vector<char> &get_code_object() const;
    [[nodiscard]] double get_compile_seconds() const;
    // "unrolled" / "cluster ..." / "table ...": which code generator was selected, and why.
    [[nodiscard]] std::string get_codegen_info() const;

    [[nodiscard]] const std::vector<double> &get_time() const;
    [[nodiscard]] std::pair<const std::vector<double> &, const std::vector<double> &> get_dtime() const;
    void set_time(const std::vector<double> &);
    void set_time(double);
    void set_dtime(const std::vector<double> &, const std::vector<double> &);
    void set_dtime(double, double);

    [[nodiscard]] const std::vector<double> &get_state() const;
    // References to the host mirrors of state / times are being kept by the caller: refresh them after every kernel.
    void hold_host_refs() const;
    void hold_time_refs() const;
    [[nodiscard]] double *get_state_data();
    [[nodiscard]] const std::vector<double> &get_pars() const;
    [[nodiscard]] double *get_pars_data();
    // Setters by value (the C ABI's hy_tab_set_state() / hy_tab_set_pars()): the mirrors are brought up to date, overwritten
    // and marked newer than the device - no pointer leaves the object, so it does NOT switch to the eager synchronisation
    // which a handed-out mutable pointer requires (a download of the state after every launch).
    void set_state_values(const double *in);
    void set_pars_values(const double *in);
    [[nodiscard]] const std::vector<double> &get_tc() const;
    [[nodiscard]] const std::vector<double> &get_last_h() const;
    [[nodiscard]] const std::vector<double> &get_d_output() const;
    const std::vector<double> &update_d_output(const std::vector<double> &, bool rel_time);
    const std::vector<double> &update_d_output(double, bool rel_time);

    [[nodiscard]] const std::vector<std::tuple<taylor_outcome, double>> &get_step_res() const;
    [[nodiscard]] const std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> &get_propagate_res() const;

    // ---- stepping ----
    void step(bool wtc);
    void step_backward(bool wtc);
    void step(const std::vector<double> &max_delta_ts, bool wtc);

    using cb_t = std::function<bool()>;
    // ts: final times (size 1 = scalar splat, else batch_size); max_delta_ts: empty or batch_size.
    void finish_device_propagate(const std::vector<double> &ts, std::size_t max_steps, const std::vector<double> &max_delta_ts,
                                 bool wtc);
    // pre: the pre_hook() of the step callback (include/heyoka/step_callback.hpp:46-62), run once after the validation of
    // the arguments and before the first step (src/taylor_adaptive_batch.cpp:1356-1365, :1782-1791); it must not move the
    // time coordinate.
    using pre_t = std::function<void()>;
    void propagate_until(const std::vector<double> &ts, std::size_t max_steps, const std::vector<double> &max_delta_ts,
                         const cb_t &cb, bool wtc, bool c_out, const pre_t &pre = {});
    void propagate_for(const std::vector<double> &delta_ts, std::size_t max_steps,
                       const std::vector<double> &max_delta_ts, const cb_t &cb, bool wtc, bool c_out,
                       const pre_t &pre = {});
    std::vector<double> propagate_grid(std::vector<double> grid, std::size_t max_steps,
                                       const std::vector<double> &max_delta_ts, const cb_t &cb,
                                       double *d_out = nullptr, const pre_t &pre = {});
    // Device-resident loop of propagate_grid(): see taylor_adaptive_batch.cpp.
    void propagate_grid_device_loop(const std::vector<double> &grid, std::vector<double> &retval,
                                    const std::vector<dfloat> &rem, const std::vector<int> &t_dir,
                                    const std::vector<double> &max_delta_ts, std::size_t max_steps,
                                    double *d_out, const cb_t &cb);
    // ---- events (reference: taylor.hpp:1008-1014) ----
    [[nodiscard]] bool with_events() const;
    [[nodiscard]] const std::vector<core_t_event> &get_t_events() const;
    [[nodiscard]] const std::vector<core_nt_event> &get_nt_events() const;
    [[nodiscard]] const std::vector<std::vector<std::optional<std::pair<double, double>>>> &get_te_cooldowns() const;
    void reset_cooldowns();
    void reset_cooldowns(std::uint32_t batch_idx);
    // Opaque pointer handed to the event callbacks (the address of the user-facing integrator object).
    void set_callback_context(void *ctx);
    // The continuous output recorded by the last propagate_for/until() invoked with c_out = true
    // (empty if no step was taken); the object is moved out.
    std::optional<c_out_core> take_c_output();

    // ---- device-resident access (MI355X extension, used by the ensemble / benchmark paths) ----
    // Raw device pointers to the SoA arrays (array[row * batch_size + lane]). Calling any of these
    // makes the device copy authoritative until a host getter is used again.
    double *device_state();
    double *device_pars();
    double *device_time_hi();
    double *device_time_lo();
    double *device_tc();
    // 0: n_steps (uint64), 1: outcome (int64), 2: last_h (double).
    void *device_aux(int which);
    // Everything a propagation leaves behind, packed into ONE device buffer of (dim + 6) rows of batch_size 8-byte words
    // (on this integrator's device, copies enqueued on its stream): the state rows, then time_hi, time_lo, outcome (int64),
    // number of steps (uint64), min |h|, max |h| - what the ensemble gather moves between devices in one piece
    // (reference: ensemble_propagate_*() returns whole integrators, src/ensemble_propagate.cpp:193-297).
    static constexpr std::size_t n_result_rows = 6;
    void pack_results(double *d_dst);
    // Accounting of the steps with events: {steps, ms upload / buffers, ms stepper (+ event jets), ms detection, ms
    // bookkeeping + flags to the host, ms state update + records, regeneration launches of the Taylor coefficients,
    // systems which reported events}. The phase times are taken only with the timing on (a synchronisation per phase).
    void set_event_timing(bool on);
    [[nodiscard]] std::array<double, 8> get_event_stats() const;
    // Mark the device copies as modified by the caller (e.g. initial conditions written by a kernel).
    void mark_device_modified();
    void set_stream(void *hip_stream);
    // Move the integrator to another HIP device (device buffers are re-created lazily).
    void set_device(int device);
    void synchronize();
    // Total number of integration steps taken by the last propagate_*() call (sum over lanes).
    [[nodiscard]] std::uint64_t get_last_total_steps() const;
    // Durations (ms) of the last n stepper kernel launches (HIP events on the launch stream).
    [[nodiscard]] std::vector<double> get_kernel_ms_history(std::size_t n) const;
    // Stepper function-pointer ABI of the reference (include/heyoka/detail/ta_jit_data.hpp:35-44)
    // on caller-provided device buffers: state rw, h in = signed max step, out = step taken.
    void raw_step(double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                  std::uint64_t n_systems, void *d_tape = nullptr);
    // The other pointer types of that ABI (ta_jit_data.hpp:34-43): the stepper with events - jets of the state variables
    // and of the event equations, step size and max |x_i|, NO state update (taylor_add_adaptive_step_with_events(),
    // src/taylor_00.cpp:592-710) -, the dense output (taylor_add_d_out_function(), src/taylor_01.cpp:1015-1185) and the
    // size / alignment of the tape of the compact-mode steppers (c_step_f_t: the caller owns the tape).
    void raw_step_e(double *d_jet, const double *d_state, const double *d_pars, const double *d_time, double *d_h,
                    double *d_max_abs_state, std::uint64_t n_systems, void *d_tape = nullptr);
    void raw_d_out_f(double *d_out, const double *d_tc, const double *d_h, std::uint64_t n_systems);
    [[nodiscard]] std::pair<std::size_t, std::size_t> raw_tape_size_align(std::uint64_t n_systems);

private:
    struct impl;
    std::unique_ptr<impl> m_impl;
};

std::vector<double> make_vector_from(double x);

// HIP source of the post-step kernel of the device-resident propagate_grid() loop (exposed for the build-time
// compilation check).
std::string make_grid_source(std::uint32_t order, std::uint32_t dim, bool high_accuracy);

} // namespace detail

// Event classes of the batch integrator (reference: nt_event_batch<T> / t_event_batch<T>,
// include/heyoka/events.hpp:52-325). Callback signatures as in the reference:
//   non-terminal: void(taylor_adaptive_batch<T> &, T time, int d_sgn, std::uint32_t batch_idx)
//   terminal:     bool(taylor_adaptive_batch<T> &, int d_sgn, std::uint32_t batch_idx) (false -> stop).
template <typename T>
class nt_event_batch
{
public:
    using callback_t = std::function<void(taylor_adaptive_batch<T> &, T, int, std::uint32_t)>;

private:
    expression m_eq;
    callback_t m_cb;
    event_direction m_dir = event_direction::any;

public:
    // Default construction: the event equation 0 with a callback which does nothing
    // (test/batch_event_detection.cpp:1020-1027).
    nt_event_batch() : m_eq(0.), m_cb([](taylor_adaptive_batch<T> &, T, int, std::uint32_t) {}) {}
    template <typename... KwArgs>
    explicit nt_event_batch(expression e, callback_t cb, const KwArgs &...kw_args)
        : m_eq(std::move(e)), m_cb(std::move(cb)),
          m_dir(static_cast<event_direction>(kw::get(kw::direction, event_direction::any, kw_args...)))
    {
        static_assert(kw::all_named_v<KwArgs...>);
        if (!m_cb) {
            throw std::invalid_argument("Cannot construct a non-terminal event with an empty callback");
        }
        check_dir();
    }
    [[nodiscard]] const expression &get_expression() const
    {
        return m_eq;
    }
    [[nodiscard]] const callback_t &get_callback() const
    {
        return m_cb;
    }
    [[nodiscard]] event_direction get_direction() const
    {
        return m_dir;
    }

private:
    void check_dir() const
    {
        if (m_dir != event_direction::any && m_dir != event_direction::positive && m_dir != event_direction::negative) {
            throw std::invalid_argument("Invalid value selected for the direction of a non-terminal event");
        }
    }
};

template <typename T>
class t_event_batch
{
public:
    using callback_t = std::function<bool(taylor_adaptive_batch<T> &, int, std::uint32_t)>;

private:
    expression m_eq;
    callback_t m_cb;
    event_direction m_dir = event_direction::any;
    T m_cooldown = -1;

public:
    // Default construction: the event equation 0, no callback, any direction, automatic cooldown
    // (test/batch_event_detection.cpp:1818-1826).
    t_event_batch() : m_eq(0.) {}
    template <typename... KwArgs>
    explicit t_event_batch(expression e, const KwArgs &...kw_args)
        : m_eq(std::move(e)),
          m_dir(static_cast<event_direction>(kw::get(kw::direction, event_direction::any, kw_args...))),
          m_cooldown(static_cast<T>(kw::get(kw::cooldown, -1., kw_args...)))
    {
        static_assert(kw::all_named_v<KwArgs...>);
        if constexpr (kw::has_v<kw::callback_tag, KwArgs...>) {
            m_cb = kw::get(kw::callback, 0, kw_args...);
        }
        if (m_dir != event_direction::any && m_dir != event_direction::positive && m_dir != event_direction::negative) {
            throw std::invalid_argument("Invalid value selected for the direction of a terminal event");
        }
        if (!std::isfinite(m_cooldown)) {
            throw std::invalid_argument("Cannot set a non-finite cooldown value for a terminal event");
        }
    }
    [[nodiscard]] const expression &get_expression() const
    {
        return m_eq;
    }
    [[nodiscard]] const callback_t &get_callback() const
    {
        return m_cb;
    }
    [[nodiscard]] event_direction get_direction() const
    {
        return m_dir;
    }
    [[nodiscard]] T get_cooldown() const
    {
        return m_cooldown;
    }
};

// Summaries of the event objects (src/nt_event.cpp:163-173, src/t_event.cpp:138-152).
template <typename T>
inline std::ostream &operator<<(std::ostream &os, const nt_event_batch<T> &e)
{
    os << "C++ datatype   : double\n";
    os << "Event type     : non-terminal\n";
    os << "Event equation : " << e.get_expression().to_string() << '\n';
    os << "Event direction: " << e.get_direction() << '\n';
    return os;
}
template <typename T>
inline std::ostream &operator<<(std::ostream &os, const t_event_batch<T> &e)
{
    os << "C++ datatype   : double\n";
    os << "Event type     : terminal\n";
    os << "Event equation : " << e.get_expression().to_string() << '\n';
    os << "Event direction: " << e.get_direction() << '\n';
    os << "With callback  : " << (e.get_callback() ? "yes" : "no") << '\n';
    os << "Cooldown       : ";
    if (e.get_cooldown() < 0) {
        os << "auto";
    } else {
        std::ostringstream oss;
        oss.precision(17);
        oss << e.get_cooldown();
        os << oss.str();
    }
    return os << '\n';
}

template <>
class taylor_adaptive_batch<double>
{
    detail::tab_core m_core;
    std::vector<t_event_batch<double>> m_t_events;
    std::vector<nt_event_batch<double>> m_nt_events;

    template <typename... KwArgs>
    static detail::tab_core::config make_config(KwArgs &&...kw_args)
    {
        static_assert(kw::all_named_v<KwArgs...>,
                      "Only named arguments (kw::name = value) can follow the batch size in the constructor");
        detail::tab_core::config cfg;
        if constexpr (kw::has_v<kw::tol_tag, KwArgs...>) {
            const auto tol = static_cast<double>(kw::get(kw::tol, 0., kw_args...));
            // NOTE: tol == 0 is interpreted as undefined (reference: taylor.hpp:838-846).
            if (tol != 0) {
                cfg.tol = tol;
            }
        }
        cfg.high_accuracy = static_cast<bool>(kw::get(kw::high_accuracy, false, kw_args...));
        cfg.compact_mode = static_cast<bool>(kw::get(kw::compact_mode, false, kw_args...));
        cfg.emitter = static_cast<int>(kw::get(kw::emitter, 0, kw_args...));
        cfg.cluster_kernel = static_cast<int>(kw::get(kw::cluster_kernel, 0, kw_args...));
        cfg.exact_division = static_cast<bool>(kw::get(kw::exact_division, false, kw_args...));
        cfg.sum_order = static_cast<int>(kw::get(kw::sum_order, 0, kw_args...));
        cfg.events_on_cluster = static_cast<int>(kw::get(kw::events_on_cluster, 0, kw_args...));
        cfg.batch_semantics = static_cast<int>(kw::get(kw::batch_semantics, 0, kw_args...));
        cfg.parallel_mode = static_cast<bool>(kw::get(kw::parallel_mode, false, kw_args...));
        cfg.device = static_cast<int>(kw::get(kw::device, 0, kw_args...));
        if constexpr (kw::has_v<kw::pars_tag, KwArgs...>) {
            // NOTE: a std::vector<double> handed over as an rvalue is moved all the way into the integrator
            // (test/taylor_adaptive_batch.cpp:1368-1393).
            using pars_arg_t = decltype(kw::get(kw::pars, 0, std::forward<KwArgs>(kw_args)...));
            if constexpr (std::is_same_v<pars_arg_t, std::vector<double> &&>) {
                cfg.pars = kw::get(kw::pars, 0, std::forward<KwArgs>(kw_args)...);
            } else {
                for (const auto &x : kw::get(kw::pars, 0, kw_args...)) {
                    cfg.pars.push_back(static_cast<double>(x));
                }
            }
        }
        if constexpr (kw::has_v<kw::time_tag, KwArgs...>) {
            using time_t = std::decay_t<decltype(kw::get(kw::time, 0, kw_args...))>;
            if constexpr (std::is_arithmetic_v<time_t>) {
                cfg.time = {static_cast<double>(kw::get(kw::time, 0, kw_args...))};
                cfg.time_is_scalar = true;
            } else {
                for (const auto &x : kw::get(kw::time, 0, kw_args...)) {
                    cfg.time.push_back(static_cast<double>(x));
                }
            }
        }
        // Events: the callbacks are type-erased; the integrator object is handed to them through the context pointer
        // set by every stepping / propagation member.
        using self_t = taylor_adaptive_batch<double>;
        if constexpr (kw::has_v<kw::t_events_tag, KwArgs...>) {
            for (const auto &ev : kw::get(kw::t_events, 0, kw_args...)) {
                detail::core_t_event ce;
                ce.eq = ev.get_expression();
                ce.dir = ev.get_direction();
                ce.cooldown = ev.get_cooldown();
                if (const auto &cb = ev.get_callback()) {
                    ce.callback = [cb](void *ctx, int d_sgn, std::uint32_t idx) {
                        return cb(*static_cast<self_t *>(ctx), d_sgn, idx);
                    };
                }
                cfg.t_events.push_back(std::move(ce));
            }
        }
        if constexpr (kw::has_v<kw::nt_events_tag, KwArgs...>) {
            for (const auto &ev : kw::get(kw::nt_events, 0, kw_args...)) {
                detail::core_nt_event ce;
                ce.eq = ev.get_expression();
                ce.dir = ev.get_direction();
                const auto &cb = ev.get_callback();
                ce.callback = [cb](void *ctx, double tm, int d_sgn, std::uint32_t idx) {
                    cb(*static_cast<self_t *>(ctx), tm, d_sgn, idx);
                };
                cfg.nt_events.push_back(std::move(ce));
            }
        }
        return cfg;
    }

    template <typename... KwArgs>
    auto propagate_common_ops(KwArgs &&...kw_args)
    {
        static_assert(kw::all_named_v<KwArgs...>);
        const auto max_steps = static_cast<std::size_t>(kw::get(kw::max_steps, 0, kw_args...));
        std::vector<double> max_delta_ts;
        if constexpr (kw::has_v<kw::max_delta_t_tag, KwArgs...>) {
            using mdt_t = std::decay_t<decltype(kw::get(kw::max_delta_t, 0, kw_args...))>;
            if constexpr (std::is_arithmetic_v<mdt_t>) {
                max_delta_ts.assign(get_batch_size(), static_cast<double>(kw::get(kw::max_delta_t, 0, kw_args...)));
            } else {
                for (const auto &x : kw::get(kw::max_delta_t, 0, kw_args...)) {
                    max_delta_ts.push_back(static_cast<double>(x));
                }
            }
        }
        // kw::callback: one callback (anything a step_callback_batch can be built from), or a range of callbacks, which
        // becomes a step_callback_batch_set (parse_propagate_cb(), include/heyoka/taylor.hpp:237-260). The callback object
        // lives in a shared holder for the duration of the call: the core invokes it through `cb` / `pre`, and the very
        // same object - with whatever state it accumulated - is handed back to the caller.
        auto user_cb = std::make_shared<step_callback_batch<double>>();
        if constexpr (kw::has_v<kw::callback_tag, KwArgs...>) {
            using cb_arg_t = std::decay_t<decltype(kw::get(kw::callback, 0, kw_args...))>;
            // NOTE: a callback passed as an rvalue (kw::callback = std::move(cb), or a temporary) is MOVED into the holder
            // and never copied afterwards (test/taylor_adaptive_batch.cpp:497-526, :688-730).
            if constexpr (std::is_constructible_v<step_callback_batch<double>, const cb_arg_t &>) {
                *user_cb = step_callback_batch<double>(kw::get(kw::callback, 0, std::forward<KwArgs>(kw_args)...));
            } else {
                std::vector<step_callback_batch<double>> v;
                for (auto &&x : kw::get(kw::callback, 0, std::forward<KwArgs>(kw_args)...)) {
                    v.emplace_back(std::forward<decltype(x)>(x));
                }
                *user_cb = step_callback_batch<double>(step_callback_batch_set<double>(std::move(v)));
            }
        }
        detail::tab_core::cb_t cb;
        detail::tab_core::pre_t pre;
        if (*user_cb) {
            cb = [this, user_cb]() { return (*user_cb)(*this); };
            pre = [this, user_cb]() { user_cb->pre_hook(*this); };
        }
        const auto wtc = static_cast<bool>(kw::get(kw::write_tc, false, kw_args...));
        const auto c_out = static_cast<bool>(kw::get(kw::c_output, false, kw_args...));
        return std::tuple{max_steps, std::move(max_delta_ts), std::move(cb), std::move(user_cb), wtc, c_out, std::move(pre)};
    }

public:
    using sys_t = detail::tab_core::sys_t;
    // (include/heyoka/taylor.hpp:789-795.)
    using value_type = double;
    using nt_event_t = nt_event_batch<double>;
    using t_event_t = t_event_batch<double>;

    // Default construction leaves the object in an invalid state: only destruction, assignment and copy are allowed
    // (include/heyoka/taylor.hpp:901, test/taylor_adaptive_batch.cpp:1253-1263).
    taylor_adaptive_batch() noexcept = default;

    template <typename... KwArgs>
    taylor_adaptive_batch(sys_t sys, std::vector<double> state, std::uint32_t batch_size, KwArgs &&...kw_args)
        : m_core(std::move(sys), std::move(state), batch_size, make_config(std::forward<KwArgs>(kw_args)...))
    {
        // The user-facing event objects, for get_t_events() / get_nt_events() (include/heyoka/taylor.hpp:1009-1024).
        if constexpr (kw::has_v<kw::t_events_tag, KwArgs...>) {
            for (const auto &ev : kw::get(kw::t_events, 0, kw_args...)) {
                m_t_events.push_back(ev);
            }
        }
        if constexpr (kw::has_v<kw::nt_events_tag, KwArgs...>) {
            for (const auto &ev : kw::get(kw::nt_events, 0, kw_args...)) {
                m_nt_events.push_back(ev);
            }
        }
    }
    template <typename... KwArgs>
    taylor_adaptive_batch(sys_t sys, std::initializer_list<double> state, std::uint32_t batch_size,
                          KwArgs &&...kw_args)
        : taylor_adaptive_batch(std::move(sys), std::vector<double>(state), batch_size, std::forward<KwArgs>(kw_args)...)
    {
    }
    // Construction without an initial state (zero-initialised, reference: taylor.hpp:917-929).
    template <typename... KwArgs>
    taylor_adaptive_batch(sys_t sys, std::uint32_t batch_size, KwArgs &&...kw_args)
        : taylor_adaptive_batch(std::move(sys), std::vector<double>{}, batch_size, std::forward<KwArgs>(kw_args)...)
    {
    }

    [[nodiscard]] const taylor_dc_t &get_decomposition() const
    {
        return m_core.get_decomposition();
    }
    // (Both throw "No events were defined for this integrator" without events: src/taylor_adaptive_batch.cpp:2127-2150.)
    [[nodiscard]] const std::vector<t_event_batch<double>> &get_t_events() const
    {
        (void)m_core.get_t_events();
        return m_t_events;
    }
    [[nodiscard]] const std::vector<nt_event_batch<double>> &get_nt_events() const
    {
        (void)m_core.get_nt_events();
        return m_nt_events;
    }
    [[nodiscard]] std::uint32_t get_batch_size() const
    {
        return m_core.get_batch_size();
    }
    [[nodiscard]] std::uint32_t get_order() const
    {
        return m_core.get_order();
    }
    [[nodiscard]] double get_tol() const
    {
        return m_core.get_tol();
    }
    [[nodiscard]] bool get_high_accuracy() const
    {
        return m_core.get_high_accuracy();
    }
    [[nodiscard]] bool get_compact_mode() const
    {
        return m_core.get_compact_mode();
    }
    // (Not in the reference, which reports these cases through its logger.)
    [[nodiscard]] std::uint64_t get_event_detection_failures() const
    {
        return m_core.get_event_detection_failures();
    }
    [[nodiscard]] std::uint32_t get_dim() const
    {
        return m_core.get_dim();
    }
    [[nodiscard]] std::uint32_t get_n_orig_sv() const noexcept
    {
        return m_core.get_dim();
    }
    [[nodiscard]] const sys_t &get_sys() const noexcept
    {
        return m_core.get_sys();
    }
    [[nodiscard]] bool with_events() const
    {
        return m_core.with_events();
    }
    void reset_cooldowns()
    {
        m_core.reset_cooldowns();
    }
    void reset_cooldowns(std::uint32_t i)
    {
        m_core.reset_cooldowns(i);
    }
    [[nodiscard]] const std::vector<std::vector<std::optional<std::pair<double, double>>>> &get_te_cooldowns() const
    {
        return m_core.get_te_cooldowns();
    }
    [[nodiscard]] bool is_variational() const noexcept
    {
        return false;
    }

    // NOTE: the getters of state and times return references to host mirrors which stay valid AND current across steps,
    // like the members they stand for in the reference (its benchmark/outer_ss_long_term_batch.cpp keeps
    // `const auto &times_v = ta.get_time()` across its stepping loop): from the first call on, the mirrors are refreshed
    // after every kernel (a reference to the times alone costs 16 B per system and kernel, one to the state the whole state).
    // Code which cares about the transfers uses the device views instead.
    [[nodiscard]] const std::vector<double> &get_time() const
    {
        m_core.hold_time_refs();
        return m_core.get_time();
    }
    [[nodiscard]] const double *get_time_data() const
    {
        m_core.hold_time_refs();
        return m_core.get_time().data();
    }
    void set_time(const std::vector<double> &t)
    {
        m_core.set_time(t);
    }
    void set_time(double t)
    {
        m_core.set_time(t);
    }
    [[nodiscard]] std::pair<const std::vector<double> &, const std::vector<double> &> get_dtime() const
    {
        m_core.hold_time_refs();
        return m_core.get_dtime();
    }
    [[nodiscard]] std::pair<const double *, const double *> get_dtime_data() const
    {
        const auto p = m_core.get_dtime();
        return {p.first.data(), p.second.data()};
    }
    void set_dtime(const std::vector<double> &hi, const std::vector<double> &lo)
    {
        m_core.set_dtime(hi, lo);
    }
    void set_dtime(double hi, double lo)
    {
        m_core.set_dtime(hi, lo);
    }

    [[nodiscard]] const std::vector<double> &get_state() const
    {
        m_core.hold_host_refs();
        return m_core.get_state();
    }
    [[nodiscard]] const double *get_state_data() const
    {
        m_core.hold_host_refs();
        return m_core.get_state().data();
    }
    [[nodiscard]] double *get_state_data()
    {
        return m_core.get_state_data();
    }
    [[nodiscard]] const std::vector<double> &get_pars() const
    {
        return m_core.get_pars();
    }
    // NOTE: ranges over the (eagerly synchronised, see get_state_data()) host mirrors, reference taylor.hpp:984-990.
    [[nodiscard]] std::span<double> get_state_range()
    {
        return {m_core.get_state_data(), m_core.get_state().size()};
    }
    [[nodiscard]] std::span<double> get_pars_range()
    {
        return {m_core.get_pars_data(), m_core.get_pars().size()};
    }
    [[nodiscard]] const double *get_pars_data() const
    {
        return m_core.get_pars().data();
    }
    [[nodiscard]] double *get_pars_data()
    {
        return m_core.get_pars_data();
    }
    [[nodiscard]] const std::vector<double> &get_tc() const
    {
        return m_core.get_tc();
    }
    [[nodiscard]] const std::vector<double> &get_last_h() const
    {
        return m_core.get_last_h();
    }
    [[nodiscard]] const std::vector<double> &get_d_output() const
    {
        return m_core.get_d_output();
    }
    const std::vector<double> &update_d_output(const std::vector<double> &t, bool rel_time = false)
    {
        return m_core.update_d_output(t, rel_time);
    }
    const std::vector<double> &update_d_output(double t, bool rel_time = false)
    {
        return m_core.update_d_output(t, rel_time);
    }

    void step(bool wtc = false)
    {
        m_core.set_callback_context(this);
        m_core.step(wtc);
    }
    void step_backward(bool wtc = false)
    {
        m_core.set_callback_context(this);
        m_core.step_backward(wtc);
    }
    void step(const std::vector<double> &max_delta_ts, bool wtc = false)
    {
        m_core.set_callback_context(this);
        m_core.step(max_delta_ts, wtc);
    }
    [[nodiscard]] const std::vector<std::tuple<taylor_outcome, double>> &get_step_res() const
    {
        return m_core.get_step_res();
    }
    [[nodiscard]] const std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> &get_propagate_res() const
    {
        return m_core.get_propagate_res();
    }

    // NOTE: as in the reference (taylor.hpp:1064-1107), the first element of the returned tuple is the
    // continuous output object if kw::c_output = true was passed (and at least one step was taken).
    std::optional<continuous_output_batch<double>> make_c_out()
    {
        auto c = m_core.take_c_output();
        if (c) {
            return continuous_output_batch<double>(std::move(*c));
        }
        return std::nullopt;
    }

public:
    template <typename... KwArgs>
    std::tuple<std::optional<continuous_output_batch<double>>, step_callback_batch<double>> propagate_until(const std::vector<double> &ts,
                                                                           KwArgs &&...kw_args)
    {
        auto [max_steps, mdts, cb, user_cb, wtc, c_out, pre] = propagate_common_ops(std::forward<KwArgs>(kw_args)...);
        m_core.set_callback_context(this);
        m_core.propagate_until(ts, max_steps, mdts, cb, wtc, c_out, pre);
        return {make_c_out(), std::move(*user_cb)};
    }
    template <typename... KwArgs>
    std::tuple<std::optional<continuous_output_batch<double>>, step_callback_batch<double>> propagate_until(double t, KwArgs &&...kw_args)
    {
        auto [max_steps, mdts, cb, user_cb, wtc, c_out, pre] = propagate_common_ops(std::forward<KwArgs>(kw_args)...);
        m_core.set_callback_context(this);
        m_core.propagate_until(std::vector<double>{t}, max_steps, mdts, cb, wtc, c_out, pre);
        return {make_c_out(), std::move(*user_cb)};
    }
    template <typename... KwArgs>
    std::tuple<std::optional<continuous_output_batch<double>>, step_callback_batch<double>> propagate_for(const std::vector<double> &dts,
                                                                         KwArgs &&...kw_args)
    {
        auto [max_steps, mdts, cb, user_cb, wtc, c_out, pre] = propagate_common_ops(std::forward<KwArgs>(kw_args)...);
        m_core.set_callback_context(this);
        m_core.propagate_for(dts, max_steps, mdts, cb, wtc, c_out, pre);
        return {make_c_out(), std::move(*user_cb)};
    }
    template <typename... KwArgs>
    std::tuple<std::optional<continuous_output_batch<double>>, step_callback_batch<double>> propagate_for(double dt, KwArgs &&...kw_args)
    {
        auto [max_steps, mdts, cb, user_cb, wtc, c_out, pre] = propagate_common_ops(std::forward<KwArgs>(kw_args)...);
        m_core.set_callback_context(this);
        m_core.propagate_for(std::vector<double>{dt}, max_steps, mdts, cb, wtc, c_out, pre);
        return {make_c_out(), std::move(*user_cb)};
    }
    template <typename... KwArgs>
    std::tuple<step_callback_batch<double>, std::vector<double>> propagate_grid(std::vector<double> grid,
                                                                               KwArgs &&...kw_args)
    {
        auto [max_steps, mdts, cb, user_cb, wtc, c_out, pre] = propagate_common_ops(std::forward<KwArgs>(kw_args)...);
        m_core.set_callback_context(this);
        auto ret = m_core.propagate_grid(std::move(grid), max_steps, mdts, cb, nullptr, pre);
        return {std::move(*user_cb), std::move(ret)};
    }

    // MI355X extensions.
    [[nodiscard]] detail::tab_core &core()
    {
        return m_core;
    }
    [[nodiscard]] const detail::tab_core &core() const
    {
        return m_core;
    }
};

// Class template argument deduction from the type of the initial state (include/heyoka/taylor.hpp:1124-1150,
// test/taylor_adaptive_batch.cpp:2048-2066): taylor_adaptive_batch({prime(x) = v, ...}, std::vector{0., 1.}, 1u).
template <typename... KwArgs>
taylor_adaptive_batch(std::vector<std::pair<expression, expression>>, std::vector<double>, std::uint32_t, KwArgs &&...)
    -> taylor_adaptive_batch<double>;
template <typename... KwArgs>
taylor_adaptive_batch(std::vector<std::pair<expression, expression>>, std::initializer_list<double>, std::uint32_t,
                      KwArgs &&...) -> taylor_adaptive_batch<double>;

// Human-readable summary (reference: taylor_adaptive_batch_stream_impl(), src/taylor_stream_ops.cpp:110-170): the same
// fields and labels, plus the code generator that produced the device kernels.
template <typename T>
inline std::ostream &operator<<(std::ostream &os, const taylor_adaptive_batch<T> &ta)
{
    std::ostringstream oss;
    oss.imbue(std::locale::classic());
    oss << std::boolalpha;
    oss.precision(std::numeric_limits<T>::max_digits10);
    const auto list = [&oss](const char *label, const std::vector<T> &v) {
        oss << label << '[';
        for (std::size_t i = 0; i < v.size(); ++i) {
            oss << v[i] << (i + 1u == v.size() ? "" : ", ");
        }
        oss << "]\n";
    };
    oss << "C++ datatype            : double\n";
    oss << "Tolerance               : " << ta.get_tol() << '\n';
    oss << "High accuracy           : " << ta.get_high_accuracy() << '\n';
    oss << "Compact mode            : " << ta.get_compact_mode() << '\n';
    oss << "Taylor order            : " << ta.get_order() << '\n';
    oss << "Dimension               : " << ta.get_dim() << '\n';
    oss << "Batch size              : " << ta.get_batch_size() << '\n';
    list("Time                    : ", ta.get_time());
    list("State                   : ", ta.get_state());
    if (!ta.get_pars().empty()) {
        list("Parameters              : ", ta.get_pars());
    }
    if (ta.with_events()) {
        if (!ta.get_t_events().empty()) {
            oss << "N of terminal events    : " << ta.get_t_events().size() << '\n';
        }
        if (!ta.get_nt_events().empty()) {
            oss << "N of non-terminal events: " << ta.get_nt_events().size() << '\n';
        }
    }
    oss << "Code generator (gfx950) : " << ta.core().get_codegen_info() << '\n';
    return os << oss.str();
}

} // namespace heyoka_amd
