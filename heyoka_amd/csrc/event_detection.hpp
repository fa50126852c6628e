// Event detection for the batch integrator.
//
// Reference: include/heyoka/events.hpp (t_event_batch / nt_event_batch), src/detail/event_detection.cpp
// (ed_data_batch<T>::detect_events(), :1733-2173), src/taylor_adaptive_batch.cpp:727-1030 (the event branch of
// step_impl()). The reference detects events on the host, one batch lane after the other, with JIT-compiled
// helpers for the polynomial translations. Here the detection runs on the device, one lane per system:
// fast exclusion check (interval Horner enclosure), Collins-Akritas root isolation with Descartes' rule on
// the translated polynomials, bracketed root finding, direction filter and terminal-event cooldowns; the
// (few) detected events are then copied to the host, which runs the reference's sequential logic (ordering,
// step truncation at the first terminal event, callbacks, cooldown bookkeeping, outcomes).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

#include "expression.hpp"

namespace heyoka_amd
{

enum class event_direction : int { negative = -1, any = 0, positive = 1 };

// (src/nt_event.cpp:42-54.)
inline std::ostream &operator<<(std::ostream &os, event_direction dir)
{
    switch (dir) {
        case event_direction::any:
            return os << "event_direction::any";
        case event_direction::positive:
            return os << "event_direction::positive";
        case event_direction::negative:
            return os << "event_direction::negative";
    }
    return os << "event_direction::??";
}

namespace detail
{

// Type-erased events as stored in the integrator core. ctx is the address of the user-facing integrator object
// (taylor_adaptive_batch<double> *, or the C handle), supplied at call time.
struct core_nt_event {
    expression eq;
    std::function<void(void *ctx, double time, int d_sgn, std::uint32_t batch_idx)> callback;
    event_direction dir = event_direction::any;
    // (See core_t_event::native_counter: hy_event_counter_nt.)
    std::uint64_t *native_counter = nullptr;
};

struct core_t_event {
    expression eq;
    std::function<bool(void *ctx, int d_sgn, std::uint32_t batch_idx)> callback; // empty -> always stop
    event_direction dir = event_direction::any;
    double cooldown = -1; // < 0: deduced automatically (taylor_deduce_cooldown())
    // The callback is the library's own counting callback (hy_event_counter_t of the C ABI: increments *counter and lets
    // the integration continue): nothing of the caller's runs, so the step applies it on the device.
    std::uint64_t *native_counter = nullptr;
};

// One detected event of a lane: (event index, root (time from the beginning of the step), sign of the time
// derivative of the event equation, absolute value of the derivative).
struct detected_event {
    std::uint32_t idx;
    double root;
    int d_sgn;
    double abs_der;
};

// Capacity of the per-lane list of detected events of one class (terminal / non-terminal) in a step: every event
// equation is a polynomial of degree `order` over the step, so (order + 1) entries per event of the class hold
// whatever a successful root isolation can produce (the reference's lists are unbounded vectors).
std::uint32_t ed_max_detected(std::uint32_t order, std::uint32_t n_te, std::uint32_t n_nte);

// Limit on the working list of the root isolation: the reference's (src/detail/event_detection.cpp:1399, :2082).
inline constexpr std::uint32_t ed_work_list_cap = 250;
// Bytes of working list per resident thread ("slot") of hy_detect_events.
std::size_t ed_work_list_bytes_per_slot(std::uint32_t order);

// HIP source of the detection kernel for a given Taylor order and list capacity.
std::string make_event_detection_source(std::uint32_t order, std::uint32_t max_detected);

struct ed_kargs {
    const double *ev_tc;      // [(event * (order + 1) + k) * N + lane], terminal events first
    const double *h;          // [N] step taken
    const double *g_eps;      // [N]
    const int *dirs;          // [n_te + n_nte]
    const double *cd_first;   // [n_te * N] terminal-event cooldowns (valid if cd_active != 0)
    const double *cd_second;  // [n_te * N]
    const int *cd_active;     // [n_te * N]
    double *out;              // [2][N][ed_max_detected()][4]: idx, root, d_sgn, abs_der
    unsigned *counts;         // [2][N]
    unsigned *flags;          // [0] root isolations which failed, [1] event lists which overflowed, [2] root findings which
                              // failed (event, lane) - the reference logs a warning and ignores the event
    unsigned long long N;
    unsigned n_te, n_nte;
    // Device-side error bound of the Taylor series of the event equations (src/taylor_adaptive_batch.cpp:744-767):
    // when mas != nullptr, g_eps is computed from max |x_i| and the tolerance and written to g_eps_out.
    const double *mas;
    double *g_eps_out;
    double tol;
    double *wl;               // working lists of the root isolation, ed_work_list_bytes_per_slot() per launched thread
    unsigned long long wl_slots; // number of launched threads (grid-stride loop over the lanes)
    // Optional [N]: 0.0 for the lanes in which the stepper has already ruled out events in this step (its conservative
    // exclusion test on the jets it computed, emitted_module::events_in_stepper): they are skipped without reading their
    // event jets.
    const double *maybe;
};

// Per-lane bookkeeping of a step with events on the device (src/taylor_adaptive_batch.cpp:771-1030): hy_ev_pre
// (truncation of the step at the first terminal event, size of the record buffer), the dense-output kernel, hy_ev_post
// (time update, non-finite check, cooldown update, outcomes, compaction of the lanes with events into records),
// hy_ev_scatter (cooldowns / outcomes set by the host for the lanes whose terminal events triggered).
// Record of a lane with events: 8 doubles (lane, n_te, n_nte, g_eps, h, new time hi, new time lo, 0) followed by
// 4 doubles per event (idx, root, d_sgn, |derivative|), terminal events first, in detection order.
struct ep_kargs {
    const double *h;          // [N] step sizes of the stepper
    const double *ed_out;
    const unsigned *counts;
    double *dout_h;           // [N] final step sizes
    const double *g_eps;      // [N]
    const double *state;      // [dim * N], after the update
    double *time_hi, *time_lo;
    const double *lim;
    double *cd_first, *cd_second;
    int *cd_active;
    long long *outcome;
    double *last_h;
    double *rec;
    unsigned long long *cursor; // [0] record doubles needed (hy_ev_pre), [1] record doubles written (hy_ev_post)
    const double *upd;          // hy_ev_scatter: n_cd x (position, first, second) then n_oc x (lane, outcome)
    unsigned long long N;
    unsigned n_te, n_nte, dim, n_cd, n_oc, pad;
    // Every callback is the library's counting callback (core_*_event::native_counter): hy_ev_post applies the events of
    // a lane itself - counts per event in ev_counts[n_te + n_nte] (terminal events first), cooldown and outcome of the
    // first terminal event, cooldown settings in te_cd[n_te] (< 0: deduced) - and writes no records.
    int native;
    unsigned long long *ev_counts;
    const double *te_cd;
};

} // namespace detail

} // namespace heyoka_amd
