// Host driver of the MI355X batch Taylor integrator. See taylor_adaptive_batch.hpp.
#include "logging.hpp"
#include "taylor_adaptive_batch.hpp"

#include <algorithm>
#include <array>
#include <cassert>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>

#include "dfloat.hpp"
#include "hip_backend.hpp"
#include "hip_emit.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd::detail
{

namespace
{

// Shortest representation which round-trips, like the reference's fmt::format("{}", x) (fp_to_string(),
// src/detail/string_conv.cpp:64-80).
std::string fp_to_string(double x)
{
    char buf[64];
    const auto res = std::to_chars(buf, buf + sizeof(buf), x);
    return std::string(buf, res.ptr);
}

// Argument block of the post-step kernels (hy_grid_post / hy_until_post, see make_grid_source()).
struct grid_kargs {
    const double *grid;
    double *out;
    const double *tc;
    const double *thi;
    const double *tlo;
    const double *last_h;
    const long long *outcome;
    double *rem_hi;
    double *rem_lo;
    const double *mdt;
    const int *t_dir;
    double *lim;
    unsigned *gidx;
    double *min_h;
    double *max_h;
    unsigned long long *n_steps;
    unsigned *counters;
    unsigned long long N;
    unsigned n_grid;
    // (Next grid time of every lane, +-inf once the lane is through its grid: hy_kargs::pad bit 2 of the next sweep.)
    double *next_tg;
    // Launches which take every lane from one grid point to the next (emitted_module::grid_multi_step): the stepper leaves
    // the counters / extrema of ITS launch in n_steps / min_h / max_h, which are accumulated here (null: single-step sweeps).
    unsigned long long *acc_n_steps;
    double *acc_min_h;
    double *acc_max_h;
    // (... and whether the lane's last step was clamped to its remaining time: hy_kargs::grid_done.)
    const double *grid_done;
};

// Code generator from the configuration field (0 automatic: the wave-cluster generator is tried first and falls back
// by itself, see emit_hip_module()).
emit_mode choose_mode(int emitter)
{
    switch (emitter) {
        case 1:
            return emit_mode::unrolled;
        case 3:
            return emit_mode::table;
        case 4:
            return emit_mode::block;
        default:
            return emit_mode::cluster;
    }
}

} // namespace

struct tab_core::impl {
    sys_t sys;
    taylor_dc_t dc;
    taylor_program prog;
    std::uint32_t order = 0;
    double tol = 0;
    bool high_accuracy = false;
    bool compact_mode = false;
    // MI355X extensions of the configuration (tab_core::config): code generator, cluster generator, exact divisions,
    // steppers used with events, outcome semantics of propagate_for / propagate_until.
    int emitter = 0, cluster_kernel = 0, events_on_cluster = 0, batch_semantics = 0, sum_order = 0;
    bool exact_division = false;
    std::uint32_t N = 0; // batch size == number of systems.
    std::uint32_t dim = 0;
    int device = 0;

    emitted_module emitted;
    std::shared_ptr<const compiled_module> cmod;

    // Host mirrors (mutable: refreshed lazily from const getters).
    mutable std::vector<double> state, pars, time_hi, time_lo, tc, last_h, d_out;
    mutable std::vector<std::tuple<taylor_outcome, double>> step_res;
    mutable std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> prop_res;

    // Device side (created lazily at the first operation needing the GPU).
    mutable std::unique_ptr<device_module> dmod;
    mutable device_buffer d_state, d_pars, d_thi, d_tlo, d_lim, d_tfhi, d_tflo, d_lasth, d_outcome, d_minh, d_maxh,
        d_nsteps, d_tc, d_counters, d_dout, d_douth;
    mutable void *stream = nullptr;

    // Synchronisation state.
    mutable bool host_newer = true;      // state/pars/time on the host must be uploaded.
    mutable bool dev_newer = false;      // state/time on the device must be downloaded.
    mutable bool tc_dev_newer = false;   // tc on the device is newer than the host mirror.
    mutable bool lasth_dev_newer = false;
    mutable bool step_res_dev_newer = false;
    mutable bool prop_res_dev_newer = false;
    bool sticky_host_ptr = false; // a mutable host pointer was handed out: sync eagerly.
    // The C++ interface handed out a reference / pointer to the host mirror of the state or of the times (the
    // reference's getters return references to members which every step updates in place, and its own benchmark keeps
    // one across steps: benchmark/outer_ss_long_term_batch.cpp, `const auto &times_v = ta.get_time()`): the mirrors are
    // refreshed after every kernel from then on.
    mutable bool sticky_const_refs = false;
    mutable bool sticky_time_refs = false; // (a reference to the times only: the state stays on the device)
    // Stepper with events on the wave-cluster kernels: the Taylor coefficients of order >= 1 of the state variables defined
    // by another state variable are not written by the stepper (emitted_module::compact_tc); hy_tc_expand fills them in
    // before anybody reads the full array.
    mutable bool tc_expand_pending = false;
    void ensure_tc_expanded() const;
    // Stepper with events which evaluates the event equations itself (emitted_module::events_in_stepper): the Taylor
    // coefficients of a step are stored only for the workgroups in which an event may have happened; the state and time
    // before the step are kept, and whoever reads coefficients which were not stored (get_tc(), update_d_output(),
    // continuous output, propagate_grid()) triggers a second launch of the stepper on the snapshot which stores nothing
    // but them (bit-identical: the same kernel on the same input). ev_all_tc: store them in every step (lock-step loops
    // which consume them step by step).
    device_buffer evs_state, evs_thi, evs_tlo, evs_pars;
    // Accounting of the steps with events (tab_core::set_event_timing() / get_event_stats(); bench.py's events leg): number
    // of steps, wall-clock ms of the five phases (only with the timing switched on: a stream synchronisation after each
    // phase), regeneration launches of the Taylor coefficients, systems which reported events.
    bool ev_timing = false;
    double ev_ms[5] = {0, 0, 0, 0, 0};
    std::uint64_t ev_steps = 0, ev_systems = 0;
    mutable std::uint64_t tc_regens = 0;
    mutable bool tc_partial = false;
    // propagate_grid() with Taylor coefficients on demand (emitted_module::tc_by_threshold) which was interrupted by a
    // non-finite state: d_tc mixes the coefficients of different steps. Cleared by the next step which stores them.
    mutable bool tc_stale = false;
    void check_tc_not_stale() const
    {
        if (tc_stale) {
            throw std::runtime_error("The Taylor coefficients of the last step are not available: the last propagate_grid() "
                                     "stored them on demand and was interrupted by a non-finite state");
        }
    }
    bool ev_all_tc = false;
    // (A caller who read the coefficients of the previous step - a step callback with dense output, say - will probably read
    // those of the next one: that step stores them all instead of paying for a second launch again.)
    mutable bool tc_regenerated = false;
    void ensure_tc_complete() const;
    std::uint64_t last_total_steps = 0;
    // Set by the lock-step propagate loop to override the device outcomes.
    mutable std::optional<taylor_outcome> prop_res_override;
    // Reference outcome semantics on the device-resident propagation (config::batch_semantics == 0): snapshot of the
    // state / times taken before the launch (a batch in which a lane goes non-finite is rolled back and re-run through
    // the lock-step loop: src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467) and the flag which makes a step-limited
    // batch report step_limit in every lane (:1516) when its results are fetched.
    mutable device_buffer snap_state, snap_thi, snap_tlo;
    mutable bool fix_step_limit = false;
    bool force_lockstep = false;
    void snapshot_for_rollback()
    {
        const auto sb = d_state.bytes(), tb = d_thi.bytes();
        if (snap_state.bytes() != sb) {
            snap_state = device_buffer(sb, device);
            snap_thi = device_buffer(tb, device);
            snap_tlo = device_buffer(tb, device);
        }
        device_copy(snap_state.get(), d_state.get(), sb, device, stream);
        device_copy(snap_thi.get(), d_thi.get(), tb, device, stream);
        device_copy(snap_tlo.get(), d_tlo.get(), tb, device, stream);
    }
    void rollback_to_snapshot()
    {
        device_copy(d_state.get(), snap_state.get(), d_state.bytes(), device, stream);
        device_copy(d_thi.get(), snap_thi.get(), d_thi.bytes(), device, stream);
        device_copy(d_tlo.get(), snap_tlo.get(), d_tlo.bytes(), device, stream);
        dev_newer = true;
        times_fresh = false;
        host_newer = false;
        // With a host pointer handed out (get_state_data(), hy_tab_set_state(), the Python state setter) the host mirrors
        // are refreshed after every launch and re-uploaded before the next one: they hold the state at the END of the
        // rolled-back propagation, which would overwrite the restored snapshot in the re-run. Bring them back as well.
        if (sticky_host_ptr || sticky_const_refs) {
            to_host();
        } else if (sticky_time_refs) {
            times_to_host();
        }
    }
    // Continuous output produced by the last propagate_for/until() with c_output = true.
    std::optional<c_out_core> last_c_out;
    // Post-step kernel of the device-resident propagate_grid() loop (created on first use).
    mutable std::unique_ptr<aux_module> grid_mod;

    // ---- event detection (see event_detection.hpp) ----
    std::vector<core_t_event> tes;
    std::vector<core_nt_event> ntes;
    // te_cooldowns[lane][event]: (time elapsed since the trigger, cooldown duration).
    mutable std::vector<std::vector<std::optional<std::pair<double, double>>>> te_cooldowns;
    void *cb_ctx = nullptr;
    mutable std::unique_ptr<aux_module> ed_mod;
    mutable device_buffer d_ev_tc, d_mas, d_geps, d_dirs, d_cd_first, d_cd_second, d_cd_active, d_ed_out, d_ed_counts,
        d_ed_flags, d_ed_wl;
    std::uint64_t ed_slots = 0;
    std::uint64_t ed_failures = 0;
    // Events on the wave-cluster steppers: the main stepper is built from the system alone and runs in mode 4 (jets of
    // the state variables, no update); hy_ev_jets (emit_event_jets()) derives the jets of the event equations and the
    // final step size from them.
    bool cluster_events = false;
    emitted_module ev_emitted;
    std::shared_ptr<const compiled_module> ev_cmod;
    mutable std::unique_ptr<aux_module> evj_mod;
    mutable device_buffer d_selnorms;
    // Set by propagate_for() only: propagate_until() then accepts 2 * N double-length (hi, lo) final times.
    bool dl_times_ok = false;
    // Incremented by set_time() / set_dtime(): lets the device-driven loops detect callbacks that touch the time
    // coordinate without moving the times to the host after every sweep.
    std::uint64_t time_gen = 0;

    [[nodiscard]] bool has_events() const
    {
        return !tes.empty() || !ntes.empty();
    }
    void step_with_events(const std::vector<double> &lims, bool wtc);
    // (lims == nullptr: the step limits are already in d_lim - device-driven loops.)
    void step_with_events_device(const std::vector<double> *lims);
    void ensure_event_buffers();
    void launch_event_stepper(const std::vector<double> *lims);
    unsigned launch_event_detection(bool device_g_eps);
    // Terminal-event cooldowns: the device arrays (d_cd_*) are authoritative between steps with events (updated by
    // hy_ev_post / hy_ev_scatter); te_cooldowns is the lazily synchronised host mirror.
    mutable bool cd_dev_newer = false;
    bool cd_host_newer = true;
    // Cooldowns set by the terminal events of the step being processed (position, first, second), not yet on the device:
    // a callback which reads or resets the cooldowns sees them (the reference sets the cooldown before it invokes the
    // callback, src/taylor_adaptive_batch.cpp:875-890).
    mutable std::vector<double> pending_cd;
    void cooldowns_to_host() const;
    void cooldowns_to_device();
    mutable device_buffer d_ev_cursor, d_ev_rec, d_ev_upd, d_ev_counts, d_te_cd;
    // Every event callback is the library's counting callback: hy_ev_post applies the events itself (ep_kargs::native).
    mutable bool ev_native = false;
    // (Page-locked landing area of the event records of a step: see pinned_buffer.)
    mutable pinned_buffer h_ev_rec;
    [[nodiscard]] bool is_cluster() const
    {
        // NOTE: true whenever the stepper does not need the tc buffer as its jet scratch (cluster / table
        // kernels, unrolled kernels with register-resident jets): tc is then written only on request.
        return emitted.tc_optional;
    }

    void ensure_tc() const
    {
        if (d_tc.bytes() == 0u) {
            d_tc = device_buffer(static_cast<std::size_t>(dim) * (order + 1u) * N * sizeof(double), device);
            // The Taylor coefficients read as zeros until a step writes them (the reference value-initialises m_tc:
            // test/taylor_adaptive_batch.cpp:741-746 checks it from a step callback).
            d_tc.zero(stream);
        }
    }

    void ensure_device() const
    {
        if (dmod) {
            return;
        }
        dmod = std::make_unique<device_module>(cmod, device);
        dmod->set_stream(stream);
        const auto n = static_cast<std::size_t>(N);
        const auto dsz = sizeof(double);
        d_state = device_buffer(state.size() * dsz, device);
        d_pars = device_buffer(pars.size() * dsz, device);
        d_thi = device_buffer(n * dsz, device);
        d_tlo = device_buffer(n * dsz, device);
        d_lim = device_buffer(n * dsz, device);
        d_tfhi = device_buffer(n * dsz, device);
        d_tflo = device_buffer(n * dsz, device);
        d_lasth = device_buffer(n * dsz, device);
        d_outcome = device_buffer(n * sizeof(long long), device);
        d_minh = device_buffer(n * dsz, device);
        d_maxh = device_buffer(n * dsz, device);
        d_nsteps = device_buffer(n * sizeof(unsigned long long), device);
        if (!is_cluster()) {
            // Unrolled mode: the tc buffer doubles as the jet scratch of the kernel.
            ensure_tc();
        }
        d_counters = device_buffer(16u * sizeof(unsigned), device);
        host_newer = true;
    }

    void to_device() const
    {
        ensure_device();
        if (host_newer) {
            d_state.upload(state.data(), state.size() * sizeof(double), stream);
            d_pars.upload(pars.data(), pars.size() * sizeof(double), stream);
            d_thi.upload(time_hi.data(), time_hi.size() * sizeof(double), stream);
            d_tlo.upload(time_lo.data(), time_lo.size() * sizeof(double), stream);
            host_newer = false;
        }
    }

    void to_host() const
    {
        if (dev_newer) {
            d_state.download(state.data(), state.size() * sizeof(double), stream);
            d_thi.download(time_hi.data(), time_hi.size() * sizeof(double), stream);
            d_tlo.download(time_lo.data(), time_lo.size() * sizeof(double), stream);
            dev_newer = false;
        }
    }

    // (The times alone: 16 B per system where the state is 8 * dim. dev_newer stays set - the state is still pending -
    // and times_fresh remembers that the mirror of the times is current until the next kernel.)
    mutable bool times_fresh = false;
    void times_to_host() const
    {
        if (dev_newer && !times_fresh) {
            d_thi.download(time_hi.data(), time_hi.size() * sizeof(double), stream);
            d_tlo.download(time_lo.data(), time_lo.size() * sizeof(double), stream);
            times_fresh = true;
        }
    }

    // tc_written: the launch was asked to write the Taylor coefficients. The reference's get_tc() holds the coefficients
    // of the last step taken with write_tc (zeros before the first one, src/taylor_adaptive_batch.cpp:756-760): steppers
    // which keep their jets in the tc buffer anyway do not count.
    void after_kernel(bool tc_written = true)
    {
        dev_newer = true;
        times_fresh = false;
        if (tc_written) {
            tc_dev_newer = true;
            // (A launch which stored the coefficients of every lane ends the "mixed steps" state of tc_stale.)
            if (tc_threshold == nullptr) {
                tc_stale = false;
            }
        }
        lasth_dev_newer = true;
        if (sticky_host_ptr || sticky_const_refs) {
            to_host();
        } else if (sticky_time_refs) {
            times_to_host();
        }
    }

    // get_tc() holds the coefficients of the last step taken with write_tc (src/taylor_adaptive_batch.cpp:756-760). The
    // steppers which are not wave-cluster kernels use the tc buffer as their jet scratch on EVERY step: before a launch
    // without write_tc overwrites it, a pending (lazily downloaded) set of coefficients is brought to the host mirror.
    void keep_written_tc(bool wtc)
    {
        if (wtc || !tc_dev_newer || is_cluster() || !dmod || d_tc.bytes() == 0u) {
            return;
        }
        const auto sz = static_cast<std::size_t>(dim) * (order + 1u) * N;
        if (tc.size() != sz) {
            tc.assign(sz, 0.);
        }
        d_tc.download(tc.data(), sz * sizeof(double), stream);
        tc_dev_newer = false;
    }

    void before_kernel()
    {
        if (sticky_host_ptr) {
            // The user may have written through a previously-obtained pointer.
            host_newer = true;
        }
        to_device();
    }

    hy_kargs base_args() const
    {
        hy_kargs a{};
        a.state = d_state.as<double>();
        a.pars = d_pars.as<double>();
        a.time_hi = d_thi.as<double>();
        a.time_lo = d_tlo.as<double>();
        a.lim = d_lim.as<double>();
        a.tfin_hi = d_tfhi.as<double>();
        a.tfin_lo = d_tflo.as<double>();
        a.last_h = d_lasth.as<double>();
        a.outcome = d_outcome.as<long long>();
        a.min_h = d_minh.as<double>();
        a.max_h = d_maxh.as<double>();
        a.n_steps = d_nsteps.as<unsigned long long>();
        a.tc = is_cluster() ? nullptr : d_tc.as<double>();
        a.N = N;
        a.max_steps = 0;
        a.mode = 0;
        a.counters = d_counters.as<unsigned>();
        return a;
    }

    // step() / step_backward(): the limits are +-infinity for every lane - kept in two vectors built once, uploaded only
    // when d_lim does not hold them already (8 MB per call for 1 048 576 systems otherwise).
    std::vector<double> lims_pinf, lims_ninf;
    const double *d_lim_src = nullptr;
    const std::vector<double> &inf_lims(bool forward)
    {
        auto &v = forward ? lims_pinf : lims_ninf;
        if (v.size() != N) {
            v.assign(N, forward ? std::numeric_limits<double>::infinity() : -std::numeric_limits<double>::infinity());
        }
        return v;
    }
    void upload_lims(const std::vector<double> &lims)
    {
        const bool cached = lims.data() == lims_pinf.data() || lims.data() == lims_ninf.data();
        if (cached && lims.data() == d_lim_src) {
            return;
        }
        d_lim.upload(lims.data(), lims.size() * sizeof(double), stream);
        d_lim_src = cached ? lims.data() : nullptr;
    }

    // One lock-step sweep: a single step for every lane with the per-lane signed limits 'lims'.
    // NOTE: lims == nullptr -> the step limits are already in d_lim (device-driven loops).
    void run_step(const std::vector<double> &lims, bool wtc)
    {
        run_step_impl(&lims, wtc);
    }
    // (Set by the lock-step loop of propagate_grid(): per-lane times below which a step does not store its Taylor
    // coefficients - emitted_module::tc_by_threshold.)
    const double *tc_threshold = nullptr;
    void run_step_impl(const std::vector<double> *lims, bool wtc)
    {
        before_kernel();
        if (lims != nullptr) {
            upload_lims(*lims);
        } else {
            d_lim_src = nullptr;
        }
        d_counters.zero(stream);
        keep_written_tc(wtc);
        auto a = base_args();
        if (wtc && is_cluster()) {
            ensure_tc();
            a.tc = d_tc.as<double>();
        }
        a.mode = 0;
        if (tc_threshold != nullptr && wtc) {
            a.tfin_hi = tc_threshold;
            a.pad = 4;
        }
        dmod->launch_taylor(a);
        after_kernel(wtc);
        step_res_dev_newer = true;
    }

    void fetch_step_res() const
    {
        if (!step_res_dev_newer) {
            return;
        }
        std::vector<long long> oc(N);
        std::vector<double> h(N);
        d_outcome.download(oc.data(), oc.size() * sizeof(long long), stream);
        d_lasth.download(h.data(), h.size() * sizeof(double), stream);
        for (std::uint32_t i = 0; i < N; ++i) {
            step_res[i] = std::tuple{static_cast<taylor_outcome>(oc[i]), h[i]};
        }
        last_h = h;
        lasth_dev_newer = false;
        step_res_dev_newer = false;
    }

    void fetch_prop_res() const
    {
        if (!prop_res_dev_newer) {
            return;
        }
        std::vector<long long> oc(N);
        std::vector<double> mn(N), mx(N);
        std::vector<unsigned long long> ns(N);
        d_outcome.download(oc.data(), oc.size() * sizeof(long long), stream);
        d_minh.download(mn.data(), mn.size() * sizeof(double), stream);
        d_maxh.download(mx.data(), mx.size() * sizeof(double), stream);
        d_nsteps.download(ns.data(), ns.size() * sizeof(unsigned long long), stream);
        if (fix_step_limit) {
            // The reference stops the whole batch when the iteration counter reaches max_steps and reports step_limit in
            // EVERY lane (src/taylor_adaptive_batch.cpp:1516): the lanes which were done earlier took zero-length steps
            // in the meantime, so their states, times and counters are what the device-resident loop left.
            fix_step_limit = false;
            const auto sl = static_cast<long long>(taylor_outcome::step_limit);
            if (std::find(oc.begin(), oc.end(), sl) != oc.end()) {
                std::fill(oc.begin(), oc.end(), sl);
                d_outcome.upload(oc.data(), oc.size() * sizeof(long long), stream);
            }
        }
        for (std::uint32_t i = 0; i < N; ++i) {
            prop_res[i] = std::tuple{static_cast<taylor_outcome>(oc[i]), mn[i], mx[i], static_cast<std::size_t>(ns[i])};
        }
        prop_res_dev_newer = false;
    }
};

// Reference: finalise_ctor_impl(), src/taylor_adaptive_batch.cpp:78-427.
tab_core::tab_core(sys_t sys, std::vector<double> state, std::uint32_t batch_size, config cfg)
    : m_impl(std::make_unique<impl>())
{
    auto &d = *m_impl;

    validate_ode_sys(sys);

    d.N = batch_size;
    d.high_accuracy = cfg.high_accuracy;
    d.compact_mode = cfg.compact_mode;
    d.device = cfg.device;
    d.emitter = cfg.emitter;
    d.cluster_kernel = cfg.cluster_kernel;
    d.exact_division = cfg.exact_division;
    d.sum_order = cfg.sum_order;
    if (d.sum_order < 0 || d.sum_order > 2) {
        throw std::invalid_argument("Invalid order of summation selected in an adaptive Taylor integrator in batch mode: "
                                    + std::to_string(d.sum_order) + " (0 automatic, 1 pairwise, 2 running sums)");
    }
    d.events_on_cluster = cfg.events_on_cluster;
    d.batch_semantics = cfg.batch_semantics;
    if (d.emitter < 0 || d.emitter > 4) {
        throw std::invalid_argument("Invalid code generator selected in an adaptive Taylor integrator in batch mode: "
                                    + std::to_string(d.emitter) + " (0 automatic, 1 unrolled, 2 cluster, 3 table, 4 block)");
    }
    if (d.cluster_kernel != 0 && d.cluster_kernel != 5 && d.cluster_kernel != 3 && d.cluster_kernel != 2
        && d.cluster_kernel != 1) {
        throw std::invalid_argument("Invalid wave-cluster generator selected in an adaptive Taylor integrator in batch mode: "
                                    + std::to_string(d.cluster_kernel) + " (0 automatic, or 5, 3, 2, 1)");
    }
    if (d.batch_semantics < 0 || d.batch_semantics > 2) {
        throw std::invalid_argument("Invalid batch semantics selected in an adaptive Taylor integrator in batch mode: "
                                    + std::to_string(d.batch_semantics) + " (0 reference, 1 lock-step loop, 2 per lane)");
    }
    // Developer overrides from the environment (experiments and the test matrix): they map onto the same fields.
    if (const char *m = std::getenv("HEYOKA_AMD_EMIT_MODE")) {
        const std::string ms(m);
        d.emitter = ms == "unrolled" ? 1 : (ms == "cluster" ? 2 : (ms == "table" ? 3 : (ms == "block" ? 4 : d.emitter)));
    }
    if (const char *ev = std::getenv("HEYOKA_AMD_ONE_LANE"); ev != nullptr && std::atoi(ev) == 0 && d.cluster_kernel == 0) {
        d.cluster_kernel = 3;
    }
    if (const char *ev = std::getenv("HEYOKA_AMD_PAIR_SPLIT"); ev != nullptr && std::atoi(ev) == 0
        && (d.cluster_kernel == 0 || d.cluster_kernel == 3)) {
        d.cluster_kernel = 2;
    }
    if (const char *ev = std::getenv("HEYOKA_AMD_EVENTS_ON_CLUSTER"); ev != nullptr && std::atoi(ev) == 0) {
        d.events_on_cluster = 1;
    }

    if (d.N == 0u) {
        throw std::invalid_argument("The batch size in an adaptive Taylor integrator cannot be zero");
    }

    if (state.size() % d.N != 0u) {
        throw std::invalid_argument("Invalid size detected in the initialization of an adaptive Taylor "
                                    "integrator: the state vector has a size of "
                                    + std::to_string(state.size()) + ", which is not a multiple of the batch size ("
                                    + std::to_string(d.N) + ")");
    }

    if (state.empty()) {
        state.resize(sys.size() * static_cast<std::size_t>(d.N));
    }

    if (state.size() / d.N != sys.size()) {
        throw std::invalid_argument("Inconsistent sizes detected in the initialization of an adaptive Taylor "
                                    "integrator: the state vector has a dimension of "
                                    + std::to_string(state.size() / d.N) + " and a batch size of "
                                    + std::to_string(d.N) + ", while the number of equations is "
                                    + std::to_string(sys.size()));
    }

    // Time.
    if (cfg.time.empty()) {
        d.time_hi.assign(d.N, 0.);
    } else if (cfg.time_is_scalar) {
        d.time_hi.assign(d.N, cfg.time[0]);
    } else {
        d.time_hi = std::move(cfg.time);
    }
    if (d.time_hi.size() != d.N) {
        throw std::invalid_argument("Invalid size detected in the initialization of an adaptive Taylor "
                                    "integrator: the time vector has a size of "
                                    + std::to_string(d.time_hi.size()) + ", which is not equal to the batch size ("
                                    + std::to_string(d.N) + ")");
    }
    d.time_lo.assign(d.N, 0.);

    if (cfg.tol && (!std::isfinite(*cfg.tol) || *cfg.tol < 0)) {
        throw std::invalid_argument("The tolerance in an adaptive Taylor integrator must be finite and positive, "
                                    "but it is "
                                    + fp_to_string(*cfg.tol) + " instead");
    }

    if (cfg.parallel_mode && !cfg.compact_mode) {
        throw std::invalid_argument("Parallel mode can be activated only in conjunction with compact mode");
    }

    d.tol = cfg.tol ? *cfg.tol : std::numeric_limits<double>::epsilon();
    d.dim = static_cast<std::uint32_t>(sys.size());
    d.state = std::move(state);

    // Decomposition + flattened program (with the event equations as extra functions, terminal events first,
    // reference: src/taylor_adaptive_batch.cpp:280-330).
    const detail::stopwatch sw_dc;
    d.tes = std::move(cfg.t_events);
    d.ntes = std::move(cfg.nt_events);
    if (d.has_events()) {
        for (const auto &ev : d.tes) {
            if (!std::isfinite(ev.cooldown)) {
                throw std::invalid_argument("Cannot set a non-finite cooldown value for a terminal event");
            }
        }
        for (const auto &ev : d.ntes) {
            if (!ev.callback) {
                throw std::invalid_argument("Cannot construct a non-terminal event with an empty callback");
            }
        }
        std::vector<expression> ev_eqs;
        for (const auto &ev : d.tes) {
            ev_eqs.push_back(ev.eq);
        }
        for (const auto &ev : d.ntes) {
            ev_eqs.push_back(ev.eq);
        }
        std::vector<std::uint32_t> ev_u;
        d.dc = taylor_decompose_sys(sys, ev_eqs, ev_u);
        d.prog = make_program(d.dc, d.dim);
        d.prog.ev_u = std::move(ev_u);
        d.te_cooldowns.assign(d.N, std::vector<std::optional<std::pair<double, double>>>(d.tes.size()));
    } else {
        d.dc = taylor_decompose_sys(sys);
        d.prog = make_program(d.dc, d.dim);
    }

    // (Stage log, in the reference's wording: src/taylor_01.cpp:439-440, :968.)
    detail::log_message(log_level::debug, "Taylor decomposition of " + std::to_string(d.dim) + " equations: " + std::to_string(d.dc.size())
                                              + " entries (" + std::to_string(d.prog.n_u) + " u variables, "
                                              + std::to_string(d.prog.nodes.size()) + " elementary functions)");
    detail::log_message(log_level::trace, "Taylor decomposition construction runtime: " + sw_dc.str());

    // Parameters.
    const auto tot_n_pars = d.prog.n_par;
    const auto pars_req = static_cast<std::size_t>(tot_n_pars) * d.N;
    if (cfg.pars.empty()) {
        cfg.pars.resize(pars_req);
    } else if (cfg.pars.size() != pars_req) {
        throw std::invalid_argument("Invalid number of parameter values passed to the constructor of an adaptive "
                                    "Taylor integrator in batch mode: "
                                    + std::to_string(cfg.pars.size())
                                    + " parameter value(s) were passed, but the ODE system contains "
                                    + std::to_string(tot_n_pars) + " parameter(s) (in batches of "
                                    + std::to_string(d.N) + ")");
    }
    d.pars = std::move(cfg.pars);

    d.order = taylor_order_from_tol(d.tol);

    // Index-range checks (the generated code indexes with 64-bit integers, but the public
    // interface shares the 32-bit batch size of the reference).
    if (static_cast<std::uint64_t>(d.dim) * (d.order + 1u) > std::numeric_limits<std::uint32_t>::max()) {
        throw std::overflow_error(
            "An overflow condition was detected in the computation of a jet of Taylor derivatives");
    }

    // Code generation + hiprtc compilation (works without a GPU).
    emit_options eo;
    eo.dev = dev_switches::from_env();
    eo.order = d.order;
    eo.high_accuracy = d.high_accuracy;
    eo.batch_size = d.N;
    eo.cluster_kernel = d.cluster_kernel;
    eo.exact_division = d.exact_division;
    eo.sum_order = d.sum_order;
    // NOTE: the stepper with events (mode 4) is implemented by the one-system-per-lane kernels: fully unrolled for
    // small decompositions, table-driven otherwise (HEYOKA_AMD_EMIT_MODE=table forces the latter).
    if (d.has_events()) {
        eo.mode = (d.prog.nodes.size() > 150u || choose_mode(d.emitter) == emit_mode::table) ? emit_mode::table
                                                                                            : emit_mode::unrolled;
        // Wave-cluster stepper for the system itself + the event equations from its jets, when both apply (event
        // equations which depend on a small part of the decomposition: distances, coordinates, angles, ...).
        // HEYOKA_AMD_EVENTS_ON_CLUSTER=0: always the one-system-per-lane steppers with events.
        if (choose_mode(d.emitter) == emit_mode::cluster && d.events_on_cluster == 0) {
            const auto prog0 = make_program(taylor_decompose_sys(sys), d.dim);
            auto eo2 = eo;
            eo2.mode = emit_mode::cluster;
            eo2.event_stepper = true;
            // The stepper may evaluate the event equations itself, take the final step size and update the state
            // (emit_options::ev_prog); lanes whose step is truncated at a terminal event are redone from the Taylor
            // coefficients afterwards.
            eo2.ev_prog = &d.prog;
            eo2.n_t_events = static_cast<std::uint32_t>(d.tes.size());
            auto m = emit_hip_module(prog0, eo2);
            std::string why;
            if (m.cluster_mode4) {
                auto eo_ev = eo;
                eo_ev.compact_tc = m.compact_tc;
                // (Event equations inside the stepper: the companion module only serves the compact Taylor coefficients.)
                eo_ev.ev_helpers_only = m.events_in_stepper && m.compact_tc;
                auto evm = emit_event_jets(d.prog, eo_ev, why);
                if (!evm.source.empty()) {
                    d.cluster_events = true;
                    d.emitted = std::move(m);
                    d.emitted.notes += "; events: " + evm.notes;
                    d.ev_emitted = std::move(evm);
                    d.ev_cmod = hiprtc_compile(d.ev_emitted);
                }
            }
        }
    } else {
        eo.mode = choose_mode(d.emitter);
        // kw::compact_mode = true. In the reference it is a code-size / compile-time knob which also changes the order of
        // the additions inside the convolutions (running sums, src/math/prod.cpp:686-698, instead of products + pairwise
        // sum, :386-395) and costs little at run time. Here: the on-chip kernels stay whenever the planner can shape the
        // decomposition (their convolutions are FMA chains - running sums - already; code size does not grow with the
        // number of bodies); the straight-line generator adds in the compact order (sum_order = 2) and is kept only for
        // small decompositions; everything else runs on the rolled, table-driven steppers (one device function per
        // elementary function: the analogue of src/taylor_02.cpp:1194-1260). HEYOKA_AMD_EMIT_MODE / kw::emitter override.
        if (d.compact_mode && d.emitter == 0) {
            eo.unroll_max_nodes = 40;
        }
    }
    // (The order of the additions of compact mode, whatever generator is in charge - the staged table stepper deals the
    // terms of a convolution to several lanes otherwise.)
    if (d.compact_mode && eo.sum_order == 0) {
        eo.sum_order = 2;
    }
    const detail::stopwatch sw_gen;
    if (!d.cluster_events) {
        d.emitted = emit_hip_module(d.prog, eo);
    }
    // The planner's verdict: which generator, and - in its notes - why the others did not apply (the reasons of every
    // planner which was tried travel in emitted_module::notes; get_codegen_info() shows the same text).
    detail::log_message(log_level::info, "Taylor batch code generation: " + get_codegen_info());
    if (d.emitted.mode == emit_mode::table || d.emitted.notes.find("cluster mode not applicable") != std::string::npos) {
        // A decomposition which left the on-chip steppers for a generic one is worth a line at the default level only
        // when it is big enough to matter.
        detail::log_message(d.prog.nodes.size() > 150u && !d.compact_mode && d.emitter == 0 ? log_level::warn : log_level::debug,
                            "the decomposition (" + std::to_string(d.prog.nodes.size())
                                + " elementary functions) runs on a generic stepper instead of a wave-cluster kernel: " + d.emitted.notes);
    }
    detail::log_message(log_level::trace, "Taylor batch code generation runtime: " + sw_gen.str());
    const detail::stopwatch sw_jit;
    d.cmod = hiprtc_compile(d.emitted);
    detail::log_message(log_level::trace, "Taylor batch hiprtc compilation runtime: " + sw_jit.str() + " ("
                                              + std::to_string(d.emitted.source.size()) + " bytes of HIP source)");

    d.sys = std::move(sys);
    d.last_h.assign(d.N, 0.);
    d.d_out.assign(static_cast<std::size_t>(d.dim) * d.N, 0.);
    d.step_res.assign(d.N, std::tuple{taylor_outcome::success, 0.});
    d.prop_res.assign(d.N, std::tuple{taylor_outcome::success, 0., 0., std::size_t(0)});
}

tab_core::tab_core() noexcept = default;

tab_core::tab_core(const tab_core &o) : m_impl(o.m_impl ? std::make_unique<impl>() : nullptr)
{
    if (!o.m_impl) {
        return;
    }
    const auto &s = *o.m_impl;
    // Bring the host mirrors of the source up to date, then deep-copy them. The compiled module
    // is shared (reference: shared_ptr<ta_jit_data>, include/heyoka/detail/i_data.hpp:125).
    s.to_host();
    s.fetch_step_res();
    s.fetch_prop_res();
    if (s.lasth_dev_newer) {
        s.d_lasth.download(s.last_h.data(), s.last_h.size() * sizeof(double), s.stream);
        s.lasth_dev_newer = false;
    }
    if (s.tc_dev_newer && s.dmod && s.d_tc.bytes() != 0u) {
        s.ensure_tc_expanded();
        s.tc.resize(static_cast<std::size_t>(s.dim) * (s.order + 1u) * s.N);
        s.d_tc.download(s.tc.data(), s.tc.size() * sizeof(double), s.stream);
        s.tc_dev_newer = false;
    }
    auto &d = *m_impl;
    d.sys = s.sys;
    d.dc = s.dc;
    d.prog = s.prog;
    d.order = s.order;
    d.tol = s.tol;
    d.high_accuracy = s.high_accuracy;
    d.compact_mode = s.compact_mode;
    d.emitter = s.emitter;
    d.cluster_kernel = s.cluster_kernel;
    d.exact_division = s.exact_division;
    d.sum_order = s.sum_order;
    d.events_on_cluster = s.events_on_cluster;
    d.batch_semantics = s.batch_semantics;
    d.N = s.N;
    d.dim = s.dim;
    d.device = s.device;
    d.emitted = s.emitted;
    d.cmod = s.cmod;
    d.cluster_events = s.cluster_events;
    d.ev_emitted = s.ev_emitted;
    d.ev_cmod = s.ev_cmod;
    d.state = s.state;
    d.pars = s.pars;
    d.time_hi = s.time_hi;
    d.time_lo = s.time_lo;
    d.tc = s.tc;
    d.last_h = s.last_h;
    d.d_out = s.d_out;
    d.step_res = s.step_res;
    d.prop_res = s.prop_res;
    d.stream = s.stream;
    d.last_total_steps = s.last_total_steps;
    d.tes = s.tes;
    d.ntes = s.ntes;
    s.cooldowns_to_host();
    d.te_cooldowns = s.te_cooldowns;
    d.host_newer = true;
}

tab_core::tab_core(tab_core &&) noexcept = default;

tab_core &tab_core::operator=(const tab_core &o)
{
    if (this != &o) {
        *this = tab_core(o);
    }
    return *this;
}

tab_core &tab_core::operator=(tab_core &&) noexcept = default;

tab_core::~tab_core() = default;

const taylor_dc_t &tab_core::get_decomposition() const
{
    return m_impl->dc;
}
const taylor_program &tab_core::get_program() const
{
    return m_impl->prog;
}
std::uint32_t tab_core::get_batch_size() const
{
    return m_impl->N;
}
std::uint32_t tab_core::get_order() const
{
    return m_impl->order;
}
double tab_core::get_tol() const
{
    return m_impl->tol;
}
bool tab_core::get_high_accuracy() const
{
    return m_impl->high_accuracy;
}
std::uint64_t tab_core::get_event_detection_failures() const
{
    return m_impl->ed_failures;
}

bool tab_core::get_compact_mode() const
{
    return m_impl->compact_mode;
}
std::uint32_t tab_core::get_dim() const
{
    return m_impl->dim;
}
const tab_core::sys_t &tab_core::get_sys() const
{
    return m_impl->sys;
}
int tab_core::get_device() const
{
    return m_impl->device;
}
const std::vector<char> &tab_core::get_code_object() const
{
    return m_impl->cmod->code;
}

const std::string &tab_core::get_hip_source() const
{
    return m_impl->emitted.source;
}
const std::string &tab_core::get_internal_program() const
{
    return m_impl->emitted.internal_program;
}
std::string tab_core::get_codegen_info() const
{
    const auto &m = m_impl->emitted;
    const char *mode = m.mode == emit_mode::cluster
                           ? "cluster"
                           : (m.mode == emit_mode::table ? "table" : (m.mode == emit_mode::block ? "block" : "unrolled"));
    return std::string(mode) + " [lanes per system: " + std::to_string(m.lanes_per_system)
           + ", statements: " + std::to_string(m.n_statements) + "] " + m.notes;
}

double tab_core::get_compile_seconds() const
{
    return m_impl->cmod->compile_seconds;
}

const std::vector<double> &tab_core::get_time() const
{
    // (Polling the time after every step - benchmark/outer_ss_long_term_batch.cpp does - must not drag the state along.)
    m_impl->times_to_host();
    return m_impl->time_hi;
}

std::pair<const std::vector<double> &, const std::vector<double> &> tab_core::get_dtime() const
{
    m_impl->times_to_host();
    return {m_impl->time_hi, m_impl->time_lo};
}

void tab_core::set_time(const std::vector<double> &t)
{
    auto &d = *m_impl;
    if (t.size() != d.N) {
        throw std::invalid_argument("Invalid number of new times specified in a Taylor integrator in batch mode: the "
                                    "batch size is "
                                    + std::to_string(d.N) + ", but the number of specified times is "
                                    + std::to_string(t.size()));
    }
    d.to_host();
    d.time_hi = t;
    std::fill(d.time_lo.begin(), d.time_lo.end(), 0.);
    d.host_newer = true;
    ++d.time_gen;
}

void tab_core::set_time(double t)
{
    auto &d = *m_impl;
    d.to_host();
    std::fill(d.time_hi.begin(), d.time_hi.end(), t);
    std::fill(d.time_lo.begin(), d.time_lo.end(), 0.);
    d.host_newer = true;
    ++d.time_gen;
}

void tab_core::set_dtime(const std::vector<double> &hi, const std::vector<double> &lo)
{
    auto &d = *m_impl;
    if (hi.size() != d.N || lo.size() != d.N) {
        throw std::invalid_argument("Invalid number of new times specified in a Taylor integrator in batch mode: the "
                                    "batch size is "
                                    + std::to_string(d.N) + ", but the number of specified times is ("
                                    + std::to_string(hi.size()) + ", " + std::to_string(lo.size()) + ")");
    }
    // Checks on the values before anything is touched (dtime_checks(), include/heyoka/detail/taylor_common.hpp:232-249;
    // src/taylor_adaptive_batch.cpp:576-580).
    for (std::uint32_t i = 0; i < d.N; ++i) {
        if (!std::isfinite(hi[i]) || !std::isfinite(lo[i])) {
            throw std::invalid_argument("The components of the double-length representation of the time coordinate must "
                                        "both be finite, but they are "
                                        + fp_to_string(hi[i]) + " and " + fp_to_string(lo[i]) + " instead");
        }
        if (std::abs(hi[i]) < std::abs(lo[i])) {
            throw std::invalid_argument("The first component of the double-length representation of the time coordinate ("
                                        + fp_to_string(hi[i])
                                        + ") must not be smaller in magnitude than the second component ("
                                        + fp_to_string(lo[i]) + ")");
        }
    }
    d.to_host();
    for (std::uint32_t i = 0; i < d.N; ++i) {
        // Normalise (reference: normalise(), include/heyoka/detail/dfloat.hpp:125-139).
        const auto [u, v] = eft_add_dekker(hi[i], lo[i]);
        d.time_hi[i] = u;
        d.time_lo[i] = v;
    }
    d.host_newer = true;
    ++d.time_gen;
}

void tab_core::set_dtime(double hi, double lo)
{
    auto &d = *m_impl;
    set_dtime(std::vector<double>(d.N, hi), std::vector<double>(d.N, lo));
}

void tab_core::hold_host_refs() const
{
    m_impl->sticky_const_refs = true;
}

void tab_core::hold_time_refs() const
{
    m_impl->sticky_time_refs = true;
}

const std::vector<double> &tab_core::get_state() const
{
    m_impl->to_host();
    return m_impl->state;
}

double *tab_core::get_state_data()
{
    auto &d = *m_impl;
    d.to_host();
    d.host_newer = true;
    d.sticky_host_ptr = true;
    return d.state.data();
}

const std::vector<double> &tab_core::get_pars() const
{
    return m_impl->pars;
}

double *tab_core::get_pars_data()
{
    auto &d = *m_impl;
    d.to_host();
    d.host_newer = true;
    d.sticky_host_ptr = true;
    return d.pars.data();
}

void tab_core::set_state_values(const double *in)
{
    auto &d = *m_impl;
    d.to_host();
    std::copy(in, in + d.state.size(), d.state.begin());
    d.host_newer = true;
}

void tab_core::set_pars_values(const double *in)
{
    auto &d = *m_impl;
    d.to_host();
    std::copy(in, in + d.pars.size(), d.pars.begin());
    d.host_newer = true;
}

void tab_core::impl::ensure_tc_complete() const
{
    if (!tc_partial) {
        return;
    }
    tc_partial = false;
    tc_regenerated = true;
    ++tc_regens;
    auto &self = const_cast<impl &>(*this);
    auto a = self.base_args();
    a.state = evs_state.as<double>();
    a.time_hi = evs_thi.as<double>();
    a.time_lo = evs_tlo.as<double>();
    a.tc = d_tc.as<double>();
    // (The regeneration launch stores the Taylor coefficients and NOTHING else: the outputs of the step proper - event
    // jets, selector norms, max |x| - get null pointers, so that a generator regression faults instead of silently
    // rewriting them; the parameters are the ones the step ran with, snapshot below.)
    a.ev_tc = nullptr;
    a.max_abs_state = nullptr;
    a.sel_norms = nullptr;
    if (evs_pars.bytes() != 0u) {
        a.pars = evs_pars.as<double>();
    }
    a.mode = 4;
    a.pad = 3; // every workgroup stores its coefficients, nothing else is stored
    self.d_counters.zero(stream);
    dmod->launch_taylor(a);
    tc_expand_pending = emitted.compact_tc;
}

void tab_core::impl::ensure_tc_expanded() const
{
    ensure_tc_complete();
    if (!tc_expand_pending || !evj_mod) {
        return;
    }
    const struct {
        double *out;
        const double *tc;
        const double *hs;
        unsigned long long N;
        const double *hfull;
    } ea{d_tc.as<double>(), d_tc.as<double>(), nullptr, N, nullptr};
    evj_mod->launch("hy_tc_expand", N, 256, &ea, sizeof(ea), stream);
    tc_expand_pending = false;
}

const std::vector<double> &tab_core::get_tc() const
{
    auto &d = *m_impl;
    d.check_tc_not_stale();
    d.ensure_tc_expanded();
    const auto sz = static_cast<std::size_t>(d.dim) * (d.order + 1u) * d.N;
    if (d.tc.size() != sz) {
        d.tc.assign(sz, 0.);
    }
    if (d.tc_dev_newer && d.dmod && d.d_tc.bytes() != 0u) {
        d.d_tc.download(d.tc.data(), sz * sizeof(double), d.stream);
        d.tc_dev_newer = false;
    }
    return d.tc;
}

const std::vector<double> &tab_core::get_last_h() const
{
    auto &d = *m_impl;
    if (d.lasth_dev_newer && d.dmod) {
        d.d_lasth.download(d.last_h.data(), d.last_h.size() * sizeof(double), d.stream);
        d.lasth_dev_newer = false;
    }
    return d.last_h;
}

const std::vector<double> &tab_core::get_d_output() const
{
    return m_impl->d_out;
}

// Reference: update_d_output(), src/taylor_adaptive_batch.cpp:2251-2327.
const std::vector<double> &tab_core::update_d_output(const std::vector<double> &t, bool rel_time)
{
    auto &d = *m_impl;
    if (t.size() != d.N) {
        throw std::invalid_argument("Invalid number of time coordinates specified for the dense output in a Taylor "
                                    "integrator in batch mode: the batch size is "
                                    + std::to_string(d.N) + ", but the number of time coordinates is "
                                    + std::to_string(t.size()));
    }
    d.ensure_device();
    d.ensure_tc();
    d.check_tc_not_stale();
    std::vector<double> hs(d.N);
    if (rel_time) {
        hs = t;
    } else {
        d.times_to_host();
        const auto &lh = get_last_h();
        for (std::uint32_t i = 0; i < d.N; ++i) {
            // h' = t - (t_now - last_h), in double-length arithmetic.
            const auto t0 = dfloat(d.time_hi[i], d.time_lo[i]) - lh[i];
            hs[i] = static_cast<double>(dfloat(t[i]) - t0);
        }
    }
    if (d.d_dout.bytes() == 0u) {
        d.d_dout = device_buffer(d.d_out.size() * sizeof(double), d.device);
        d.d_douth = device_buffer(static_cast<std::size_t>(d.N) * sizeof(double), d.device);
    }
    d.d_douth.upload(hs.data(), hs.size() * sizeof(double), d.stream);
    d.ensure_tc_expanded();
    d.dmod->launch_dout(d.d_dout.as<double>(), d.d_tc.as<double>(), d.d_douth.as<double>(), d.N);
    d.d_dout.download(d.d_out.data(), d.d_out.size() * sizeof(double), d.stream);
    return d.d_out;
}

const std::vector<double> &tab_core::update_d_output(double t, bool rel_time)
{
    return update_d_output(std::vector<double>(m_impl->N, t), rel_time);
}

const std::vector<std::tuple<taylor_outcome, double>> &tab_core::get_step_res() const
{
    m_impl->fetch_step_res();
    return m_impl->step_res;
}

const std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> &tab_core::get_propagate_res() const
{
    auto &d = *m_impl;
    d.fetch_prop_res();
    if (d.prop_res_override) {
        for (auto &r : d.prop_res) {
            std::get<0>(r) = *d.prop_res_override;
        }
        d.prop_res_override.reset();
    d.fix_step_limit = false;
    }
    return d.prop_res;
}

// ---- events ----
bool tab_core::with_events() const
{
    return m_impl->has_events();
}
const std::vector<core_t_event> &tab_core::get_t_events() const
{
    if (!m_impl->has_events()) {
        throw std::invalid_argument("No events were defined for this integrator");
    }
    return m_impl->tes;
}
const std::vector<core_nt_event> &tab_core::get_nt_events() const
{
    if (!m_impl->has_events()) {
        throw std::invalid_argument("No events were defined for this integrator");
    }
    return m_impl->ntes;
}
const std::vector<std::vector<std::optional<std::pair<double, double>>>> &tab_core::get_te_cooldowns() const
{
    if (!m_impl->has_events()) {
        throw std::invalid_argument("No events were defined for this integrator");
    }
    m_impl->cooldowns_to_host();
    return m_impl->te_cooldowns;
}
void tab_core::reset_cooldowns()
{
    for (std::uint32_t i = 0; i < m_impl->N; ++i) {
        reset_cooldowns(i);
    }
}
void tab_core::reset_cooldowns(std::uint32_t i)
{
    if (!m_impl->has_events()) {
        throw std::invalid_argument("No events were defined for this integrator");
    }
    if (i >= m_impl->N) {
        throw std::invalid_argument("Cannot reset the cooldowns at batch index " + std::to_string(i)
                                    + ": the batch size for this integrator is only " + std::to_string(m_impl->N));
    }
    m_impl->cooldowns_to_host();
    for (auto &cd : m_impl->te_cooldowns[i]) {
        cd.reset();
    }
    m_impl->cd_host_newer = true;
}
void tab_core::set_callback_context(void *ctx)
{
    m_impl->cb_ctx = ctx;
}

// One step with event detection: the event branch of step_impl(), src/taylor_adaptive_batch.cpp:727-1030.
// Device: stepper with events (jets of the state and of the event equations, step size, no state update), event
// detection kernel, dense-output kernel for the state update at the (possibly truncated) step. Host: the
// reference's sequential per-lane logic on the few detected events.
void tab_core::impl::cooldowns_to_host() const
{
    if (!cd_dev_newer) {
        return;
    }
    const auto n = static_cast<std::size_t>(N);
    const auto n_te = tes.size();
    std::vector<double> cf(n_te * n), cs(n_te * n);
    std::vector<int> ca(n_te * n);
    d_cd_first.download(cf.data(), cf.size() * sizeof(double), stream);
    d_cd_second.download(cs.data(), cs.size() * sizeof(double), stream);
    d_cd_active.download(ca.data(), ca.size() * sizeof(int), stream);
    for (std::size_t i = 0; i < n; ++i) {
        for (std::size_t e = 0; e < n_te; ++e) {
            if (ca[e * n + i] != 0) {
                te_cooldowns[i][e].emplace(cf[e * n + i], cs[e * n + i]);
            } else {
                te_cooldowns[i][e].reset();
            }
        }
    }
    for (std::size_t q = 0; q + 2u < pending_cd.size(); q += 3u) {
        const auto pos = static_cast<std::size_t>(pending_cd[q]);
        te_cooldowns[pos % n][pos / n].emplace(pending_cd[q + 1u], pending_cd[q + 2u]);
    }
    cd_dev_newer = false;
}

void tab_core::impl::cooldowns_to_device()
{
    if (!cd_host_newer || tes.empty()) {
        cd_host_newer = false;
        return;
    }
    cooldowns_to_host();
    const auto n = static_cast<std::size_t>(N);
    const auto n_te = tes.size();
    std::vector<double> cf(n_te * n, 0.), cs(n_te * n, 0.);
    std::vector<int> ca(n_te * n, 0);
    for (std::size_t i = 0; i < n; ++i) {
        for (std::size_t e = 0; e < n_te; ++e) {
            if (const auto &cd = te_cooldowns[i][e]) {
                cf[e * n + i] = cd->first;
                cs[e * n + i] = cd->second;
                ca[e * n + i] = 1;
            }
        }
    }
    d_cd_first.upload(cf.data(), cf.size() * sizeof(double), stream);
    d_cd_second.upload(cs.data(), cs.size() * sizeof(double), stream);
    d_cd_active.upload(ca.data(), ca.size() * sizeof(int), stream);
    cd_host_newer = false;
}

void tab_core::impl::ensure_event_buffers()
{
    const auto n = static_cast<std::size_t>(N);
    const auto dsz = sizeof(double);
    const auto n_te = static_cast<std::uint32_t>(tes.size()), n_nte = static_cast<std::uint32_t>(ntes.size());
    const auto n_ev = n_te + n_nte;
    const auto maxd = ed_max_detected(order, n_te, n_nte);
    ensure_tc();
    if (d_ev_tc.bytes() == 0u) {
        // (One spare block: the stepper which takes close-encounter events from the lanes of their pairs lets the lanes
        // WITHOUT an event store there - every statement unconditional.)
        d_ev_tc = device_buffer((static_cast<std::size_t>(n_ev) + 1u) * (order + 1u) * n * dsz, device);
        d_mas = device_buffer(n * dsz, device);
        d_geps = device_buffer(n * dsz, device);
        d_dirs = device_buffer(std::max<std::size_t>(n_ev, 1u) * sizeof(int), device);
        const auto ncd = std::max<std::size_t>(n_te, 1u) * n;
        d_cd_first = device_buffer(ncd * dsz, device);
        d_cd_second = device_buffer(ncd * dsz, device);
        d_cd_active = device_buffer(ncd * sizeof(int), device);
        d_ed_out = device_buffer(2u * n * maxd * 4u * dsz, device);
        d_ed_counts = device_buffer(2u * n * sizeof(unsigned), device);
        d_ed_flags = device_buffer(4u * sizeof(unsigned), device);
        // Working lists of the root isolation: one column per launched thread of hy_detect_events (a grid-stride loop
        // over the lanes). Sized from what the device keeps in flight, not from the ensemble: at least one wavefront per
        // compute unit (64 x 256 columns), at most 1 GiB (42 KB per column at order 20: ~25 000 columns; almost every
        // lane leaves the kernel at the exclusion test and never touches its column) - the previous 4 GiB budget made a
        // large-N integrator with events fail at allocation where nothing needed the space. The lists of detected events
        // (d_ed_out) are 2 * N * (order + 1) * max(n_te, n_nte) * 32 B: 1.4 GB per million systems at order 20, the price
        // of the reference's bound of (order + 1) detections per event and step.
        const auto per_slot = ed_work_list_bytes_per_slot(order);
        const std::uint64_t max_slots = std::clamp<std::uint64_t>((std::uint64_t(1) << 30) / per_slot / 64u * 64u, 64u * 256u, 64u * 256u * 8u);
        ed_slots = std::min<std::uint64_t>((static_cast<std::uint64_t>(n) + 63u) / 64u * 64u, max_slots);
        d_ed_wl = device_buffer(static_cast<std::size_t>(ed_slots) * per_slot, device);
        d_ev_cursor = device_buffer(4u * sizeof(unsigned long long), device);
        // Library-side counting callbacks only (core_*_event::native_counter): the events are applied on the device.
        ev_native = true;
        std::vector<double> te_cd;
        for (const auto &ev : tes) {
            ev_native = ev_native && ev.native_counter != nullptr;
            te_cd.push_back(ev.cooldown);
        }
        for (const auto &ev : ntes) {
            ev_native = ev_native && ev.native_counter != nullptr;
        }
        if (ev_native) {
            d_ev_counts = device_buffer((tes.size() + ntes.size()) * sizeof(unsigned long long), device);
            d_te_cd = device_buffer(std::max<std::size_t>(te_cd.size(), 1u) * sizeof(double), device);
            if (!te_cd.empty()) {
                d_te_cd.upload(te_cd.data(), te_cd.size() * sizeof(double), stream);
            }
        }
        std::vector<int> dirs;
        for (const auto &ev : tes) {
            dirs.push_back(static_cast<int>(ev.dir));
        }
        for (const auto &ev : ntes) {
            dirs.push_back(static_cast<int>(ev.dir));
        }
        d_dirs.upload(dirs.data(), dirs.size() * sizeof(int), stream);
        ed_mod = std::make_unique<aux_module>(hiprtc_compile_source(make_event_detection_source(order, maxd)), device);
        cd_host_newer = true;
    }
    if (d_dout.bytes() == 0u) {
        d_dout = device_buffer(d_out.size() * dsz, device);
        d_douth = device_buffer(n * dsz, device);
    }
}

// Stepper with events: jets of the state and of the event equations, step sizes, max |x_i|, no state update.
void tab_core::impl::launch_event_stepper(const std::vector<double> *lims)
{
    const auto n = static_cast<std::size_t>(N);
    const auto dsz = sizeof(double);
    if (lims != nullptr) {
        upload_lims(*lims);
    } else {
        d_lim_src = nullptr;
    }
    d_counters.zero(stream);
    auto a = base_args();
    a.tc = d_tc.as<double>();
    a.ev_tc = d_ev_tc.as<double>();
    a.max_abs_state = d_mas.as<double>();
    a.mode = 4;
    a.pad = 1;
    if (cluster_events && emitted.events_in_stepper) {
        tc_partial = false; // (nobody asked for the coefficients of the previous step: this step replaces them)
        const bool all_now = ev_all_tc || tc_regenerated;
        tc_regenerated = false;
        if (!all_now) {
            if (evs_state.bytes() == 0u) {
                evs_state = device_buffer(d_state.bytes(), device);
                evs_thi = device_buffer(d_thi.bytes(), device);
                evs_tlo = device_buffer(d_tlo.bytes(), device);
            }
            // (One copy kernel of the event-detection module: see hy_copy_arrays in event_detection.cpp.)
            struct {
                double *dst[4];
                const double *src[4];
                unsigned long long n[4];
            } ca{{evs_state.as<double>(), evs_thi.as<double>(), evs_tlo.as<double>(), nullptr},
                 {d_state.as<double>(), d_thi.as<double>(), d_tlo.as<double>(), nullptr},
                 {d_state.bytes() / dsz, d_thi.bytes() / dsz, d_tlo.bytes() / dsz, 0u}};
            // (Runtime parameters: a callback of this step may change them before somebody asks for the coefficients.)
            if (prog.n_par != 0u && d_pars.bytes() != 0u) {
                if (evs_pars.bytes() != d_pars.bytes()) {
                    evs_pars = device_buffer(d_pars.bytes(), device);
                }
                ca.dst[3] = evs_pars.as<double>();
                ca.src[3] = d_pars.as<double>();
                ca.n[3] = d_pars.bytes() / dsz;
            }
            ed_mod->launch("hy_copy_arrays", std::min<std::uint64_t>(ca.n[0], std::uint64_t(256) * 256u * 16u), 256, &ca, sizeof(ca), stream);
            a.pad = 0;
        }
    }

    if (cluster_events) {
        if (d_selnorms.bytes() == 0u) {
            d_selnorms = device_buffer(3u * n * dsz, device);
            evj_mod = std::make_unique<aux_module>(ev_cmod, device);
        }
        a.sel_norms = d_selnorms.as<double>();
    }
    dmod->launch_taylor(a);
    tc_expand_pending = cluster_events && emitted.compact_tc;
    if (cluster_events && !emitted.events_in_stepper) {
        // Jets of the event equations, extended norms and final step sizes from the jets of the state variables.
        evj_mod->launch("hy_ev_jets", N, 256, &a, sizeof(a), stream);
    }
}

// Event detection on the device; returns the number of lanes whose event lists overflowed / whose root isolation failed.
unsigned tab_core::impl::launch_event_detection(bool device_g_eps)
{
    const auto n_te = static_cast<std::uint32_t>(tes.size()), n_nte = static_cast<std::uint32_t>(ntes.size());
    d_ed_flags.zero(stream);
    const ed_kargs ea{d_ev_tc.as<double>(),   d_lasth.as<double>(),     d_geps.as<double>(),     d_dirs.as<int>(),
                      d_cd_first.as<double>(), d_cd_second.as<double>(), d_cd_active.as<int>(),   d_ed_out.as<double>(),
                      d_ed_counts.as<unsigned>(), d_ed_flags.as<unsigned>(), N, n_te, n_nte,
                      device_g_eps ? d_mas.as<double>() : nullptr, d_geps.as<double>(), tol,
                      d_ed_wl.as<double>(), ed_slots,
                      // (The stepper which evaluates the event equations itself leaves a flag per system in the buffer of the
                      // selector norms, which it does not use: 0 = no event possible in this step.)
                      (cluster_events && emitted.events_in_stepper) ? d_selnorms.as<double>() : nullptr};
    ed_mod->launch("hy_detect_events", ed_slots, 64, &ea, sizeof(ea), stream);
    return 0;
}

void tab_core::impl::step_with_events(const std::vector<double> &lims, bool wtc)
{
    (void)wtc; // The Taylor coefficients are always written by the stepper with events (:756-757).
    step_with_events_device(&lims);
}

namespace
{

void report_ed_failures(std::uint64_t &ed_failures, const unsigned (&flags)[3])
{
    const auto total = static_cast<std::uint64_t>(flags[0]) + flags[1] + flags[2];
    if (total != 0u) {
        // The reference logs a warning through its logger and ignores the event for the step when the root isolation
        // exceeds its limits (working list > 250 intervals or more isolating intervals than the order,
        // src/detail/event_detection.cpp:2082-2090) or when the root finder fails (:2150-2165). Here the count is kept
        // (get_event_detection_failures()) and the first occurrence is reported on stderr. The list of detected events
        // of a lane holds (order + 1) entries per event of the class: an overflow cannot come from a successful
        // isolation and is reported separately.
        if (ed_failures == 0u) {
            std::fprintf(stderr,
                         "heyoka_amd: warning: event detection: %u root isolation(s) failed (working list > 250 or more "
                         "isolating intervals than the Taylor order), %u root finding(s) failed, %u event list(s) "
                         "overflowed: the events concerned were ignored in this step\n",
                         flags[0], flags[2], flags[1]);
        }
        ed_failures += total;
    }
}

[[noreturn]] void throw_callback_exceptions(std::vector<std::pair<std::uint32_t, std::exception_ptr>> &cb_eptrs)
{
    if (cb_eptrs.size() == 1u) {
        std::rethrow_exception(cb_eptrs[0].second);
    }
    std::string exc_msg = "Two or more exceptions were raised during the execution of event callbacks in a "
                          "batch integrator:\n\n";
    for (auto &[i, eptr] : cb_eptrs) {
        exc_msg += "Batch index #" + std::to_string(i) + ":\n";
        try {
            std::rethrow_exception(eptr);
        } catch (const std::exception &ex) {
            exc_msg += std::string("    Exception message: ") + ex.what() + "\n";
        } catch (...) {
            exc_msg += "    Exception type: unknown\n    Exception message: unknown\n";
        }
        exc_msg += '\n';
    }
    throw std::runtime_error(exc_msg);
}

} // namespace

// One step with events, per-lane bookkeeping on the device: only the lanes with detected events reach the host (compact
// records), which runs the callbacks and the logic that depends on them (src/taylor_adaptive_batch.cpp:837-1030) in
// the order of the batch index; state, times, step sizes, outcomes and cooldowns stay on the device.
void tab_core::impl::step_with_events_device(const std::vector<double> *lims)
{
    const auto n = static_cast<std::size_t>(N);
    const auto dsz = sizeof(double);
    const auto n_te = static_cast<std::uint32_t>(tes.size()), n_nte = static_cast<std::uint32_t>(ntes.size());

    // HEYOKA_AMD_EVENTS_TIMING=1: wall-clock time of the phases (with a stream synchronisation after each of them).
    static const bool timing_env = std::getenv("HEYOKA_AMD_EVENTS_TIMING") != nullptr;
    const bool timing = timing_env || ev_timing;
    auto t_last = std::chrono::steady_clock::now();
    int lap_idx = 0;
    const auto lap = [&](const char *what) {
        if (timing) {
            stream_synchronize(device, stream);
            const auto now = std::chrono::steady_clock::now();
            const auto ms = std::chrono::duration<double, std::milli>(now - t_last).count();
            if (timing_env) {
                std::fprintf(stderr, "[events] %-28s %8.3f ms\n", what, ms);
            }
            if (lap_idx < 5) {
                ev_ms[lap_idx] += ms;
            }
            ++lap_idx;
            t_last = now;
        }
    };
    ++ev_steps;
    before_kernel();
    ensure_event_buffers();
    cooldowns_to_device();
    lap("upload / buffers");

    launch_event_stepper(lims);
    lap("stepper (+ event jets)");
    launch_event_detection(true);
    lap("detection");

    ep_kargs pa{};
    pa.h = d_lasth.as<double>();
    pa.ed_out = d_ed_out.as<double>();
    pa.counts = d_ed_counts.as<unsigned>();
    pa.dout_h = d_douth.as<double>();
    pa.g_eps = d_geps.as<double>();
    pa.state = d_state.as<double>();
    pa.time_hi = d_thi.as<double>();
    pa.time_lo = d_tlo.as<double>();
    pa.lim = d_lim.as<double>();
    pa.cd_first = d_cd_first.as<double>();
    pa.cd_second = d_cd_second.as<double>();
    pa.cd_active = d_cd_active.as<int>();
    pa.outcome = d_outcome.as<long long>();
    pa.last_h = d_lasth.as<double>();
    pa.cursor = d_ev_cursor.as<unsigned long long>();
    pa.N = N;
    pa.n_te = n_te;
    pa.n_nte = n_nte;
    pa.dim = dim;
    d_ev_cursor.zero(stream);
    ed_mod->launch("hy_ev_pre", N, 256, &pa, sizeof(pa), stream);
    unsigned flags[3] = {0, 0, 0};
    unsigned long long cur[4] = {0, 0, 0, 0};
    if (ev_native) {
        pa.native = 1;
        pa.ev_counts = d_ev_counts.as<unsigned long long>();
        pa.te_cd = d_te_cd.as<double>();
        d_ev_counts.zero(stream);
    }
    d_ed_flags.download(flags, sizeof(flags), stream);
    if (cluster_events && emitted.events_in_stepper) {
        // (Workgroups of the stepper which did not store their Taylor coefficients: none if it was asked to store all.)
        unsigned cnt[5] = {0, 0, 0, 0, 0};
        d_counters.download(cnt, sizeof(cnt), stream);
        tc_partial = cnt[4] != 0u;
    }
    d_ev_cursor.download(cur, 2u * sizeof(unsigned long long), stream);
    report_ed_failures(ed_failures, flags);
    lap("pre + flags to host");
    if (!ev_native && cur[0] * dsz > d_ev_rec.bytes()) {
        d_ev_rec = device_buffer(static_cast<std::size_t>(cur[0] + cur[0] / 2u + 1024u) * dsz, device);
    }
    pa.rec = d_ev_rec.as<double>();

    // State update via dense output at the final step sizes (:781), then times / non-finite check / cooldowns /
    // outcomes / records.
    if (cluster_events && emitted.events_in_stepper) {
        // (The stepper evaluated the event equations, took the final step size and updated the state itself.) Lanes whose
        // step is truncated at a terminal event (dout_h != h) are redone from the Taylor coefficients: their workgroup
        // stored them - a detected event is an event the stepper's exclusion test could not rule out.
        if (n_te != 0u) {
            const struct {
                double *out;
                const double *tc;
                const double *hs;
                unsigned long long N;
                const double *hfull;
            } da{d_state.as<double>(), d_tc.as<double>(), d_douth.as<double>(), N, d_lasth.as<double>()};
            evj_mod->launch("hy_dout_c", N, 256, &da, sizeof(da), stream);
        }
    } else if (tc_expand_pending) {
        // (Compact Taylor coefficients: the dense output derives the rows the stepper left out.)
        const struct {
            double *out;
            const double *tc;
            const double *hs;
            unsigned long long N;
            const double *hfull;
        } da{d_state.as<double>(), d_tc.as<double>(), d_douth.as<double>(), N, nullptr};
        evj_mod->launch("hy_dout_c", N, 256, &da, sizeof(da), stream);
    } else {
        dmod->launch_dout(d_state.as<double>(), d_tc.as<double>(), d_douth.as<double>(), N);
    }
    ed_mod->launch("hy_ev_post", N, 256, &pa, sizeof(pa), stream);
    if (ev_native && cur[0] != 0u) {
        ed_mod->launch("hy_ev_native", N, 256, &pa, sizeof(pa), stream);
    }
    const double *rec = nullptr;
    std::size_t rec_size = 0;
    if (ev_native) {
        // The events were applied by hy_ev_post (counts per event, cooldown and outcome of the first terminal event of a
        // lane): what is left of the host loop of src/taylor_adaptive_batch.cpp:837-1030 is adding the counts to the
        // callbacks' counters - no records, no per-event work. (The reference runs the callbacks one by one in batch
        // order; a counter does not see the order.)
        std::vector<unsigned long long> cnts(tes.size() + ntes.size(), 0u);
        if (cur[0] != 0u) {
            d_ev_counts.download(cnts.data(), cnts.size() * sizeof(unsigned long long), stream);
            d_ev_cursor.download(cur, sizeof(cur), stream);
            ev_systems += cur[2];
        } else {
            stream_synchronize(device, stream);
        }
        for (std::size_t e = 0; e < cnts.size(); ++e) {
            auto *ctr = e < tes.size() ? tes[e].native_counter : ntes[e - tes.size()].native_counter;
            __atomic_fetch_add(ctr, static_cast<std::uint64_t>(cnts[e]), __ATOMIC_RELAXED);
        }
        lap("dout + post + records");
        host_newer = false;
        after_kernel();
        step_res_dev_newer = true;
        cd_dev_newer = n_te != 0u;
        return;
    }
    if (cur[0] != 0u) {
        d_ev_cursor.download(cur, 2u * sizeof(unsigned long long), stream);
        rec_size = static_cast<std::size_t>(cur[1]);
        if (cur[1] != 0u) {
            auto *dst = static_cast<double *>(h_ev_rec.reserve(rec_size * dsz));
            d_ev_rec.download(dst, rec_size * dsz, stream);
            rec = dst;
        }
    } else {
        stream_synchronize(device, stream);
    }
    lap("dout + post + records");
    host_newer = false;
    after_kernel();
    step_res_dev_newer = true;
    cd_dev_newer = n_te != 0u;

    // Records in the order of the batch index (the compaction kernel appends them in the order its lanes get there): an
    // index of (lane, offset) pairs, sorted; the events of a record are unpacked into two scratch lists which are reused
    // from record to record - with 10^5 systems reporting events per step a pair of heap-allocated lists per record was
    // most of the host time of a step.
    struct rec_ref {
        std::uint32_t lane;
        std::size_t off;
    };
    std::vector<rec_ref> refs;
    for (std::size_t p = 0; p < rec_size;) {
        const auto *r = rec + p;
        refs.push_back({static_cast<std::uint32_t>(r[0]), p});
        p += 8u + 4u * (static_cast<std::size_t>(r[1]) + static_cast<std::size_t>(r[2]));
        ++ev_systems;
    }
    std::sort(refs.begin(), refs.end(), [](const auto &x, const auto &y) { return x.lane < y.lane; });
    struct lane_rec {
        std::uint32_t lane = 0;
        double g_eps = 0, h = 0, thi = 0, tlo = 0;
        std::vector<detected_event> tes, ntes;
    } lr;

    std::vector<std::pair<std::uint32_t, std::exception_ptr>> cb_eptrs;
    auto &upd_cd = pending_cd;
    upd_cd.clear();
    std::vector<double> upd_oc;
    const auto gen = time_gen;
    for (const auto &ref : refs) {
        {
            const auto *r = rec + ref.off;
            lr.lane = ref.lane;
            lr.g_eps = r[3];
            lr.h = r[4];
            lr.thi = r[5];
            lr.tlo = r[6];
            lr.tes.clear();
            lr.ntes.clear();
            const auto c_te = static_cast<unsigned>(r[1]), c_nte = static_cast<unsigned>(r[2]);
            const auto *e = r + 8;
            for (unsigned c = 0; c < c_te + c_nte; ++c, e += 4) {
                (c < c_te ? lr.tes : lr.ntes).push_back({static_cast<std::uint32_t>(e[0]), e[1], static_cast<int>(e[2]), e[3]});
            }
        }
        // (Stable, by |root|: src/detail/event_detection.cpp:771-781. Insertion sort: the lists hold one or two events and
        // std::stable_sort() asks the allocator for a buffer every time.)
        const auto sort_by_root = [](std::vector<detected_event> &v) {
            for (std::size_t a_ = 1; a_ < v.size(); ++a_) {
                const auto x = v[a_];
                auto b_ = a_;
                for (; b_ > 0u && std::abs(x.root) < std::abs(v[b_ - 1u].root); --b_) {
                    v[b_] = v[b_ - 1u];
                }
                v[b_] = x;
            }
        };
        sort_by_root(lr.tes);
        sort_by_root(lr.ntes);
        const auto i = lr.lane;
        const auto h = lr.h;
        const auto new_time = dfloat(lr.thi, lr.tlo);

        // Non-terminal events triggering before the first terminal event (:837-871).
        bool nt_cb_exception = false;
        for (const auto &ev : lr.ntes) {
            if (!lr.tes.empty() && !(std::abs(ev.root) < std::abs(h))) {
                break;
            }
            try {
                ntes[ev.idx].callback(cb_ctx, static_cast<double>(new_time - h + ev.root), ev.d_sgn, i);
            } catch (...) {
                cb_eptrs.emplace_back(i, std::current_exception());
                nt_cb_exception = true;
                break;
            }
        }
        if (nt_cb_exception || lr.tes.empty()) {
            continue;
        }

        // The first terminal event (:875-908).
        const auto &ev = lr.tes[0];
        auto &te = tes[ev.idx];
        auto cd = te.cooldown;
        if (!(cd >= 0)) {
            // taylor_deduce_cooldown(), src/detail/event_detection.cpp:519-550.
            cd = lr.g_eps / ev.abs_der * 10;
            if (!std::isfinite(cd)) {
                cd = 0;
            }
        }
        upd_cd.insert(upd_cd.end(), {static_cast<double>(static_cast<std::size_t>(ev.idx) * n + i), 0., cd});
        if (!cd_dev_newer) {
            // (An earlier callback of this step moved the cooldowns to the host: the mirror is the authoritative copy.)
            te_cooldowns[i][ev.idx].emplace(0., cd);
        }
        bool te_cb_ret = false;
        if (te.callback) {
            try {
                te_cb_ret = te.callback(cb_ctx, ev.d_sgn, i);
            } catch (...) {
                cb_eptrs.emplace_back(i, std::current_exception());
                continue;
            }
        }
        const auto ev_idx = static_cast<std::int64_t>(ev.idx);
        upd_oc.insert(upd_oc.end(), {static_cast<double>(i), static_cast<double>(te_cb_ret ? ev_idx : (-ev_idx - 1))});
    }
    if (!upd_cd.empty() || !upd_oc.empty()) {
        // NOTE: a callback may have moved the host mirrors ahead (mutable getters): the device arrays touched here
        // (cooldowns, outcomes) are not among those it can reach.
        std::vector<double> upd(upd_cd);
        upd.insert(upd.end(), upd_oc.begin(), upd_oc.end());
        if (upd.size() * dsz > d_ev_upd.bytes()) {
            d_ev_upd = device_buffer((upd.size() * 2u + 64u) * dsz, device);
        }
        d_ev_upd.upload(upd.data(), upd.size() * dsz, stream);
        pa.upd = d_ev_upd.as<double>();
        pa.n_cd = static_cast<unsigned>(upd_cd.size() / 3u);
        pa.n_oc = static_cast<unsigned>(upd_oc.size() / 2u);
        ed_mod->launch("hy_ev_scatter", pa.n_cd + pa.n_oc, 256, &pa, sizeof(pa), stream);
        stream_synchronize(device, stream);
    }
    pending_cd.clear();

    if (!cb_eptrs.empty()) {
        throw_callback_exceptions(cb_eptrs);
    }
    if (time_gen != gen) {
        // A callback went through set_time() / set_dtime(): compare the host mirror with the times of the device.
        std::vector<double> thi(n), tlo(n);
        d_thi.download(thi.data(), n * dsz, stream);
        d_tlo.download(tlo.data(), n * dsz, stream);
        for (std::uint32_t i = 0; i < N; ++i) {
            const auto same = [](double x, double y) { return x == y || (std::isnan(x) && std::isnan(y)); };
            if (!same(time_hi[i], thi[i]) || !same(time_lo[i], tlo[i])) {
                throw std::runtime_error("The invocation of one or more event callbacks resulted in the alteration of the "
                                         "time coordinate of the integrator at the batch index "
                                         + std::to_string(i) + " - this is not supported");
            }
        }
    }
}

// ---- stepping (reference: src/taylor_adaptive_batch.cpp:1039-1080) ----
void tab_core::step(bool wtc)
{
    const auto &lims = m_impl->inf_lims(true);
    if (m_impl->has_events()) {
        m_impl->step_with_events(lims, wtc);
    } else {
        m_impl->run_step(lims, wtc);
    }
}

void tab_core::step_backward(bool wtc)
{
    const auto &lims = m_impl->inf_lims(false);
    if (m_impl->has_events()) {
        m_impl->step_with_events(lims, wtc);
    } else {
        m_impl->run_step(lims, wtc);
    }
}

void tab_core::step(const std::vector<double> &max_delta_ts, bool wtc)
{
    auto &d = *m_impl;
    if (max_delta_ts.size() != d.N) {
        throw std::invalid_argument("Invalid number of max timesteps specified in a Taylor integrator in batch mode: "
                                    "the batch size is "
                                    + std::to_string(d.N) + ", but the number of specified timesteps is "
                                    + std::to_string(max_delta_ts.size()));
    }
    if (std::any_of(max_delta_ts.begin(), max_delta_ts.end(), [](double x) { return std::isnan(x); })) {
        throw std::invalid_argument("Cannot invoke the step() function of an adaptive Taylor integrator in batch "
                                    "mode if one of the max timesteps is nan");
    }
    if (d.has_events()) {
        d.step_with_events(max_delta_ts, wtc);
    } else {
        d.run_step(max_delta_ts, wtc);
    }
}

// Reference: propagate_for_impl(), src/taylor_adaptive_batch.cpp:1082-1118.
// Reference outcomes on the device-resident propagation (config::batch_semantics == 0, the default). In the reference
// every iteration of propagate_until() steps ALL the lanes of the batch; a lane which produces a non-finite state stops
// the whole batch at that iteration (src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467) and max_steps counts iterations of
// the batch (:1516). The device-resident loop runs every lane on its own. Its results are the reference's whenever no lane
// goes non-finite (finished lanes take zero-length steps in the reference: nothing changes) up to the outcome of a
// step-limited batch, which is fixed when the results are fetched (impl::fetch_prop_res()). A batch WITH a non-finite
// lane - an error path - is rolled back to the snapshot taken before the launch and re-run through the lock-step loop,
// which implements the reference's semantics iteration by iteration.
void tab_core::finish_device_propagate(const std::vector<double> &ts, std::size_t max_steps,
                                       const std::vector<double> &max_delta_ts, bool wtc)
{
    auto &d = *m_impl;
    if (d.batch_semantics != 0) {
        return;
    }
    d.fix_step_limit = max_steps != 0u;
    unsigned nf = 0;
    // (One 4-byte download per call: it waits for the launch, i.e. propagate_*() is synchronous in this mode;
    // batch_semantics = 2 keeps the fully asynchronous per-lane behaviour.)
    d.d_counters.download(&nf, sizeof(unsigned), d.stream);
    if (nf == 0u) {
        return;
    }
    d.fix_step_limit = false;
    d.rollback_to_snapshot();
    struct flag_guard {
        bool &f;
        explicit flag_guard(bool &x) : f(x)
        {
            f = true;
        }
        ~flag_guard()
        {
            f = false;
        }
    } guard(d.force_lockstep);
    propagate_until(ts, max_steps, max_delta_ts, {}, wtc, false);
}

void tab_core::propagate_for(const std::vector<double> &delta_ts, std::size_t max_steps,
                             const std::vector<double> &max_delta_ts, const cb_t &cb, bool wtc, bool c_out,
                             const pre_t &pre)
{
    auto &d = *m_impl;
    if (delta_ts.size() != 1u && delta_ts.size() != d.N) {
        throw std::invalid_argument("Invalid number of time intervals specified in a Taylor integrator in batch "
                                    "mode: the batch size is "
                                    + std::to_string(d.N) + ", but the number of specified time intervals is "
                                    + std::to_string(delta_ts.size()));
    }
    d.times_to_host();
    std::vector<double> ts(2u * static_cast<std::size_t>(d.N));
    for (std::uint32_t i = 0; i < d.N; ++i) {
        const auto dt = delta_ts.size() == 1u ? delta_ts[0] : delta_ts[i];
        const auto tf = dfloat(d.time_hi[i], d.time_lo[i]) + dt;
        ts[i] = tf.hi;
        ts[d.N + i] = tf.lo;
    }
    // NOTE: double-length final times travel as a vector of size 2 * N: a form accepted only from here (a public
    // propagate_until() call with any size other than N throws like the reference).
    struct dl_guard {
        bool &flag;
        explicit dl_guard(bool &f) : flag(f)
        {
            flag = true;
        }
        ~dl_guard()
        {
            flag = false;
        }
    } guard(d.dl_times_ok);
    propagate_until(ts, max_steps, max_delta_ts, cb, wtc, c_out, pre);
}

// Reference: propagate_until_impl(), src/taylor_adaptive_batch.cpp:1137-1534.
void tab_core::propagate_until(const std::vector<double> &ts_, std::size_t max_steps,
                               const std::vector<double> &max_delta_ts, const cb_t &cb, bool wtc, bool c_out,
                               const pre_t &pre)
{
    auto &d = *m_impl;
    const auto N = d.N;

    const auto check_mdts = [&]() {
        if (!max_delta_ts.empty() && max_delta_ts.size() != N) {
            throw std::invalid_argument("Invalid number of max timesteps specified in a Taylor integrator in batch "
                                        "mode: the batch size is "
                                        + std::to_string(N) + ", but the number of specified timesteps is "
                                        + std::to_string(max_delta_ts.size()));
        }
        for (const auto dt : max_delta_ts) {
            if (std::isnan(dt)) {
                throw std::invalid_argument("A nan max_delta_t was passed to the propagate_until() function of an "
                                            "adaptive Taylor integrator in batch mode");
            }
            if (dt <= 0) {
                throw std::invalid_argument("A non-positive max_delta_t was passed to the propagate_until() function "
                                            "of an adaptive Taylor integrator in batch mode");
            }
        }
    };

    // Fast path: state and time live on the device (they were produced by a previous kernel), scalar final
    // time, no callback -> nothing to move or to inspect on the host. The per-lane checks of the reference on
    // the *current* times are subsumed by the kernel: a lane whose time is already non-finite (it can only
    // come from an earlier err_nf_state) reports err_nf_state again instead of raising an exception.
    d.last_c_out.reset();
    if (!cb && !c_out && !d.has_events() && ts_.size() == 1u && d.dev_newer && !d.host_newer && !d.sticky_host_ptr
        && d.dmod && d.batch_semantics != 1 && !d.force_lockstep) {
        if (!std::isfinite(ts_[0])) {
            throw std::invalid_argument("A non-finite time was passed to the propagate_until() function of an "
                                        "adaptive Taylor integrator in batch mode");
        }
        check_mdts();
        d.prop_res_override.reset();
    d.fix_step_limit = false;
        d.d_counters.zero(d.stream);
        auto a = d.base_args();
        a.tfin_hi = nullptr;
        a.tfin_lo = nullptr;
        a.tfin_s_hi = ts_[0];
        a.tfin_s_lo = 0.;
        if (max_delta_ts.empty()) {
            a.lim = nullptr;
        } else {
            d.d_lim.upload(max_delta_ts.data(), max_delta_ts.size() * sizeof(double), d.stream);
            d.d_lim_src = nullptr;
        }
        if (wtc && d.is_cluster()) {
            d.ensure_tc();
            a.tc = d.d_tc.as<double>();
        }
        a.mode = 1;
        a.max_steps = max_steps;
        d.keep_written_tc(wtc);
        if (d.batch_semantics == 0) {
            d.snapshot_for_rollback();
        }
        d.dmod->launch_taylor(a);
        d.after_kernel(wtc);
        d.prop_res_dev_newer = true;
        d.step_res_dev_newer = false;
        finish_device_propagate(ts_, max_steps, max_delta_ts, wtc);
        return;
    }

    std::vector<double> tf_hi(N), tf_lo(N, 0.);
    if (ts_.size() == 1u) {
        std::fill(tf_hi.begin(), tf_hi.end(), ts_[0]);
    } else if (ts_.size() == N) {
        tf_hi = ts_;
    } else if (d.dl_times_ok && ts_.size() == 2u * static_cast<std::size_t>(N)) {
        std::copy(ts_.begin(), ts_.begin() + N, tf_hi.begin());
        std::copy(ts_.begin() + N, ts_.end(), tf_lo.begin());
    } else {
        throw std::invalid_argument("Invalid number of time limits specified in a Taylor integrator in batch mode: "
                                    "the batch size is "
                                    + std::to_string(N) + ", but the number of specified time limits is "
                                    + std::to_string(ts_.size()));
    }

    d.times_to_host();
    const auto nonfinite = [](double t) { return !std::isfinite(t); };
    if (std::any_of(d.time_hi.begin(), d.time_hi.end(), nonfinite)
        || std::any_of(d.time_lo.begin(), d.time_lo.end(), nonfinite)) {
        throw std::invalid_argument("Cannot invoke the propagate_until() function of an adaptive Taylor integrator "
                                    "in batch mode if one of the current times is not finite");
    }
    if (std::any_of(tf_hi.begin(), tf_hi.end(), nonfinite) || std::any_of(tf_lo.begin(), tf_lo.end(), nonfinite)) {
        throw std::invalid_argument("A non-finite time was passed to the propagate_until() function of an adaptive "
                                    "Taylor integrator in batch mode");
    }
    if (!max_delta_ts.empty() && max_delta_ts.size() != N) {
        throw std::invalid_argument("Invalid number of max timesteps specified in a Taylor integrator in batch mode: "
                                    "the batch size is "
                                    + std::to_string(N) + ", but the number of specified timesteps is "
                                    + std::to_string(max_delta_ts.size()));
    }
    for (const auto dt : max_delta_ts) {
        if (std::isnan(dt)) {
            throw std::invalid_argument("A nan max_delta_t was passed to the propagate_until() function of an "
                                        "adaptive Taylor integrator in batch mode");
        }
        if (dt <= 0) {
            throw std::invalid_argument("A non-positive max_delta_t was passed to the propagate_until() function of "
                                        "an adaptive Taylor integrator in batch mode");
        }
    }
    std::vector<dfloat> rem(N);
    for (std::uint32_t i = 0; i < N; ++i) {
        rem[i] = dfloat(tf_hi[i], tf_lo[i]) - dfloat(d.time_hi[i], d.time_lo[i]);
        if (!isfinite(rem[i])) {
            throw std::invalid_argument("The final time passed to the propagate_until() function of an adaptive "
                                        "Taylor integrator in batch mode results in an overflow condition");
        }
    }

    d.prop_res_override.reset();
    d.fix_step_limit = false;

    // The reference's batch-wide semantics (src/taylor_adaptive_batch.cpp:1404-1407, :1462-1467, :1516): a non-finite lane
    // stops the whole batch at that iteration, max_steps counts lock-step iterations of the batch and the lanes which are
    // done keep taking zero-length steps. batch_semantics = 1 routes propagate_until() / propagate_for() through the
    // lock-step loop (one step of every lane per sweep), which implements exactly that; the default (0) runs every lane's
    // own loop on the device and falls back to the lock-step loop only where the outcomes would differ (DESIGN.md,
    // "Outcome semantics").
    const bool ref_semantics = d.batch_semantics == 1 || d.force_lockstep;

    if (!cb && !c_out && !d.has_events() && !ref_semantics) {
        // Device-resident propagation: every lane runs its own adaptive loop to completion
        // (or to max_steps) inside a single kernel launch.
        d.before_kernel();
        d.d_tfhi.upload(tf_hi.data(), tf_hi.size() * sizeof(double), d.stream);
        d.d_tflo.upload(tf_lo.data(), tf_lo.size() * sizeof(double), d.stream);
        d.d_counters.zero(d.stream);
        auto a = d.base_args();
        if (max_delta_ts.empty()) {
            a.lim = nullptr;
        } else {
            d.d_lim.upload(max_delta_ts.data(), max_delta_ts.size() * sizeof(double), d.stream);
            d.d_lim_src = nullptr;
        }
        if (wtc && d.is_cluster()) {
            d.ensure_tc();
            a.tc = d.d_tc.as<double>();
        }
        a.mode = 1;
        a.max_steps = max_steps;
        d.keep_written_tc(wtc);
        if (d.batch_semantics == 0) {
            d.snapshot_for_rollback();
        }
        d.dmod->launch_taylor(a);
        d.after_kernel(wtc);
        d.prop_res_dev_newer = true;
        d.step_res_dev_newer = false;
        finish_device_propagate(ts_, max_steps, max_delta_ts, wtc);
        return;
    }

    // Lock-step propagation with a callback executed after every sweep and/or the recording of the
    // continuous output: the reference's loop, one single-step kernel launch per iteration.
    // The pre_hook() of the step callback, once, before the first step (src/taylor_adaptive_batch.cpp:1356-1365).
    if (cb && pre) {
        const auto gen = d.time_gen;
        pre();
        if (d.time_gen != gen) {
            throw std::runtime_error("The invocation of the callback passed to propagate_until() resulted in the "
                                     "alteration of the time coordinate of the integrator - this is not supported");
        }
        // (The hook may have changed the state: the remaining times only depend on the times.)
    }
    // If c_out is true, we always need to write the Taylor coefficients (:1243-1244).
    wtc = wtc || c_out;
    std::unique_ptr<c_out_builder> cob;
    if (c_out) {
        d.ensure_device();
        cob = std::make_unique<c_out_builder>(N, d.order, d.dim, d.high_accuracy, d.device, d.stream, d.time_hi,
                                              d.time_lo);
    }
    std::vector<int> t_dir(N);
    std::vector<double> min_abs_h(N, std::numeric_limits<double>::infinity()), max_abs_h(N, 0.);
    std::vector<double> cur_max(N);
    for (std::uint32_t i = 0; i < N; ++i) {
        t_dir[i] = rem[i] >= dfloat(0.);
    }
    const auto pinf = std::numeric_limits<double>::infinity();
    std::size_t iter_counter = 0;

    {
        // Device-driven loop: the per-lane bookkeeping runs in a post-step kernel, the host reads three counters per
        // sweep, runs the callback and (for the continuous output) appends the coefficients device-to-device.
        d.ensure_device();
        d.ensure_tc();
        if (!d.grid_mod) {
            d.grid_mod = std::make_unique<aux_module>(
                hiprtc_compile_source(make_grid_source(d.order, d.dim, d.high_accuracy)), d.device);
        }
        const auto dsz = sizeof(double);
        device_buffer b_rem_hi(N * dsz, d.device), b_rem_lo(N * dsz, d.device), b_mdt(N * dsz, d.device);
        device_buffer b_tdir(N * sizeof(int), d.device), b_cnt(4u * sizeof(unsigned), d.device);
        std::vector<double> rhi(N), rlo(N), mdts(N);
        const std::vector<unsigned long long> ns0(N, 0u);
        for (std::uint32_t i = 0; i < N; ++i) {
            rhi[i] = rem[i].hi;
            rlo[i] = rem[i].lo;
            mdts[i] = max_delta_ts.empty() ? pinf : max_delta_ts[i];
            const auto dt_limit
                = t_dir[i] != 0 ? std::min(dfloat(mdts[i]), rem[i]) : std::max(dfloat(-mdts[i]), rem[i]);
            cur_max[i] = static_cast<double>(dt_limit);
        }
        b_rem_hi.upload(rhi.data(), N * dsz, d.stream);
        b_rem_lo.upload(rlo.data(), N * dsz, d.stream);
        b_mdt.upload(mdts.data(), N * dsz, d.stream);
        b_tdir.upload(t_dir.data(), N * sizeof(int), d.stream);
        d.d_tfhi.upload(tf_hi.data(), N * dsz, d.stream);
        d.d_tflo.upload(tf_lo.data(), N * dsz, d.stream);
        d.d_lim.upload(cur_max.data(), N * dsz, d.stream);
        d.d_lim_src = nullptr;
        d.d_minh.upload(min_abs_h.data(), N * dsz, d.stream);
        d.d_maxh.upload(max_abs_h.data(), N * dsz, d.stream);
        d.d_nsteps.upload(ns0.data(), N * sizeof(unsigned long long), d.stream);
        const auto make_c_out = [&]() {
            if (cob) {
                d.last_c_out = cob->finish(t_dir);
            }
        };
        // (Continuous output consumes the Taylor coefficients of every step.)
        const struct all_tc_guard {
            bool &flag;
            bool old;
            all_tc_guard(bool &f, bool v) : flag(f), old(f)
            {
                flag = v;
            }
            ~all_tc_guard()
            {
                flag = old;
            }
        } tc_guard(d.ev_all_tc, static_cast<bool>(cob));
        while (true) {
            if (d.has_events()) {
                // (Callbacks of the events run inside: state, times, outcomes and cooldowns stay on the device.)
                d.step_with_events_device(nullptr);
            } else {
                d.run_step_impl(nullptr, wtc);
            }
            b_cnt.zero(d.stream);
            const grid_kargs a{d.d_tfhi.as<double>(), d.d_tflo.as<double>(), nullptr, d.d_thi.as<double>(),
                               d.d_tlo.as<double>(), d.d_lasth.as<double>(), d.d_outcome.as<long long>(),
                               b_rem_hi.as<double>(), b_rem_lo.as<double>(), b_mdt.as<double>(), b_tdir.as<int>(),
                               d.d_lim.as<double>(), nullptr, d.d_minh.as<double>(), d.d_maxh.as<double>(),
                               d.d_nsteps.as<unsigned long long>(), b_cnt.as<unsigned>(), N, 0u, nullptr, nullptr, nullptr, nullptr, nullptr};
            d.grid_mod->launch("hy_until_post", N, 256, &a, sizeof(a), d.stream);
            unsigned cnt[3] = {0, 0, 0};
            b_cnt.download(cnt, sizeof(cnt), d.stream);
            // Outcomes of the last sweep + accumulated statistics: on the device.
            d.prop_res_dev_newer = true;
            d.step_res_dev_newer = true;
            if (cnt[1] != 0u) {
                make_c_out();
                return;
            }
            if (cob) {
                d.times_to_host();
                d.ensure_tc_expanded();
                cob->append(d.d_tc.as<double>(), d.time_hi, d.time_lo);
            }
            ++iter_counter;
            if (cb) {
                const auto gen = d.time_gen;
                const auto ret_cb = cb();
                if (d.time_gen != gen) {
                    throw std::runtime_error("The invocation of the callback passed to propagate_until() resulted in "
                                             "the alteration of the time coordinate of the integrator - this is not "
                                             "supported");
                }
                if (!ret_cb) {
                    d.prop_res_override = taylor_outcome::cb_stop;
                    make_c_out();
                    return;
                }
            }
            // (cnt[2]: lanes stopped by a terminal event - the propagation of the whole batch ends, :1411, :1429.)
            if (cnt[0] == N || cnt[2] != 0u) {
                make_c_out();
                return;
            }
            if (iter_counter == max_steps) {
                d.prop_res_override = taylor_outcome::step_limit;
                make_c_out();
                return;
            }
        }
    }

}

std::optional<c_out_core> tab_core::take_c_output()
{
    auto ret = std::move(m_impl->last_c_out);
    m_impl->last_c_out.reset();
    return ret;
}

namespace
{


} // namespace

// Post-step kernel of the device-resident propagate_grid() loop: the per-lane body of the reference's loop
// (src/taylor_adaptive_batch.cpp:1760-2040) - step counters, remaining time, dense output at every grid
// point inside the step just taken (h' = t_grid - (t_now - last_h) in double-length arithmetic, Horner or
// compensated summation as in taylor_add_d_out_function()), limit of the next step.
std::string make_grid_source(std::uint32_t order, std::uint32_t dim, bool ha)
{
    std::ostringstream src;
    src << emit_detail::prelude;
    src << "#define HY_ORDER " << order << "u\n#define HY_DIM " << dim << "u\n#define HY_HA " << (ha ? 1 : 0) << "\n";
    src << R"HIP(
struct hy_grid_args {
    const double *grid;
    double *out;
    const double *tc;
    const double *thi;
    const double *tlo;
    const double *last_h;
    const i64 *outcome;
    double *rem_hi;
    double *rem_lo;
    const double *mdt;
    const int *t_dir;
    double *lim;
    unsigned *gidx;
    double *min_h;
    double *max_h;
    u64 *n_steps;
    unsigned *counters;
    u64 N;
    unsigned n_grid;
    double *next_tg;
    u64 *acc_n_steps;
    double *acc_min_h;
    double *acc_max_h;
    const double *grid_done;
};

// Post-step kernel of the device-driven propagate_until() lock-step loop (callbacks / continuous output): the
// per-lane bookkeeping of src/taylor_adaptive_batch.cpp:1395-1440 (step counters, min/max |h|, remaining time, limit of
// the next step). counters[0] = lanes done in this sweep, counters[1] = lanes with a non-finite state. The final
// times are in the (double-length) grid row 0: grid[i] = hi, out[i] = lo.
// One atomic per wavefront instead of one per lane: the lanes which reach a call site with pred set elect the lowest of
// them, which adds their number. (hy_grid_post counts the lanes which are NOT through their grid - every lane of every
// sweep: 262 144 atomics on one address were 6 ms of a 6.5-ms sweep, profiles/r05_grid_sweeps.log.)
__device__ __forceinline__ void hy_count(unsigned *p, bool pred)
{
    const u64 m = __builtin_amdgcn_ballot_w64(pred);
    if (pred && (unsigned)__builtin_ctzll(m) == (threadIdx.x & 63u)) atomicAdd(p, (unsigned)__builtin_popcountll(m));
}

extern "C" __global__ void __launch_bounds__(256) hy_until_post(const hy_grid_args a)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    if (i >= N) return;
    const i64 oc = a.outcome[i];
    const double h = a.last_h[i];
    if (oc == HY_OC_ERR_NF_STATE) {
        hy_count(a.counters + 1, true);
        return;
    }
    a.n_steps[i] += (h != 0.0) ? 1u : 0u;
    if (oc == HY_OC_SUCCESS) {
        const double ah = fabs(h);
        a.min_h[i] = hy_min(a.min_h[i], ah);
        a.max_h[i] = hy_max(a.max_h[i], ah);
    }
    // Stopping terminal event: outcome -index - 1 (src/taylor_adaptive_batch.cpp:1411).
    hy_count(a.counters + 2, oc > HY_OC_SUCCESS && oc < 0);
    hy_df rem; rem.hi = a.rem_hi[i]; rem.lo = a.rem_lo[i];
    hy_count(a.counters, h == rem.hi);
    if (h == rem.hi) {
        rem.hi = 0.0; rem.lo = 0.0;
    } else {
        hy_df tcur; tcur.hi = a.thi[i]; tcur.lo = a.tlo[i];
        hy_df tf; tf.hi = a.grid[i]; tf.lo = a.out[i];
        rem = hy_df_sub(tf, tcur);
    }
    a.rem_hi[i] = rem.hi; a.rem_lo[i] = rem.lo;
    hy_df m; m.lo = 0.0;
    double lim;
    if (a.t_dir[i] != 0) { m.hi = a.mdt[i]; lim = hy_df_lt(rem, m) ? rem.hi : m.hi; }
    else { m.hi = -a.mdt[i]; lim = hy_df_lt(m, rem) ? rem.hi : m.hi; }
    a.lim[i] = lim;
}

extern "C" __global__ void __launch_bounds__(256) hy_grid_post(const hy_grid_args a)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    if (i >= N) return;
    const i64 oc = a.outcome[i];
    const double h = a.last_h[i];
    if (oc == HY_OC_ERR_NF_STATE) {
        hy_count(a.counters + 1, true);
        return;
    }
    if (a.acc_n_steps != nullptr) {
        // (A launch of several steps per lane: its own counters and extrema.)
        a.acc_n_steps[i] += a.n_steps[i];
        a.acc_min_h[i] = hy_min(a.acc_min_h[i], a.min_h[i]);
        a.acc_max_h[i] = hy_max(a.acc_max_h[i], a.max_h[i]);
    } else {
        a.n_steps[i] += (h != 0.0) ? 1u : 0u;
        if (oc == HY_OC_SUCCESS) {
            const double ah = fabs(h);
            a.min_h[i] = hy_min(a.min_h[i], ah);
            a.max_h[i] = hy_max(a.max_h[i], ah);
        }
    }
    // Stopping terminal event: outcome -index - 1 (:1903-1908).
    hy_count(a.counters + 2, oc > HY_OC_SUCCESS && oc < 0);
    hy_df tcur; tcur.hi = a.thi[i]; tcur.lo = a.tlo[i];
    hy_df rem; rem.hi = a.rem_hi[i]; rem.lo = a.rem_lo[i];
    const unsigned ng = a.n_grid;
    // (A launch of several steps per lane: the stored remaining time is the one before its FIRST step - the stepper says
    // whether its last step was the one clamped to the remaining time.)
    const bool clamped_to_rem = (a.grid_done != nullptr) ? (a.grid_done[i] != 0.0) : (h == rem.hi);
    if (clamped_to_rem) {
        rem.hi = 0.0; rem.lo = 0.0;
    } else {
        hy_df tl; tl.hi = a.grid[(u64)(ng - 1u) * N + i]; tl.lo = 0.0;
        rem = hy_df_sub(tl, tcur);
    }
    a.rem_hi[i] = rem.hi; a.rem_lo[i] = rem.lo;
    // Time interval covered by the step, and the start of the step for the dense output.
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df tstart = hy_df_sub(tcur, hh);
    const bool fwd = !hy_df_lt(tcur, tstart);
    const hy_df t0 = fwd ? tstart : tcur, t1 = fwd ? tcur : tstart;
    const bool done_lane = (rem.hi == 0.0 && rem.lo == 0.0);
    unsigned g = a.gidx[i];
    while (g < ng) {
        hy_df tg; tg.hi = a.grid[(u64)g * N + i]; tg.lo = 0.0;
        const bool avail = (!hy_df_lt(tg, t0) && !hy_df_lt(t1, tg)) || done_lane;
        if (!avail) break;
        const double hd = hy_df_sub(tg, tstart).hi;
        for (unsigned v = 0; v < HY_DIM; ++v) {
            const double *c = a.tc + (u64)v * (HY_ORDER + 1u) * N + i;
#if HY_HA
            double res = c[0], comp = 0.0, cur_h = hd;
            for (unsigned k = 1; k <= HY_ORDER; ++k) {
                const double tmp = c[(u64)k * N] * cur_h;
                const double y = tmp - comp;
                const double t = res + y;
                comp = (t - res) - y;
                res = t;
                cur_h = cur_h * hd;
            }
#else
            double res = c[(u64)HY_ORDER * N];
            for (unsigned k = 1; k <= HY_ORDER; ++k) {
                res = c[(u64)(HY_ORDER - k) * N] + res * hd;
            }
#endif
            a.out[((u64)g * HY_DIM + v) * N + i] = res;
        }
        ++g;
    }
    a.gidx[i] = g;
    // (The next grid time of the lane: the steps which do not reach it need not store their Taylor coefficients.)
    if (a.next_tg != nullptr) {
        a.next_tg[i] = (g < ng) ? a.grid[(u64)g * N + i] : ((a.t_dir[i] != 0) ? __builtin_inf() : -__builtin_inf());
    }
    // Limit of the next step.
    hy_df m; m.lo = 0.0;
    double lim;
    if (a.t_dir[i] != 0) { m.hi = a.mdt[i]; lim = hy_df_lt(rem, m) ? rem.hi : m.hi; }
    else { m.hi = -a.mdt[i]; lim = hy_df_lt(m, rem) ? rem.hi : m.hi; }
    a.lim[i] = lim;
    hy_count(a.counters, g < ng);
}
)HIP";
    return src.str();
}

void tab_core::propagate_grid_device_loop(const std::vector<double> &grid, std::vector<double> &retval,
                                          const std::vector<dfloat> &rem, const std::vector<int> &t_dir,
                                          const std::vector<double> &max_delta_ts, std::size_t max_steps,
                                          double *d_out, const cb_t &cb)
{
    auto &d = *m_impl;
    const auto N = d.N;
    const auto dim = d.dim;
    const auto n_grid = static_cast<std::uint32_t>(grid.size() / N);
    const auto pinf = std::numeric_limits<double>::infinity();
    const auto dsz = sizeof(double);

    d.ensure_device();
    d.ensure_tc();
    if (!d.grid_mod) {
        d.grid_mod = std::make_unique<aux_module>(hiprtc_compile_source(make_grid_source(d.order, dim, d.high_accuracy)),
                                                  d.device);
    }

    const auto out_doubles = grid.size() * dim;
    device_buffer b_grid(grid.size() * dsz, d.device), b_out(d_out != nullptr ? 0u : out_doubles * dsz, d.device);
    double *const out_ptr = d_out != nullptr ? d_out : b_out.as<double>();
    device_buffer b_rem_hi(N * dsz, d.device), b_rem_lo(N * dsz, d.device), b_mdt(N * dsz, d.device);
    device_buffer b_tdir(N * sizeof(int), d.device), b_gidx(N * sizeof(unsigned), d.device), b_cnt(4u * sizeof(unsigned), d.device);
    b_grid.upload(grid.data(), grid.size() * dsz, d.stream);
    // Row 0 = current state, everything else NaN until reached.
    if (d_out == nullptr) {
        b_out.upload(retval.data(), retval.size() * dsz, d.stream);
    } else {
        // NOTE: the all-ones byte pattern is a (quiet) NaN.
        device_fill_bytes(out_ptr, 0xFF, out_doubles * dsz, d.device, d.stream);
        d.to_device();
        device_copy(out_ptr, d.d_state.get(), static_cast<std::size_t>(dim) * N * dsz, d.device, d.stream);
    }
    std::vector<double> rhi(N), rlo(N), lim(N), mn(N, pinf), mx(N, 0.);
    std::vector<unsigned> gidx(N, 1u);
    std::vector<unsigned long long> ns(N, 0u);
    for (std::uint32_t i = 0; i < N; ++i) {
        rhi[i] = rem[i].hi;
        rlo[i] = rem[i].lo;
        const auto dt_limit
            = t_dir[i] != 0 ? std::min(dfloat(max_delta_ts[i]), rem[i]) : std::max(dfloat(-max_delta_ts[i]), rem[i]);
        lim[i] = static_cast<double>(dt_limit);
    }
    b_rem_hi.upload(rhi.data(), N * dsz, d.stream);
    b_rem_lo.upload(rlo.data(), N * dsz, d.stream);
    b_mdt.upload(max_delta_ts.data(), N * dsz, d.stream);
    b_tdir.upload(t_dir.data(), N * sizeof(int), d.stream);
    b_gidx.upload(gidx.data(), N * sizeof(unsigned), d.stream);
    d.d_lim.upload(lim.data(), N * dsz, d.stream);
    d.d_lim_src = nullptr;
    d.d_minh.upload(mn.data(), N * dsz, d.stream);
    d.d_maxh.upload(mx.data(), N * dsz, d.stream);
    d.d_nsteps.upload(ns.data(), N * sizeof(unsigned long long), d.stream);

    d.prop_res_override.reset();
    d.fix_step_limit = false;
    std::size_t iter_counter = 0;
    bool any_step = false;
    // Taylor coefficients on demand: dense output is evaluated only in the steps which reach a grid point, so a stepper
    // which can tell (emitted_module::tc_by_threshold) stores the coefficients of those steps only - unless a step callback
    // may look at them, or the stepper with events is in charge (its own on-demand logic is switched off below).
    const bool tc_on_demand = !cb && !d.has_events() && d.emitted.tc_by_threshold && n_grid > 1u;
    // From grid point to grid point in ONE launch per lane (emitted_module::grid_multi_step, hy_kargs::tc_thr): without a
    // callback and without events nothing happens on the host between two sweeps, and the lanes are independent - every lane
    // runs its own steps inside a propagate-mode launch until the step which reaches its next grid time, whose coefficients
    // it stores; hy_grid_post then evaluates the dense output of that step. A launch per grid interval instead of a launch
    // per step: the lock-step loop was at 0.7 of the rate of the propagation loop (ramp-up / drain and clock of 2-ms
    // launches). max_steps counts lock-step iterations of the batch: with a step limit the single-step sweeps stay.
    const bool multi_step = tc_on_demand && d.emitted.grid_multi_step && max_steps == 0u && d.batch_semantics != 1;
    device_buffer b_acc_ns(multi_step ? N * sizeof(unsigned long long) : 0u, d.device), b_acc_min(multi_step ? N * dsz : 0u, d.device),
        b_acc_max(multi_step ? N * dsz : 0u, d.device), b_grid_done(multi_step ? N * dsz : 0u, d.device);
    if (multi_step) {
        std::vector<double> tl(N), zero(N, 0.);
        for (std::uint32_t i = 0; i < N; ++i) {
            tl[i] = grid[static_cast<std::size_t>(n_grid - 1u) * N + i];
        }
        d.d_tfhi.upload(tl.data(), N * dsz, d.stream);
        d.d_tflo.upload(zero.data(), N * dsz, d.stream);
        b_acc_ns.upload(ns.data(), N * sizeof(unsigned long long), d.stream);
        b_acc_min.upload(mn.data(), N * dsz, d.stream);
        b_acc_max.upload(mx.data(), N * dsz, d.stream);
    }
    device_buffer b_next_tg(tc_on_demand ? N * dsz : 0u, d.device);
    if (tc_on_demand) {
        std::vector<double> tg(N);
        for (std::uint32_t i = 0; i < N; ++i) {
            tg[i] = grid[static_cast<std::size_t>(N) + i];
        }
        b_next_tg.upload(tg.data(), N * dsz, d.stream);
        d.tc_threshold = b_next_tg.as<double>();
    }
    const struct thr_reset {
        const double *&p;
        ~thr_reset()
        {
            p = nullptr;
        }
    } thr_guard{d.tc_threshold};
    // (The dense output over the grid consumes the Taylor coefficients of every step.)
    d.ev_all_tc = true;
    const struct all_tc_reset {
        bool &flag;
        ~all_tc_reset()
        {
            flag = false;
        }
    } tc_reset{d.ev_all_tc};
    d.tc_stale = false;
    while (n_grid > 1u) {
        // (The sweep after which max_steps ends the loop stores the coefficients of EVERY lane: the reference leaves the
        // Taylor coefficients of the last step behind, src/taylor_adaptive_batch.cpp:1546-2055.)
        if (tc_on_demand && max_steps != 0u && iter_counter + 1u == max_steps) {
            d.tc_threshold = nullptr;
        }
        if (d.has_events()) {
            d.step_with_events_device(nullptr);
        } else if (multi_step) {
            d.before_kernel();
            d.d_counters.zero(d.stream);
            auto ka = d.base_args();
            ka.tfin_hi = d.d_tfhi.as<double>();
            ka.tfin_lo = d.d_tflo.as<double>();
            ka.lim = b_mdt.as<double>();
            ka.tc = d.d_tc.as<double>();
            ka.tc_thr = b_next_tg.as<double>();
            ka.grid_done = b_grid_done.as<double>();
            ka.mode = 1;
            ka.pad = 4;
            ka.max_steps = 0;
            d.dmod->launch_taylor(ka);
            d.after_kernel(true);
            d.step_res_dev_newer = true;
        } else {
            d.run_step_impl(nullptr, true);
        }
        any_step = true;
        d.ensure_tc_expanded();
        b_cnt.zero(d.stream);
        const grid_kargs a{b_grid.as<double>(),    out_ptr,     d.d_tc.as<double>(),   d.d_thi.as<double>(),
                           d.d_tlo.as<double>(),   d.d_lasth.as<double>(), d.d_outcome.as<long long>(),
                           b_rem_hi.as<double>(),  b_rem_lo.as<double>(),  b_mdt.as<double>(),    b_tdir.as<int>(),
                           d.d_lim.as<double>(),   b_gidx.as<unsigned>(),  d.d_minh.as<double>(), d.d_maxh.as<double>(),
                           d.d_nsteps.as<unsigned long long>(), b_cnt.as<unsigned>(), N, n_grid,
                           tc_on_demand ? b_next_tg.as<double>() : nullptr,
                           multi_step ? b_acc_ns.as<unsigned long long>() : nullptr, multi_step ? b_acc_min.as<double>() : nullptr,
                           multi_step ? b_acc_max.as<double>() : nullptr, multi_step ? b_grid_done.as<double>() : nullptr};
        d.grid_mod->launch("hy_grid_post", N, 256, &a, sizeof(a), d.stream);
        unsigned cnt[3] = {0, 0, 0};
        b_cnt.download(cnt, sizeof(cnt), d.stream);
        if (cnt[1] != 0u) {
            // A non-finite state was detected: stop (the outcomes of the last step are reported). With coefficients on
            // demand the lanes which did not reach a grid point in this sweep hold the coefficients of OLDER steps:
            // get_tc() / update_d_output() refuse to hand those out as the last step's (tc_stale).
            d.tc_stale = tc_on_demand && d.tc_threshold != nullptr;
            break;
        }
        ++iter_counter;
        if (cb) {
            // The step callback, once per sweep (src/taylor_adaptive_batch.cpp:2003-2040); it may read or write the state
            // through the lazily synchronised mirrors, but not move the time coordinate (generation counter).
            d.prop_res_dev_newer = true;
            d.step_res_dev_newer = true;
            const auto gen = d.time_gen;
            const auto ret_cb = cb();
            if (d.time_gen != gen) {
                throw std::runtime_error("The invocation of the callback passed to propagate_grid() resulted in the "
                                         "alteration of the time coordinate of the integrator - this is not supported");
            }
            if (!ret_cb) {
                d.prop_res_override = taylor_outcome::cb_stop;
                break;
            }
        }
        // (cnt[2]: lanes stopped by a terminal event - they interrupt the propagation of the whole batch.)
        if (cnt[0] == 0u || cnt[2] != 0u) {
            break;
        }
        if (iter_counter == max_steps) {
            d.prop_res_override = taylor_outcome::step_limit;
            break;
        }
    }
    if (multi_step && any_step) {
        // (The accumulated counters / extrema take the place of the last launch's own.)
        device_copy(d.d_nsteps.get(), b_acc_ns.get(), N * sizeof(unsigned long long), d.device, d.stream);
        device_copy(d.d_minh.get(), b_acc_min.get(), N * dsz, d.device, d.stream);
        device_copy(d.d_maxh.get(), b_acc_max.get(), N * dsz, d.device, d.stream);
    }
    if (any_step) {
        // Outcomes of the last sweep + the accumulated statistics live on the device.
        d.prop_res_dev_newer = true;
        d.step_res_dev_newer = true;
    }
    if (d_out == nullptr) {
        b_out.download(retval.data(), retval.size() * dsz, d.stream);
    } else {
        stream_synchronize(d.device, d.stream);
    }
}

// Reference: propagate_grid_impl(), src/taylor_adaptive_batch.cpp:1546-2055. Host-driven lock-step loop:
// single-step kernel launches (always with the Taylor coefficients) interleaved with dense-output launches.
// grid[point * N + lane]; return value ret[(point * dim + var) * N + lane], NaN where not reached.
std::vector<double> tab_core::propagate_grid(std::vector<double> grid, std::size_t max_steps,
                                             const std::vector<double> &max_delta_ts_, const cb_t &cb, double *d_out,
                                             const pre_t &pre)
{
    auto &d = *m_impl;
    const auto N = d.N;
    const auto dim = d.dim;
    const auto pinf = std::numeric_limits<double>::infinity();

    if (grid.empty()) {
        throw std::invalid_argument(
            "Cannot invoke propagate_grid() in an adaptive Taylor integrator in batch mode if the time grid is empty");
    }
    if (grid.size() % N != 0u) {
        throw std::invalid_argument("Invalid grid size detected in propagate_grid() for an adaptive Taylor integrator "
                                    "in batch mode: the grid has a size of "
                                    + std::to_string(grid.size()) + ", which is not a multiple of the batch size ("
                                    + std::to_string(N) + ")");
    }
    // The current time coordinates (src/taylor_adaptive_batch.cpp:1588-1593).
    d.times_to_host();
    if (std::any_of(d.time_hi.begin(), d.time_hi.end(), [](double t) { return !std::isfinite(t); })
        || std::any_of(d.time_lo.begin(), d.time_lo.end(), [](double t) { return !std::isfinite(t); })) {
        throw std::invalid_argument("Cannot invoke propagate_grid() in an adaptive Taylor integrator in batch mode if "
                                    "the current time is not finite");
    }
    const std::vector<double> max_delta_ts = max_delta_ts_.empty() ? std::vector<double>(N, pinf) : max_delta_ts_;
    if (max_delta_ts.size() != N) {
        throw std::invalid_argument("Invalid number of max timesteps specified in a Taylor integrator in batch mode: "
                                    "the batch size is "
                                    + std::to_string(N) + ", but the number of specified timesteps is "
                                    + std::to_string(max_delta_ts.size()));
    }
    for (const auto dt : max_delta_ts) {
        if (std::isnan(dt)) {
            throw std::invalid_argument("A nan max_delta_t was passed to the propagate_grid() function of an adaptive "
                                        "Taylor integrator in batch mode");
        }
        if (dt <= 0) {
            throw std::invalid_argument("A non-positive max_delta_t was passed to the propagate_grid() function of an "
                                        "adaptive Taylor integrator in batch mode");
        }
    }

    const auto n_grid_points = grid.size() / N;
    const auto *const gp = grid.data();
    const auto is_nf = [](double t) { return !std::isfinite(t); };
    const char *nf_err_msg
        = "A non-finite time value was passed to propagate_grid() in an adaptive Taylor integrator in batch mode";
    const char *ig_err_msg = "A non-monotonic time grid was passed to propagate_grid() in an adaptive "
                             "Taylor integrator in batch mode";
    if (std::any_of(gp, gp + N, is_nf)) {
        throw std::invalid_argument(nf_err_msg);
    }
    if (n_grid_points > 1u) {
        if (std::any_of(gp + N, gp + 2u * N, is_nf)) {
            throw std::invalid_argument(nf_err_msg);
        }
        if (gp[N] == gp[0]) {
            throw std::invalid_argument(ig_err_msg);
        }
        const auto grid_direction = gp[N] > gp[0];
        for (std::uint32_t i = 1; i < N; ++i) {
            if ((gp[N + i] > gp[i]) != grid_direction) {
                throw std::invalid_argument(ig_err_msg);
            }
        }
        // (Row by row: finiteness of the whole row first, then the ordering - src/taylor_adaptive_batch.cpp:1652-1661.)
        for (std::size_t k = 2; k < n_grid_points; ++k) {
            if (std::any_of(gp + k * N, gp + (k + 1u) * N, is_nf)) {
                throw std::invalid_argument(nf_err_msg);
            }
            for (std::uint32_t i = 0; i < N; ++i) {
                if ((gp[k * N + i] > gp[(k - 1u) * N + i]) != grid_direction) {
                    throw std::invalid_argument(ig_err_msg);
                }
            }
        }
    }
    d.to_host();
    for (std::uint32_t i = 0; i < N; ++i) {
        if (d.time_hi[i] != gp[i]) {
            throw std::invalid_argument("When invoking propagate_grid(), the first element of the time grid "
                                        "must match the current time coordinate - however, the first element of the "
                                        "time grid at batch index "
                                        + std::to_string(i) + " has a value of " + fp_to_string(gp[i])
                                        + ", while the current time coordinate is " + fp_to_string(d.time_hi[i]));
        }
    }

    // NOTE: with a caller-provided device output (MI355X extension, no callback) nothing of size n_grid * dim * N
    // is ever materialised on the host: the samples go straight to d_out and an empty vector is returned.
    if (d_out != nullptr && cb) {
        throw std::invalid_argument("propagate_grid() with a device output buffer does not support callbacks");
    }
    std::vector<double> retval(d_out != nullptr ? 0u : grid.size() * dim, std::numeric_limits<double>::quiet_NaN());
    std::vector<double> pgrid_tmp(gp, gp + N);

    // Propagate up to the first grid point (absorbs the low part of the double-length time).
    propagate_until(pgrid_tmp, max_steps, max_delta_ts, {}, true, false);
    d.fetch_prop_res();
    if (std::any_of(d.prop_res.begin(), d.prop_res.end(),
                    [](const auto &t) { return std::get<0>(t) != taylor_outcome::time_limit; })) {
        for (auto &[oc, min_h, max_h, ts_count] : d.prop_res) {
            (void)oc;
            min_h = pinf;
            max_h = 0;
            ts_count = 0;
        }
        return retval;
    }
    if (d_out == nullptr) {
        d.to_host();
        std::copy(d.state.begin(), d.state.end(), retval.begin());
    } else {
        d.times_to_host();
    }

    std::vector<dfloat> rem(N), t0(N), t1(N);
    std::vector<int> t_dir(N);
    for (std::uint32_t i = 0; i < N; ++i) {
        rem[i] = dfloat(gp[(n_grid_points - 1u) * N + i]) - dfloat(d.time_hi[i], d.time_lo[i]);
        if (!isfinite(rem[i])) {
            throw std::invalid_argument("The final time passed to the propagate_grid() function of an adaptive Taylor "
                                        "integrator in batch mode results in an overflow condition");
        }
        t_dir[i] = rem[i] >= dfloat(0.);
    }

    // The pre_hook() of the step callback (src/taylor_adaptive_batch.cpp:1782-1791).
    if (cb && pre) {
        const auto gen = d.time_gen;
        pre();
        if (d.time_gen != gen) {
            throw std::runtime_error("The invocation of the callback passed to propagate_grid() resulted in the "
                                     "alteration of the time coordinate of the integrator - this is not supported");
        }
    }
    // Device-resident lock-step loop: the step kernel and a post-step kernel (bookkeeping of the reference's loop, dense
    // output at the grid points covered by the step, next step limit) alternate without any per-lane host work; the host
    // reads three counters per sweep and runs the callback, if any.
    propagate_grid_device_loop(grid, retval, rem, t_dir, max_delta_ts, max_steps, d_out, cb);
    return retval;
}

double *tab_core::device_state()
{
    m_impl->to_device();
    return m_impl->d_state.as<double>();
}
double *tab_core::device_pars()
{
    m_impl->to_device();
    return m_impl->d_pars.as<double>();
}
double *tab_core::device_time_hi()
{
    m_impl->to_device();
    return m_impl->d_thi.as<double>();
}
double *tab_core::device_time_lo()
{
    m_impl->to_device();
    return m_impl->d_tlo.as<double>();
}
double *tab_core::device_tc()
{
    m_impl->ensure_device();
    m_impl->ensure_tc();
    m_impl->ensure_tc_expanded();
    return m_impl->d_tc.as<double>();
}

void *tab_core::device_aux(int which)
{
    m_impl->ensure_device();
    switch (which) {
        case 0:
            return m_impl->d_nsteps.get();
        case 1:
            return m_impl->d_outcome.get();
        default:
            return m_impl->d_lasth.get();
    }
}

void tab_core::pack_results(double *dst)
{
    auto &d = *m_impl;
    d.to_device();
    const auto n = static_cast<std::size_t>(d.N), w = sizeof(double);
    // The results of the last propagation: on the device after a device-resident propagation (a step-limited batch first
    // gets the reference's batch-wide outcome, see fetch_prop_res()), otherwise uploaded from the host records.
    if (d.prop_res_dev_newer && (d.fix_step_limit || d.prop_res_override)) {
        d.fetch_prop_res();
    }
    if (d.prop_res_override) {
        // (A lock-step propagation which ended with step_limit / cb_stop: the batch-wide outcome get_propagate_res()
        // reports, src/taylor_adaptive_batch.cpp:1516.)
        for (auto &r : d.prop_res) {
            std::get<0>(r) = *d.prop_res_override;
        }
        d.prop_res_override.reset();
    }
    if (!d.prop_res_dev_newer) {
        std::vector<long long> oc(n);
        std::vector<double> mn(n), mx(n);
        std::vector<unsigned long long> ns(n);
        for (std::size_t i = 0; i < n; ++i) {
            const auto &[o, a, b, c] = d.prop_res[i];
            oc[i] = static_cast<long long>(o);
            mn[i] = a;
            mx[i] = b;
            ns[i] = static_cast<unsigned long long>(c);
        }
        d.d_outcome.upload(oc.data(), n * w, d.stream);
        d.d_minh.upload(mn.data(), n * w, d.stream);
        d.d_maxh.upload(mx.data(), n * w, d.stream);
        d.d_nsteps.upload(ns.data(), n * w, d.stream);
    }
    const auto cp = [&](std::size_t row, const void *src, std::size_t rows) {
        device_copy(dst + row * n, src, rows * n * w, d.device, d.stream);
    };
    cp(0, d.d_state.get(), d.dim);
    cp(d.dim, d.d_thi.get(), 1);
    cp(d.dim + 1u, d.d_tlo.get(), 1);
    cp(d.dim + 2u, d.d_outcome.get(), 1);
    cp(d.dim + 3u, d.d_nsteps.get(), 1);
    cp(d.dim + 4u, d.d_minh.get(), 1);
    cp(d.dim + 5u, d.d_maxh.get(), 1);
}

void tab_core::set_event_timing(bool on)
{
    m_impl->ev_timing = on;
}

std::array<double, 8> tab_core::get_event_stats() const
{
    const auto &d = *m_impl;
    return {static_cast<double>(d.ev_steps), d.ev_ms[0], d.ev_ms[1], d.ev_ms[2], d.ev_ms[3], d.ev_ms[4],
            static_cast<double>(d.tc_regens), static_cast<double>(d.ev_systems)};
}

void tab_core::mark_device_modified()
{
    m_impl->to_device();
    m_impl->dev_newer = true;
    m_impl->times_fresh = false;
}

void tab_core::set_stream(void *s)
{
    m_impl->stream = s;
    if (m_impl->dmod) {
        m_impl->dmod->set_stream(s);
    }
}

void tab_core::set_device(int device)
{
    auto &d = *m_impl;
    if (device == d.device) {
        return;
    }
    // Bring everything back to the host, drop the device objects, switch.
    d.to_host();
    d.fetch_step_res();
    d.fetch_prop_res();
    d.cooldowns_to_host();
    (void)get_last_h();
    if (d.tc_dev_newer && d.dmod) {
        (void)get_tc();
    }
    d.dmod.reset();
    d.d_state = {};
    d.d_pars = {};
    d.d_thi = {};
    d.d_tlo = {};
    d.d_lim = {};
    d.d_lim_src = nullptr;
    d.d_tfhi = {};
    d.d_tflo = {};
    d.d_lasth = {};
    d.d_outcome = {};
    d.d_minh = {};
    d.d_maxh = {};
    d.d_nsteps = {};
    d.d_tc = {};
    d.d_counters = {};
    d.d_dout = {};
    d.d_douth = {};
    // (The rollback snapshot lives on the old device too; a pending step-limit fix-up / forced lock-step flag belonged to a
    // propagation which has been fetched above.)
    d.snap_state = {};
    d.snap_thi = {};
    d.snap_tlo = {};
    d.fix_step_limit = false;
    d.force_lockstep = false;
    // The auxiliary modules (event detection, post-step kernels of the lock-step loops) and the event buffers belong to
    // the old device as well: they are recreated on first use. A caller-provided stream belonged to the old device:
    // back to the default stream of the new one (set_stream() again if needed).
    d.ed_mod.reset();
    d.grid_mod.reset();
    d.evj_mod.reset();
    d.tc_expand_pending = false;
    d.tc_partial = false;
    d.evs_state = {};
    d.evs_pars = {};
    d.evs_thi = {};
    d.evs_tlo = {};
    d.d_selnorms = {};
    d.d_ev_cursor = {};
    d.d_ev_rec = {};
    d.d_ev_upd = {};
    d.cd_host_newer = true;
    d.d_ev_tc = {};
    d.d_mas = {};
    d.d_geps = {};
    d.d_dirs = {};
    d.d_cd_first = {};
    d.d_cd_second = {};
    d.d_cd_active = {};
    d.d_ed_out = {};
    d.d_ed_counts = {};
    d.d_ed_flags = {};
    d.d_ed_wl = {};
    d.stream = nullptr;
    d.device = device;
    d.host_newer = true;
    d.dev_newer = false;
    d.tc_dev_newer = false;
    d.lasth_dev_newer = false;
}

void tab_core::synchronize()
{
    if (m_impl->dmod) {
        m_impl->dmod->synchronize();
    }
}

std::uint64_t tab_core::get_last_total_steps() const
{
    auto &d = *m_impl;
    d.fetch_prop_res();
    std::uint64_t tot = 0;
    for (const auto &r : d.prop_res) {
        tot += std::get<3>(r);
    }
    return tot;
}

std::vector<double> tab_core::get_kernel_ms_history(std::size_t n) const
{
    if (!m_impl->dmod) {
        return {};
    }
    return m_impl->dmod->kernel_ms_history(n);
}

void tab_core::raw_step(double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                        std::uint64_t n_systems, void *d_tape)
{
    auto &d = *m_impl;
    d.ensure_device();
    // Scratch for the outputs the raw ABI does not expose.
    device_buffer tlo(n_systems * sizeof(double), d.device), oc(n_systems * sizeof(long long), d.device),
        lh(n_systems * sizeof(double), d.device);
    device_buffer tc_scratch;
    tlo.zero(d.stream);
    hy_kargs a{};
    a.state = d_state;
    a.pars = d_pars;
    // NOTE: mode 2 does not write the time back (the caller advances it, like step_impl() in the reference).
    a.time_hi = const_cast<double *>(d_time);
    a.time_lo = tlo.as<double>();
    a.lim = d_h;
    a.last_h = lh.as<double>();
    a.outcome = oc.as<long long>();
    d.d_counters.zero(d.stream);
    if (d_tc == nullptr && !d.is_cluster()) {
        tc_scratch = device_buffer(static_cast<std::size_t>(d.dim) * (d.order + 1u) * n_systems * sizeof(double),
                                   d.device);
        a.tc = tc_scratch.as<double>();
    } else {
        a.tc = d_tc;
    }
    a.N = n_systems;
    a.mode = 2;
    a.counters = d.d_counters.as<unsigned>();
    d.dmod->launch_taylor(a, d_tape);
    d.dmod->synchronize();
}

std::pair<std::size_t, std::size_t> tab_core::raw_tape_size_align(std::uint64_t n_systems)
{
    auto &d = *m_impl;
    d.ensure_device();
    // (0 bytes: this stepper keeps its Taylor coefficients on chip. 256: the alignment of a device allocation.)
    return {d.dmod->tape_bytes(n_systems), 256u};
}

void tab_core::raw_d_out_f(double *d_out, const double *d_tc, const double *d_h, std::uint64_t n_systems)
{
    auto &d = *m_impl;
    d.ensure_device();
    d.dmod->launch_dout(d_out, d_tc, d_h, n_systems);
    d.dmod->synchronize();
}

void tab_core::raw_step_e(double *d_jet, const double *d_state, const double *d_pars, const double *d_time, double *d_h,
                          double *d_max_abs_state, std::uint64_t n_systems, void *d_tape)
{
    auto &d = *m_impl;
    if (!with_events()) {
        throw std::invalid_argument("raw_step_e(): the stepper with events exists only in an integrator constructed with events");
    }
    d.ensure_device();
    d.ensure_event_buffers();
    const auto n = static_cast<std::size_t>(n_systems), w = sizeof(double);
    const auto n_ev = d.tes.size() + d.ntes.size();
    const auto tc_words = static_cast<std::size_t>(d.dim) * (d.order + 1u) * n;
    // Scratch for what the ABI does not expose. The stepper which evaluates the event equations itself also updates the
    // state it is given: it works on a copy (step_e leaves the state alone).
    device_buffer st(static_cast<std::size_t>(d.dim) * n * w, d.device), tlo(n * w, d.device), oc(n * sizeof(long long), d.device),
        lh(n * w, d.device), seln(3u * n * w, d.device), cnt(16u * sizeof(unsigned), d.device);
    device_copy(st.get(), d_state, st.bytes(), d.device, d.stream);
    tlo.zero(d.stream);
    cnt.zero(d.stream);
    hy_kargs a{};
    a.state = st.as<double>();
    a.pars = d_pars;
    a.time_hi = const_cast<double *>(d_time);
    a.time_lo = tlo.as<double>();
    a.lim = d_h;
    a.last_h = lh.as<double>();
    a.outcome = oc.as<long long>();
    a.tc = d_jet;
    a.ev_tc = d_jet + tc_words;
    a.max_abs_state = d_max_abs_state;
    a.sel_norms = seln.as<double>();
    a.N = n_systems;
    a.mode = 4;
    a.pad = 1; // the Taylor coefficients of every system
    a.counters = cnt.as<unsigned>();
    (void)n_ev;
    d.dmod->launch_taylor(a, d_tape);
    if (d.cluster_events && !d.evj_mod && d.ev_cmod) {
        d.evj_mod = std::make_unique<aux_module>(d.ev_cmod, d.device);
    }
    if (d.cluster_events && !d.emitted.events_in_stepper) {
        // (Jets of the event equations, extended norms and the step size from the jets of the state variables.)
        d.evj_mod->launch("hy_ev_jets", n_systems, 256, &a, sizeof(a), d.stream);
    }
    if (d.emitted.compact_tc && d.evj_mod) {
        // (The stepper left the rows of the variables defined by another state variable to be derived: x^[k] = v^[k-1] / k.)
        const struct {
            double *out;
            const double *tc;
            const double *hs;
            unsigned long long N;
            const double *hfull;
        } ea{d_jet, d_jet, nullptr, n_systems, nullptr};
        d.evj_mod->launch("hy_tc_expand", n_systems, 256, &ea, sizeof(ea), d.stream);
    }
    // The step size which was taken.
    device_copy(d_h, lh.get(), n * w, d.device, d.stream);
    d.dmod->synchronize();
}

std::vector<double> make_vector_from(double x)
{
    return {x};
}

} // namespace heyoka_amd::detail
