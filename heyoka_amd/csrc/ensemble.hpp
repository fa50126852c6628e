// ensemble_propagate_*_batch(): n_iter independent propagations of copies of a batch integrator
// (reference: include/heyoka/ensemble_propagate.hpp:222-271, src/ensemble_propagate.cpp:193-297).
// The TBB parallel_for of the reference becomes: iteration i runs on HIP device i % n_devices, all
// launches are asynchronous (one device-resident kernel per iteration), one final synchronisation.
// Multi-process / multi-node sharding with an RCCL gather of the final states lives in the Python
// layer (heyoka_amd/ensemble.py), one process per GPU.
#pragma once

#include <algorithm>
#include <cstddef>
#include <exception>
#include <functional>
#include <optional>
#include <thread>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "hip_backend.hpp"
#include "taylor_adaptive_batch.hpp"

namespace heyoka_amd
{

namespace detail
{

enum class ensemble_kind { until, for_ };

// gen(copy, i): sets up iteration i on a copy of the template integrator.
std::vector<tab_core> ensemble_propagate_core(const tab_core &ta, double t, std::size_t n_iter,
                                              const std::function<void(tab_core &, std::size_t)> &gen,
                                              std::size_t max_steps, int n_devices, ensemble_kind kind);
int ensemble_visible_devices();

} // namespace detail

// The final states of the integrators of an ensemble, gathered into ONE buffer on one device:
// data()[row * n_total() + offset(i) + lane] = state row `row`, lane `lane` of integrator i. The reference's
// ensemble_propagate_*() leaves its results in host memory next to each other (src/ensemble_propagate.cpp:193-297);
// here the iterations finish on (up to) 8 devices, and a caller of the drop-in who wants the ensemble in one place asks
// for it with kw::gather = &g (or ensemble_gather_states() / C ABI hy_ensemble_gather_states()).
// Transport: RCCL over xGMI (librccl loaded at run time: one communicator per device of the process,
// ncclSend / ncclRecv of every integrator's state block inside one group, then a strided placement on the destination)
// when there is more than one device or HEYOKA_AMD_GATHER_RCCL=1; device-to-device copies otherwise (and as the fallback
// when librccl is not available).
class ensemble_gathered
{
public:
    ensemble_gathered() = default;
    // The gathered buffer: (dim + 6) rows of n_total 8-byte words, [row * n_total + offset_i + lane]. Rows [0, dim): the
    // states (data() alone is the states-only view of earlier rounds); then time_hi, time_lo, outcome (int64), number
    // of steps (uint64), min |h|, max |h| of the last propagation - what the reference's returned integrators hold
    // (src/ensemble_propagate.cpp:193-297: m_state, m_time_hi / m_time_lo, m_prop_res).
    enum row : std::size_t { time_hi = 0, time_lo = 1, outcome = 2, n_steps = 3, min_h = 4, max_h = 5, n_result_rows = 6 };
    [[nodiscard]] const double *data() const
    {
        return m_buf.as<double>();
    }
    [[nodiscard]] std::size_t n_rows() const
    {
        return m_dim + n_result_rows;
    }
    // (Device pointer to one of the record rows.)
    [[nodiscard]] const double *result_row(row r) const
    {
        return m_buf.as<double>() + (m_dim + static_cast<std::size_t>(r)) * m_total;
    }
    // Host copies: everything ((dim + 6) * n_total words), or the records alone as typed vectors.
    [[nodiscard]] std::vector<double> all_to_host() const;
    [[nodiscard]] std::vector<double> times_hi() const;
    [[nodiscard]] std::vector<double> times_lo() const;
    [[nodiscard]] std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> propagate_res() const;
    [[nodiscard]] int device() const
    {
        return m_device;
    }
    [[nodiscard]] std::size_t dim() const
    {
        return m_dim;
    }
    [[nodiscard]] std::size_t n_total() const
    {
        return m_total;
    }
    [[nodiscard]] std::size_t offset(std::size_t i) const
    {
        return m_off.at(i);
    }
    [[nodiscard]] bool used_rccl() const
    {
        return m_rccl;
    }
    [[nodiscard]] std::vector<double> to_host() const;

private:
    friend ensemble_gathered detail_gather(const std::vector<detail::tab_core *> &, int);
    device_buffer m_buf;
    int m_device = 0;
    std::size_t m_dim = 0, m_total = 0;
    std::vector<std::size_t> m_off;
    bool m_rccl = false;
};
ensemble_gathered detail_gather(const std::vector<detail::tab_core *> &, int dst_device);

// Gather the states of a range of integrators (e.g. the first tuple elements of an ensemble_propagate_*_batch() result).
template <typename Range>
ensemble_gathered ensemble_gather_states(Range &tas, int dst_device = 0)
{
    std::vector<detail::tab_core *> cores;
    for (auto &x : tas) {
        if constexpr (requires { x.core(); }) {
            cores.push_back(&x.core());
        } else {
            cores.push_back(&std::get<0>(x).core());
        }
    }
    return detail_gather(cores, dst_device);
}

// Reference signatures: ensemble_propagate_{until,for,grid}_batch(ta, t | delta_t | grid, n_iter, gen, kw...)
// (include/heyoka/ensemble_propagate.hpp:222-271) with
// gen: taylor_adaptive_batch<T>(taylor_adaptive_batch<T>, std::size_t), invoked here serially on the
// calling thread (the reference may invoke it concurrently, src/ensemble_propagate.cpp:45-49).
// Accepted kwargs: max_steps, max_delta_t, callback, write_tc, c_output (not for grid), plus the MI355X
// extension kw::device = number of HIP devices to spread the iterations over (0 = all visible).
// Return values as in the reference: one (integrator, optional<continuous_output_batch>, callback)
// tuple per iteration, or (integrator, callback, grid output) for the grid variant.
//
// Without callback / continuous output every iteration is a single device-resident kernel: all the
// launches are issued asynchronously and synchronised once at the end. Otherwise the propagation is a
// host-driven lock-step loop, and the iterations are distributed over one host thread per device.
namespace detail
{

enum class ensemble_tmpl_kind { until, for_, grid };

template <typename F>
void ensemble_run_per_device(std::size_t n_iter, int n_dev, const F &f)
{
    const auto n_thr = static_cast<std::size_t>(std::max(1, n_dev));
    if (n_thr == 1u || n_iter == 1u) {
        for (std::size_t i = 0; i < n_iter; ++i) {
            f(i);
        }
        return;
    }
    std::vector<std::exception_ptr> errs(n_thr);
    std::vector<std::thread> thr;
    for (std::size_t d = 0; d < n_thr; ++d) {
        thr.emplace_back([&, d]() {
            try {
                for (std::size_t i = d; i < n_iter; i += n_thr) {
                    f(i);
                }
            } catch (...) {
                errs[d] = std::current_exception();
            }
        });
    }
    for (auto &t : thr) {
        t.join();
    }
    for (auto &e : errs) {
        if (e) {
            std::rethrow_exception(e);
        }
    }
}

template <ensemble_tmpl_kind Kind, typename TimeArg, typename Gen, typename... KwArgs>
auto ensemble_propagate_tmpl(const taylor_adaptive_batch<double> &ta, const TimeArg &t, std::size_t n_iter,
                             const Gen &gen, KwArgs &&...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    constexpr bool is_grid = (Kind == ensemble_tmpl_kind::grid);
    static_assert(!is_grid || (!kw::has_v<kw::c_output_tag, KwArgs...> && !kw::has_v<kw::write_tc_tag, KwArgs...>),
                  "kw::c_output and kw::write_tc are not accepted by ensemble_propagate_grid_batch()");
    // NOTE: zero iterations: an empty result (test/ensemble_propagate.cpp:90-99).
    const auto batch_size = ta.get_batch_size();
    const auto max_steps = static_cast<std::size_t>(kw::get(kw::max_steps, 0, kw_args...));
    std::vector<double> max_delta_ts;
    if constexpr (kw::has_v<kw::max_delta_t_tag, KwArgs...>) {
        using mdt_t = std::decay_t<decltype(kw::get(kw::max_delta_t, 0, kw_args...))>;
        if constexpr (std::is_arithmetic_v<mdt_t>) {
            max_delta_ts.assign(batch_size, static_cast<double>(kw::get(kw::max_delta_t, 0, kw_args...)));
        } else {
            for (const auto &x : kw::get(kw::max_delta_t, 0, kw_args...)) {
                max_delta_ts.push_back(static_cast<double>(x));
            }
        }
    }
    // NOTE: the callback is moved in when it arrives as an rvalue, then copied ONCE per iteration
    // (test/ensemble_propagate.cpp:380-405).
    step_callback_batch<double> cb;
    if constexpr (kw::has_v<kw::callback_tag, KwArgs...>) {
        cb = step_callback_batch<double>(kw::get(kw::callback, 0, std::forward<KwArgs>(kw_args)...));
    }
    const auto wtc = static_cast<bool>(kw::get(kw::write_tc, false, kw_args...));
    const auto c_out = static_cast<bool>(kw::get(kw::c_output, false, kw_args...));
    // kw::device = number of HIP devices to use (0: all the visible ones; one host thread per device). A NEGATIVE value
    // -k asks for k host threads spread round-robin over the visible devices - more workers than devices: the threaded
    // path on a single-GPU box, and a way to overlap expensive generators / callbacks.
    auto n_dev = static_cast<int>(kw::get(kw::device, 0, kw_args...));
    const auto visible = ensemble_visible_devices();
    int n_workers = 0;
    if (n_dev < 0) {
        n_workers = -n_dev;
        n_dev = visible;
    } else {
        if (n_dev == 0 || n_dev > visible) {
            n_dev = visible;
        }
        n_workers = n_dev;
    }

    // Generate the integrators (serially) and pin them to their devices.
    std::vector<taylor_adaptive_batch<double>> tas;
    tas.reserve(n_iter);
    for (std::size_t i = 0; i < n_iter; ++i) {
        tas.push_back(gen(ta, i));
        if (n_dev > 0) {
            const auto worker = i % static_cast<std::size_t>(std::max(1, n_workers));
            tas.back().core().set_device(static_cast<int>(worker % static_cast<std::size_t>(n_dev)));
        }
    }
    n_dev = n_workers; // (below: the number of worker threads)

    if constexpr (is_grid) {
        // Splat out the time grid (src/ensemble_propagate.cpp:266-273).
        std::vector<double> grid;
        grid.reserve(t.size() * batch_size);
        for (const auto gval : t) {
            for (std::uint32_t i = 0; i < batch_size; ++i) {
                grid.push_back(static_cast<double>(gval));
            }
        }
        std::vector<std::tuple<step_callback_batch<double>, std::vector<double>>> res(n_iter);
        ensemble_run_per_device(n_iter, n_dev, [&](std::size_t i) {
            res[i] = tas[i].propagate_grid(grid, kw::max_steps = max_steps, kw::max_delta_t = max_delta_ts,
                                           kw::callback = cb);
        });
        if constexpr (kw::has_v<kw::gather_tag, KwArgs...>) {
            *kw::get(kw::gather, 0, kw_args...) = ensemble_gather_states(tas, 0);
        }
        std::vector<std::tuple<taylor_adaptive_batch<double>, step_callback_batch<double>, std::vector<double>>> ret;
        ret.reserve(n_iter);
        for (std::size_t i = 0; i < n_iter; ++i) {
            ret.emplace_back(std::move(tas[i]), std::move(std::get<0>(res[i])), std::move(std::get<1>(res[i])));
        }
        return ret;
    } else {
        using res_t = std::tuple<std::optional<continuous_output_batch<double>>, step_callback_batch<double>>;
        std::vector<res_t> res(n_iter);
        const auto run = [&](std::size_t i) {
            if constexpr (Kind == ensemble_tmpl_kind::until) {
                res[i] = tas[i].propagate_until(static_cast<double>(t), kw::max_steps = max_steps,
                                                kw::max_delta_t = max_delta_ts, kw::callback = cb,
                                                kw::write_tc = wtc, kw::c_output = c_out);
            } else {
                res[i] = tas[i].propagate_for(static_cast<double>(t), kw::max_steps = max_steps,
                                              kw::max_delta_t = max_delta_ts, kw::callback = cb, kw::write_tc = wtc,
                                              kw::c_output = c_out);
            }
        };
        if (!cb && !c_out && n_dev <= 1) {
            // One device: the device-resident propagations of the iterations, one after the other.
            for (std::size_t i = 0; i < n_iter; ++i) {
                run(i);
            }
            for (auto &x : tas) {
                x.core().synchronize();
            }
        } else {
            ensemble_run_per_device(n_iter, n_dev, run);
        }
        if constexpr (kw::has_v<kw::gather_tag, KwArgs...>) {
            *kw::get(kw::gather, 0, kw_args...) = ensemble_gather_states(tas, 0);
        }
        std::vector<std::tuple<taylor_adaptive_batch<double>, std::optional<continuous_output_batch<double>>,
                               step_callback_batch<double>>>
            ret;
        ret.reserve(n_iter);
        for (std::size_t i = 0; i < n_iter; ++i) {
            ret.emplace_back(std::move(tas[i]), std::move(std::get<0>(res[i])), std::move(std::get<1>(res[i])));
        }
        return ret;
    }
}

} // namespace detail

// (The reference's call sites name the value type explicitly: ensemble_propagate_until_batch<double>(...),
// include/heyoka/ensemble_propagate.hpp:222-271.)
template <typename T = double, typename Gen, typename... KwArgs>
    requires std::is_same_v<T, double>
auto ensemble_propagate_until_batch(const taylor_adaptive_batch<double> &ta, double t, std::size_t n_iter,
                                    const Gen &gen, KwArgs &&...kw_args)
{
    return detail::ensemble_propagate_tmpl<detail::ensemble_tmpl_kind::until>(ta, t, n_iter, gen, std::forward<KwArgs>(kw_args)...);
}

template <typename T = double, typename Gen, typename... KwArgs>
    requires std::is_same_v<T, double>
auto ensemble_propagate_for_batch(const taylor_adaptive_batch<double> &ta, double delta_t, std::size_t n_iter,
                                  const Gen &gen, KwArgs &&...kw_args)
{
    return detail::ensemble_propagate_tmpl<detail::ensemble_tmpl_kind::for_>(ta, delta_t, n_iter, gen, std::forward<KwArgs>(kw_args)...);
}

template <typename T = double, typename Gen, typename... KwArgs>
    requires std::is_same_v<T, double>
auto ensemble_propagate_grid_batch(const taylor_adaptive_batch<double> &ta, const std::vector<double> &grid,
                                   std::size_t n_iter, const Gen &gen, KwArgs &&...kw_args)
{
    return detail::ensemble_propagate_tmpl<detail::ensemble_tmpl_kind::grid>(ta, grid, n_iter, gen, std::forward<KwArgs>(kw_args)...);
}

} // namespace heyoka_amd
