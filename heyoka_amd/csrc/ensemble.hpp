// ensemble_propagate_*_batch(): n_iter independent propagations of copies of a batch integrator
// (reference: include/heyoka/ensemble_propagate.hpp:222-271, src/ensemble_propagate.cpp:193-297).
// The TBB parallel_for of the reference becomes: iteration i runs on HIP device i % n_devices, all
// launches are asynchronous (one device-resident kernel per iteration), one final synchronisation.
// Multi-process / multi-node sharding with an RCCL gather of the final states lives in the Python
// layer (heyoka_amd/ensemble.py), one process per GPU.
#pragma once

#include <cstddef>
#include <functional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

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

// Reference signature: ensemble_propagate_until_batch(ta, t, n_iter, gen, kw...) with
// gen: taylor_adaptive_batch<T>(taylor_adaptive_batch<T>, std::size_t), invoked here serially on the
// calling thread (the reference may invoke it concurrently, src/ensemble_propagate.cpp:45-49).
// Returns one (integrator, callback) tuple per iteration; the continuous-output slot of the
// reference's return type is not available (see kw::c_output). kw::device selects the number of HIP
// devices to spread the iterations over (0 = all visible).
namespace detail
{

template <bool Until, typename Gen, typename... KwArgs>
std::vector<std::tuple<taylor_adaptive_batch<double>, step_callback_batch<double>>>
ensemble_propagate_tmpl(const taylor_adaptive_batch<double> &ta, double t, std::size_t n_iter, const Gen &gen,
                        const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    if (n_iter == 0u) {
        throw std::invalid_argument(std::string("Cannot perform an ensemble propagate_") + (Until ? "until" : "for")
                                    + "() if the number of iterations is zero");
    }
    const auto max_steps = static_cast<std::size_t>(kw::get(kw::max_steps, 0, kw_args...));
    auto n_dev = static_cast<int>(kw::get(kw::device, 0, kw_args...));
    const auto visible = ensemble_visible_devices();
    if (n_dev <= 0 || n_dev > visible) {
        n_dev = visible;
    }

    std::vector<std::tuple<taylor_adaptive_batch<double>, step_callback_batch<double>>> ret;
    ret.reserve(n_iter);
    for (std::size_t i = 0; i < n_iter; ++i) {
        ret.emplace_back(gen(ta, i), step_callback_batch<double>{});
        if (n_dev > 0) {
            std::get<0>(ret.back()).core().set_device(static_cast<int>(i % static_cast<std::size_t>(n_dev)));
        }
    }
    // Asynchronous launches (one device-resident propagation per iteration), then one sync each.
    for (auto &r : ret) {
        if constexpr (Until) {
            std::get<0>(r).propagate_until(t, kw::max_steps = max_steps);
        } else {
            std::get<0>(r).propagate_for(t, kw::max_steps = max_steps);
        }
    }
    for (auto &r : ret) {
        std::get<0>(r).core().synchronize();
    }
    return ret;
}

} // namespace detail

template <typename Gen, typename... KwArgs>
auto ensemble_propagate_until_batch(const taylor_adaptive_batch<double> &ta, double t, std::size_t n_iter,
                                    const Gen &gen, const KwArgs &...kw_args)
{
    return detail::ensemble_propagate_tmpl<true>(ta, t, n_iter, gen, kw_args...);
}

template <typename Gen, typename... KwArgs>
auto ensemble_propagate_for_batch(const taylor_adaptive_batch<double> &ta, double delta_t, std::size_t n_iter,
                                  const Gen &gen, const KwArgs &...kw_args)
{
    return detail::ensemble_propagate_tmpl<false>(ta, delta_t, n_iter, gen, kw_args...);
}

} // namespace heyoka_amd
