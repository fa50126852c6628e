// Ensemble driver. See ensemble.hpp.
#include "ensemble.hpp"

#include "hip_backend.hpp"

namespace heyoka_amd::detail
{

int ensemble_visible_devices()
{
    return hip_device_count();
}

std::vector<tab_core> ensemble_propagate_core(const tab_core &ta, double t, std::size_t n_iter,
                                              const std::function<void(tab_core &, std::size_t)> &gen,
                                              std::size_t max_steps, int n_devices, ensemble_kind kind)
{
    if (n_iter == 0u) {
        // Reference: an empty result (test/ensemble_propagate.cpp:90-99).
        return {};
    }

    const auto visible = hip_device_count();
    if (n_devices <= 0 || n_devices > visible) {
        n_devices = visible;
    }
    if (n_devices <= 0) {
        throw std::runtime_error("heyoka_amd: no HIP device is available for ensemble propagation");
    }

    std::vector<tab_core> ret;
    ret.reserve(n_iter);
    for (std::size_t i = 0; i < n_iter; ++i) {
        ret.emplace_back(ta);
        gen(ret.back(), i);
        ret.back().set_device(static_cast<int>(i % static_cast<std::size_t>(n_devices)));
    }

    // Asynchronous launches: one device-resident propagation per iteration.
    const std::vector<double> ts{t};
    for (auto &c : ret) {
        if (kind == ensemble_kind::until) {
            c.propagate_until(ts, max_steps, {}, {}, false, false);
        } else {
            c.propagate_for(ts, max_steps, {}, {}, false, false);
        }
    }
    for (auto &c : ret) {
        c.synchronize();
    }
    return ret;
}

} // namespace heyoka_amd::detail
