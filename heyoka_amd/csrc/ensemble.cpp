// Ensemble driver. See ensemble.hpp.
#include "ensemble.hpp"

#include <exception>
#include <optional>
#include <thread>

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

    // n_devices: number of HIP devices (0: all the visible ones), one host thread per device; a NEGATIVE value -k asks
    // for k host threads spread round-robin over the visible devices (more workers than devices: how the threaded path
    // runs on a single-GPU box; also useful when the generator is expensive).
    const auto visible = hip_device_count();
    if (visible <= 0) {
        throw std::runtime_error("heyoka_amd: no HIP device is available for ensemble propagation");
    }
    int n_workers = 0;
    if (n_devices < 0) {
        n_workers = -n_devices;
    } else {
        n_workers = (n_devices == 0 || n_devices > visible) ? visible : n_devices;
    }
    const auto device_of = [visible](int worker) { return worker % visible; };
    n_devices = n_workers;

    // One host thread per device (the reference runs the iterations inside a TBB parallel_for,
    // src/ensemble_propagate.cpp:203-219, and documents that the generator is invoked concurrently): thread d copies
    // the template integrator, runs the generator, uploads and launches the device-resident propagation of the
    // iterations i = d, d + n_devices, ... on device d, so that neither the generator nor the uploads of one device
    // wait for another device. Results keep the iteration order.
    // The copy constructor brings the (mutable, lazily synchronised) host mirrors of its source up to date before it
    // copies them. The worker threads below all copy `ta` concurrently: one throwaway copy on the calling thread
    // leaves nothing for them to synchronise, so that their copies only READ the template.
    if (n_devices > 1) {
        const tab_core synced(ta);
        (void)synced;
    }
    std::vector<std::optional<tab_core>> slots(n_iter);
    std::vector<std::exception_ptr> errors(static_cast<std::size_t>(n_devices));
    const std::vector<double> ts{t};
    const auto worker = [&](int dev) {
        try {
            for (std::size_t i = static_cast<std::size_t>(dev); i < n_iter; i += static_cast<std::size_t>(n_devices)) {
                slots[i].emplace(ta);
                gen(*slots[i], i);
                slots[i]->set_device(device_of(dev));
                // Asynchronous launch: one device-resident propagation per iteration.
                if (kind == ensemble_kind::until) {
                    slots[i]->propagate_until(ts, max_steps, {}, {}, false, false);
                } else {
                    slots[i]->propagate_for(ts, max_steps, {}, {}, false, false);
                }
            }
            for (std::size_t i = static_cast<std::size_t>(dev); i < n_iter; i += static_cast<std::size_t>(n_devices)) {
                slots[i]->synchronize();
            }
        } catch (...) {
            errors[static_cast<std::size_t>(dev)] = std::current_exception();
        }
    };
    if (n_devices == 1) {
        worker(0);
    } else {
        std::vector<std::thread> threads;
        threads.reserve(static_cast<std::size_t>(n_devices));
        for (int d = 0; d < n_devices; ++d) {
            threads.emplace_back(worker, d);
        }
        for (auto &th : threads) {
            th.join();
        }
    }
    for (const auto &ep : errors) {
        if (ep) {
            std::rethrow_exception(ep);
        }
    }
    std::vector<tab_core> ret;
    ret.reserve(n_iter);
    for (auto &sl : slots) {
        ret.emplace_back(std::move(*sl));
    }
    return ret;
}

} // namespace heyoka_amd::detail
