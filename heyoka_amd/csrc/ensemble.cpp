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

// ---------------------------------------------------------------------------------------------------------------------
// Gather of the final states (see ensemble.hpp).
// ---------------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <string>

namespace heyoka_amd
{

namespace
{

void hip_ok(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        throw std::runtime_error(std::string("heyoka_amd: ") + what + " failed: " + hipGetErrorString(e));
    }
}

// The handful of RCCL entry points used, resolved at run time (the library is not linked: the single-GPU path has no use
// for it, and the C ABI must load on machines without it).
struct rccl_api {
    using comm_t = void *;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, std::size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, std::size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

const rccl_api &rccl()
{
    static const rccl_api api = [] {
        rccl_api a;
        void *h = nullptr;
        for (const auto *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h != nullptr) {
                break;
            }
        }
        if (h == nullptr) {
            return a;
        }
        const auto sym = [h](const char *n) { return dlsym(h, n); };
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.CommInitAll != nullptr && a.CommDestroy != nullptr && a.GroupStart != nullptr && a.GroupEnd != nullptr
               && a.Send != nullptr && a.Recv != nullptr;
        return a;
    }();
    return api;
}

void nccl_ok(int rc, const char *what)
{
    if (rc != 0) {
        const auto &a = rccl();
        throw std::runtime_error(std::string("heyoka_amd: RCCL ") + what + " failed: "
                                 + (a.GetErrorString != nullptr ? std::string(a.GetErrorString(rc)) : std::to_string(rc)));
    }
}

// The device which owns a buffer must be the one the plan says: a block packed on the wrong device (or a stale pointer)
// would otherwise be read over xGMI from somewhere else - or fault inside a collective, where nothing names the culprit.
void assert_owner(const void *ptr, int device, const char *what)
{
    if (ptr == nullptr) {
        return;
    }
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        throw std::runtime_error(std::string("heyoka_amd: the gather cannot identify the owner of the ") + what);
    }
    if (attr.device != device) {
        throw std::runtime_error(std::string("heyoka_amd: the ") + what + " lives on device " + std::to_string(attr.device)
                                 + ", expected device " + std::to_string(device));
    }
}

// A 2D copy (dim rows of n doubles) into the columns [off, off + n) of the result.
void place_block(double *dst, std::size_t n_total, std::size_t off, const double *src, std::size_t n, std::size_t dim,
                 hipStream_t stream)
{
    hip_ok(hipMemcpy2DAsync(dst + off, n_total * sizeof(double), src, n * sizeof(double), n * sizeof(double), dim,
                            hipMemcpyDefault, stream),
           "hipMemcpy2DAsync (gather)");
}

} // namespace

namespace
{

// RCCL communicators are expensive to create (hundreds of ms on 8 GPUs): one set per list of devices, kept for the
// lifetime of the process.
struct comm_cache {
    std::mutex mtx;
    // Held for the whole group of sends / receives of a gather: concurrent gathers would otherwise interleave their
    // operations on the same (cached) communicators.
    std::mutex use_mtx;
    std::map<std::vector<int>, std::vector<rccl_api::comm_t>> comms;
};
comm_cache &comm_store()
{
    static comm_cache c;
    return c;
}
std::vector<rccl_api::comm_t> comms_for(const std::vector<int> &devs)
{
    auto &c = comm_store();
    std::lock_guard<std::mutex> lock(c.mtx);
    auto it = c.comms.find(devs);
    if (it == c.comms.end()) {
        std::vector<rccl_api::comm_t> v(devs.size(), nullptr);
        nccl_ok(rccl().CommInitAll(v.data(), static_cast<int>(devs.size()), devs.data()), "ncclCommInitAll");
        it = c.comms.emplace(devs, std::move(v)).first;
    }
    return it->second;
}

// (The caller's current device is put back on every way out.)
struct device_restore {
    int dev = 0;
    bool ok = false;
    device_restore()
    {
        ok = hipGetDevice(&dev) == hipSuccess;
    }
    ~device_restore()
    {
        if (ok) {
            (void)hipSetDevice(dev);
        }
    }
};

} // namespace

ensemble_gathered detail_gather(const std::vector<detail::tab_core *> &tabs, int dst_device)
{
    ensemble_gathered g;
    g.m_device = dst_device;
    if (tabs.empty()) {
        return g;
    }
    const device_restore restore;
    g.m_dim = tabs[0]->get_dim();
    const auto n_rows = g.m_dim + detail::tab_core::n_result_rows;
    for (auto *t : tabs) {
        if (t->get_dim() != g.m_dim) {
            throw std::invalid_argument("Cannot gather the states of integrators of different dimensions");
        }
        g.m_off.push_back(g.m_total);
        g.m_total += t->get_batch_size();
    }
    // 1. Every integrator packs its state, times and propagation records into one block on ITS device (copies on its own
    // stream, uploads of host-side records included); only then are the integrators synchronised - the transfers below
    // run on other streams and must not start before the blocks are complete.
    std::vector<device_buffer> packs;
    for (auto *t : tabs) {
        packs.emplace_back(n_rows * t->get_batch_size() * sizeof(double), t->get_device());
        t->pack_results(packs.back().as<double>());
    }
    for (auto *t : tabs) {
        t->synchronize();
    }
    g.m_buf = device_buffer(n_rows * g.m_total * sizeof(double), dst_device);
    auto *const out = g.m_buf.as<double>();
    // (First contact with a multi-GPU node: every buffer is checked against the device it is supposed to live on before
    // anything is sent or copied between devices.)
    assert_owner(out, dst_device, "gathered result");
    for (std::size_t i = 0; i < tabs.size(); ++i) {
        assert_owner(packs[i].get(), tabs[i]->get_device(), "packed results of an integrator");
        assert_owner(tabs[i]->device_state(), tabs[i]->get_device(), "state of an integrator");
    }

    // Which devices take part (the destination is rank 0).
    std::vector<int> devs{dst_device};
    for (auto *t : tabs) {
        if (std::find(devs.begin(), devs.end(), t->get_device()) == devs.end()) {
            devs.push_back(t->get_device());
        }
    }
    const char *force = std::getenv("HEYOKA_AMD_GATHER_RCCL");
    const bool want_rccl = force != nullptr ? std::atoi(force) != 0 : devs.size() > 1u;
    bool done = false;
    if (want_rccl && rccl().ok) {
        const auto &a = rccl();
        const std::lock_guard<std::mutex> comm_use(comm_store().use_mtx);
        const auto comms = comms_for(devs);
        std::vector<hipStream_t> streams(devs.size(), nullptr);
        std::vector<device_buffer> staging; // contiguous landing blocks on the destination, one per integrator
        const auto cleanup = [&]() {
            for (std::size_t d = 0; d < devs.size(); ++d) {
                (void)hipSetDevice(devs[d]);
                if (streams[d] != nullptr) {
                    (void)hipStreamDestroy(streams[d]);
                }
            }
        };
        try {
            for (std::size_t d = 0; d < devs.size(); ++d) {
                hip_ok(hipSetDevice(devs[d]), "hipSetDevice");
                hip_ok(hipStreamCreateWithFlags(&streams[d], hipStreamNonBlocking), "hipStreamCreate");
            }
            const auto rank_of = [&](int dev) {
                return static_cast<std::size_t>(std::find(devs.begin(), devs.end(), dev) - devs.begin());
            };
            for (auto *t : tabs) {
                staging.emplace_back(n_rows * t->get_batch_size() * sizeof(double), dst_device);
                assert_owner(staging.back().get(), dst_device, "staging block of the gather");
            }
            constexpr int nccl_f64 = 8; // ncclFloat64 (ncclDataType_t, nccl.h)
            nccl_ok(a.GroupStart(), "ncclGroupStart");
            for (std::size_t i = 0; i < tabs.size(); ++i) {
                const auto cnt = n_rows * tabs[i]->get_batch_size();
                const auto r = rank_of(tabs[i]->get_device());
                nccl_ok(a.Send(packs[i].get(), cnt, nccl_f64, 0, comms[r], streams[r]), "ncclSend");
                nccl_ok(a.Recv(staging[i].get(), cnt, nccl_f64, static_cast<int>(r), comms[0], streams[0]), "ncclRecv");
            }
            nccl_ok(a.GroupEnd(), "ncclGroupEnd");
            // (The placements follow the receives on the destination's stream.)
            hip_ok(hipSetDevice(dst_device), "hipSetDevice");
            for (std::size_t i = 0; i < tabs.size(); ++i) {
                place_block(out, g.m_total, g.m_off[i], staging[i].as<double>(), tabs[i]->get_batch_size(), n_rows, streams[0]);
            }
            for (std::size_t d = 0; d < devs.size(); ++d) {
                hip_ok(hipSetDevice(devs[d]), "hipSetDevice");
                hip_ok(hipStreamSynchronize(streams[d]), "hipStreamSynchronize");
            }
            done = true;
            g.m_rccl = true;
        } catch (...) {
            cleanup();
            throw;
        }
        cleanup();
    }
    if (!done) {
        // Device-to-device copies (unified addressing: the runtime routes peer copies over xGMI), all of them on ONE stream
        // of the destination device, which is then synchronised.
        hip_ok(hipSetDevice(dst_device), "hipSetDevice");
        hipStream_t st = nullptr;
        hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        try {
            for (std::size_t i = 0; i < tabs.size(); ++i) {
                hip_ok(hipSetDevice(dst_device), "hipSetDevice");
                place_block(out, g.m_total, g.m_off[i], packs[i].as<double>(), tabs[i]->get_batch_size(), n_rows, st);
            }
            hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
        } catch (...) {
            (void)hipStreamDestroy(st);
            throw;
        }
        (void)hipStreamDestroy(st);
    }
    return g;
}

std::vector<double> ensemble_gathered::all_to_host() const
{
    std::vector<double> ret(n_rows() * m_total);
    if (!ret.empty()) {
        m_buf.download(ret.data(), ret.size() * sizeof(double), nullptr);
    }
    return ret;
}

namespace
{

std::vector<double> row_to_host(const device_buffer &buf, std::size_t first_row, std::size_t n_rows_, std::size_t n_total)
{
    std::vector<double> ret(n_rows_ * n_total);
    if (!ret.empty()) {
        hip_ok(hipMemcpy(ret.data(), buf.as<double>() + first_row * n_total, ret.size() * sizeof(double), hipMemcpyDeviceToHost),
               "hipMemcpy (gathered records)");
    }
    return ret;
}

} // namespace

std::vector<double> ensemble_gathered::times_hi() const
{
    return row_to_host(m_buf, m_dim + time_hi, 1, m_total);
}

std::vector<double> ensemble_gathered::times_lo() const
{
    return row_to_host(m_buf, m_dim + time_lo, 1, m_total);
}

std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> ensemble_gathered::propagate_res() const
{
    const auto raw = row_to_host(m_buf, m_dim + outcome, 4, m_total);
    std::vector<std::tuple<taylor_outcome, double, double, std::size_t>> ret(m_total);
    for (std::size_t i = 0; i < m_total; ++i) {
        long long oc = 0;
        unsigned long long ns = 0;
        std::memcpy(&oc, &raw[i], sizeof(oc));
        std::memcpy(&ns, &raw[m_total + i], sizeof(ns));
        ret[i] = std::tuple{static_cast<taylor_outcome>(oc), raw[2u * m_total + i], raw[3u * m_total + i], static_cast<std::size_t>(ns)};
    }
    return ret;
}

std::vector<double> ensemble_gathered::to_host() const
{
    std::vector<double> ret(m_dim * m_total);
    if (!ret.empty()) {
        m_buf.download(ret.data(), ret.size() * sizeof(double), nullptr);
    }
    return ret;
}

} // namespace heyoka_amd
