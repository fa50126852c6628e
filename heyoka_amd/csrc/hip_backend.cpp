// hiprtc + HIP module runtime. See hip_backend.hpp.
#include "hip_backend.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>

namespace heyoka_amd
{

namespace
{

void hip_check(hipError_t err, const char *what)
{
    if (err != hipSuccess) {
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(err));
    }
}

void rtc_check(hiprtcResult res, const char *what, const std::string &log = {})
{
    if (res != HIPRTC_SUCCESS) {
        throw std::runtime_error(std::string("hiprtc error in ") + what + ": " + hiprtcGetErrorString(res)
                                 + (log.empty() ? std::string{} : "\n" + log));
    }
}

std::mutex cache_mutex;
std::unordered_map<std::string, std::shared_ptr<const compiled_module>> module_cache;

} // namespace

std::shared_ptr<const compiled_module> hiprtc_compile(const emitted_module &m)
{
    // NOTE: the extra compiler flags (HEYOKA_AMD_HIPRTC_FLAGS, e.g. "-ffp-contract=off" in the parity tests) are part
    // of the cache key.
    const std::string cache_key = [&]() {
        const char *ex = std::getenv("HEYOKA_AMD_HIPRTC_FLAGS");
        return (ex != nullptr ? std::string(ex) + "\n" : std::string{}) + m.compile_flags + "\n" + m.source;
    }();
    {
        std::lock_guard lock(cache_mutex);
        if (const auto it = module_cache.find(cache_key); it != module_cache.end()) {
            return it->second;
        }
    }

    const auto t0 = std::chrono::steady_clock::now();

    hiprtcProgram prog = nullptr;
    rtc_check(hiprtcCreateProgram(&prog, m.source.c_str(), "heyoka_amd_taylor.hip", 0, nullptr, nullptr),
              "hiprtcCreateProgram");

    // NOTE: contraction into FMA is allowed (the reference enables the 'contract' fast-math flag,
    // src/llvm_state.cpp:843-845); nothing else is relaxed.
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=fast", "-std=c++17"};
    const char *extra = std::getenv("HEYOKA_AMD_HIPRTC_FLAGS");
    std::vector<std::string> extra_store;
    if (extra != nullptr || !m.compile_flags.empty()) {
        // (Module flags first: the environment can override them.)
        std::string s = m.compile_flags + " " + (extra != nullptr ? extra : "");
        std::size_t pos = 0;
        while (pos < s.size()) {
            const auto next = s.find(' ', pos);
            const auto tok = s.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
            if (!tok.empty()) {
                extra_store.push_back(tok);
            }
            if (next == std::string::npos) {
                break;
            }
            pos = next + 1u;
        }
        for (const auto &t : extra_store) {
            opts.push_back(t.c_str());
        }
    }

    const auto cres = hiprtcCompileProgram(prog, static_cast<int>(opts.size()), opts.data());

    std::string log;
    std::size_t log_size = 0;
    if (hiprtcGetProgramLogSize(prog, &log_size) == HIPRTC_SUCCESS && log_size > 1u) {
        log.resize(log_size);
        hiprtcGetProgramLog(prog, log.data());
    }

    if (cres != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        rtc_check(cres, "hiprtcCompileProgram", log);
    }

    std::size_t code_size = 0;
    rtc_check(hiprtcGetCodeSize(prog, &code_size), "hiprtcGetCodeSize");
    auto ret = std::make_shared<compiled_module>();
    ret->code.resize(code_size);
    rtc_check(hiprtcGetCode(prog, ret->code.data()), "hiprtcGetCode");
    hiprtcDestroyProgram(&prog);

    ret->meta = m;
    ret->log = std::move(log);
    ret->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::lock_guard lock(cache_mutex);
    // (Bounded: beyond 1024 entries the compiled modules which only the cache still refers to are dropped.)
    if (module_cache.size() >= 1024u) {
        for (auto it = module_cache.begin(); it != module_cache.end();) {
            it = it->second.use_count() == 1 ? module_cache.erase(it) : std::next(it);
        }
    }
    module_cache.emplace(cache_key, ret);
    return ret;
}

int hip_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

namespace
{

// Loaded code objects, per (device, compiled module), reference counted: integrators built from the same system share
// the hipModule_t (construction does not reload it). A module nobody uses any more is NOT unloaded at once: with this
// toolchain (ROCm 7.x) a process which loads and unloads many run-time modules and then runs the first kernels of
// another HIP client (PyTorch's lazily-loaded kernels) intermittently took a GPU memory fault inside that client's
// kernel. Idle modules stay resident as a cache - a later integrator of the same system gets them back without a reload -
// up to a bound (256 by default; HEYOKA_AMD_KEEP_MODULES = number, or "all" for the unbounded behaviour of the earlier
// rounds): beyond it the longest-idle one is unloaded, so that a long-lived process which builds many distinct
// integrators holds a bounded number of code objects (a few tens of KB each, plus their LDS / scratch descriptors).
struct loaded_entry {
    hipModule_t mod = nullptr;
    std::shared_ptr<const compiled_module> cm;
    std::uint64_t users = 0, idle_since = 0;
};
std::mutex loaded_mutex;
std::map<std::pair<int, const compiled_module *>, loaded_entry> loaded_modules;
std::uint64_t loaded_tick = 0;

std::size_t idle_module_bound()
{
    static const std::size_t bound = [] {
        const char *ev = std::getenv("HEYOKA_AMD_KEEP_MODULES");
        if (ev == nullptr) {
            return std::size_t(256);
        }
        if (std::string(ev) == "all") {
            return std::numeric_limits<std::size_t>::max();
        }
        return static_cast<std::size_t>(std::max(0, std::atoi(ev)));
    }();
    return bound;
}

hipModule_t load_module_cached(const std::shared_ptr<const compiled_module> &cm, int device)
{
    std::lock_guard lock(loaded_mutex);
    const auto key = std::make_pair(device, cm.get());
    if (const auto it = loaded_modules.find(key); it != loaded_modules.end()) {
        ++it->second.users;
        return it->second.mod;
    }
    hip_check(hipSetDevice(device), "hipSetDevice");
    hipModule_t mod = nullptr;
    hip_check(hipModuleLoadData(&mod, cm->code.data()), "hipModuleLoadData");
    loaded_modules.emplace(key, loaded_entry{mod, cm, 1, 0});
    return mod;
}

// One user less; unloads the longest-idle modules beyond the bound. Never throws (called from destructors, possibly
// while the process winds down).
void release_module(const std::shared_ptr<const compiled_module> &cm, int device) noexcept
{
    try {
        std::lock_guard lock(loaded_mutex);
        const auto it = loaded_modules.find(std::make_pair(device, cm.get()));
        if (it == loaded_modules.end() || it->second.users == 0u) {
            return;
        }
        if (--it->second.users != 0u) {
            return;
        }
        it->second.idle_since = ++loaded_tick;
        std::size_t n_idle = 0;
        for (const auto &[k, e] : loaded_modules) {
            (void)k;
            n_idle += e.users == 0u ? 1u : 0u;
        }
        while (n_idle > idle_module_bound()) {
            auto victim = loaded_modules.end();
            for (auto jt = loaded_modules.begin(); jt != loaded_modules.end(); ++jt) {
                if (jt->second.users == 0u && (victim == loaded_modules.end() || jt->second.idle_since < victim->second.idle_since)) {
                    victim = jt;
                }
            }
            if (victim == loaded_modules.end()) {
                break;
            }
            if (hipSetDevice(victim->first.first) == hipSuccess) {
                // (Kernels of the module may still be in flight on a stream of its last owner: drain the device first.)
                (void)hipDeviceSynchronize();
                (void)hipModuleUnload(victim->second.mod);
            }
            loaded_modules.erase(victim);
            --n_idle;
        }
    } catch (...) {
    }
}

} // namespace

struct device_module::impl {
    std::shared_ptr<const compiled_module> cm;
    int device = 0;
    hipModule_t mod = nullptr;
    hipFunction_t fn_taylor = nullptr;
    hipFunction_t fn_dout = nullptr;
    hipFunction_t fn_taylor_tc = nullptr; // optional variant writing the Taylor coefficients
    hipStream_t stream = nullptr;
    // Ring of HIP event pairs bracketing the stepper launches.
    static constexpr int n_ev = 64;
    hipEvent_t ev_start[n_ev] = {}, ev_stop[n_ev] = {};
    std::uint64_t n_launches = 0;
    // Persistent (cluster mode) launches: resident grid size and jet scratch.
    unsigned max_grid = 0;
    void *scratch = nullptr;
    std::size_t scratch_bytes = 0;
};

device_module::device_module(std::shared_ptr<const compiled_module> cm, int device) : m_impl(std::make_unique<impl>())
{
    if (hip_device_count() <= device) {
        throw std::runtime_error("heyoka_amd: no HIP device " + std::to_string(device)
                                 + " is available (the MI355X/gfx950 code path has no CPU fallback)");
    }
    m_impl->cm = std::move(cm);
    m_impl->device = device;
    hip_check(hipSetDevice(device), "hipSetDevice");
    m_impl->mod = load_module_cached(m_impl->cm, device);
    // (If anything below throws the destructor does not run: give the user count of the module back.)
    struct load_guard {
        impl *p;
        ~load_guard()
        {
            if (p != nullptr) {
                for (int i = 0; i < impl::n_ev; ++i) {
                    if (p->ev_start[i] != nullptr) (void)hipEventDestroy(p->ev_start[i]);
                    if (p->ev_stop[i] != nullptr) (void)hipEventDestroy(p->ev_stop[i]);
                }
                release_module(p->cm, p->device);
                p->mod = nullptr;
            }
        }
    } guard{m_impl.get()};
    hip_check(hipModuleGetFunction(&m_impl->fn_taylor, m_impl->mod, m_impl->cm->meta.kernel_name.c_str()),
              "hipModuleGetFunction(taylor)");
    hip_check(hipModuleGetFunction(&m_impl->fn_dout, m_impl->mod, m_impl->cm->meta.dout_name.c_str()),
              "hipModuleGetFunction(dout)");
    if (!m_impl->cm->meta.tc_kernel_name.empty()) {
        hip_check(hipModuleGetFunction(&m_impl->fn_taylor_tc, m_impl->mod, m_impl->cm->meta.tc_kernel_name.c_str()),
                  "hipModuleGetFunction(taylor_tc)");
    }
    if (m_impl->cm->meta.persistent) {
        int n_cu = 0, per_cu = 0;
        hip_check(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device),
                  "hipDeviceGetAttribute");
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, m_impl->fn_taylor,
                                                               static_cast<int>(m_impl->cm->meta.block_size),
                                                               m_impl->cm->meta.lds_bytes)
            != hipSuccess) {
            // The occupancy query is advisory (and known to be unreliable for register-heavy kernels): the
            // cluster kernels use the whole register file, i.e. one 256-thread block per CU.
            (void)hipGetLastError();
            per_cu = 1;
        }
        m_impl->max_grid = static_cast<unsigned>(std::max(1, n_cu) * std::max(1, per_cu));
    }
    for (int i = 0; i < impl::n_ev; ++i) {
        hip_check(hipEventCreate(&m_impl->ev_start[i]), "hipEventCreate");
        hip_check(hipEventCreate(&m_impl->ev_stop[i]), "hipEventCreate");
    }
    guard.p = nullptr;
}

device_module::~device_module()
{
    if (m_impl && m_impl->mod != nullptr) {
        for (int i = 0; i < impl::n_ev; ++i) {
            (void)hipEventDestroy(m_impl->ev_start[i]);
            (void)hipEventDestroy(m_impl->ev_stop[i]);
        }
        if (m_impl->scratch != nullptr) {
            (void)hipFree(m_impl->scratch);
        }
        release_module(m_impl->cm, m_impl->device);
    }
}

int device_module::device() const
{
    return m_impl->device;
}

void device_module::set_stream(void *s)
{
    m_impl->stream = static_cast<hipStream_t>(s);
}

void *device_module::stream() const
{
    return m_impl->stream;
}

std::size_t device_module::tape_bytes(std::uint64_t n_systems) const
{
    const auto &meta = m_impl->cm->meta;
    if (!meta.persistent || meta.scratch_per_wave == 0u || n_systems == 0u) {
        return 0;
    }
    const auto bs = static_cast<std::uint64_t>(meta.block_size);
    const auto grid = std::min<std::uint64_t>((n_systems * meta.lanes_per_system + bs - 1u) / bs, m_impl->max_grid);
    return static_cast<std::size_t>(grid) * (bs / 64u) * meta.scratch_per_wave * sizeof(double);
}

void device_module::launch_taylor(const hy_kargs &args, void *user_tape)
{
    if (args.N == 0u) {
        return;
    }
    hip_check(hipSetDevice(m_impl->device), "hipSetDevice");
    const auto &meta = m_impl->cm->meta;
    const std::uint64_t threads = args.N * meta.lanes_per_system;
    const auto bs = static_cast<std::uint64_t>(meta.block_size);
    auto grid = (threads + bs - 1u) / bs;
    if (grid > 0x7fffffffull) {
        throw std::overflow_error("heyoka_amd: grid size overflow");
    }

    hy_kargs a = args;
    if (meta.persistent) {
        // Persistent blocks pulling work from a device-side queue: the grid covers the machine once,
        // and the jet scratch is sized by the number of resident waves.
        grid = std::min<std::uint64_t>(grid, m_impl->max_grid);
        if (meta.scratch_per_wave != 0u && user_tape != nullptr) {
            // (The caller's tape: tape_bytes() of this many systems, the stepper function-pointer ABI with the tape argument.)
            a.scratch = static_cast<double *>(user_tape);
        } else if (meta.scratch_per_wave != 0u) {
            // Bound the scratch (table mode keeps the whole tape of every resident thread in HBM):
            // 48 GiB by default out of the 288 GB of an MI355X.
            double budget_gib = 48.;
            {
                // Default: 60 % of the free device memory (288 GB of HBM3E on an MI355X).
                std::size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b != 0u) {
                    budget_gib = 0.6 * static_cast<double>(free_b + m_impl->scratch_bytes) / 1073741824.;
                }
            }
            if (const char *env = std::getenv("HEYOKA_AMD_SCRATCH_GIB")) {
                budget_gib = std::max(0.25, std::atof(env));
            }
            const auto per_block = static_cast<double>(bs / 64u) * static_cast<double>(meta.scratch_per_wave) * 8.;
            const auto max_blocks = static_cast<std::uint64_t>(budget_gib * 1073741824. / per_block);
            grid = std::max<std::uint64_t>(1u, std::min<std::uint64_t>(grid, max_blocks));
        }
        const auto need = static_cast<std::size_t>(grid) * (bs / 64u) * meta.scratch_per_wave * sizeof(double);
        if (user_tape == nullptr || meta.scratch_per_wave == 0u) {
            if (need > m_impl->scratch_bytes && need != 0u) {
                if (m_impl->scratch != nullptr) {
                    hip_check(hipFree(m_impl->scratch), "hipFree");
                    m_impl->scratch = nullptr;
                }
                hip_check(hipMalloc(&m_impl->scratch, need), "hipMalloc(scratch)");
                m_impl->scratch_bytes = need;
            }
            a.scratch = static_cast<double *>(m_impl->scratch);
        }
    }
    std::size_t sz = sizeof(a);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    // HIP events on the launch stream bracket exactly the kernel (used for the roofline figure).
    const auto slot = static_cast<int>(m_impl->n_launches % impl::n_ev);
    hip_check(hipEventRecord(m_impl->ev_start[slot], m_impl->stream), "hipEventRecord");
    // Kernels with register-resident jets come with a second variant used when the caller wants the TCs.
    const auto fn = (a.tc != nullptr && m_impl->fn_taylor_tc != nullptr) ? m_impl->fn_taylor_tc : m_impl->fn_taylor;
    hip_check(hipModuleLaunchKernel(fn, static_cast<unsigned>(grid), 1, 1,
                                    static_cast<unsigned>(bs), 1, 1, meta.lds_bytes, m_impl->stream, nullptr, config),
              "hipModuleLaunchKernel(taylor)");
    hip_check(hipEventRecord(m_impl->ev_stop[slot], m_impl->stream), "hipEventRecord");
    ++m_impl->n_launches;
}

std::vector<double> device_module::kernel_ms_history(std::size_t n)
{
    std::vector<double> ret;
    hip_check(hipSetDevice(m_impl->device), "hipSetDevice");
    n = std::min<std::size_t>({n, static_cast<std::size_t>(impl::n_ev), static_cast<std::size_t>(m_impl->n_launches)});
    for (std::size_t i = 0; i < n; ++i) {
        const auto slot = static_cast<int>((m_impl->n_launches - n + i) % impl::n_ev);
        hip_check(hipEventSynchronize(m_impl->ev_stop[slot]), "hipEventSynchronize");
        float ms = 0;
        hip_check(hipEventElapsedTime(&ms, m_impl->ev_start[slot], m_impl->ev_stop[slot]), "hipEventElapsedTime");
        ret.push_back(static_cast<double>(ms));
    }
    return ret;
}

void device_module::launch_dout(double *out, const double *tc, const double *hs, std::uint64_t N)
{
    if (N == 0u) {
        return;
    }
    hip_check(hipSetDevice(m_impl->device), "hipSetDevice");
    struct {
        double *out;
        const double *tc;
        const double *hs;
        unsigned long long N;
    } a{out, tc, hs, N};
    std::size_t sz = sizeof(a);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    const auto grid = (N + 255u) / 256u;
    hip_check(hipModuleLaunchKernel(m_impl->fn_dout, static_cast<unsigned>(grid), 1, 1, 256, 1, 1, 0, m_impl->stream,
                                    nullptr, config),
              "hipModuleLaunchKernel(dout)");
}

void device_module::synchronize()
{
    hip_check(hipSetDevice(m_impl->device), "hipSetDevice");
    hip_check(hipStreamSynchronize(m_impl->stream), "hipStreamSynchronize");
}

struct aux_module::impl {
    std::shared_ptr<const compiled_module> cm;
    int device = 0;
    hipModule_t mod = nullptr;
    std::unordered_map<std::string, hipFunction_t> fns;
};

aux_module::aux_module(std::shared_ptr<const compiled_module> cm, int device) : m_impl(std::make_unique<impl>())
{
    m_impl->cm = std::move(cm);
    m_impl->device = device;
    hip_check(hipSetDevice(device), "hipSetDevice");
    m_impl->mod = load_module_cached(m_impl->cm, device);
}

aux_module::~aux_module()
{
    if (m_impl && m_impl->mod != nullptr) {
        release_module(m_impl->cm, m_impl->device);
    }
}

int aux_module::device() const
{
    return m_impl->device;
}

void aux_module::launch(const char *name, std::uint64_t n_threads, unsigned block, const void *args,
                        std::size_t args_size, void *stream)
{
    if (n_threads == 0u) {
        return;
    }
    hip_check(hipSetDevice(m_impl->device), "hipSetDevice");
    auto it = m_impl->fns.find(name);
    if (it == m_impl->fns.end()) {
        hipFunction_t fn = nullptr;
        hip_check(hipModuleGetFunction(&fn, m_impl->mod, name), "hipModuleGetFunction(aux)");
        it = m_impl->fns.emplace(name, fn).first;
    }
    std::size_t sz = args_size;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                      HIP_LAUNCH_PARAM_END};
    const auto grid = (n_threads + block - 1u) / block;
    hip_check(hipModuleLaunchKernel(it->second, static_cast<unsigned>(grid), 1, 1, block, 1, 1, 0,
                                    static_cast<hipStream_t>(stream), nullptr, config),
              "hipModuleLaunchKernel(aux)");
}

std::shared_ptr<const compiled_module> hiprtc_compile_source(const std::string &source)
{
    emitted_module m;
    m.source = source;
    return hiprtc_compile(m);
}

void device_copy(void *dst, const void *src, std::size_t bytes, int device, void *stream)
{
    if (bytes == 0u) {
        return;
    }
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)), "hipMemcpyAsync");
}

void device_fill_bytes(void *dst, int value, std::size_t bytes, int device, void *stream)
{
    if (bytes == 0u) {
        return;
    }
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)), "hipMemsetAsync");
}

void stream_synchronize(int device, void *stream)
{
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}

device_buffer::device_buffer(std::size_t bytes, int device) : m_bytes(bytes), m_device(device)
{
    if (bytes != 0u) {
        hip_check(hipSetDevice(device), "hipSetDevice");
        hip_check(hipMalloc(&m_ptr, bytes), "hipMalloc");
    }
}

device_buffer::~device_buffer()
{
    if (m_ptr != nullptr) {
        (void)hipFree(m_ptr);
    }
}

device_buffer::device_buffer(device_buffer &&o) noexcept : m_ptr(o.m_ptr), m_bytes(o.m_bytes), m_device(o.m_device)
{
    o.m_ptr = nullptr;
    o.m_bytes = 0;
}

device_buffer &device_buffer::operator=(device_buffer &&o) noexcept
{
    if (this != &o) {
        if (m_ptr != nullptr) {
            (void)hipFree(m_ptr);
        }
        m_ptr = o.m_ptr;
        m_bytes = o.m_bytes;
        m_device = o.m_device;
        o.m_ptr = nullptr;
        o.m_bytes = 0;
    }
    return *this;
}

void device_buffer::upload(const void *src, std::size_t bytes, void *stream)
{
    if (bytes == 0u) {
        return;
    }
    hip_check(hipMemcpyAsync(m_ptr, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)),
              "hipMemcpyAsync(H2D)");
    // NOTE: the sources are pageable and often temporaries of the caller: an H2D copy from pageable memory may
    // still be reading the source after hipMemcpyAsync() returns (the runtime can pin it in place), so the copy
    // is completed here. No upload is on a timed path (the propagate loops are device-driven).
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}

void device_buffer::download(void *dst, std::size_t bytes, void *stream) const
{
    if (bytes == 0u) {
        return;
    }
    hip_check(hipMemcpyAsync(dst, m_ptr, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)),
              "hipMemcpyAsync(D2H)");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}

pinned_buffer::~pinned_buffer()
{
    if (m_ptr != nullptr) {
        (void)hipHostFree(m_ptr);
    }
}

void *pinned_buffer::reserve(std::size_t bytes)
{
    if (bytes > m_bytes) {
        if (m_ptr != nullptr) {
            (void)hipHostFree(m_ptr);
            m_ptr = nullptr;
            m_bytes = 0;
        }
        const auto want = bytes + bytes / 2u + 4096u;
        hip_check(hipHostMalloc(&m_ptr, want, hipHostMallocDefault), "hipHostMalloc");
        m_bytes = want;
    }
    return m_ptr;
}

void device_buffer::zero(void *stream)
{
    if (m_bytes == 0u) {
        return;
    }
    hip_check(hipMemsetAsync(m_ptr, 0, m_bytes, static_cast<hipStream_t>(stream)), "hipMemsetAsync");
}

} // namespace heyoka_amd
