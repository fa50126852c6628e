// hiprtc + HIP module runtime for the generated Taylor kernels.
//
// Plays the role of the reference's JIT layer (llvm_state::compile() / jit_lookup(),
// src/llvm_state.cpp:1507, :1606) for the HIP source modules produced by hip_emit.cpp.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "hip_emit.hpp"

namespace heyoka_amd
{

// A compiled (device-independent) code object for gfx950.
struct compiled_module {
    emitted_module meta;
    std::vector<char> code;
    std::string log;
    double compile_seconds = 0;
};

// Compile HIP source with hiprtc for gfx950. Works without a GPU (used by the CPU build check).
// Results are cached in-process by source text.
std::shared_ptr<const compiled_module> hiprtc_compile(const emitted_module &m);

// Number of visible HIP devices (0 if no GPU / no driver).
int hip_device_count();

// A module loaded on a specific device + the buffers of one integrator.
class device_module
{
    struct impl;
    std::unique_ptr<impl> m_impl;

public:
    device_module(std::shared_ptr<const compiled_module>, int device);
    ~device_module();
    device_module(const device_module &) = delete;
    device_module &operator=(const device_module &) = delete;

    [[nodiscard]] int device() const;
    void set_stream(void *hip_stream);
    [[nodiscard]] void *stream() const;

    // Launch the stepper over N systems.
    // (user_tape: the tape / scratch of the launch in caller-owned device memory of tape_bytes(args.N) bytes instead of the
    // module's own allocation - the c_step(..., void *tape) flavour of the reference's stepper ABI.)
    void launch_taylor(const hy_kargs &args, void *user_tape = nullptr);
    [[nodiscard]] std::size_t tape_bytes(std::uint64_t n_systems) const;
    // Durations (ms) of the last n stepper kernels, oldest first, from HIP events recorded on the launch
    // stream right around each launch (at most 64 are kept); synchronises on their completion.
    std::vector<double> kernel_ms_history(std::size_t n);
    // Launch the dense-output kernel.
    void launch_dout(double *out, const double *tc, const double *hs, std::uint64_t N);
    void synchronize();
};

// A generic hiprtc module loaded on a device: used for the auxiliary kernels (continuous output,
// compiled functions) that live outside the stepper module. Kernels take a single struct argument.
class aux_module
{
    struct impl;
    std::unique_ptr<impl> m_impl;

public:
    aux_module(std::shared_ptr<const compiled_module>, int device);
    ~aux_module();
    aux_module(const aux_module &) = delete;
    aux_module &operator=(const aux_module &) = delete;

    [[nodiscard]] int device() const;
    // Launch kernel `name` with n_threads threads in blocks of `block`, passing the args_size bytes at args.
    void launch(const char *name, std::uint64_t n_threads, unsigned block, const void *args, std::size_t args_size,
                void *stream);
};

// Plain source -> code object helper for the auxiliary modules (cached like hiprtc_compile()).
std::shared_ptr<const compiled_module> hiprtc_compile_source(const std::string &source);

// Device-to-device / host copies on a stream + stream synchronisation, for code that does not include HIP headers.
void device_copy(void *dst, const void *src, std::size_t bytes, int device, void *stream);
void stream_synchronize(int device, void *stream);
void device_fill_bytes(void *dst, int value, std::size_t bytes, int device, void *stream);

// Thin RAII device buffer.
class device_buffer
{
    void *m_ptr = nullptr;
    std::size_t m_bytes = 0;
    int m_device = 0;

public:
    device_buffer() = default;
    device_buffer(std::size_t bytes, int device);
    ~device_buffer();
    device_buffer(device_buffer &&) noexcept;
    device_buffer &operator=(device_buffer &&) noexcept;
    device_buffer(const device_buffer &) = delete;
    device_buffer &operator=(const device_buffer &) = delete;

    [[nodiscard]] void *get() const
    {
        return m_ptr;
    }
    template <typename T>
    [[nodiscard]] T *as() const
    {
        return static_cast<T *>(m_ptr);
    }
    [[nodiscard]] std::size_t bytes() const
    {
        return m_bytes;
    }
    void upload(const void *src, std::size_t bytes, void *stream);
    void download(void *dst, std::size_t bytes, void *stream) const;
    void zero(void *stream);
};

// Page-locked host memory (hipHostMalloc): the target of the per-step downloads of a loop. A device-to-host copy into
// pageable memory makes the runtime pin and unpin the destination around the transfer; with a fresh 20 MB vector per step
// (the event records of 10^5 systems) the next submission to the device waited 15 ... 25 ms for it
// (profiles/r05_events_leg_laps.log).
class pinned_buffer
{
    void *m_ptr = nullptr;
    std::size_t m_bytes = 0;

public:
    pinned_buffer() = default;
    ~pinned_buffer();
    pinned_buffer(const pinned_buffer &) = delete;
    pinned_buffer &operator=(const pinned_buffer &) = delete;
    // Grows (never shrinks) to at least 'bytes'; the contents are not preserved.
    void *reserve(std::size_t bytes);
    [[nodiscard]] std::size_t bytes() const
    {
        return m_bytes;
    }
};

} // namespace heyoka_amd
