// Continuous (dense) output for the batch integrator.
//
// Reference: class continuous_output_batch<T>, include/heyoka/continuous_output.hpp:151-204 and
// src/continuous_output.cpp:602-1236. The reference stores, for every lock-step sweep of
// propagate_until(), the Taylor coefficients of all the batch lanes and the double-length times, and
// JIT-compiles a function which, for a vector of target times, locates the step with a vectorised
// upper_bound and evaluates the Taylor polynomials.
//
// MI355X design: everything stays in HBM. Each sweep's coefficients are appended with one
// device-to-device copy (n_eq * (order + 1) * N doubles, the layout written by the stepper,
// tc[(var * (order + 1) + k) * N + lane]); the evaluation is one kernel with one lane per batch element
// (binary search over the lane's column of the times array + Horner / compensated summation, cf.
// src/continuous_output.cpp:700-984). The host vectors mirror the reference's getters and are
// materialised lazily.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <ostream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace heyoka_amd
{

namespace detail
{

class c_out_core
{
    struct data;
    // Immutable after construction, shared between copies.
    std::shared_ptr<const data> m_data;
    // Per-object mutable state (output buffers).
    struct scratch;
    std::shared_ptr<scratch> m_scratch;
    std::vector<double> m_output;

    friend class c_out_builder;

    void check_valid() const;
    void run(const double *host_tm, bool scalar, double tm_s);

public:
    c_out_core();
    c_out_core(const c_out_core &);
    c_out_core(c_out_core &&) noexcept;
    c_out_core &operator=(const c_out_core &);
    c_out_core &operator=(c_out_core &&) noexcept;
    ~c_out_core();

    [[nodiscard]] bool valid() const
    {
        return static_cast<bool>(m_data);
    }

    const std::vector<double> &call(const double *);
    const std::vector<double> &call(const std::vector<double> &);
    const std::vector<double> &call(double);
    // MI355X extension: target times and output both in device memory (N and dim * N doubles);
    // asynchronous on the integrator's stream.
    void call_device(const double *d_tm, double *d_out);

    [[nodiscard]] const std::vector<double> &get_output() const;
    [[nodiscard]] const std::vector<double> &get_times() const;
    [[nodiscard]] const std::vector<double> &get_times_lo() const;
    [[nodiscard]] const std::vector<double> &get_tcs() const;
    [[nodiscard]] std::uint32_t get_batch_size() const;
    [[nodiscard]] std::pair<std::vector<double>, std::vector<double>> get_bounds() const;
    [[nodiscard]] std::size_t get_n_steps() const;
    [[nodiscard]] std::uint32_t get_dim() const;
    [[nodiscard]] std::uint32_t get_order() const;
    void stream_to(std::ostream &) const;
};

// HIP source of the evaluation kernel (exposed for the build-time compilation check).
std::string make_cout_source(std::uint32_t order, std::uint32_t dim, bool high_accuracy);

// Accumulates the per-sweep data during propagate_until().
class c_out_builder
{
    struct impl;
    std::unique_ptr<impl> m_impl;

public:
    c_out_builder(std::uint32_t N, std::uint32_t order, std::uint32_t dim, bool high_accuracy, int device,
                  void *stream, const std::vector<double> &time_hi, const std::vector<double> &time_lo);
    ~c_out_builder();
    // Append the coefficients of the sweep just taken (device pointer) and the new times.
    void append(const double *d_tc, const std::vector<double> &time_hi, const std::vector<double> &time_lo);
    // t_dir[i] != 0 -> forward integration for lane i.
    std::optional<c_out_core> finish(const std::vector<int> &t_dir);
};

} // namespace detail

template <typename T>
class continuous_output_batch
{
    static_assert(std::is_same_v<T, double>, "The MI355X build supports double precision only.");

    detail::c_out_core m_core;

public:
    continuous_output_batch() = default;
    explicit continuous_output_batch(detail::c_out_core c) : m_core(std::move(c)) {}

    const std::vector<T> &operator()(const T *tm)
    {
        return m_core.call(tm);
    }
    const std::vector<T> &operator()(const std::vector<T> &tm)
    {
        return m_core.call(tm);
    }
    const std::vector<T> &operator()(T tm)
    {
        return m_core.call(tm);
    }

    [[nodiscard]] const std::vector<T> &get_output() const
    {
        return m_core.get_output();
    }
    [[nodiscard]] const std::vector<T> &get_times() const
    {
        return m_core.get_times();
    }
    [[nodiscard]] const std::vector<T> &get_tcs() const
    {
        return m_core.get_tcs();
    }
    [[nodiscard]] std::uint32_t get_batch_size() const
    {
        return m_core.get_batch_size();
    }
    [[nodiscard]] std::pair<std::vector<T>, std::vector<T>> get_bounds() const
    {
        return m_core.get_bounds();
    }
    [[nodiscard]] std::size_t get_n_steps() const
    {
        return m_core.get_n_steps();
    }

    // MI355X extensions.
    [[nodiscard]] detail::c_out_core &core()
    {
        return m_core;
    }
    [[nodiscard]] const detail::c_out_core &core() const
    {
        return m_core;
    }
};

template <typename T>
inline std::ostream &operator<<(std::ostream &os, const continuous_output_batch<T> &co)
{
    co.core().stream_to(os);
    return os;
}

} // namespace heyoka_amd
