// Compiled functions: batched evaluation of vector functions of the state on the device.
//
// Reference: class cfunc<T>, include/heyoka/expression.hpp:735-970; function_decompose() and
// add_cfunc(), src/expression_cfunc.cpp:723-900, :2180-2400. The reference JIT-compiles, for a list of
// expressions fn(vars), three functions (unstrided / strided / batch) evaluated with SIMD over
// `nevals` input columns. Here the same decomposition (minus the Taylor-specific rewrites) is emitted
// as one straight-line HIP kernel with one lane per evaluation, operating directly on the SoA arrays
// of the batch integrator (in[var * nevals + eval], i.e. the row-major 2D layout of the reference's
// multi-evaluation call operator, expression.hpp:898-960). This is the building block of the
// device-side invariant monitors (e.g. model::nbody_energy over a whole ensemble without a D2H copy).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "decompose.hpp"
#include "expression.hpp"
#include "kw.hpp"

namespace heyoka_amd
{

namespace detail
{

class cfunc_core
{
    struct impl;
    std::shared_ptr<impl> m_impl;

    void check_valid(const char *) const;

public:
    cfunc_core();
    cfunc_core(std::vector<expression> fn, std::vector<expression> vars, int device);
    cfunc_core(const cfunc_core &);
    cfunc_core(cfunc_core &&) noexcept;
    cfunc_core &operator=(const cfunc_core &);
    cfunc_core &operator=(cfunc_core &&) noexcept;
    ~cfunc_core();

    [[nodiscard]] bool is_valid() const noexcept;
    [[nodiscard]] const std::vector<expression> &get_fn() const;
    [[nodiscard]] const std::vector<expression> &get_vars() const;
    [[nodiscard]] const taylor_dc_t &get_dc() const;
    [[nodiscard]] std::uint32_t get_nparams() const;
    [[nodiscard]] std::uint32_t get_nvars() const;
    [[nodiscard]] std::uint32_t get_nouts() const;
    [[nodiscard]] bool is_time_dependent() const;
    [[nodiscard]] const std::string &get_hip_source() const;
    [[nodiscard]] int get_device() const;
    void set_stream(void *hip_stream);

    // Evaluation on device-resident arrays: out[nouts * nevals], in[nvars * nevals],
    // pars[nparams * nevals] (may be null if nparams == 0), time[nevals] (may be null if the function is
    // not time-dependent). Asynchronous on the stream.
    void call_device(double *d_out, const double *d_in, const double *d_pars, const double *d_time,
                     std::uint64_t nevals) const;
    // Host arrays (same layout): upload, evaluate, download, synchronise.
    void call_host(double *out, std::size_t out_size, const double *in, std::size_t in_size, const double *pars,
                   std::size_t pars_size, const double *time, std::size_t time_size) const;
};

} // namespace detail

template <typename T>
class cfunc
{
    static_assert(std::is_same_v<T, double>, "The MI355X build supports double precision only.");

    detail::cfunc_core m_core;

public:
    cfunc() = default;
    // kwargs of the reference are accepted (high_accuracy, compact_mode, parallel_mode, batch_size and the
    // llvm_state options are no-ops here: the kernel is always one lane per evaluation); kw::device
    // selects the HIP device.
    template <typename... KwArgs>
    explicit cfunc(std::vector<expression> fn, std::vector<expression> vars, const KwArgs &...kw_args)
        : m_core(std::move(fn), std::move(vars), static_cast<int>(kw::get(kw::device, 0, kw_args...)))
    {
        static_assert(kw::all_named_v<KwArgs...>);
    }

    [[nodiscard]] bool is_valid() const noexcept
    {
        return m_core.is_valid();
    }
    [[nodiscard]] const std::vector<expression> &get_fn() const
    {
        return m_core.get_fn();
    }
    [[nodiscard]] const std::vector<expression> &get_vars() const
    {
        return m_core.get_vars();
    }
    [[nodiscard]] const taylor_dc_t &get_dc() const
    {
        return m_core.get_dc();
    }
    [[nodiscard]] std::uint32_t get_nparams() const
    {
        return m_core.get_nparams();
    }
    [[nodiscard]] std::uint32_t get_nvars() const
    {
        return m_core.get_nvars();
    }
    [[nodiscard]] std::uint32_t get_nouts() const
    {
        return m_core.get_nouts();
    }
    [[nodiscard]] bool is_time_dependent() const
    {
        return m_core.is_time_dependent();
    }
    // Reference getters without a counterpart on the device (include/heyoka/expression.hpp:735-970): the evaluation is
    // always one lane per evaluation, in double precision, with the function fully unrolled.
    [[nodiscard]] bool get_parallel_mode() const
    {
        return false;
    }
    [[nodiscard]] bool get_compact_mode() const
    {
        return false;
    }
    [[nodiscard]] bool get_high_accuracy() const
    {
        return false;
    }

    // Evaluation over host vectors. Single evaluation: inputs.size() == nvars; multiple evaluations:
    // inputs.size() == nvars * nevals with the row-major layout inputs[var * nevals + eval] (the 2D
    // mdspan overload of the reference). kw::pars = vector (nparams * nevals), kw::time = scalar (single
    // evaluation) or vector (nevals).
    template <typename... KwArgs>
    void operator()(std::vector<T> &outputs, const std::vector<T> &inputs, const KwArgs &...kw_args) const
    {
        static_assert(kw::all_named_v<KwArgs...>);
        std::vector<T> pars, tm;
        if constexpr (kw::has_v<kw::pars_tag, KwArgs...>) {
            for (const auto &x : kw::get(kw::pars, 0, kw_args...)) {
                pars.push_back(static_cast<T>(x));
            }
        }
        bool with_time = false;
        if constexpr (kw::has_v<kw::time_tag, KwArgs...>) {
            with_time = true;
            using time_t = std::decay_t<decltype(kw::get(kw::time, 0, kw_args...))>;
            if constexpr (std::is_arithmetic_v<time_t>) {
                tm.push_back(static_cast<T>(kw::get(kw::time, 0, kw_args...)));
            } else {
                for (const auto &x : kw::get(kw::time, 0, kw_args...)) {
                    tm.push_back(static_cast<T>(x));
                }
            }
        }
        m_core.call_host(outputs.data(), outputs.size(), inputs.data(), inputs.size(),
                         kw::has_v<kw::pars_tag, KwArgs...> ? pars.data() : nullptr, pars.size(),
                         with_time ? tm.data() : nullptr, tm.size());
    }

    // MI355X extensions.
    [[nodiscard]] detail::cfunc_core &core()
    {
        return m_core;
    }
    [[nodiscard]] const detail::cfunc_core &core() const
    {
        return m_core;
    }
};

} // namespace heyoka_amd
