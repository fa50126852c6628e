// Continuous output for the batch integrator: device-resident storage + evaluation kernel.
// See continuous_output.hpp for the design; reference: src/continuous_output.cpp:602-1306.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <locale>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "continuous_output.hpp"
#include "dfloat.hpp"
#include "hip_backend.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd::detail
{

namespace
{

struct cout_kargs {
    double *out;
    const double *tm;
    const double *const *tcs;
    const double *thi;
    const double *tlo;
    unsigned long long N;
    unsigned n_times;
    unsigned scalar_tm;
    double tm_s;
};

std::string fp_str(double x)
{
    std::ostringstream oss;
    oss.imbue(std::locale::classic());
    oss.precision(std::numeric_limits<double>::max_digits10);
    oss << x;
    return oss.str();
}

} // namespace

// The evaluation kernel. One lane per batch element:
// - upper_bound over the lane's column of the (hi, lo) times (src/continuous_output.cpp:700-800; the last
//   row is the +-inf padding which makes the search well defined),
// - tc_idx = first - (first != 0) - (first == n_times - 1) (:803-813),
// - h = (tm - times[tc_idx]) in double-length arithmetic, hi part (:835-842),
// - Horner from the top order, or compensated summation in high accuracy mode (:851-975).
std::string make_cout_source(std::uint32_t order, std::uint32_t dim, bool ha)
{
    std::ostringstream src;
    src << emit_detail::prelude;
    src << "#define HY_ORDER " << order << "u\n#define HY_DIM " << dim << "u\n#define HY_HA " << (ha ? 1 : 0) << "\n";
    src << R"HIP(
struct hy_cout_args {
    double *out;
    const double *tm;
    const double *const *tcs;
    const double *thi;
    const double *tlo;
    u64 N;
    unsigned n_times;
    unsigned scalar_tm;
    double tm_s;
};

extern "C" __global__ void __launch_bounds__(256) hy_c_out(const hy_cout_args a)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.N;
    if (i >= N) return;
    hy_df tm;
    tm.hi = a.scalar_tm ? a.tm_s : a.tm[i];
    tm.lo = 0.0;
    const unsigned nt = a.n_times;
    hy_df t0, t1;
    t0.hi = a.thi[i]; t0.lo = a.tlo[i];
    t1.hi = a.thi[(u64)(nt - 1u) * N + i]; t1.lo = a.tlo[(u64)(nt - 1u) * N + i];
    const bool dir = hy_df_lt(t0, t1);
    unsigned first = 0, count = nt;
    while (count != 0u) {
        const unsigned step = count / 2u;
        const unsigned idx = first + step;
        hy_df tv;
        tv.hi = a.thi[(u64)idx * N + i];
        tv.lo = a.tlo[(u64)idx * N + i];
        const bool cond = dir ? !hy_df_lt(tm, tv) : !hy_df_lt(tv, tm);
        if (cond) {
            first = idx + 1u;
            count -= step + 1u;
        } else {
            count = step;
        }
    }
    const unsigned tc_idx = first - (first != 0u ? 1u : 0u) - (first == nt - 1u ? 1u : 0u);
    hy_df ts;
    ts.hi = a.thi[(u64)tc_idx * N + i];
    ts.lo = a.tlo[(u64)tc_idx * N + i];
    const double h = hy_df_sub(tm, ts).hi;
    const double *tc = a.tcs[tc_idx] + i;
    for (unsigned v = 0; v < HY_DIM; ++v) {
        const double *c = tc + (u64)v * (HY_ORDER + 1u) * N;
#if HY_HA
        double res = c[0], comp = 0.0, cur_h = h;
        for (unsigned k = 1; k <= HY_ORDER; ++k) {
            const double tmp = c[(u64)k * N] * cur_h;
            const double y = tmp - comp;
            const double t = res + y;
            comp = (t - res) - y;
            res = t;
            cur_h = cur_h * h;
        }
#else
        double res = c[(u64)HY_ORDER * N];
        for (unsigned k = 1; k <= HY_ORDER; ++k) {
            res = c[(u64)(HY_ORDER - k) * N] + res * h;
        }
#endif
        a.out[(u64)v * N + i] = res;
    }
}
)HIP";
    return src.str();
}

struct c_out_core::data {
    std::uint32_t N = 0, order = 0, dim = 0;
    bool ha = false;
    int device = 0;
    void *stream = nullptr;
    // Host copies of the times, including the padding row.
    std::vector<double> times_hi, times_lo;
    // Device storage.
    std::vector<device_buffer> tcs; // one per sweep
    device_buffer d_ptrs, d_thi, d_tlo;
    std::unique_ptr<aux_module> mod;
    // Lazily materialised host copy of the coefficients.
    mutable std::vector<double> tcs_host;

    [[nodiscard]] std::size_t n_rows() const
    {
        return times_hi.size() / N;
    }
};

struct c_out_core::scratch {
    device_buffer d_out, d_tm;
};

c_out_core::c_out_core() = default;
c_out_core::c_out_core(const c_out_core &o) : m_data(o.m_data), m_output(o.m_output) {}
c_out_core::c_out_core(c_out_core &&) noexcept = default;
c_out_core &c_out_core::operator=(const c_out_core &o)
{
    if (this != &o) {
        *this = c_out_core(o);
    }
    return *this;
}
c_out_core &c_out_core::operator=(c_out_core &&) noexcept = default;
c_out_core::~c_out_core() = default;

void c_out_core::check_valid() const
{
    if (!m_data) {
        throw std::invalid_argument("Cannot use a default-constructed continuous_output_batch object");
    }
}

void c_out_core::call_device(const double *d_tm, double *d_out)
{
    check_valid();
    const auto &d = *m_data;
    cout_kargs a{d_out, d_tm, d.d_ptrs.as<const double *>(), d.d_thi.as<double>(), d.d_tlo.as<double>(), d.N,
                 static_cast<unsigned>(d.n_rows()), 0u, 0.};
    d.mod->launch("hy_c_out", d.N, 256, &a, sizeof(a), d.stream);
}

void c_out_core::run(const double *host_tm, bool scalar, double tm_s)
{
    const auto &d = *m_data;
    if (!m_scratch) {
        m_scratch = std::make_shared<scratch>();
        m_scratch->d_out = device_buffer(static_cast<std::size_t>(d.dim) * d.N * sizeof(double), d.device);
        m_scratch->d_tm = device_buffer(static_cast<std::size_t>(d.N) * sizeof(double), d.device);
    }
    if (!scalar) {
        m_scratch->d_tm.upload(host_tm, static_cast<std::size_t>(d.N) * sizeof(double), d.stream);
    }
    cout_kargs a{m_scratch->d_out.as<double>(), m_scratch->d_tm.as<double>(), d.d_ptrs.as<const double *>(),
                 d.d_thi.as<double>(),          d.d_tlo.as<double>(),         d.N,
                 static_cast<unsigned>(d.n_rows()), scalar ? 1u : 0u, tm_s};
    d.mod->launch("hy_c_out", d.N, 256, &a, sizeof(a), d.stream);
    m_output.resize(static_cast<std::size_t>(d.dim) * d.N);
    m_scratch->d_out.download(m_output.data(), m_output.size() * sizeof(double), d.stream);
    stream_synchronize(d.device, d.stream);
}

const std::vector<double> &c_out_core::call(const double *t)
{
    check_valid();
    const auto N = m_data->N;
    for (std::uint32_t i = 0; i < N; ++i) {
        if (!std::isfinite(t[i])) {
            throw std::invalid_argument("Cannot compute the continuous output in batch mode for the batch index "
                                        + std::to_string(i) + " at the non-finite time " + fp_str(t[i]));
        }
    }
    // NOTE: the upload snapshots the times, so aliasing with the output vector is harmless.
    const std::vector<double> tmp(t, t + N);
    run(tmp.data(), false, 0.);
    return m_output;
}

const std::vector<double> &c_out_core::call(const std::vector<double> &tm)
{
    check_valid();
    if (tm.size() != m_data->N) {
        throw std::invalid_argument("An invalid time vector was passed to the call operator of "
                                    "continuous_output_batch: the vector size is "
                                    + std::to_string(tm.size()) + ", but a size of " + std::to_string(m_data->N)
                                    + " was expected instead");
    }
    return call(tm.data());
}

const std::vector<double> &c_out_core::call(double tm)
{
    check_valid();
    if (!std::isfinite(tm)) {
        throw std::invalid_argument("Cannot compute the continuous output in batch mode at the non-finite time "
                                    + fp_str(tm));
    }
    run(nullptr, true, tm);
    return m_output;
}

const std::vector<double> &c_out_core::get_output() const
{
    return m_output;
}

namespace
{

const std::vector<double> empty_vec;

}

const std::vector<double> &c_out_core::get_times() const
{
    return m_data ? m_data->times_hi : empty_vec;
}

const std::vector<double> &c_out_core::get_times_lo() const
{
    return m_data ? m_data->times_lo : empty_vec;
}

const std::vector<double> &c_out_core::get_tcs() const
{
    if (!m_data) {
        return empty_vec;
    }
    const auto &d = *m_data;
    if (d.tcs_host.empty() && !d.tcs.empty()) {
        const auto chunk = static_cast<std::size_t>(d.dim) * (d.order + 1u) * d.N;
        d.tcs_host.resize(chunk * d.tcs.size());
        for (std::size_t s = 0; s < d.tcs.size(); ++s) {
            d.tcs[s].download(d.tcs_host.data() + s * chunk, chunk * sizeof(double), d.stream);
        }
        stream_synchronize(d.device, d.stream);
    }
    return d.tcs_host;
}

std::uint32_t c_out_core::get_batch_size() const
{
    return m_data ? m_data->N : 0u;
}

std::uint32_t c_out_core::get_dim() const
{
    return m_data ? m_data->dim : 0u;
}

std::uint32_t c_out_core::get_order() const
{
    return m_data ? m_data->order : 0u;
}

std::pair<std::vector<double>, std::vector<double>> c_out_core::get_bounds() const
{
    check_valid();
    const auto &d = *m_data;
    std::vector<double> lb(d.N), ub(d.N);
    for (std::uint32_t i = 0; i < d.N; ++i) {
        lb[i] = d.times_hi[i];
        // NOTE: take into account the padding.
        ub[i] = d.times_hi[d.times_hi.size() - 2u * d.N + i];
    }
    return {std::move(lb), std::move(ub)};
}

std::size_t c_out_core::get_n_steps() const
{
    check_valid();
    // NOTE: account for padding.
    return m_data->n_rows() - 2u;
}

// Reference: c_out_batch_stream_impl(), src/continuous_output.cpp:1253-1303.
void c_out_core::stream_to(std::ostream &os) const
{
    std::ostringstream oss;
    oss.imbue(std::locale::classic());
    oss << std::showpoint;
    oss.precision(std::numeric_limits<double>::max_digits10);
    oss << "C++ datatype: double\n";
    if (!m_data) {
        oss << "Default-constructed continuous_output_batch";
    } else {
        const auto &d = *m_data;
        const auto N = d.N;
        const auto get_se = [&](std::uint32_t i) {
            return std::pair{dfloat(d.times_hi[i], d.times_lo[i]),
                             dfloat(d.times_hi[d.times_hi.size() - 2u * N + i],
                                    d.times_lo[d.times_lo.size() - 2u * N + i])};
        };
        oss << "Directions  : [";
        for (std::uint32_t i = 0; i < N; ++i) {
            const auto [s, e] = get_se(i);
            oss << ((s < e) ? "forward" : "backward");
            if (i != N - 1u) {
                oss << ", ";
            }
        }
        oss << "]\n";
        oss << "Time ranges : [";
        for (std::uint32_t i = 0; i < N; ++i) {
            const auto [s, e] = get_se(i);
            if (s < e) {
                oss << "[" << fp_str(s.hi) << ", " << fp_str(e.hi) << ")";
            } else {
                oss << "(" << fp_str(e.hi) << ", " << fp_str(s.hi) << "]";
            }
            if (i != N - 1u) {
                oss << ", ";
            }
        }
        oss << "]\n";
        oss << "N of steps  : " << get_n_steps() << '\n';
    }
    os << oss.str();
}

// ---------------------------------------------------------------------------------------------------

struct c_out_builder::impl {
    std::shared_ptr<c_out_core::data> d;
};

c_out_builder::c_out_builder(std::uint32_t N, std::uint32_t order, std::uint32_t dim, bool high_accuracy, int device,
                             void *stream, const std::vector<double> &time_hi, const std::vector<double> &time_lo)
    : m_impl(std::make_unique<impl>())
{
    auto d = std::make_shared<c_out_core::data>();
    d->N = N;
    d->order = order;
    d->dim = dim;
    d->ha = high_accuracy;
    d->device = device;
    d->stream = stream;
    // Push in the starting time (src/taylor_adaptive_batch.cpp:1249-1253).
    d->times_hi = time_hi;
    d->times_lo = time_lo;
    m_impl->d = std::move(d);
}

c_out_builder::~c_out_builder() = default;

// Reference: update_c_out, src/taylor_adaptive_batch.cpp:1317-1346.
void c_out_builder::append(const double *d_tc, const std::vector<double> &time_hi, const std::vector<double> &time_lo)
{
    auto &d = *m_impl->d;
    d.times_hi.insert(d.times_hi.end(), time_hi.begin(), time_hi.end());
    d.times_lo.insert(d.times_lo.end(), time_lo.begin(), time_lo.end());
    const auto bytes = static_cast<std::size_t>(d.dim) * (d.order + 1u) * d.N * sizeof(double);
    device_buffer buf(bytes, d.device);
    device_copy(buf.get(), d_tc, bytes, d.device, d.stream);
    d.tcs.push_back(std::move(buf));
}

// Reference: make_c_out, src/taylor_adaptive_batch.cpp:1276-1314.
std::optional<c_out_core> c_out_builder::finish(const std::vector<int> &t_dir)
{
    auto &d = *m_impl->d;
    if (d.times_hi.size() / d.N < 2u) {
        // NOTE: this means that no successful steps were taken.
        return {};
    }
    // Padding row: +-inf, which makes the upper_bound search well defined.
    for (std::uint32_t i = 0; i < d.N; ++i) {
        d.times_hi.push_back(t_dir[i] != 0 ? std::numeric_limits<double>::infinity()
                                           : -std::numeric_limits<double>::infinity());
        d.times_lo.push_back(0.);
    }
    if (d.times_hi.size() / d.N > std::numeric_limits<std::uint32_t>::max()) {
        throw std::overflow_error(
            "Overflow detected while adding continuous output to a Taylor integrator in batch mode");
    }
    d.d_thi = device_buffer(d.times_hi.size() * sizeof(double), d.device);
    d.d_tlo = device_buffer(d.times_lo.size() * sizeof(double), d.device);
    d.d_thi.upload(d.times_hi.data(), d.times_hi.size() * sizeof(double), d.stream);
    d.d_tlo.upload(d.times_lo.data(), d.times_lo.size() * sizeof(double), d.stream);
    std::vector<const double *> ptrs;
    for (const auto &b : d.tcs) {
        ptrs.push_back(b.as<const double>());
    }
    d.d_ptrs = device_buffer(ptrs.size() * sizeof(const double *), d.device);
    d.d_ptrs.upload(ptrs.data(), ptrs.size() * sizeof(const double *), d.stream);
    stream_synchronize(d.device, d.stream);
    d.mod = std::make_unique<aux_module>(hiprtc_compile_source(make_cout_source(d.order, d.dim, d.ha)), d.device);

    c_out_core ret;
    ret.m_data = std::move(m_impl->d);
    ret.m_output.resize(static_cast<std::size_t>(ret.m_data->dim) * ret.m_data->N);
    return ret;
}

} // namespace heyoka_amd::detail
