// Compiled functions on the device. See cfunc.hpp.
#include "cfunc.hpp"

#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hip_backend.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd::detail
{

namespace
{

struct cf_kargs {
    double *out;
    const double *in;
    const double *pars;
    const double *tm;
    unsigned long long n;
};

// One lane per evaluation; straight-line order-0 evaluation of the decomposition with the same node
// emitters as the Taylor stepper (hip_emit_detail.hpp, ssa_emitter::node(i, 0)).
std::string emit_cfunc_source(const taylor_program &p)
{
    // NOTE: the decomposition is sorted breadth-first (all the nodes of a dependency level before the next one),
    // which is the right order for the Taylor recursions but, for a straight-line evaluation, keeps every value of
    // a level alive until the next level: thousands of live values in one basic block, minutes of register
    // allocation (model::nbody_energy(32): 88 s, nbody_energy(64): 9 minutes). The nodes are therefore emitted
    // depth-first from the outputs (operands right before their consumer), and the inputs are read where they are
    // used.
    emit_detail::ssa_emitter e(p, 0);
    for (std::uint32_t i = 0; i < p.n_eq; ++i) {
        e.val(i, 0) = "a.in[(u64)" + std::to_string(i) + "u * N + s]";
    }
    std::vector<char> done(p.nodes.size(), 0);
    const auto emit_from = [&](std::uint32_t root) {
        // Iterative post-order traversal over the u variables >= n_eq.
        std::vector<std::pair<std::uint32_t, bool>> stack{{root, false}};
        while (!stack.empty()) {
            const auto [u, expanded] = stack.back();
            stack.pop_back();
            const auto i = u - p.n_eq;
            if (done[i] != 0) {
                continue;
            }
            if (expanded) {
                e.node(i, 0);
                done[i] = 1;
                continue;
            }
            stack.emplace_back(u, true);
            const auto &n = p.nodes[i];
            for (auto it = n.args.rbegin(); it != n.args.rend(); ++it) {
                if (it->type == operand::kind::uvar && it->idx >= p.n_eq && done[it->idx - p.n_eq] == 0) {
                    stack.emplace_back(it->idx, false);
                }
            }
        }
    };
    for (const auto &d : p.sv_defs) {
        if (d.type == operand::kind::uvar && d.idx >= p.n_eq) {
            emit_from(d.idx);
        }
    }
    std::ostringstream src;
    src << emit_detail::prelude << emit_detail::rules_source(p);
    src << R"HIP(
struct hy_cf_args {
    double *out;
    const double *in;
    const double *pars;
    const double *tm;
    u64 n;
};

extern "C" __global__ void __launch_bounds__(256) hy_cfunc(const hy_cf_args a)
{
    const u64 s = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 N = a.n;
    if (s >= N) return;
)HIP";
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        src << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
    }
    if (p.time_dependent) {
        src << "const double t_hi = a.tm[s];\n";
    }
    src << e.os.str();
    for (std::size_t o = 0; o < p.sv_defs.size(); ++o) {
        const auto &d = p.sv_defs[o];
        src << "a.out[(u64)" << o << "u * N + s] = "
            << (d.type == operand::kind::uvar ? e.val(d.idx, 0) : e.numpar(d)) << ";\n";
    }
    src << "}\n";
    return src.str();
}

} // namespace

struct cfunc_core::impl {
    std::vector<expression> fn, vars;
    taylor_dc_t dc;
    taylor_program prog;
    std::string source;
    std::shared_ptr<const compiled_module> cmod;
    int device = 0;
    void *stream = nullptr;
    // Created lazily at the first evaluation (construction works without a GPU).
    std::unique_ptr<aux_module> mod;
};

cfunc_core::cfunc_core() = default;

// Reference: cfunc<T>::cfunc(), src/cfunc_class.cpp:180-330.
cfunc_core::cfunc_core(std::vector<expression> fn, std::vector<expression> vars, int device)
    : m_impl(std::make_shared<impl>())
{
    auto &d = *m_impl;
    d.dc = function_decompose(fn, vars);
    d.prog = make_program(d.dc, static_cast<std::uint32_t>(vars.size()), static_cast<std::uint32_t>(fn.size()));
    d.fn = std::move(fn);
    d.vars = std::move(vars);
    d.device = device;
    d.source = emit_cfunc_source(d.prog);
    d.cmod = hiprtc_compile_source(d.source);
}

cfunc_core::cfunc_core(const cfunc_core &) = default;
cfunc_core::cfunc_core(cfunc_core &&) noexcept = default;
cfunc_core &cfunc_core::operator=(const cfunc_core &) = default;
cfunc_core &cfunc_core::operator=(cfunc_core &&) noexcept = default;
cfunc_core::~cfunc_core() = default;

void cfunc_core::check_valid(const char *name) const
{
    if (!m_impl) {
        throw std::invalid_argument(std::string("The function '") + name
                                    + "' cannot be invoked on an invalid cfunc object");
    }
}

bool cfunc_core::is_valid() const noexcept
{
    return static_cast<bool>(m_impl);
}
const std::vector<expression> &cfunc_core::get_fn() const
{
    check_valid(__func__);
    return m_impl->fn;
}
const std::vector<expression> &cfunc_core::get_vars() const
{
    check_valid(__func__);
    return m_impl->vars;
}
const taylor_dc_t &cfunc_core::get_dc() const
{
    check_valid(__func__);
    return m_impl->dc;
}
std::uint32_t cfunc_core::get_nparams() const
{
    check_valid(__func__);
    return m_impl->prog.n_par;
}
std::uint32_t cfunc_core::get_nvars() const
{
    check_valid(__func__);
    return static_cast<std::uint32_t>(m_impl->vars.size());
}
std::uint32_t cfunc_core::get_nouts() const
{
    check_valid(__func__);
    return static_cast<std::uint32_t>(m_impl->fn.size());
}
bool cfunc_core::is_time_dependent() const
{
    check_valid(__func__);
    return m_impl->prog.time_dependent;
}
const std::string &cfunc_core::get_hip_source() const
{
    check_valid(__func__);
    return m_impl->source;
}
int cfunc_core::get_device() const
{
    check_valid(__func__);
    return m_impl->device;
}
void cfunc_core::set_stream(void *s)
{
    check_valid(__func__);
    m_impl->stream = s;
}

void cfunc_core::call_device(double *d_out, const double *d_in, const double *d_pars, const double *d_time,
                             std::uint64_t nevals) const
{
    check_valid(__func__);
    auto &d = *m_impl;
    if (d.prog.n_par != 0u && d_pars == nullptr) {
        throw std::invalid_argument(
            "An array of parameter values must be passed in order to evaluate a function with parameters");
    }
    if (d.prog.time_dependent && d_time == nullptr) {
        throw std::invalid_argument(
            "An array of time values must be provided in order to evaluate a time-dependent function");
    }
    if (!d.mod) {
        d.mod = std::make_unique<aux_module>(d.cmod, d.device);
    }
    const cf_kargs a{d_out, d_in, d_pars, d_time, nevals};
    d.mod->launch("hy_cfunc", nevals, 256, &a, sizeof(a), d.stream);
}

// Reference: cfunc<T>::single_eval() / multi_eval(), src/cfunc_class.cpp:544-640, :875-1000.
void cfunc_core::call_host(double *out, std::size_t out_size, const double *in, std::size_t in_size, const double *pars,
                           std::size_t pars_size, const double *time, std::size_t time_size) const
{
    check_valid(__func__);
    auto &d = *m_impl;
    const auto nouts = d.fn.size(), nvars = d.vars.size();
    const std::size_t npars = d.prog.n_par;

    if (out_size == 0u || out_size % nouts != 0u) {
        throw std::invalid_argument("Invalid outputs array passed to a cfunc: the number of function outputs is "
                                    + std::to_string(nouts) + ", but the outputs array has a size of "
                                    + std::to_string(out_size));
    }
    const auto ncols = out_size / nouts;
    if (ncols == 1u) {
        if (in_size != nvars) {
            throw std::invalid_argument("Invalid inputs array passed to a cfunc: the number of function inputs is "
                                        + std::to_string(nvars) + ", but the inputs array has a size of "
                                        + std::to_string(in_size));
        }
    } else if (in_size != nvars * ncols) {
        throw std::invalid_argument("Invalid inputs array passed to a cfunc: the expected number of columns deduced "
                                    "from the outputs array is "
                                    + std::to_string(ncols) + ", but the inputs array has a size of "
                                    + std::to_string(in_size) + " for " + std::to_string(nvars) + " row(s)");
    }
    if (npars != 0u && pars == nullptr) {
        throw std::invalid_argument(
            "An array of parameter values must be passed in order to evaluate a function with parameters");
    }
    if (pars != nullptr && pars_size != npars * ncols) {
        throw std::invalid_argument("The array of parameter values provided for the evaluation of a compiled function "
                                    "has "
                                    + std::to_string(pars_size) + " element(s), but the number of parameters in the "
                                    "function is "
                                    + std::to_string(npars)
                                    + (ncols == 1u ? std::string{} : " (x " + std::to_string(ncols) + " columns)"));
    }
    if (d.prog.time_dependent && time == nullptr) {
        throw std::invalid_argument(ncols == 1u ? "A time value must be provided in order to evaluate a "
                                                  "time-dependent function"
                                                : "An array of time values must be provided in order to evaluate a "
                                                  "time-dependent function");
    }
    if (time != nullptr && time_size != ncols) {
        throw std::invalid_argument("The array of time values provided for the evaluation of a compiled function has "
                                    "a size of "
                                    + std::to_string(time_size) + ", but the expected size deduced from the outputs "
                                    "array is "
                                    + std::to_string(ncols));
    }

    const auto dsz = sizeof(double);
    device_buffer b_out(out_size * dsz, d.device), b_in(in_size * dsz, d.device);
    device_buffer b_pars((pars != nullptr && npars != 0u) ? pars_size * dsz : 0u, d.device);
    device_buffer b_tm((time != nullptr && d.prog.time_dependent) ? time_size * dsz : 0u, d.device);
    b_in.upload(in, in_size * dsz, d.stream);
    if (b_pars.bytes() != 0u) {
        b_pars.upload(pars, pars_size * dsz, d.stream);
    }
    if (b_tm.bytes() != 0u) {
        b_tm.upload(time, time_size * dsz, d.stream);
    }
    call_device(b_out.as<double>(), b_in.as<double>(), b_pars.as<double>(), b_tm.as<double>(), ncols);
    b_out.download(out, out_size * dsz, d.stream);
    stream_synchronize(d.device, d.stream);
}

} // namespace heyoka_amd::detail
