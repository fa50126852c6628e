// C ABI of the MI355X batch Taylor integrator. See include/heyoka_amd.h.
#include "logging.hpp"
#include "_build/build_id.h"
#include "../../include/heyoka_amd.h"

#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>

#include "cfunc.hpp"
#include "decompose.hpp"
#include "ensemble.hpp"
#include "expression.hpp"
#include "hip_backend.hpp"
#include "model.hpp"
#include "node_rule.hpp"
#include "taylor_adaptive_batch.hpp"

using namespace heyoka_amd;

struct hy_expr_s {
    expression ex;
};

struct hy_sys_s {
    std::vector<std::pair<expression, expression>> sys;
};

struct hy_tab_s {
    detail::tab_core core;
};

struct hy_cout_s {
    detail::c_out_core core;
};

struct hy_cfunc_s {
    detail::cfunc_core core;
};

namespace
{

thread_local std::string last_error;
thread_local int last_error_code = HY_OK;

int handle_exception_impl();

int handle_exception()
{
    last_error_code = handle_exception_impl();
    return last_error_code;
}

int handle_exception_impl()
{
    try {
        throw;
    } catch (const not_implemented_error &e) {
        last_error = e.what();
        return HY_ERR_NOT_IMPLEMENTED;
    } catch (const std::invalid_argument &e) {
        last_error = e.what();
        return HY_ERR_INVALID_ARGUMENT;
    } catch (const std::overflow_error &e) {
        last_error = e.what();
        return HY_ERR_OVERFLOW;
    } catch (const std::exception &e) {
        last_error = e.what();
        return HY_ERR_RUNTIME;
    } catch (...) {
        last_error = "unknown exception";
        return HY_ERR_RUNTIME;
    }
}

char *dup_str(const std::string &s)
{
    auto *p = static_cast<char *>(std::malloc(s.size() + 1u));
    if (p != nullptr) {
        std::memcpy(p, s.c_str(), s.size() + 1u);
    }
    return p;
}

template <typename F>
hy_expr make_expr(const F &f)
{
    try {
        return new hy_expr_s{f()};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}

template <typename F>
int guarded(const F &f)
{
    try {
        f();
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}

std::string dc_to_string(const taylor_dc_t &dc)
{
    std::ostringstream oss;
    for (std::size_t i = 0; i < dc.size(); ++i) {
        oss << dc[i].first.to_string();
        for (const auto d : dc[i].second) {
            oss << " [dep " << d << "]";
        }
        oss << '\n';
    }
    return oss.str();
}

std::vector<double> vec_from(const double *p, std::size_t n)
{
    if (p == nullptr || n == 0u) {
        return {};
    }
    return std::vector<double>(p, p + n);
}

// Expand a (pointer, n) kw::max_delta_t argument: 0 -> none, 1 -> splat, else as is.
std::vector<double> expand_mdt(const double *p, std::size_t n, std::uint32_t batch_size)
{
    if (p == nullptr || n == 0u) {
        return {};
    }
    if (n == 1u) {
        return std::vector<double>(batch_size, p[0]);
    }
    return std::vector<double>(p, p + n);
}

detail::tab_core::cb_t wrap_cb(hy_tab tab, hy_step_callback cb, void *cb_data)
{
    if (cb == nullptr) {
        return {};
    }
    return [tab, cb, cb_data]() { return cb(tab, cb_data) != 0; };
}

// A set of C step callbacks with optional pre-hooks (hy_step_callback_desc): every member runs at every step, the
// results are and-ed (src/step_callback.cpp:108-127). Returns the pair (call, pre_hook) for the core.
std::pair<detail::tab_core::cb_t, detail::tab_core::pre_t> wrap_cbs(hy_tab tab, const hy_step_callback_desc *cbs, size_t n_cbs)
{
    if (n_cbs == 0u || (n_cbs == 1u && cbs[0].call == nullptr)) {
        return {};
    }
    std::vector<hy_step_callback_desc> v(cbs, cbs + n_cbs);
    for (const auto &c : v) {
        if (c.call == nullptr) {
            throw std::invalid_argument("Cannot construct a callback set containing one or more empty callbacks");
        }
    }
    detail::tab_core::cb_t call = [tab, v]() {
        bool ret = true;
        for (const auto &c : v) {
            const auto rc = c.call(tab, c.user_data);
            if (rc < 0) {
                // An error inside the callback (e.g. a Python exception recorded by the binding): the other members of the
                // set do not run any more, the propagation stops.
                return false;
            }
            ret = (rc != 0) && ret;
        }
        return ret;
    };
    detail::tab_core::pre_t pre = [tab, v]() {
        for (const auto &c : v) {
            if (c.pre_hook != nullptr && c.pre_hook(tab, c.user_data) != 0) {
                throw std::runtime_error("The pre_hook() of a step callback failed: the propagation was not started");
            }
        }
    };
    return {std::move(call), std::move(pre)};
}

} // namespace

extern "C" {

const char *hy_last_error(void)
{
    return last_error.c_str();
}

int hy_last_error_code(void)
{
    return last_error_code;
}

void hy_free_str(char *s)
{
    std::free(s);
}

char *hy_version(void)
{
    int major = 0, minor = 0;
    hiprtcVersion(&major, &minor);
    // (The HIP runtime's own version as well: the kernels are compiled by whatever hiprtc the box carries - ROCm 7.2 in the
    // authoring container, 7.0.x on the driver's GPU boxes - and the bench line records it.)
    int rt = 0;
    (void)hipRuntimeGetVersion(&rt);
    return dup_str("heyoka_amd 0.1.0; target gfx950; hiprtc " + std::to_string(major) + "." + std::to_string(minor)
                   + "; hip runtime " + std::to_string(rt));
}

int hy_device_count(void)
{
    return hip_device_count();
}

// ---- expressions ----
hy_expr hy_expr_var(const char *name)
{
    return make_expr([&] { return expression{std::string(name)}; });
}
hy_expr hy_expr_num(double v)
{
    return make_expr([&] { return expression{v}; });
}
hy_expr hy_expr_par(uint32_t i)
{
    return make_expr([&] { return par[i]; });
}
hy_expr hy_expr_time(void)
{
    return make_expr([&] { return heyoka_amd::time; });
}
hy_expr hy_expr_neg(hy_expr a)
{
    return make_expr([&] { return -a->ex; });
}
hy_expr hy_expr_add(hy_expr a, hy_expr b)
{
    return make_expr([&] { return a->ex + b->ex; });
}
hy_expr hy_expr_sub(hy_expr a, hy_expr b)
{
    return make_expr([&] { return a->ex - b->ex; });
}
hy_expr hy_expr_mul(hy_expr a, hy_expr b)
{
    return make_expr([&] { return a->ex * b->ex; });
}
hy_expr hy_expr_div(hy_expr a, hy_expr b)
{
    return make_expr([&] { return a->ex / b->ex; });
}
hy_expr hy_expr_pow(hy_expr a, hy_expr b)
{
    return make_expr([&] { return pow(a->ex, b->ex); });
}
hy_expr hy_expr_atan2(hy_expr y, hy_expr x)
{
    return make_expr([&] { return atan2(y->ex, x->ex); });
}
hy_expr hy_expr_kepE(hy_expr e, hy_expr M)
{
    return make_expr([&] { return kepE(e->ex, M->ex); });
}
hy_expr hy_expr_kepF(hy_expr h, hy_expr k, hy_expr lam)
{
    return make_expr([&] { return kepF(h->ex, k->ex, lam->ex); });
}
hy_expr hy_expr_kepDE(hy_expr s0, hy_expr c0, hy_expr DM)
{
    return make_expr([&] { return kepDE(s0->ex, c0->ex, DM->ex); });
}
hy_expr hy_expr_pi(void)
{
    return make_expr([&] { return pi_constant(); });
}
hy_expr hy_expr_custom(const char *name, const hy_expr *args, size_t n)
{
    return make_expr([&] {
        std::vector<expression> v;
        for (size_t i = 0; i < n; ++i) {
            v.push_back(args[i]->ex);
        }
        return custom_func(name, std::move(v));
    });
}
int hy_node_rule_register(const hy_node_rule_desc *d)
{
    try {
        if (d == nullptr || d->name == nullptr || d->hip_source == nullptr) {
            throw std::invalid_argument("hy_node_rule_register(): null descriptor, name or source");
        }
        if ((d->n_hidden != 0u) != (d->decompose != nullptr)) {
            throw std::invalid_argument("hy_node_rule_register(): a decomposition callback is needed iff n_hidden != 0");
        }
        node_rule r;
        r.name = d->name;
        r.n_args = d->n_args;
        r.hip_source = d->hip_source;
        r.deps.assign(d->deps, d->deps + (d->deps == nullptr ? 0u : d->n_deps));
        for (const auto x : r.deps) {
            if (x >= d->n_hidden) {
                throw std::invalid_argument("hy_node_rule_register(): a hidden dependency of the node is out of range");
            }
        }
        if (d->n_hidden != 0u) {
            const auto n_hidden = d->n_hidden;
            const auto cb = d->decompose;
            auto *const ctx = d->ctx;
            std::vector<std::int32_t> hdeps(static_cast<std::size_t>(n_hidden) * 4u, -1);
            if (d->hidden_deps != nullptr) {
                hdeps.assign(d->hidden_deps, d->hidden_deps + static_cast<std::size_t>(n_hidden) * 4u);
                // (At most 4 dependencies per hidden definition, -1 = unused: checked here, not at the first decomposition.)
                for (const auto x : hdeps) {
                    if (x < -1 || x >= static_cast<std::int32_t>(n_hidden)) {
                        throw std::invalid_argument("hy_node_rule_register(): an entry of hidden_deps is out of range");
                    }
                }
            }
            const auto rname = r.name;
            r.decompose = [n_hidden, cb, ctx, hdeps, rname](const expression &self, const std::vector<expression> &args,
                                                            const std::function<expression(std::uint32_t)> &hidden) {
                hy_expr_s self_h{self};
                std::vector<hy_expr_s> arg_h, hid_h;
                for (const auto &a : args) {
                    arg_h.push_back(hy_expr_s{a});
                }
                for (std::uint32_t j = 0; j < n_hidden; ++j) {
                    hid_h.push_back(hy_expr_s{hidden(j)});
                }
                std::vector<hy_expr> arg_p, hid_p, out(n_hidden, nullptr);
                for (auto &x : arg_h) {
                    arg_p.push_back(&x);
                }
                for (auto &x : hid_h) {
                    hid_p.push_back(&x);
                }
                const auto rc = cb(ctx, &self_h, arg_p.data(), static_cast<std::uint32_t>(arg_p.size()), hid_p.data(), out.data());
                std::vector<hidden_def> defs;
                bool ok = rc == 0;
                for (std::uint32_t j = 0; j < n_hidden; ++j) {
                    if (out[j] == nullptr) {
                        ok = false;
                        continue;
                    }
                    hidden_def hd;
                    hd.ex = out[j]->ex;
                    for (unsigned q = 0; q < 4u; ++q) {
                        if (hdeps[j * 4u + q] >= 0) {
                            hd.deps.push_back(static_cast<std::uint32_t>(hdeps[j * 4u + q]));
                        }
                    }
                    defs.push_back(std::move(hd));
                    delete out[j];
                }
                if (!ok) {
                    throw std::invalid_argument("The decomposition callback of the node rule '" + rname + "' failed");
                }
                return defs;
            };
        }
        register_node_rule(std::move(r));
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}
hy_expr hy_expr_relu(hy_expr x, double slope)
{
    return make_expr([&] { return relu(x->ex, slope); });
}
hy_expr hy_expr_relup(hy_expr x, double slope)
{
    return make_expr([&] { return relup(x->ex, slope); });
}
hy_expr hy_expr_select(hy_expr c, hy_expr t, hy_expr f)
{
    return make_expr([&] { return select(c->ex, t->ex, f->ex); });
}
hy_expr hy_expr_logical(int is_and, const hy_expr *args, size_t n)
{
    return make_expr([&] {
        std::vector<expression> v;
        for (size_t i = 0; i < n; ++i) {
            v.push_back(args[i]->ex);
        }
        return is_and != 0 ? logical_and(std::move(v)) : logical_or(std::move(v));
    });
}
hy_expr hy_expr_rel(int op, hy_expr a, hy_expr b)
{
    return make_expr([&] {
        switch (op) {
            case 0:
                return eq(a->ex, b->ex);
            case 1:
                return neq(a->ex, b->ex);
            case 2:
                return lt(a->ex, b->ex);
            case 3:
                return gt(a->ex, b->ex);
            case 4:
                return lte(a->ex, b->ex);
            case 5:
                return gte(a->ex, b->ex);
            default:
                throw std::invalid_argument("Invalid relational operator code: " + std::to_string(op));
        }
    });
}
hy_expr hy_expr_sqrt(hy_expr a)
{
    return make_expr([&] { return sqrt(a->ex); });
}
hy_expr hy_expr_sin(hy_expr a)
{
    return make_expr([&] { return sin(a->ex); });
}
hy_expr hy_expr_cos(hy_expr a)
{
    return make_expr([&] { return cos(a->ex); });
}
hy_expr hy_expr_exp(hy_expr a)
{
    return make_expr([&] { return exp(a->ex); });
}
hy_expr hy_expr_log(hy_expr a)
{
    return make_expr([&] { return log(a->ex); });
}
hy_expr hy_expr_tan(hy_expr a)
{
    return make_expr([&] { return tan(a->ex); });
}
hy_expr hy_expr_tanh(hy_expr a)
{
    return make_expr([&] { return tanh(a->ex); });
}
hy_expr hy_expr_sinh(hy_expr a)
{
    return make_expr([&] { return sinh(a->ex); });
}
hy_expr hy_expr_cosh(hy_expr a)
{
    return make_expr([&] { return cosh(a->ex); });
}
hy_expr hy_expr_asin(hy_expr a)
{
    return make_expr([&] { return asin(a->ex); });
}
hy_expr hy_expr_acos(hy_expr a)
{
    return make_expr([&] { return acos(a->ex); });
}
hy_expr hy_expr_atan(hy_expr a)
{
    return make_expr([&] { return atan(a->ex); });
}
hy_expr hy_expr_asinh(hy_expr a)
{
    return make_expr([&] { return asinh(a->ex); });
}
hy_expr hy_expr_acosh(hy_expr a)
{
    return make_expr([&] { return acosh(a->ex); });
}
hy_expr hy_expr_atanh(hy_expr a)
{
    return make_expr([&] { return atanh(a->ex); });
}
hy_expr hy_expr_erf(hy_expr a)
{
    return make_expr([&] { return erf(a->ex); });
}
hy_expr hy_expr_sigmoid(hy_expr a)
{
    return make_expr([&] { return sigmoid(a->ex); });
}
hy_expr hy_expr_sum(const hy_expr *v, size_t n)
{
    return make_expr([&] {
        std::vector<expression> args;
        for (size_t i = 0; i < n; ++i) {
            args.push_back(v[i]->ex);
        }
        return sum(std::move(args));
    });
}
hy_expr hy_expr_prod(const hy_expr *v, size_t n)
{
    return make_expr([&] {
        std::vector<expression> args;
        for (size_t i = 0; i < n; ++i) {
            args.push_back(v[i]->ex);
        }
        return prod(std::move(args));
    });
}
void hy_expr_free(hy_expr e)
{
    delete e;
}
char *hy_expr_str(hy_expr e)
{
    return dup_str(e->ex.to_string());
}

// ---- systems ----
hy_sys hy_sys_new(void)
{
    return new hy_sys_s{};
}
int hy_sys_add(hy_sys s, hy_expr lhs, hy_expr rhs)
{
    return guarded([&] { s->sys.emplace_back(lhs->ex, rhs->ex); });
}
size_t hy_sys_size(hy_sys s)
{
    return s->sys.size();
}
void hy_sys_free(hy_sys s)
{
    delete s;
}
hy_sys hy_model_nbody(uint32_t n, const double *masses, size_t n_masses, double Gconst)
{
    try {
        std::vector<expression> mv;
        if (masses == nullptr) {
            mv.resize(n, expression{1.});
        } else {
            for (size_t i = 0; i < n_masses; ++i) {
                mv.emplace_back(masses[i]);
            }
        }
        return new hy_sys_s{model::detail::nbody_impl(n, expression{Gconst}, mv)};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
namespace
{

std::pair<expression, std::vector<expression>> model_args(uint32_t n, const hy_expr *masses, size_t n_masses,
                                                          hy_expr Gconst)
{
    std::vector<expression> mv;
    if (masses == nullptr) {
        mv.resize(n, expression{1.});
    } else {
        for (size_t i = 0; i < n_masses; ++i) {
            mv.push_back(masses[i]->ex);
        }
    }
    return {Gconst == nullptr ? expression{1.} : Gconst->ex, std::move(mv)};
}

} // namespace

hy_sys hy_model_nbody_ex(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    try {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return new hy_sys_s{model::detail::nbody_impl(n, G, mv)};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
hy_expr hy_model_nbody_energy(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    return make_expr([&] {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return model::detail::nbody_energy_impl(n, G, mv);
    });
}
hy_expr hy_model_nbody_potential(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    return make_expr([&] {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return model::detail::nbody_potential_impl(n, G, mv);
    });
}
extern "C++" {
namespace
{

std::vector<expression> expr_vector(const hy_expr *v, size_t n)
{
    std::vector<expression> ret;
    for (size_t i = 0; v != nullptr && i < n; ++i) {
        ret.push_back(v[i]->ex);
    }
    return ret;
}

template <typename F>
hy_sys make_sys(F &&f)
{
    try {
        return new hy_sys_s{f()};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}

expression or_default(hy_expr e, double def)
{
    return e == nullptr ? expression{def} : e->ex;
}

} // namespace
} // extern "C++"

hy_sys hy_model_np1body(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    return make_sys([&] {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return model::detail::np1body_impl(n, G, mv);
    });
}
hy_expr hy_model_np1body_energy(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    return make_expr([&] {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return model::detail::np1body_energy_impl(n, G, mv);
    });
}
hy_expr hy_model_np1body_potential(uint32_t n, const hy_expr *masses, size_t n_masses, hy_expr Gconst)
{
    return make_expr([&] {
        auto [G, mv] = model_args(n, masses, n_masses, Gconst);
        return model::detail::np1body_potential_impl(n, G, mv);
    });
}
hy_sys hy_model_cr3bp(hy_expr mu)
{
    return make_sys([&] { return model::detail::cr3bp_impl(or_default(mu, 1e-3)); });
}
hy_expr hy_model_cr3bp_jacobi(hy_expr mu)
{
    return make_expr([&] { return model::detail::cr3bp_jacobi_impl(or_default(mu, 1e-3)); });
}
hy_sys hy_model_fixed_centres(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                              size_t n_positions)
{
    return make_sys([&] {
        return model::detail::fixed_centres_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                                 expr_vector(positions, n_positions));
    });
}
hy_expr hy_model_fixed_centres_energy(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                                      size_t n_positions)
{
    return make_expr([&] {
        return model::detail::fixed_centres_energy_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                                        expr_vector(positions, n_positions));
    });
}
hy_expr hy_model_fixed_centres_potential(hy_expr Gconst, const hy_expr *masses, size_t n_masses,
                                         const hy_expr *positions, size_t n_positions)
{
    return make_expr([&] {
        return model::detail::fixed_centres_potential_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                                           expr_vector(positions, n_positions));
    });
}
hy_sys hy_model_rotating(const hy_expr *omega, size_t n_omega)
{
    return make_sys([&] { return model::detail::rotating_impl(expr_vector(omega, n_omega)); });
}
hy_expr hy_model_rotating_energy(const hy_expr *omega, size_t n_omega)
{
    return make_expr([&] { return model::detail::rotating_energy_impl(expr_vector(omega, n_omega)); });
}
hy_expr hy_model_rotating_potential(const hy_expr *omega, size_t n_omega)
{
    return make_expr([&] { return model::detail::rotating_potential_impl(expr_vector(omega, n_omega)); });
}
hy_sys hy_model_mascon(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                       size_t n_positions, const hy_expr *omega, size_t n_omega)
{
    return make_sys([&] {
        return model::detail::mascon_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                          expr_vector(positions, n_positions), expr_vector(omega, n_omega));
    });
}
hy_expr hy_model_mascon_energy(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                               size_t n_positions, const hy_expr *omega, size_t n_omega)
{
    return make_expr([&] {
        return model::detail::mascon_energy_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                                 expr_vector(positions, n_positions), expr_vector(omega, n_omega));
    });
}
hy_expr hy_model_mascon_potential(hy_expr Gconst, const hy_expr *masses, size_t n_masses, const hy_expr *positions,
                                  size_t n_positions, const hy_expr *omega, size_t n_omega)
{
    return make_expr([&] {
        return model::detail::mascon_potential_impl(or_default(Gconst, 1.), expr_vector(masses, n_masses),
                                                    expr_vector(positions, n_positions), expr_vector(omega, n_omega));
    });
}
hy_expr hy_model_pendulum_energy(double gconst, double length)
{
    return make_expr([&] { return model::detail::pendulum_energy_impl(expression{gconst}, expression{length}); });
}
int hy_sys_get_vars(hy_sys s, hy_expr *out)
{
    return guarded([&] {
        for (std::size_t i = 0; i < s->sys.size(); ++i) {
            out[i] = new hy_expr_s{s->sys[i].first};
        }
    });
}
hy_sys hy_model_pendulum(double gconst, double length)
{
    try {
        return new hy_sys_s{model::detail::pendulum_impl(expression{gconst}, expression{length})};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
char *hy_sys_decomposition_str(hy_sys s)
{
    try {
        validate_ode_sys(s->sys);
        return dup_str(dc_to_string(taylor_decompose_sys(s->sys)));
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}

// ---- integrator ----
hy_tab hy_tab_create(hy_sys sys, const double *state, size_t n_state, uint32_t batch_size, const hy_tab_config *cfg)
{
    return hy_tab_create_with_events(sys, state, n_state, batch_size, cfg, nullptr, 0, nullptr, 0);
}
hy_tab hy_tab_create_with_events(hy_sys sys, const double *state, size_t n_state, uint32_t batch_size,
                                 const hy_tab_config *cfg, const hy_t_event *tes, size_t n_tes, const hy_nt_event *ntes,
                                 size_t n_ntes)
{
    try {
        detail::tab_core::config c;
        for (size_t i = 0; i < n_tes; ++i) {
            detail::core_t_event e;
            e.eq = tes[i].eq->ex;
            e.dir = static_cast<event_direction>(tes[i].direction);
            e.cooldown = tes[i].cooldown;
            if (tes[i].cb != nullptr) {
                const auto cb = tes[i].cb;
                auto *const user = tes[i].user;
                e.callback = [cb, user](void *ctx, int d_sgn, std::uint32_t idx) {
                    return cb(static_cast<hy_tab>(ctx), d_sgn, idx, user) != 0;
                };
                if (cb == &hy_event_counter_t && user != nullptr) {
                    e.native_counter = static_cast<std::uint64_t *>(user);
                }
            }
            c.t_events.push_back(std::move(e));
        }
        for (size_t i = 0; i < n_ntes; ++i) {
            detail::core_nt_event e;
            e.eq = ntes[i].eq->ex;
            e.dir = static_cast<event_direction>(ntes[i].direction);
            if (ntes[i].cb != nullptr) {
                const auto cb = ntes[i].cb;
                auto *const user = ntes[i].user;
                e.callback = [cb, user](void *ctx, double tm, int d_sgn, std::uint32_t idx) {
                    cb(static_cast<hy_tab>(ctx), tm, d_sgn, idx, user);
                };
                if (cb == &hy_event_counter_nt && user != nullptr) {
                    e.native_counter = static_cast<std::uint64_t *>(user);
                }
            }
            c.nt_events.push_back(std::move(e));
        }
        if (cfg != nullptr) {
            if (cfg->tol != 0) {
                c.tol = cfg->tol;
            }
            c.high_accuracy = cfg->high_accuracy != 0;
            c.compact_mode = cfg->compact_mode != 0;
            c.parallel_mode = cfg->parallel_mode != 0;
            c.pars = vec_from(cfg->pars, cfg->n_pars);
            c.time = vec_from(cfg->time, cfg->n_time);
            c.time_is_scalar = (cfg->n_time == 1u && batch_size != 1u) || (cfg->n_time == 1u);
            c.device = cfg->device;
            c.emitter = cfg->emitter;
            c.cluster_kernel = cfg->cluster_kernel;
            c.exact_division = cfg->exact_division != 0;
            c.sum_order = cfg->sum_order;
            c.events_on_cluster = cfg->events_on_cluster;
            c.batch_semantics = cfg->batch_semantics;
        }
        auto *ret = new hy_tab_s{detail::tab_core(sys->sys, vec_from(state, n_state), batch_size, std::move(c))};
        // The event callbacks receive the handle itself.
        ret->core.set_callback_context(ret);
        return ret;
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
int hy_tab_set_event_timing(hy_tab t, int on)
{
    try {
        t->core.set_event_timing(on != 0);
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}

int hy_tab_get_event_stats(hy_tab t, double *out8)
{
    try {
        const auto st = t->core.get_event_stats();
        std::copy(st.begin(), st.end(), out8);
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}

// Native callbacks which count their invocations in the 64-bit integer behind `user` (the callbacks of an integrator run
// serially on the host).
// (Copies of an integrator made by ensemble_propagate_*() share `user` and run on one host thread per device: atomic.)
void hy_event_counter_nt(hy_tab, double, int, uint32_t, void *user)
{
    __atomic_fetch_add(static_cast<std::uint64_t *>(user), std::uint64_t(1), __ATOMIC_RELAXED);
}
int hy_event_counter_t(hy_tab, int, uint32_t, void *user)
{
    __atomic_fetch_add(static_cast<std::uint64_t *>(user), std::uint64_t(1), __ATOMIC_RELAXED);
    return 1;
}

// Logger (include/heyoka/logging.hpp:19-24): level 0 trace ... 5 critical, 6 off; default 3 (warn). A sink of the caller's
// receives (level, message) instead of stderr.
int hy_set_logger_level(int level)
{
    return guarded([&] {
        if (level < 0 || level > 6) {
            throw std::invalid_argument("Invalid logger level: " + std::to_string(level) + " (0 trace ... 5 critical, 6 off)");
        }
        heyoka_amd::set_logger_level(static_cast<heyoka_amd::log_level>(level));
    });
}
int hy_get_logger_level(void)
{
    return static_cast<int>(heyoka_amd::get_logger_level());
}
void hy_set_log_callback(hy_log_callback_t cb, void *user)
{
    heyoka_amd::set_log_sink(cb, user);
}

// Build id of the library: the first 16 hex digits of the SHA-256 of its sources (Makefile, _build/build_id.h).
const char *hy_build_id(void)
{
    return HY_BUILD_ID;
}

int hy_tab_with_events(hy_tab t)
{
    return t->core.with_events() ? 1 : 0;
}
int hy_tab_reset_cooldowns(hy_tab t, int64_t batch_idx)
{
    return guarded([&] {
        if (batch_idx < 0) {
            t->core.reset_cooldowns();
        } else {
            t->core.reset_cooldowns(static_cast<std::uint32_t>(batch_idx));
        }
    });
}
int hy_tab_get_te_cooldowns(hy_tab t, double *first, double *second, int *active)
{
    return guarded([&] {
        const auto &cds = t->core.get_te_cooldowns();
        for (std::size_t i = 0; i < cds.size(); ++i) {
            for (std::size_t e = 0; e < cds[i].size(); ++e) {
                const auto k = i * cds[i].size() + e;
                active[k] = cds[i][e] ? 1 : 0;
                first[k] = cds[i][e] ? cds[i][e]->first : 0.;
                second[k] = cds[i][e] ? cds[i][e]->second : 0.;
            }
        }
    });
}
hy_tab hy_tab_copy(hy_tab t)
{
    try {
        auto *ret = new hy_tab_s{detail::tab_core(t->core)};
        ret->core.set_callback_context(ret);
        return ret;
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
void hy_tab_free(hy_tab t)
{
    delete t;
}

uint32_t hy_tab_get_batch_size(hy_tab t)
{
    return t->core.get_batch_size();
}
uint32_t hy_tab_get_order(hy_tab t)
{
    return t->core.get_order();
}
uint32_t hy_tab_get_dim(hy_tab t)
{
    return t->core.get_dim();
}
uint32_t hy_tab_get_n_pars(hy_tab t)
{
    return t->core.get_program().n_par;
}
uint32_t hy_tab_get_n_uvars(hy_tab t)
{
    return t->core.get_program().n_u;
}
double hy_tab_get_tol(hy_tab t)
{
    return t->core.get_tol();
}
int hy_tab_get_high_accuracy(hy_tab t)
{
    return t->core.get_high_accuracy() ? 1 : 0;
}
unsigned long long hy_tab_get_event_detection_failures(hy_tab t)
{
    return t->core.get_event_detection_failures();
}

int hy_tab_get_compact_mode(hy_tab t)
{
    return t->core.get_compact_mode() ? 1 : 0;
}
double hy_tab_get_compile_seconds(hy_tab t)
{
    return t->core.get_compile_seconds();
}
char *hy_tab_get_hip_source(hy_tab t)
{
    return dup_str(t->core.get_hip_source());
}
char *hy_tab_get_internal_program(hy_tab t)
{
    return dup_str(t->core.get_internal_program());
}
int hy_tab_get_code_object(hy_tab t, void *out, size_t *size)
{
    return guarded([&] {
        const auto &co = t->core.get_code_object();
        if (out != nullptr) {
            if (*size < co.size()) {
                throw std::invalid_argument("hy_tab_get_code_object(): the output buffer is too small");
            }
            std::memcpy(out, co.data(), co.size());
        }
        *size = co.size();
    });
}
int hy_hiprtc_compile(const char *source, void *out, size_t *size)
{
    return guarded([&] {
        const auto cm = hiprtc_compile_source(source);
        if (out != nullptr) {
            if (*size < cm->code.size()) {
                throw std::invalid_argument("hy_hiprtc_compile(): the output buffer is too small");
            }
            std::memcpy(out, cm->code.data(), cm->code.size());
        }
        *size = cm->code.size();
    });
}
char *hy_tab_get_codegen_info(hy_tab t)
{
    return dup_str(t->core.get_codegen_info());
}
char *hy_tab_get_decomposition_str(hy_tab t)
{
    return dup_str(dc_to_string(t->core.get_decomposition()));
}

int hy_tab_get_state(hy_tab t, double *out)
{
    return guarded([&] {
        const auto &s = t->core.get_state();
        std::memcpy(out, s.data(), s.size() * sizeof(double));
    });
}
int hy_tab_set_state(hy_tab t, const double *in)
{
    return guarded([&] {
        t->core.set_state_values(in);
    });
}
double *hy_tab_get_state_data(hy_tab t)
{
    double *ret = nullptr;
    (void)guarded([&] { ret = t->core.get_state_data(); });
    return ret;
}
double *hy_tab_get_pars_data(hy_tab t)
{
    double *ret = nullptr;
    (void)guarded([&] { ret = t->core.get_pars_data(); });
    return ret;
}
int hy_tab_get_pars(hy_tab t, double *out)
{
    return guarded([&] {
        const auto &s = t->core.get_pars();
        std::memcpy(out, s.data(), s.size() * sizeof(double));
    });
}
int hy_tab_set_pars(hy_tab t, const double *in)
{
    return guarded([&] {
        t->core.set_pars_values(in);
    });
}
int hy_tab_get_dtime(hy_tab t, double *hi, double *lo)
{
    return guarded([&] {
        const auto p = t->core.get_dtime();
        std::memcpy(hi, p.first.data(), p.first.size() * sizeof(double));
        if (lo != nullptr) {
            std::memcpy(lo, p.second.data(), p.second.size() * sizeof(double));
        }
    });
}
int hy_tab_set_time(hy_tab t, const double *tm, size_t n)
{
    return guarded([&] {
        if (n == 1u) {
            t->core.set_time(tm[0]);
        } else {
            t->core.set_time(vec_from(tm, n));
        }
    });
}
int hy_tab_set_dtime(hy_tab t, const double *hi, const double *lo, size_t n)
{
    return guarded([&] {
        if (n == 1u) {
            t->core.set_dtime(hi[0], lo[0]);
        } else {
            t->core.set_dtime(vec_from(hi, n), vec_from(lo, n));
        }
    });
}
int hy_tab_get_tc(hy_tab t, double *out)
{
    return guarded([&] {
        const auto &s = t->core.get_tc();
        std::memcpy(out, s.data(), s.size() * sizeof(double));
    });
}
int hy_tab_get_last_h(hy_tab t, double *out)
{
    return guarded([&] {
        const auto &s = t->core.get_last_h();
        std::memcpy(out, s.data(), s.size() * sizeof(double));
    });
}
int hy_tab_update_d_output(hy_tab t, const double *tm, size_t n, int rel_time, double *out)
{
    return guarded([&] {
        const auto &s = (n == 1u) ? t->core.update_d_output(tm[0], rel_time != 0)
                                  : t->core.update_d_output(vec_from(tm, n), rel_time != 0);
        std::memcpy(out, s.data(), s.size() * sizeof(double));
    });
}

int hy_tab_step(hy_tab t, int wtc)
{
    return guarded([&] { t->core.step(wtc != 0); });
}
int hy_tab_step_backward(hy_tab t, int wtc)
{
    return guarded([&] { t->core.step_backward(wtc != 0); });
}
int hy_tab_step_limited(hy_tab t, const double *mdts, size_t n, int wtc)
{
    return guarded([&] { t->core.step(vec_from(mdts, n), wtc != 0); });
}
int hy_tab_get_step_res(hy_tab t, int64_t *outcome, double *h)
{
    return guarded([&] {
        const auto &r = t->core.get_step_res();
        for (std::size_t i = 0; i < r.size(); ++i) {
            outcome[i] = static_cast<int64_t>(std::get<0>(r[i]));
            h[i] = std::get<1>(r[i]);
        }
    });
}

int hy_tab_propagate_until(hy_tab t, const double *ts, size_t n_ts, uint64_t max_steps, const double *mdts,
                           size_t n_mdt, hy_step_callback cb, void *cb_data, int wtc, int c_out)
{
    return guarded([&] {
        t->core.propagate_until(vec_from(ts, n_ts), static_cast<std::size_t>(max_steps),
                                expand_mdt(mdts, n_mdt, t->core.get_batch_size()), wrap_cb(t, cb, cb_data), wtc != 0,
                                c_out != 0);
    });
}
int hy_tab_propagate_for(hy_tab t, const double *dts, size_t n_dts, uint64_t max_steps, const double *mdts,
                         size_t n_mdt, hy_step_callback cb, void *cb_data, int wtc, int c_out)
{
    return guarded([&] {
        t->core.propagate_for(vec_from(dts, n_dts), static_cast<std::size_t>(max_steps),
                              expand_mdt(mdts, n_mdt, t->core.get_batch_size()), wrap_cb(t, cb, cb_data), wtc != 0,
                              c_out != 0);
    });
}
int hy_tab_propagate_until_cbs(hy_tab t, const double *ts, size_t n_ts, uint64_t max_steps, const double *mdts, size_t n_mdt,
                               const hy_step_callback_desc *cbs, size_t n_cbs, int wtc, int c_out)
{
    return guarded([&] {
        auto [call, pre] = wrap_cbs(t, cbs, n_cbs);
        t->core.propagate_until(vec_from(ts, n_ts), static_cast<std::size_t>(max_steps),
                                expand_mdt(mdts, n_mdt, t->core.get_batch_size()), call, wtc != 0, c_out != 0, pre);
    });
}
int hy_tab_propagate_for_cbs(hy_tab t, const double *dts, size_t n_dts, uint64_t max_steps, const double *mdts, size_t n_mdt,
                             const hy_step_callback_desc *cbs, size_t n_cbs, int wtc, int c_out)
{
    return guarded([&] {
        auto [call, pre] = wrap_cbs(t, cbs, n_cbs);
        t->core.propagate_for(vec_from(dts, n_dts), static_cast<std::size_t>(max_steps),
                              expand_mdt(mdts, n_mdt, t->core.get_batch_size()), call, wtc != 0, c_out != 0, pre);
    });
}
int hy_tab_propagate_grid_cbs(hy_tab t, const double *grid, size_t n_grid, uint64_t max_steps, const double *mdts,
                              size_t n_mdt, const hy_step_callback_desc *cbs, size_t n_cbs, double *out)
{
    return guarded([&] {
        const auto bs = t->core.get_batch_size();
        auto [call, pre] = wrap_cbs(t, cbs, n_cbs);
        auto ret = t->core.propagate_grid(vec_from(grid, n_grid * bs), static_cast<std::size_t>(max_steps),
                                          expand_mdt(mdts, n_mdt, bs), call, nullptr, pre);
        std::memcpy(out, ret.data(), ret.size() * sizeof(double));
    });
}
int hy_tab_take_c_output(hy_tab t, hy_cout *out)
{
    return guarded([&] {
        *out = nullptr;
        auto c = t->core.take_c_output();
        if (c) {
            *out = new hy_cout_s{std::move(*c)};
        }
    });
}
void hy_cout_free(hy_cout c)
{
    delete c;
}
hy_cout hy_cout_clone(hy_cout c)
{
    try {
        return new hy_cout_s{c->core};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
int hy_cout_eval(hy_cout c, const double *tm, size_t n_tm, double *out)
{
    return guarded([&] {
        const auto &r = (n_tm == 1u && c->core.get_batch_size() != 1u) ? c->core.call(tm[0])
                                                                       : c->core.call(vec_from(tm, n_tm));
        std::memcpy(out, r.data(), r.size() * sizeof(double));
    });
}
int hy_cout_eval_device(hy_cout c, const double *d_tm, double *d_out)
{
    return guarded([&] { c->core.call_device(d_tm, d_out); });
}
uint32_t hy_cout_get_batch_size(hy_cout c)
{
    return c->core.get_batch_size();
}
uint32_t hy_cout_get_dim(hy_cout c)
{
    return c->core.get_dim();
}
uint32_t hy_cout_get_order(hy_cout c)
{
    return c->core.get_order();
}
int hy_cout_get_n_steps(hy_cout c, size_t *n)
{
    return guarded([&] { *n = c->core.get_n_steps(); });
}
int hy_cout_get_bounds(hy_cout c, double *lb, double *ub)
{
    return guarded([&] {
        const auto [l, u] = c->core.get_bounds();
        std::memcpy(lb, l.data(), l.size() * sizeof(double));
        std::memcpy(ub, u.data(), u.size() * sizeof(double));
    });
}
int hy_cout_get_times(hy_cout c, double *hi, double *lo)
{
    return guarded([&] {
        const auto &h = c->core.get_times();
        std::memcpy(hi, h.data(), h.size() * sizeof(double));
        if (lo != nullptr) {
            const auto &l = c->core.get_times_lo();
            std::memcpy(lo, l.data(), l.size() * sizeof(double));
        }
    });
}
int hy_cout_get_tcs(hy_cout c, double *out)
{
    return guarded([&] {
        const auto &t = c->core.get_tcs();
        std::memcpy(out, t.data(), t.size() * sizeof(double));
    });
}
char *hy_cout_to_string(hy_cout c)
{
    try {
        std::ostringstream oss;
        c->core.stream_to(oss);
        return dup_str(oss.str());
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
int hy_compile_aux_kernels(uint32_t order, uint32_t dim, int high_accuracy)
{
    return guarded([&] {
        hiprtc_compile_source(detail::make_cout_source(order, dim, high_accuracy != 0));
        hiprtc_compile_source(detail::make_grid_source(order, dim, high_accuracy != 0));
        hiprtc_compile_source(detail::make_event_detection_source(order, detail::ed_max_detected(order, 1, 1)));
    });
}
hy_cfunc hy_cfunc_new(const hy_expr *fn, size_t n_fn, const hy_expr *vars, size_t n_vars, int device)
{
    try {
        std::vector<expression> f, v;
        for (size_t i = 0; i < n_fn; ++i) {
            f.push_back(fn[i]->ex);
        }
        for (size_t i = 0; i < n_vars; ++i) {
            v.push_back(vars[i]->ex);
        }
        return new hy_cfunc_s{detail::cfunc_core(std::move(f), std::move(v), device)};
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
void hy_cfunc_free(hy_cfunc c)
{
    delete c;
}
uint32_t hy_cfunc_get_nparams(hy_cfunc c)
{
    return c->core.get_nparams();
}
uint32_t hy_cfunc_get_nvars(hy_cfunc c)
{
    return c->core.get_nvars();
}
uint32_t hy_cfunc_get_nouts(hy_cfunc c)
{
    return c->core.get_nouts();
}
int hy_cfunc_is_time_dependent(hy_cfunc c)
{
    return c->core.is_time_dependent() ? 1 : 0;
}
char *hy_cfunc_decomposition_str(hy_cfunc c)
{
    try {
        return dup_str(dc_to_string(c->core.get_dc()));
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
char *hy_cfunc_get_hip_source(hy_cfunc c)
{
    try {
        return dup_str(c->core.get_hip_source());
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
int hy_cfunc_set_stream(hy_cfunc c, void *stream)
{
    return guarded([&] { c->core.set_stream(stream); });
}
int hy_cfunc_eval(hy_cfunc c, double *out, size_t out_size, const double *in, size_t in_size, const double *pars,
                  size_t pars_size, const double *time, size_t time_size)
{
    return guarded([&] { c->core.call_host(out, out_size, in, in_size, pars, pars_size, time, time_size); });
}
int hy_cfunc_eval_device(hy_cfunc c, double *d_out, const double *d_in, const double *d_pars, const double *d_time,
                         uint64_t nevals)
{
    return guarded([&] { c->core.call_device(d_out, d_in, d_pars, d_time, nevals); });
}
int hy_tab_propagate_grid(hy_tab t, const double *grid, size_t n_grid, uint64_t max_steps, const double *mdts,
                          size_t n_mdt, hy_step_callback cb, void *cb_data, double *out)
{
    return guarded([&] {
        const auto bs = t->core.get_batch_size();
        auto ret = t->core.propagate_grid(vec_from(grid, n_grid * bs), static_cast<std::size_t>(max_steps),
                                          expand_mdt(mdts, n_mdt, bs), wrap_cb(t, cb, cb_data));
        std::memcpy(out, ret.data(), ret.size() * sizeof(double));
    });
}
int hy_tab_propagate_grid_device(hy_tab t, const double *grid, size_t n_grid, int scalar_grid, uint64_t max_steps,
                                 const double *mdts, size_t n_mdt, double *d_out)
{
    return guarded([&] {
        const auto bs = t->core.get_batch_size();
        std::vector<double> g;
        if (scalar_grid != 0) {
            g.reserve(n_grid * bs);
            for (size_t k = 0; k < n_grid; ++k) {
                g.insert(g.end(), bs, grid[k]);
            }
        } else {
            g = vec_from(grid, n_grid * bs);
        }
        if (d_out == nullptr) {
            throw std::invalid_argument("hy_tab_propagate_grid_device(): the device output pointer is null");
        }
        t->core.propagate_grid(std::move(g), static_cast<std::size_t>(max_steps), expand_mdt(mdts, n_mdt, bs), {}, d_out);
    });
}
int hy_tab_get_propagate_res(hy_tab t, int64_t *outcome, double *min_h, double *max_h, uint64_t *n_steps)
{
    return guarded([&] {
        const auto &r = t->core.get_propagate_res();
        for (std::size_t i = 0; i < r.size(); ++i) {
            outcome[i] = static_cast<int64_t>(std::get<0>(r[i]));
            min_h[i] = std::get<1>(r[i]);
            max_h[i] = std::get<2>(r[i]);
            n_steps[i] = std::get<3>(r[i]);
        }
    });
}

void *hy_tab_device_ptr(hy_tab t, int which)
{
    try {
        switch (which) {
            case HY_BUF_STATE:
                return t->core.device_state();
            case HY_BUF_PARS:
                return t->core.device_pars();
            case HY_BUF_TIME_HI:
                return t->core.device_time_hi();
            case HY_BUF_TIME_LO:
                return t->core.device_time_lo();
            case HY_BUF_TC:
                return t->core.device_tc();
            case HY_BUF_N_STEPS:
                return t->core.device_aux(0);
            case HY_BUF_OUTCOME:
                return t->core.device_aux(1);
            case HY_BUF_LAST_H:
                return t->core.device_aux(2);
            default:
                throw std::invalid_argument("Invalid device buffer identifier: " + std::to_string(which));
        }
    } catch (...) {
        handle_exception();
        return nullptr;
    }
}
int hy_tab_mark_device_modified(hy_tab t)
{
    return guarded([&] { t->core.mark_device_modified(); });
}
int hy_tab_set_stream(hy_tab t, void *s)
{
    return guarded([&] { t->core.set_stream(s); });
}
int hy_tab_synchronize(hy_tab t)
{
    return guarded([&] { t->core.synchronize(); });
}
uint64_t hy_tab_get_last_total_steps(hy_tab t)
{
    try {
        return t->core.get_last_total_steps();
    } catch (...) {
        handle_exception();
        return 0;
    }
}
size_t hy_tab_get_kernel_ms_history(hy_tab t, double *out, size_t n)
{
    try {
        const auto v = t->core.get_kernel_ms_history(n);
        for (std::size_t i = 0; i < v.size(); ++i) {
            out[i] = v[i];
        }
        return v.size();
    } catch (...) {
        handle_exception();
        return 0;
    }
}
int hy_tab_raw_step(hy_tab t, double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                    uint64_t n)
{
    return guarded([&] { t->core.raw_step(d_state, d_pars, d_time, d_h, d_tc, n); });
}

int hy_tab_raw_step_tape(hy_tab t, double *d_state, const double *d_pars, const double *d_time, double *d_h, double *d_tc,
                         void *d_tape, uint64_t n)
{
    return guarded([&] { t->core.raw_step(d_state, d_pars, d_time, d_h, d_tc, n, d_tape); });
}

int hy_tab_raw_step_e(hy_tab t, double *d_jet, const double *d_state, const double *d_pars, const double *d_time, double *d_h,
                      double *d_max_abs_state, uint64_t n)
{
    return guarded([&] { t->core.raw_step_e(d_jet, d_state, d_pars, d_time, d_h, d_max_abs_state, n, nullptr); });
}

int hy_tab_raw_step_e_tape(hy_tab t, double *d_jet, const double *d_state, const double *d_pars, const double *d_time,
                           double *d_h, double *d_max_abs_state, void *d_tape, uint64_t n)
{
    return guarded([&] { t->core.raw_step_e(d_jet, d_state, d_pars, d_time, d_h, d_max_abs_state, n, d_tape); });
}

int hy_tab_raw_d_out_f(hy_tab t, double *d_out, const double *d_tc, const double *d_h, uint64_t n)
{
    return guarded([&] { t->core.raw_d_out_f(d_out, d_tc, d_h, n); });
}

int hy_tab_tape_size_align(hy_tab t, uint64_t n, size_t *size, size_t *align)
{
    return guarded([&] {
        const auto sa = t->core.raw_tape_size_align(n);
        if (size != nullptr) {
            *size = sa.first;
        }
        if (align != nullptr) {
            *align = sa.second;
        }
    });
}

static int ensemble_impl(hy_tab ta, double tm, size_t n_iter, hy_ensemble_gen gen, void *gen_data, uint64_t max_steps,
                         int n_devices, hy_tab *out, detail::ensemble_kind kind)
{
    return guarded([&] {
        // NOTE: the generator receives a full-blown handle so that it can use the whole C API.
        std::vector<hy_tab> handles;
        auto res = detail::ensemble_propagate_core(
            ta->core, tm, n_iter,
            [&](detail::tab_core &c, std::size_t i) {
                if (gen != nullptr) {
                    hy_tab_s tmp{std::move(c)};
                    const auto rc = gen(&tmp, i, gen_data);
                    c = std::move(tmp.core);
                    if (rc != 0) {
                        throw std::runtime_error("The generator of an ensemble propagation returned the error code "
                                                 + std::to_string(rc) + " at iteration " + std::to_string(i));
                    }
                }
            },
            static_cast<std::size_t>(max_steps), n_devices, kind);
        for (std::size_t i = 0; i < n_iter; ++i) {
            out[i] = new hy_tab_s{std::move(res[i])};
        }
    });
}

int hy_ensemble_propagate_until_batch(hy_tab ta, double tm, size_t n_iter, hy_ensemble_gen gen, void *gen_data,
                                      uint64_t max_steps, int n_devices, hy_tab *out)
{
    return ensemble_impl(ta, tm, n_iter, gen, gen_data, max_steps, n_devices, out, detail::ensemble_kind::until);
}
int hy_ensemble_propagate_for_batch(hy_tab ta, double dt, size_t n_iter, hy_ensemble_gen gen, void *gen_data,
                                    uint64_t max_steps, int n_devices, hy_tab *out)
{
    return ensemble_impl(ta, dt, n_iter, gen, gen_data, max_steps, n_devices, out, detail::ensemble_kind::for_);
}

int hy_ensemble_gather_states(const hy_tab *tabs, size_t n, int dst_device, double *out, size_t out_doubles,
                              int out_is_device, int *used_rccl)
{
    try {
        std::vector<detail::tab_core *> cores;
        for (size_t i = 0; i < n; ++i) {
            cores.push_back(&tabs[i]->core);
        }
        const auto g = detail_gather(cores, dst_device);
        if (out_doubles < g.dim() * g.n_total()) {
            throw std::invalid_argument("hy_ensemble_gather_states(): the output buffer holds " + std::to_string(out_doubles)
                                        + " doubles, " + std::to_string(g.dim() * g.n_total()) + " are needed");
        }
        if (used_rccl != nullptr) {
            *used_rccl = g.used_rccl() ? 1 : 0;
        }
        if (g.n_total() != 0u) {
            if (out_is_device != 0) {
                if (hipMemcpy(out, g.data(), g.dim() * g.n_total() * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess) {
                    throw std::runtime_error("heyoka_amd: copy of the gathered states failed");
                }
            } else {
                const auto h = g.to_host();
                std::memcpy(out, h.data(), h.size() * sizeof(double));
            }
        }
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}

int hy_ensemble_gather_results(const hy_tab *tabs, size_t n, int dst_device, double *out, size_t out_words, int out_is_device,
                               int *used_rccl)
{
    try {
        std::vector<detail::tab_core *> cores;
        for (size_t i = 0; i < n; ++i) {
            cores.push_back(&tabs[i]->core);
        }
        const auto g = detail_gather(cores, dst_device);
        const auto need = g.n_rows() * g.n_total();
        if (out_words < need) {
            throw std::invalid_argument("hy_ensemble_gather_results(): the output buffer holds " + std::to_string(out_words)
                                        + " 8-byte words, " + std::to_string(need) + " are needed");
        }
        if (used_rccl != nullptr) {
            *used_rccl = g.used_rccl() ? 1 : 0;
        }
        if (g.n_total() != 0u) {
            if (out_is_device != 0) {
                if (hipMemcpy(out, g.data(), need * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess) {
                    throw std::runtime_error("heyoka_amd: copy of the gathered results failed");
                }
            } else {
                const auto h = g.all_to_host();
                std::memcpy(out, h.data(), h.size() * sizeof(double));
            }
        }
        return HY_OK;
    } catch (...) {
        return handle_exception();
    }
}

} // extern "C"
