// ODE right-hand-side builders: the benchmark configurations (reference: src/model/nbody.cpp:53-174,
// include/heyoka/model/nbody.hpp:33-78, src/model/pendulum.cpp:23-33, include/heyoka/model/pendulum.hpp) and the
// other point-mass models of SURVEY section 8f-4: np1body (src/model/nbody.cpp:236-445), cr3bp
// (src/model/cr3bp.cpp), fixed_centres, rotating, mascon (src/model/{fixed_centres,rotating,mascon}.cpp), each with
// its conserved quantity (energy / potential / Jacobi constant) for compiled-function monitors.
#pragma once

#include <cstdint>
#include <tuple>
#include <utility>
#include <vector>

#include "expression.hpp"
#include "kw.hpp"

namespace heyoka_amd::model
{

namespace detail
{

std::vector<std::pair<expression, expression>> nbody_impl(std::uint32_t n, const expression &Gconst,
                                                          const std::vector<expression> &masses);
expression nbody_potential_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses);
expression nbody_energy_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses);
std::vector<std::pair<expression, expression>> pendulum_impl(const expression &gconst, const expression &length);
expression pendulum_energy_impl(const expression &gconst, const expression &length);
std::vector<std::pair<expression, expression>> np1body_impl(std::uint32_t n, const expression &Gconst,
                                                            const std::vector<expression> &masses);
expression np1body_potential_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses);
expression np1body_energy_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses);
std::vector<std::pair<expression, expression>> cr3bp_impl(const expression &mu);
expression cr3bp_jacobi_impl(const expression &mu);
std::vector<std::pair<expression, expression>>
fixed_centres_impl(const expression &G, const std::vector<expression> &masses, const std::vector<expression> &positions);
expression fixed_centres_energy_impl(const expression &G, const std::vector<expression> &masses,
                                     const std::vector<expression> &positions);
expression fixed_centres_potential_impl(const expression &G, const std::vector<expression> &masses,
                                        const std::vector<expression> &positions);
std::vector<std::pair<expression, expression>> rotating_impl(const std::vector<expression> &omega);
expression rotating_energy_impl(const std::vector<expression> &omega);
expression rotating_potential_impl(const std::vector<expression> &omega);
std::vector<std::pair<expression, expression>> mascon_impl(const expression &G, const std::vector<expression> &masses,
                                                           const std::vector<expression> &positions,
                                                           const std::vector<expression> &omega);
expression mascon_energy_impl(const expression &G, const std::vector<expression> &masses,
                              const std::vector<expression> &positions, const std::vector<expression> &omega);
expression mascon_potential_impl(const expression &G, const std::vector<expression> &masses,
                                 const std::vector<expression> &positions, const std::vector<expression> &omega);

// kw::name = {range of numbers / expressions} -> vector of expressions (empty if absent).
template <typename Tag, typename... KwArgs>
std::vector<expression> kw_expr_vector(kw::named_arg<Tag> na, const KwArgs &...kw_args)
{
    std::vector<expression> ret;
    if constexpr (kw::has_v<Tag, KwArgs...>) {
        for (const auto &v : kw::get(na, 0, kw_args...)) {
            ret.emplace_back(v);
        }
    }
    return ret;
}

// Reference: include/heyoka/model/fixed_centres.hpp:33-54, rotating.hpp:33-44, mascon.hpp:33-37.
template <typename... KwArgs>
auto fixed_centres_common_opts(const KwArgs &...kw_args)
{
    return std::tuple{expression(kw::get(kw::Gconst, 1., kw_args...)), kw_expr_vector(kw::masses, kw_args...),
                      kw_expr_vector(kw::positions, kw_args...)};
}
template <typename... KwArgs>
auto rotating_common_opts(const KwArgs &...kw_args)
{
    return std::tuple{kw_expr_vector(kw::omega, kw_args...)};
}
template <typename... KwArgs>
auto mascon_common_opts(const KwArgs &...kw_args)
{
    return std::tuple_cat(fixed_centres_common_opts(kw_args...), rotating_common_opts(kw_args...));
}

template <typename... KwArgs>
auto nbody_common_opts(std::uint32_t n, const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>, "nbody() accepts only named arguments after the number of bodies");
    auto Gconst = expression(kw::get(kw::Gconst, 1., kw_args...));
    std::vector<expression> masses_vec;
    if constexpr (kw::has_v<kw::masses_tag, KwArgs...>) {
        for (const auto &m : kw::get(kw::masses, 0, kw_args...)) {
            masses_vec.emplace_back(m);
        }
    } else {
        masses_vec.resize(n, expression{1.});
    }
    return std::tuple{n, std::move(Gconst), std::move(masses_vec)};
}

} // namespace detail

template <typename... KwArgs>
std::vector<std::pair<expression, expression>> nbody(std::uint32_t n, const KwArgs &...kw_args)
{
    return std::apply(detail::nbody_impl, detail::nbody_common_opts(n, kw_args...));
}

template <typename... KwArgs>
expression nbody_energy(std::uint32_t n, const KwArgs &...kw_args)
{
    return std::apply(detail::nbody_energy_impl, detail::nbody_common_opts(n, kw_args...));
}

template <typename... KwArgs>
expression nbody_potential(std::uint32_t n, const KwArgs &...kw_args)
{
    return std::apply(detail::nbody_potential_impl, detail::nbody_common_opts(n, kw_args...));
}

template <typename... KwArgs>
std::vector<std::pair<expression, expression>> pendulum(const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    return detail::pendulum_impl(expression(kw::get(kw::gconst, 1., kw_args...)),
                                 expression(kw::get(kw::length, 1., kw_args...)));
}

template <typename... KwArgs>
expression pendulum_energy(const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    return detail::pendulum_energy_impl(expression(kw::get(kw::gconst, 1., kw_args...)),
                                        expression(kw::get(kw::length, 1., kw_args...)));
}

#define HEYOKA_AMD_MODEL_NBODY_LIKE(name, ret)                                                                         \
    template <typename... KwArgs>                                                                                      \
    ret name(std::uint32_t n, const KwArgs &...kw_args)                                                                \
    {                                                                                                                  \
        return std::apply(detail::name##_impl, detail::nbody_common_opts(n, kw_args...));                             \
    }
#define HEYOKA_AMD_MODEL_KW(name, opts, ret)                                                                           \
    template <typename... KwArgs>                                                                                      \
    ret name(const KwArgs &...kw_args)                                                                                 \
    {                                                                                                                  \
        static_assert(kw::all_named_v<KwArgs...>);                                                                     \
        return std::apply(detail::name##_impl, detail::opts(kw_args...));                                             \
    }

using model_sys_t = std::vector<std::pair<expression, expression>>;

// model::np1body(n, kw::masses, kw::Gconst) (include/heyoka/model/nbody.hpp:94-120).
HEYOKA_AMD_MODEL_NBODY_LIKE(np1body, model_sys_t)
HEYOKA_AMD_MODEL_NBODY_LIKE(np1body_energy, expression)
HEYOKA_AMD_MODEL_NBODY_LIKE(np1body_potential, expression)
// model::fixed_centres(kw::Gconst, kw::masses, kw::positions), model::rotating(kw::omega),
// model::mascon(kw::Gconst, kw::masses, kw::positions, kw::omega).
HEYOKA_AMD_MODEL_KW(fixed_centres, fixed_centres_common_opts, model_sys_t)
HEYOKA_AMD_MODEL_KW(fixed_centres_energy, fixed_centres_common_opts, expression)
HEYOKA_AMD_MODEL_KW(fixed_centres_potential, fixed_centres_common_opts, expression)
HEYOKA_AMD_MODEL_KW(rotating, rotating_common_opts, model_sys_t)
HEYOKA_AMD_MODEL_KW(rotating_energy, rotating_common_opts, expression)
HEYOKA_AMD_MODEL_KW(rotating_potential, rotating_common_opts, expression)
HEYOKA_AMD_MODEL_KW(mascon, mascon_common_opts, model_sys_t)
HEYOKA_AMD_MODEL_KW(mascon_energy, mascon_common_opts, expression)
HEYOKA_AMD_MODEL_KW(mascon_potential, mascon_common_opts, expression)

#undef HEYOKA_AMD_MODEL_NBODY_LIKE
#undef HEYOKA_AMD_MODEL_KW

// model::cr3bp(kw::mu = 1e-3), model::cr3bp_jacobi() (include/heyoka/model/cr3bp.hpp:30-64).
template <typename... KwArgs>
std::vector<std::pair<expression, expression>> cr3bp(const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    return detail::cr3bp_impl(expression(kw::get(kw::mu, 1e-3, kw_args...)));
}
template <typename... KwArgs>
expression cr3bp_jacobi(const KwArgs &...kw_args)
{
    static_assert(kw::all_named_v<KwArgs...>);
    return detail::cr3bp_jacobi_impl(expression(kw::get(kw::mu, 1e-3, kw_args...)));
}

} // namespace heyoka_amd::model
