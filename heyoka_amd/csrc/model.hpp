// ODE right-hand-side builders for the benchmark configurations
// (reference: src/model/nbody.cpp:53-174, include/heyoka/model/nbody.hpp:33-78,
// src/model/pendulum.cpp:23-33, include/heyoka/model/pendulum.hpp).
#pragma once

#include <cstdint>
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

} // namespace heyoka_amd::model
