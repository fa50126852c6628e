// Minimal named-argument machinery so that reference call sites of the form
// `taylor_adaptive_batch<double>{sys, state, batch, kw::tol = 1e-9, kw::high_accuracy = true}`
// compile unchanged (reference: include/heyoka/detail/igor.hpp:92-160, include/heyoka/kw.hpp).
// From-scratch design: a tag type per keyword, `kw::name = value` yields a tagged value, and a
// compile-time parser picks values out of the argument pack.
#pragma once

#include <initializer_list>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace heyoka_amd
{

namespace kw
{

template <typename Tag, typename T>
struct tagged_arg {
    using tag_type = Tag;
    T value;
};

template <typename Tag>
struct named_arg {
    template <typename T>
    constexpr auto operator=(T &&v) const
    {
        return tagged_arg<Tag, std::decay_t<T>>{std::forward<T>(v)};
    }
    template <typename T>
    auto operator=(std::initializer_list<T> il) const
    {
        return tagged_arg<Tag, std::vector<T>>{std::vector<T>(il)};
    }
};

#define HEYOKA_AMD_KWARG(name)                                                                                         \
    struct name##_tag {                                                                                                \
    };                                                                                                                 \
    inline constexpr named_arg<name##_tag> name {}

// Integrator construction (reference: include/heyoka/taylor.hpp:178-181, :814-821).
HEYOKA_AMD_KWARG(tol);
HEYOKA_AMD_KWARG(high_accuracy);
HEYOKA_AMD_KWARG(compact_mode);
HEYOKA_AMD_KWARG(pars);
HEYOKA_AMD_KWARG(time);
HEYOKA_AMD_KWARG(parallel_mode);
HEYOKA_AMD_KWARG(parjit);
HEYOKA_AMD_KWARG(t_events);
HEYOKA_AMD_KWARG(nt_events);
// LLVM-only knobs of the reference (include/heyoka/llvm_state.hpp:243-246): accepted and ignored.
HEYOKA_AMD_KWARG(opt_level);
HEYOKA_AMD_KWARG(fast_math);
HEYOKA_AMD_KWARG(force_avx512);
HEYOKA_AMD_KWARG(slp_vectorize);
HEYOKA_AMD_KWARG(code_model);
HEYOKA_AMD_KWARG(mname);
// propagate_*() (reference: include/heyoka/taylor.hpp:263-272).
HEYOKA_AMD_KWARG(max_steps);
HEYOKA_AMD_KWARG(max_delta_t);
HEYOKA_AMD_KWARG(callback);
HEYOKA_AMD_KWARG(write_tc);
HEYOKA_AMD_KWARG(c_output);
// Events (reference: include/heyoka/events.hpp:87-100).
HEYOKA_AMD_KWARG(direction);
HEYOKA_AMD_KWARG(cooldown);
// Models.
HEYOKA_AMD_KWARG(masses);
HEYOKA_AMD_KWARG(Gconst);
HEYOKA_AMD_KWARG(gconst);
HEYOKA_AMD_KWARG(length);
HEYOKA_AMD_KWARG(mu);
HEYOKA_AMD_KWARG(positions);
HEYOKA_AMD_KWARG(omega);
// MI355X-specific extensions: the device ordinal, the code generator (emitter: 0 automatic, 1 unrolled, 2 wave-cluster,
// 3 table, 4 block; cluster_kernel: 0 automatic, 5 / 3 / 2 / 1), correctly rounded quotients in the recurrences, the
// steppers used with events, and the outcome semantics of propagate_for / propagate_until (0 reference, 1 lock-step loop,
// 2 per lane) - see tab_core::config.
HEYOKA_AMD_KWARG(device);
HEYOKA_AMD_KWARG(emitter);
HEYOKA_AMD_KWARG(cluster_kernel);
HEYOKA_AMD_KWARG(exact_division);
HEYOKA_AMD_KWARG(sum_order);
HEYOKA_AMD_KWARG(events_on_cluster);
HEYOKA_AMD_KWARG(batch_semantics);
// ensemble_propagate_*_batch(): kw::gather = &g (ensemble_gathered *) collects the final states of all the iterations in one
// buffer on device 0 (RCCL over xGMI between devices, see ensemble.hpp).
HEYOKA_AMD_KWARG(gather);

#undef HEYOKA_AMD_KWARG

namespace detail
{

template <typename T>
struct is_tagged_arg : std::false_type {
};
template <typename Tag, typename T>
struct is_tagged_arg<tagged_arg<Tag, T>> : std::true_type {
};

template <typename Tag>
constexpr bool has_impl()
{
    return false;
}
template <typename Tag, typename A, typename... Rest>
constexpr bool has_impl()
{
    if constexpr (std::is_same_v<typename std::decay_t<A>::tag_type, Tag>) {
        return true;
    } else {
        return has_impl<Tag, Rest...>();
    }
}

} // namespace detail

// True if all Args are tagged named arguments.
template <typename... Args>
inline constexpr bool all_named_v = (detail::is_tagged_arg<std::decay_t<Args>>::value && ...);

template <typename Tag, typename... Args>
inline constexpr bool has_v = detail::has_impl<Tag, Args...>();

// Fetch the value of the named argument na from args, or def if absent.
template <typename Tag, typename Def>
constexpr decltype(auto) get(named_arg<Tag>, Def &&def)
{
    return std::forward<Def>(def);
}
template <typename Tag, typename Def, typename A, typename... Rest>
constexpr decltype(auto) get(named_arg<Tag> na, Def &&def, A &&a, Rest &&...rest)
{
    if constexpr (std::is_same_v<typename std::decay_t<A>::tag_type, Tag>) {
        return (std::forward<A>(a).value);
    } else {
        return get(na, std::forward<Def>(def), std::forward<Rest>(rest)...);
    }
}

template <typename Tag, typename... Args>
constexpr bool has(named_arg<Tag>, const Args &...)
{
    return has_v<Tag, Args...>;
}

} // namespace kw

} // namespace heyoka_amd
