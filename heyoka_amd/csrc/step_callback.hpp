// Step callbacks of the propagate_*() functions.
//
// Mirrors the interface of the reference's step_callback_batch<T> / step_callback_batch_set<T>
// (include/heyoka/step_callback.hpp:46-62, :139-185; src/step_callback.cpp:40-127): a type-erased, copyable holder of
// any callable `bool(taylor_adaptive_batch<T> &)` - lambdas, function objects, function pointers, std::function,
// std::reference_wrapper - which may ALSO provide a `pre_hook(taylor_adaptive_batch<T> &)` member: propagate_*() invokes
// it once before the first step (src/taylor_adaptive_batch.cpp:1356-1365, :1782-1791). A callback *set* runs all its
// members at every step (every one of them, then the conjunction of their results) and forwards pre_hook() to each.
// From-scratch implementation (a small virtual holder instead of the reference's tanuki wrap).
#pragma once

#include <cstddef>
#include <functional>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeindex>
#include <typeinfo>
#include <utility>
#include <vector>

namespace heyoka_amd
{

namespace detail
{

template <typename F>
struct unwrap_ref {
    using type = F;
};
template <typename F>
struct unwrap_ref<std::reference_wrapper<F>> {
    using type = F;
};

template <typename F, typename TA>
concept step_cb_callable = requires(typename unwrap_ref<F>::type &f, TA &ta) {
    { f(ta) } -> std::convertible_to<bool>;
};

template <typename F, typename TA>
concept step_cb_with_pre_hook = requires(typename unwrap_ref<F>::type &f, TA &ta) { static_cast<void>(f.pre_hook(ta)); };

template <typename F>
struct is_std_function : std::false_type {
};
template <typename S>
struct is_std_function<std::function<S>> : std::true_type {
};

template <typename TA>
class step_cb_wrap
{
    struct iface {
        virtual ~iface() = default;
        virtual bool call(TA &) = 0;
        virtual void pre_hook(TA &) = 0;
        [[nodiscard]] virtual std::unique_ptr<iface> clone() const = 0;
        [[nodiscard]] virtual const std::type_info &type() const noexcept = 0;
        [[nodiscard]] virtual void *ptr() noexcept = 0;
    };
    template <typename F>
    struct holder final : iface {
        F f;
        explicit holder(F x) : f(std::move(x)) {}
        static auto &target(F &x)
        {
            if constexpr (std::is_same_v<typename unwrap_ref<F>::type, F>) {
                return x;
            } else {
                return x.get();
            }
        }
        bool call(TA &ta) override
        {
            return static_cast<bool>(target(f)(ta));
        }
        void pre_hook(TA &ta) override
        {
            // (The default is a no-op: step_cb_iface::pre_hook(), include/heyoka/step_callback.hpp:62.)
            if constexpr (step_cb_with_pre_hook<F, TA>) {
                static_cast<void>(target(f).pre_hook(ta));
            } else {
                (void)ta;
            }
        }
        [[nodiscard]] std::unique_ptr<iface> clone() const override
        {
            return std::make_unique<holder>(f);
        }
        [[nodiscard]] const std::type_info &type() const noexcept override
        {
            return typeid(F);
        }
        [[nodiscard]] void *ptr() noexcept override
        {
            return &f;
        }
    };
    std::unique_ptr<iface> m_ptr;

public:
    using ta_t = TA;

    // Empty (invalid) callback.
    step_cb_wrap() noexcept = default;
    step_cb_wrap(std::nullptr_t) noexcept {}
    // From any callable. Null function pointers and empty std::function objects give an empty callback, like the
    // reference's callable wrapper.
    template <typename F>
        requires(!std::is_same_v<std::remove_cvref_t<F>, step_cb_wrap>) && step_cb_callable<std::decay_t<F>, TA>
    step_cb_wrap(F &&f)
    {
        using D = std::decay_t<F>;
        if constexpr (std::is_pointer_v<D> || std::is_member_pointer_v<D>) {
            if (f == nullptr) {
                return;
            }
        } else if constexpr (is_std_function<D>::value) {
            if (!f) {
                return;
            }
        }
        m_ptr = std::make_unique<holder<D>>(std::forward<F>(f));
    }
    step_cb_wrap(const step_cb_wrap &o) : m_ptr(o.m_ptr ? o.m_ptr->clone() : nullptr) {}
    step_cb_wrap(step_cb_wrap &&) noexcept = default;
    step_cb_wrap &operator=(const step_cb_wrap &o)
    {
        if (this != &o) {
            m_ptr = o.m_ptr ? o.m_ptr->clone() : nullptr;
        }
        return *this;
    }
    step_cb_wrap &operator=(step_cb_wrap &&) noexcept = default;
    ~step_cb_wrap() = default;

    explicit operator bool() const noexcept
    {
        return static_cast<bool>(m_ptr);
    }
    bool operator()(TA &ta)
    {
        if (!m_ptr) {
            throw std::bad_function_call();
        }
        return m_ptr->call(ta);
    }
    void pre_hook(TA &ta)
    {
        if (!m_ptr) {
            throw std::bad_function_call();
        }
        m_ptr->pre_hook(ta);
    }
    // Access to the stored object (cf. value_type_index() / value_ptr<T>() of the reference's wrap).
    [[nodiscard]] const std::type_info &value_type() const noexcept
    {
        return m_ptr ? m_ptr->type() : typeid(void);
    }
    template <typename T>
    [[nodiscard]] T *extract() noexcept
    {
        return (m_ptr && m_ptr->type() == typeid(T)) ? static_cast<T *>(m_ptr->ptr()) : nullptr;
    }
    template <typename T>
    [[nodiscard]] const T *extract() const noexcept
    {
        return (m_ptr && m_ptr->type() == typeid(T)) ? static_cast<const T *>(m_ptr->ptr()) : nullptr;
    }
    friend void swap(step_cb_wrap &a, step_cb_wrap &b) noexcept
    {
        a.m_ptr.swap(b.m_ptr);
    }
};

// Reference: step_callback_set_impl, include/heyoka/step_callback.hpp:139-185, src/step_callback.cpp:40-127.
template <typename TA>
class step_cb_set
{
public:
    using step_cb_t = step_cb_wrap<TA>;
    using ta_t = TA;
    using size_type = typename std::vector<step_cb_t>::size_type;

private:
    std::vector<step_cb_t> m_cbs;

public:
    step_cb_set() noexcept = default;
    explicit step_cb_set(std::vector<step_cb_t> cbs) : m_cbs(std::move(cbs))
    {
        for (const auto &cb : m_cbs) {
            if (!cb) {
                throw std::invalid_argument("Cannot construct a callback set containing one or more empty callbacks");
            }
        }
    }
    step_cb_set(std::initializer_list<step_cb_t> cbs) : step_cb_set(std::vector<step_cb_t>(cbs)) {}

    [[nodiscard]] size_type size() const noexcept
    {
        return m_cbs.size();
    }
    const step_cb_t &operator[](size_type i) const
    {
        if (i >= size()) {
            throw std::out_of_range("Out of range index " + std::to_string(i) + " when accessing a step callback set of size "
                                    + std::to_string(size()));
        }
        return m_cbs[i];
    }
    step_cb_t &operator[](size_type i)
    {
        return const_cast<step_cb_t &>(static_cast<const step_cb_set &>(*this)[i]);
    }
    // Every callback runs, then the results are combined (src/step_callback.cpp:108-119).
    bool operator()(TA &ta)
    {
        bool retval = true;
        for (auto &cb : m_cbs) {
            retval = cb(ta) && retval;
        }
        return retval;
    }
    void pre_hook(TA &ta)
    {
        for (auto &cb : m_cbs) {
            cb.pre_hook(ta);
        }
    }
    friend void swap(step_cb_set &a, step_cb_set &b) noexcept
    {
        a.m_cbs.swap(b.m_cbs);
    }
};

} // namespace detail

// Access to the object stored in a callback wrapper, spelled like the reference's (tanuki) helpers
// (test/taylor_adaptive_batch.cpp:688-730: value_isa<T>(cb), value_ptr<T>(cb)).
template <typename TA>
[[nodiscard]] std::type_index value_type_index(const detail::step_cb_wrap<TA> &w) noexcept
{
    return std::type_index(w.value_type());
}
template <typename T, typename TA>
[[nodiscard]] bool value_isa(const detail::step_cb_wrap<TA> &w) noexcept
{
    return w.template extract<T>() != nullptr;
}
template <typename T, typename TA>
[[nodiscard]] T &value_ref(detail::step_cb_wrap<TA> &w)
{
    auto *p = w.template extract<T>();
    if (p == nullptr) {
        throw std::runtime_error("Invalid reference to the value stored in a step callback: the stored type is different");
    }
    return *p;
}
template <typename T, typename TA>
[[nodiscard]] T *value_ptr(detail::step_cb_wrap<TA> &w) noexcept
{
    return w.template extract<T>();
}
template <typename T, typename TA>
[[nodiscard]] const T *value_ptr(const detail::step_cb_wrap<TA> &w) noexcept
{
    return w.template extract<T>();
}

} // namespace heyoka_amd
