// Behaviours which the reference's own unit tests of the batch integrator pin down
// (test/taylor_adaptive_batch.cpp), restated against the reference's include layout and namespace. Every case
// names the TEST_CASE (file:line) whose assertions it re-expresses. usage: test_reference_cases [gpu]
// (without "gpu" only the cases which never launch a kernel run).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <limits>
#include <ranges>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include <heyoka/ensemble_propagate.hpp>
#include <heyoka/heyoka.hpp>
#include <heyoka/kw.hpp>
#include <heyoka/model/pendulum.hpp>
#include <heyoka/taylor.hpp>

using namespace heyoka;

namespace
{

int n_checks = 0;

#define CHECK(cond)                                                                                                    \
    do {                                                                                                               \
        ++n_checks;                                                                                                    \
        if (!(cond)) {                                                                                                 \
            std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond);                              \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

// The callable must throw an exception of type E whose what() is exactly msg.
template <typename E, typename F>
void throws_with(F &&f, const std::string &msg, int line)
{
    ++n_checks;
    try {
        f();
    } catch (const E &e) {
        if (msg != e.what()) {
            std::fprintf(stderr, "line %d: wrong message:\n  got:      %s\n  expected: %s\n", line, e.what(), msg.c_str());
            std::exit(1);
        }
        return;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "line %d: wrong exception type, what(): %s\n", line, e.what());
        std::exit(1);
    }
    std::fprintf(stderr, "line %d: no exception (expected: %s)\n", line, msg.c_str());
    std::exit(1);
}
#define THROWS_WITH(E, expr, msg) throws_with<E>([&]() { (void)(expr); }, msg, __LINE__)
#define THROWS(E, expr) \
    do {                                                                                                               \
        ++n_checks;                                                                                                    \
        bool hy_thrown = false;                                                                                        \
        try {                                                                                                          \
            (void)(expr);                                                                                              \
        } catch (const E &) {                                                                                          \
            hy_thrown = true;                                                                                          \
        }                                                                                                              \
        if (!hy_thrown) {                                                                                              \
            std::fprintf(stderr, "line %d: %s did not throw\n", __LINE__, #expr);                                      \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

using tab = taylor_adaptive_batch<double>;
using te_t = t_event_batch<double>;
using nte_t = nt_event_batch<double>;
using dvec = std::vector<double>;
constexpr auto inf = std::numeric_limits<double>::infinity();
constexpr auto eps = std::numeric_limits<double>::epsilon();

auto oc(const tab &ta, std::size_t i)
{
    return std::get<0>(ta.get_propagate_res()[i]);
}
auto nsteps(const tab &ta, std::size_t i)
{
    return std::get<3>(ta.get_propagate_res()[i]);
}
bool close_to(double a, double b, double tol = 100 * eps)
{
    return std::abs(a - b) <= tol * std::max(1., std::abs(b));
}

// Step callback which counts its copies: once it is inside propagate_*() it must not be copied any more
// (the cb_functor_until / cb_functor_for checks of test/taylor_adaptive_batch.cpp:497-526, :688-730).
template <int>
struct counting_cb {
    counting_cb() = default;
    counting_cb(counting_cb &&) noexcept = default;
    counting_cb(const counting_cb &)
    {
        ++n_copies;
    }
    counting_cb &operator=(const counting_cb &) = delete;
    bool operator()(taylor_adaptive_batch<double> &) const
    {
        CHECK(n_copies == n_copies_after);
        ++n_calls;
        return true;
    }
    inline static unsigned n_copies = 0, n_copies_after = 0, n_calls = 0;
};

// The callables of test/step_callback.cpp:49-78, :362-401.
bool free_cb(tab &)
{
    return true;
}
struct two_and_false {
    bool operator()(tab &ta)
    {
        ta.get_state_data()[0] = 2;
        return false;
    }
    void pre_hook(tab &ta)
    {
        ta.get_state_data()[0] = 1;
    }
};
struct only_pre_hook {
    void pre_hook(tab &) {}
};
struct sets_length_in_pre_hook {
    bool operator()(tab &)
    {
        return true;
    }
    void pre_hook(tab &ta)
    {
        ta.get_pars_data()[0] = 1.5;
        ta.get_pars_data()[1] = 1.5;
    }
};
struct moves_time_in_pre_hook {
    bool operator()(tab &)
    {
        return true;
    }
    void pre_hook(tab &ta)
    {
        ta.set_time({ta.get_time()[0] + 1, ta.get_time()[1] + 1});
    }
};

// (test/ensemble_propagate.cpp:52-64.)
struct counting_copies_cb {
    counting_copies_cb() = default;
    counting_copies_cb(counting_copies_cb &&) noexcept = default;
    counting_copies_cb(const counting_copies_cb &)
    {
        ++n_copies;
    }
    bool operator()(tab &) const
    {
        return true;
    }
    inline static unsigned long n_copies = 0;
};

const std::string ev_time_msg = "The invocation of one or more event callbacks resulted in the alteration of the time "
                                "coordinate of the integrator at the batch index 0 - this is not supported";

// ------------------------------------------------------------------------------------------------------------------
// Cases which never launch a kernel.
// ------------------------------------------------------------------------------------------------------------------
void host_cases()
{
    auto [x, v] = make_vars("x", "v");
    const auto pend = std::vector{prime(x) = v, prime(v) = -9.8 * sin(x)};

    // "state pars range" (:88-103): the mutable ranges alias state and parameters.
    {
        auto ta = tab{{prime(x) = v, prime(v) = -par[0] * sin(x)}, {1.1, 1.11, 2.2, 2.21}, 2u, kw::pars = {3.3, 3.31}};
        CHECK(std::ranges::equal(ta.get_state(), ta.get_state_range()));
        CHECK(std::ranges::equal(ta.get_pars(), ta.get_pars_range()));
        const dvec ns{4.4, 4.41, 5.5, 5.51}, np{6.6, 6.61};
        std::ranges::copy(ns, ta.get_state_range().begin());
        std::ranges::copy(np, ta.get_pars_range().begin());
        CHECK(ta.get_state() == ns);
        CHECK(ta.get_pars() == np);
    }

    // "set time" (:466-495).
    {
        auto ta = tab{{prime(x) = v, prime(v) = 1_dbl}, {0, 0, 0.1, 0.1}, 2};
        const std::string head = "Invalid number of new times specified in a Taylor integrator in batch mode: the batch "
                                 "size is 2, but the number of specified times is ";
        THROWS_WITH(std::invalid_argument, ta.set_time(dvec{}), head + "0");
        THROWS_WITH(std::invalid_argument, ta.set_time({1, 2, 3}), head + "3");
        CHECK((ta.get_time() == dvec{0., 0.}));
        ta.set_time({1, -2});
        CHECK((ta.get_time() == dvec{1., -2.}));
        ta.set_time(-1);
        CHECK((ta.get_time() == dvec{-1., -1.}));
        ta.set_time(1);
        CHECK((ta.get_time() == dvec{1., 1.}));
    }

    // "param wrong number" (:955-993), "param auto setup" (:995-1015), "param deduction from events" (:1017-1053):
    // the parameters of the event equations count.
    {
        const auto msg = [](int passed, int in_sys) {
            return "Invalid number of parameter values passed to the constructor of an adaptive Taylor integrator in "
                   "batch mode: "
                   + std::to_string(passed) + " parameter value(s) were passed, but the ODE system contains "
                   + std::to_string(in_sys) + " parameter(s) (in batches of 2)";
        };
        const dvec st{0.05, 0.06, 0.025, 0.026};
        THROWS_WITH(std::invalid_argument,
                    (tab{{prime(x) = v + par[0], prime(v) = -9.8 * sin(x)}, st, 2u, kw::pars = dvec{1., 2., 3.}}), msg(3, 1));
        THROWS_WITH(std::invalid_argument, (tab{pend, st, 2u, kw::pars = dvec{1., 2., 3.}, kw::t_events = {te_t(v - par[0])}}),
                    msg(3, 1));
        THROWS_WITH(std::invalid_argument,
                    (tab{{prime(x) = v + par[3], prime(v) = -9.8 * sin(x)}, st, 2u, kw::pars = dvec{1., 2., 3.}}), msg(3, 4));

        auto ta = tab{pend, st, 2u, kw::t_events = {te_t(v - par[3])}};
        CHECK(ta.get_pars() == dvec(8u, 0.));
        ta = tab{pend, st, 2u, kw::pars = dvec{}, kw::t_events = {te_t(v - par[3])}};
        CHECK(ta.get_pars() == dvec(8u, 0.));

        const auto nt_noop = [](auto &, double, int, std::uint32_t) {};
        CHECK((tab{pend, st, 2, kw::t_events = {te_t(v - par[0])}}.get_pars().size() == 2u));
        CHECK((tab{pend, st, 2, kw::nt_events = {nte_t(v - par[1], nt_noop)}}.get_pars().size() == 4u));
        CHECK((tab{pend, st, 2, kw::t_events = {te_t(v - par[10])}, kw::nt_events = {nte_t(v - par[1], nt_noop)}}
                   .get_pars()
                   .size()
               == 22u));
    }

    // "events error" (:1395-1454).
    {
        const auto sys = model::nbody(2, kw::masses = {1., 0.});
        const dvec st{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0};
        const std::string idx_msg
            = "Cannot reset the cooldowns at batch index 2: the batch size for this integrator is only 2";
        const std::string no_ev = "No events were defined for this integrator";
        {
            auto ta = tab{sys, st, 2, kw::t_events = {te_t("x_0"_var)}};
            CHECK(ta.with_events());
            THROWS_WITH(std::invalid_argument, ta.reset_cooldowns(2), idx_msg);
        }
        {
            // No terminal events: the cooldown lists exist and are empty.
            auto ta = tab{sys, st, 2, kw::nt_events = {nte_t("x_0"_var, [](auto &, double, int, std::uint32_t) {})}};
            CHECK(ta.with_events());
            CHECK(std::ranges::all_of(ta.get_te_cooldowns(), [](const auto &c) { return c.empty(); }));
            ta.reset_cooldowns();
            ta.reset_cooldowns(0);
            ta.reset_cooldowns(1);
            THROWS_WITH(std::invalid_argument, ta.reset_cooldowns(2), idx_msg);
        }
        {
            auto ta = tab{sys, st, 2};
            CHECK(!ta.with_events());
            THROWS_WITH(std::invalid_argument, ta.get_t_events(), no_ev);
            THROWS_WITH(std::invalid_argument, ta.get_nt_events(), no_ev);
            THROWS_WITH(std::invalid_argument, ta.get_te_cooldowns(), no_ev);
            THROWS_WITH(std::invalid_argument, ta.reset_cooldowns(), no_ev);
            THROWS_WITH(std::invalid_argument, ta.reset_cooldowns(2), no_ev);
        }
    }

    // "set_time alias bug" (:1847-1862): the argument may alias the time vector of the integrator.
    {
        auto ta = tab{pend, {0., 0., 0.5, 0.5001}, 2, kw::t_events = {te_t(x - 1e-6)}};
        ta.set_time(ta.get_time());
        CHECK(ta.get_time()[0] == 0. && ta.get_time()[1] == 0.);
    }

    // "get_set_dtime" (:1864-1942), the part without steps: normalisation, checks before anything is modified.
    {
        auto ta = tab{pend, {0, 0.01, 0.1, 0.11}, 2};
        THROWS_WITH(std::invalid_argument, ta.set_dtime(dvec{}, dvec{1.}),
                    "Invalid number of new times specified in a Taylor integrator in batch mode: the batch size is 2, but "
                    "the number of specified times is (0, 1)");
        const auto same = [&](double h0, double h1, double l0, double l1) {
            const auto [hi, lo] = ta.get_dtime();
            return hi[0] == h0 && hi[1] == h1 && lo[0] == l0 && lo[1] == l1;
        };
        ta.set_dtime({3., -7}, {2., 5.});
        CHECK(same(5, -2, 0, 0));
        ta.set_dtime({3., -3}, {eps, eps});
        CHECK(same(3, -3, eps, eps));
        ta.set_dtime(4., 3.);
        CHECK(same(7, 7, 0, 0));
        ta.set_dtime(3., eps);
        CHECK(same(3, 3, eps, eps));
        ta.set_dtime({3., 4.}, {1., 2.});
        THROWS_WITH(std::invalid_argument, ta.set_dtime(inf, 1.),
                    "The components of the double-length representation of the time coordinate must both be finite, but "
                    "they are inf and 1 instead");
        THROWS(std::invalid_argument, ta.set_dtime(1., inf));
        THROWS_WITH(std::invalid_argument, ta.set_dtime(3., 4.),
                    "The first component of the double-length representation of the time coordinate (3) must not be "
                    "smaller in magnitude than the second component (4)");
        THROWS(std::invalid_argument, ta.set_dtime({1., inf}, {1., 2.}));
        THROWS(std::invalid_argument, ta.set_dtime({1., .1}, {inf, 2.}));
        THROWS(std::invalid_argument, ta.set_dtime({1., 2.}, {1., 3.}));
        THROWS(std::invalid_argument, ta.set_dtime({4., 4.}, {8., 3.}));
        CHECK(same(4, 6, 0, 0));
    }

    // "invalid initial state" (:2231-2242), "empty init state" (:2244-2250), "scalar time ctor" (:2259-2268),
    // "def ctor" (:1253-1263).
    {
        THROWS_WITH(std::invalid_argument, (tab{{prime(x) = v, prime(v) = -x}, {0.05, 0.051}, 2}),
                    "Inconsistent sizes detected in the initialization of an adaptive Taylor integrator: the state vector "
                    "has a dimension of 1 and a batch size of 2, while the number of equations is 2");
        const auto dyn = model::pendulum();
        CHECK((tab{dyn, 2u}.get_state() == dvec{0., 0., 0., 0.}));
        CHECK((tab{dyn, 2u, kw::time = 42}.get_time() == dvec{42., 42.}));
        tab def_constructed;
        (void)def_constructed;
    }

    // "propagate for_until" (:528-570), argument checks.
    for (const auto cm : {true, false}) {
        auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2u, kw::compact_mode = cm};
        const std::string tail = " function of an adaptive Taylor integrator in batch mode";
        const auto n_msg = [](int n) {
            return "Invalid number of max timesteps specified in a Taylor integrator in batch mode: the batch size is 2, but "
                   "the number of specified timesteps is "
                   + std::to_string(n);
        };
        THROWS_WITH(std::invalid_argument, ta.propagate_until({0., inf}),
                    "A non-finite time was passed to the propagate_until()" + tail);
        THROWS_WITH(std::invalid_argument, ta.propagate_until({10., 11.}, kw::max_delta_t = dvec{1}), n_msg(1));
        THROWS_WITH(std::invalid_argument, ta.propagate_until({10., 11.}, kw::max_delta_t = {1., 2., 3.}), n_msg(3));
        THROWS_WITH(std::invalid_argument,
                    ta.propagate_until({10., 11.}, kw::max_delta_t = {1., std::numeric_limits<double>::quiet_NaN()}),
                    "A nan max_delta_t was passed to the propagate_until()" + tail);
        THROWS_WITH(std::invalid_argument, ta.propagate_until({10., 11.}, kw::max_delta_t = {1., -1.}),
                    "A non-positive max_delta_t was passed to the propagate_until()" + tail);
        ta.set_time({0., std::numeric_limits<double>::lowest()});
        THROWS_WITH(std::invalid_argument,
                    ta.propagate_until({10., std::numeric_limits<double>::max()}, kw::max_delta_t = dvec{}),
                    "The final time passed to the propagate_until()" + tail + " results in an overflow condition");
    }

    // test/step_callback.cpp "step_callback basics" (:80-195) for the batch wrapper.
    {
        using cb_t = step_callback_batch<double>;
        auto ta = tab{{prime(x) = 0_dbl, prime(v) = 0_dbl}, {0., 0.}, 1u, kw::tol = 1e-1};
        {
            cb_t c;
            CHECK(!c);
            THROWS(std::bad_function_call, c(ta));
            static_assert(std::is_nothrow_swappable_v<cb_t>);
            static_assert(!std::is_constructible_v<cb_t, void>);
            static_assert(!std::is_constructible_v<cb_t, int, int>);
            static_assert(!std::is_constructible_v<cb_t, only_pre_hook>);
            auto c2 = c;
            auto c3 = std::move(c);
            CHECK(!c2 && !c3);
            cb_t c6 = static_cast<bool (*)(tab &)>(nullptr);
            cb_t c7 = std::function<bool(tab &)>{};
            CHECK(!c6 && !c7);
        }
        {
            auto lam = [](auto &) { return true; };
            cb_t c(lam);
            CHECK(static_cast<bool>(c) && c(ta));
            c.pre_hook(ta);
            CHECK(value_type_index(c) == typeid(decltype(lam)));
            CHECK(value_ptr<decltype(lam)>(c) != nullptr && value_ptr<decltype(lam)>(std::as_const(c)) != nullptr);
            cb_t from_ptr(&free_cb), from_fn(free_cb);
            CHECK(value_type_index(from_ptr) == typeid(decltype(&free_cb)));
            CHECK(from_ptr(ta) && from_fn(ta));
            from_ptr.pre_hook(ta);
            from_fn.pre_hook(ta);
        }
        const auto call_then_hook = [&](cb_t c) {
            CHECK(ta.get_state()[0] == 0.);
            CHECK(static_cast<bool>(c) && !c(ta));
            CHECK(ta.get_state()[0] == 2.);
            c.pre_hook(ta);
            CHECK(ta.get_state()[0] == 1.);
            ta.get_state_data()[0] = 0;
        };
        call_then_hook(cb_t(two_and_false{}));
        two_and_false by_ref;
        call_then_hook(cb_t(std::ref(by_ref)));
        {
            cb_t c([](auto &t) {
                t.get_state_data()[0] = 3;
                return true;
            });
            CHECK(c(ta) && ta.get_state()[0] == 3.);
            c.pre_hook(ta); // (no pre_hook() member: nothing happens)
            CHECK(ta.get_state()[0] == 3.);
            ta.get_state_data()[0] = 0;
        }
        {
            using std::swap;
            cb_t c1(two_and_false{}), c2;
            swap(c1, c2);
            CHECK(static_cast<bool>(c2) && !c1 && value_ptr<two_and_false>(c2) != nullptr);
        }
        // "step_callback_set" (:493-545, :660-680): container interface and error messages.
        {
            using set_t = step_callback_batch_set<double>;
            using std::swap;
            static_assert(std::is_nothrow_swappable_v<set_t>);
            set_t s0;
            CHECK(s0.size() == 0u);
            THROWS_WITH(std::out_of_range, s0[0], "Out of range index 0 when accessing a step callback set of size 0");
            THROWS_WITH(std::out_of_range, std::as_const(s0)[0],
                        "Out of range index 0 when accessing a step callback set of size 0");
            auto s1 = set_t{[](const auto &) { return true; }};
            CHECK(s1.size() == 1u);
            (void)s1[0];
            THROWS_WITH(std::out_of_range, s1[10], "Out of range index 10 when accessing a step callback set of size 1");
            swap(s0, s1);
            CHECK(s0.size() == 1u && s1.size() == 0u);
            auto s3 = s0;
            auto s4 = std::move(s0);
            s0 = s4;
            s1 = std::move(s0);
            CHECK(s3.size() == 1u && s4.size() == 1u && s1.size() == 1u);
            const std::string empty_msg = "Cannot construct a callback set containing one or more empty callbacks";
            THROWS_WITH(std::invalid_argument, (set_t{cb_t{}}), empty_msg);
            THROWS_WITH(std::invalid_argument, (set_t{cb_t{}, [](const auto &) { return true; }}), empty_msg);
            THROWS_WITH(std::invalid_argument, (set_t{[](const auto &) { return true; }, cb_t{}}), empty_msg);
        }
    }

    // "ctad" (:2048-2066, :2089-2097): the value type is deduced from the initial state.
    {
        auto ta = taylor_adaptive_batch({prime(x) = v, prime(v) = -x}, std::vector{0., 1.}, 1u);
        static_assert(std::is_same_v<decltype(ta), taylor_adaptive_batch<double>>);
        CHECK(ta.get_state()[0] == 0 && ta.get_state()[1] == 1);
        ta = taylor_adaptive_batch({{v, v}, {x, -x}}, std::vector{0., 1.}, 1u);
        CHECK(ta.get_state()[0] == 0 && ta.get_state()[1] == 1);
        auto tb = taylor_adaptive_batch({prime(x) = v, prime(v) = -x}, {0., 1.}, 1u);
        static_assert(std::is_same_v<decltype(tb), taylor_adaptive_batch<double>>);
        tb = taylor_adaptive_batch({{v, v}, {x, -x}}, {0., 1.}, 1u);
        CHECK(tb.get_state()[0] == 0 && tb.get_state()[1] == 1);
    }

    // "taylor move" (:1368-1393): state and parameters handed over as rvalues are not reallocated.
    {
        auto init_state = dvec{-1., -1.1, 0., 0.1};
        auto pars = dvec{9.8, 9.9};
        const auto *s_data = init_state.data();
        const auto *p_data = pars.data();
        auto ta = tab{{prime(x) = v, prime(v) = -par[0] * sin(x)}, std::move(init_state), 2, kw::pars = std::move(pars)};
        CHECK(s_data == ta.get_state().data());
        CHECK(p_data == ta.get_pars().data());
    }

    // "stream output" (:1265-1352): the fields of the summary; the event counts appear only when there are events.
    {
        const auto sys = std::vector{prime(x) = v - par[1], prime(v) = -9.8 * sin(x + par[0])};
        const dvec st{0., 0.01, 0.5, 0.51};
        const auto text = [](const tab &ta) {
            std::ostringstream oss;
            oss << ta;
            return oss.str();
        };
        const auto has = [](const std::string &str, const char *p) { return str.find(p) != std::string::npos; };
        const auto nt_noop = [](auto &, double, int, std::uint32_t) {};
        auto str = text(tab{sys, st, 2u, kw::pars = dvec{-1e-4, -1.1e-4, 0, 0}});
        for (const auto *field : {"Tolerance", "Dimension", "Batch size", "Parameters", "High accuracy", "Compact mode"}) {
            CHECK(has(str, field));
        }
        CHECK(!has(str, "events"));
        str = text(tab{sys, st, 2u, kw::t_events = {te_t(x)}});
        CHECK(has(str, "N of terminal events") && has(str, ": 1") && !has(str, "N of non-terminal events"));
        str = text(tab{sys, st, 2u, kw::nt_events = {nte_t(x, nt_noop)}});
        CHECK(!has(str, "N of terminal events") && has(str, ": 1") && has(str, "N of non-terminal events"));
        str = text(tab{sys, st, 2u, kw::t_events = {te_t(x)}, kw::nt_events = {nte_t(x, nt_noop)}});
        CHECK(has(str, "N of terminal events") && has(str, ": 1") && has(str, "N of non-terminal events"));
    }

    // "propagate grid 2" (:768-800), argument checks.
    {
        auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2u};
        const std::string tail = " function of an adaptive Taylor integrator in batch mode";
        const auto n_msg = [](int n) {
            return "Invalid number of max timesteps specified in a Taylor integrator in batch mode: the batch size is 2, but "
                   "the number of specified timesteps is "
                   + std::to_string(n);
        };
        THROWS_WITH(std::invalid_argument, ta.propagate_grid({10., 11.}, kw::max_delta_t = dvec{1}), n_msg(1));
        THROWS_WITH(std::invalid_argument, ta.propagate_grid({10., 11.}, kw::max_delta_t = {1., 2., 3.}), n_msg(3));
        THROWS_WITH(std::invalid_argument,
                    ta.propagate_grid({10., 11.}, kw::max_delta_t = {1., std::numeric_limits<double>::quiet_NaN()}),
                    "A nan max_delta_t was passed to the propagate_grid()" + tail);
        THROWS_WITH(std::invalid_argument, ta.propagate_grid({10., 11.}, kw::max_delta_t = {1., -1.}),
                    "A non-positive max_delta_t was passed to the propagate_grid()" + tail);
    }

    // "propagate grid" (:162-245), argument checks in the reference's order.
    {
        for (const auto cm : {true, false}) {
            auto ta = tab{pend, {0.05, 0.025, 0.051, 0.0251, 0.052, 0.0252, 0.053, 0.0253}, 4u, kw::compact_mode = cm};
            const std::string batch = " in an adaptive Taylor integrator in batch mode";
            const auto size_msg = [](int n) {
                return "Invalid grid size detected in propagate_grid() for an adaptive Taylor integrator in batch mode: the "
                       "grid has a size of "
                       + std::to_string(n) + ", which is not a multiple of the batch size (4)";
            };
            const auto nf_msg = "A non-finite time value was passed to propagate_grid()" + batch;
            const auto nm_msg = "A non-monotonic time grid was passed to propagate_grid()" + batch;
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({}),
                        "Cannot invoke propagate_grid()" + batch + " if the time grid is empty");
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({1.}), size_msg(1));
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({1., 2.}), size_msg(2));
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({1., 2., 3., 4., 5.}), size_msg(5));
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 1., 4.}),
                        "When invoking propagate_grid(), the first element of the time grid must match the current time "
                        "coordinate - however, the first element of the time grid at batch index 2 has a value of 1, while "
                        "the current time coordinate is 0");
            ta.set_time({0., 0., inf, 0.});
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0.}),
                        "Cannot invoke propagate_grid()" + batch + " if the current time is not finite");
            ta.set_time({0., 0., 0., 0.});
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., inf, 0.}), nf_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 0., inf, 0., 0.}), nf_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 1., -1., 1.}), nm_msg);
            // A row with a non-finite value AND an ordering violation reports the non-finite value.
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 1., 1., 1., 0., 0., 0., inf}), nf_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 1., 1., 1., 2., 0., 0., 2.}), nm_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 0., 1., 1., 1., 2., 2., 2., 2.}), nm_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 0., 1., 1., 2., 2., 2., 2.}), nm_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 1., 1., 0., 2., 2., 2., 2.}), nm_msg);
            THROWS_WITH(std::invalid_argument, ta.propagate_grid({0., 0., 0., 0., 1., 1., 1., 1., 2., 2., 1., 2.}), nm_msg);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Cases which integrate.
// ------------------------------------------------------------------------------------------------------------------
void gpu_cases()
{
    auto [x, v] = make_vars("x", "v");
    const auto pend = std::vector{prime(x) = v, prime(v) = -9.8 * sin(x)};
    const auto osc = std::vector{prime(x) = v, prime(v) = -x};
    const auto all_time_limit = [](const tab &ta) {
        return std::ranges::all_of(ta.get_propagate_res(),
                                   [](const auto &t) { return std::get<0>(t) == taylor_outcome::time_limit; });
    };

    // References returned by the getters stay current across steps, like the members they stand for in the reference: its
    // benchmark/outer_ss_long_term_batch.cpp keeps `const auto &times_v = ta.get_time()` (and an xtensor view over
    // get_state_data()) across the stepping loop.
    {
        auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2u};
        const auto &times_v = ta.get_time();
        const auto &state_v = ta.get_state();
        const auto *state_p = ta.get_state_data();
        const auto x0 = state_v[0];
        auto n = 0;
        while (std::ranges::any_of(times_v, [](double t) { return t < 1.; })) {
            ta.step();
            CHECK(++n < 1000);
        }
        CHECK(times_v[0] >= 1. && times_v[1] >= 1. && state_v[0] != x0 && state_p == state_v.data());
        ta.propagate_until(5.);
        CHECK(times_v[0] == 5. && times_v[1] == 5.);
    }

    // "batch consistency" (:105-160): the forced damped pendulum with per-lane parameters and per-lane initial times,
    // propagated to per-lane final times; the reference compares with four scalar integrators, here: with four batches
    // of one (1000 eps).
    {
        const dvec st{0.01, 0.02, 0.03, 0.04, 1.85, 1.86, 1.87, 1.88}, pv{0.10, 0.11, 0.12, 0.13};
        const auto dyn = std::vector{prime(x) = v, prime(v) = cos(heyoka::time) - par[0] * v - sin(x)};
        auto ta = tab{dyn, st, 4u, kw::pars = pv};
        ta.set_time({0.1, 0.2, 0.3, 0.4});
        ta.propagate_until({20, 21, 22, 23});
        for (auto i = 0u; i < 4u; ++i) {
            auto t1 = tab{dyn, {st[i], st[4u + i]}, 1u, kw::pars = dvec{pv[i]}};
            t1.set_time((i + 1) / 10.);
            t1.propagate_until(20. + i);
            CHECK(close_to(t1.get_state()[0], ta.get_state()[i], 1000 * eps));
            CHECK(close_to(t1.get_state()[1], ta.get_state()[4u + i], 1000 * eps));
            CHECK(t1.get_time()[0] == ta.get_time()[i]);
        }
    }

    // "propagate trivial" (:446-464).
    {
        auto ta = tab{{prime(x) = v, prime(v) = 1_dbl}, {0, 0, 0.1, 0.1}, 2};
        ta.propagate_for({1.2, 1.3});
        CHECK(all_time_limit(ta));
        ta.propagate_until({2.3, 4.5});
        CHECK(all_time_limit(ta));
        ta = tab{{prime(x) = v, prime(v) = 1_dbl}, {0, 0, 0.1, 0.1}, 2};
        ta.propagate_grid({0., 0., 5, 6, 7, 8.});
        CHECK(all_time_limit(ta));
    }

    // "propagate for_until" (:572-730) with max_delta_t 100x larger (1 000 / 2 200 steps per leg instead of 100 000 /
    // 220 000): exact final times forwards and backwards, one callback invocation per iteration of the batch, lanes which
    // are done take steps of length zero, scalar and vector overloads agree bitwise, callbacks are moved, never copied.
    for (const auto cm : {true, false}) {
        auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2u, kw::compact_mode = cm};
        auto ta_copy = ta;
        auto counter0 = 0ul, counter1 = 0ul;
        auto cb = [&counter0, &counter1](tab &t) {
            counter0 += t.get_last_h()[0] != 0;
            counter1 += t.get_last_h()[1] != 0;
            return true;
        };
        const dvec mdt{1e-2, 5e-3};
        const auto agree = [&]() {
            for (auto i = 0u; i < 4u; ++i) {
                CHECK(close_to(ta.get_state()[i], ta_copy.get_state()[i], 1000 * eps));
            }
            CHECK(all_time_limit(ta) && all_time_limit(ta_copy));
        };
        ta.propagate_until({10., 11.}, kw::max_delta_t = mdt, kw::callback = cb);
        ta_copy.propagate_until({10., 11.});
        CHECK((ta.get_time() == dvec{10., 11.}) && (ta_copy.get_time() == dvec{10., 11.}));
        CHECK(counter0 == 1000ul && counter1 == 2200ul);
        agree();

        auto ta_copy2 = ta, ta_copy3 = ta;
        ta_copy2.propagate_until(20.);
        ta_copy3.propagate_until({20., 20.});
        CHECK(ta_copy2.get_state() == ta_copy3.get_state());
        ta_copy2.propagate_until(30., kw::max_delta_t = mdt);
        ta_copy3.propagate_until({30., 30.}, kw::max_delta_t = mdt);
        CHECK(ta_copy2.get_state() == ta_copy3.get_state());
        ta_copy2.propagate_for(20.);
        ta_copy3.propagate_for({20., 20.});
        CHECK(ta_copy2.get_state() == ta_copy3.get_state());

        ta.propagate_for({10., 11.}, kw::max_delta_t = mdt, kw::callback = cb);
        ta_copy.propagate_for({10., 11.});
        CHECK((ta.get_time() == dvec{20., 22.}) && (ta_copy.get_time() == dvec{20., 22.}));
        CHECK(counter0 == 2000ul && counter1 == 4400ul);
        agree();
        ta.propagate_for({-10., -11.}, kw::max_delta_t = mdt, kw::callback = cb);
        ta_copy.propagate_for({-10., -11.});
        CHECK((ta.get_time() == dvec{10., 11.}) && (ta_copy.get_time() == dvec{10., 11.}));
        CHECK(counter0 == 3000ul && counter1 == 6600ul);
        agree();
        ta.propagate_until({0., 0.}, kw::max_delta_t = mdt, kw::callback = cb);
        ta_copy.propagate_until({0., 0.});
        CHECK((ta.get_time() == dvec{0., 0.}) && (ta_copy.get_time() == dvec{0., 0.}));
        CHECK(counter0 == 4000ul && counter1 == 8800ul);
        agree();

        // A scalar max_delta_t is the vector with equal entries.
        ta_copy = ta;
        ta.propagate_until({10., 11.}, kw::max_delta_t = {1e-2, 1e-2});
        ta_copy.propagate_until({10., 11.}, kw::max_delta_t = 1e-2);
        CHECK(ta.get_propagate_res() == ta_copy.get_propagate_res());
        ta.propagate_for({10., 11.}, kw::max_delta_t = {1e-2, 1e-2});
        ta_copy.propagate_for({10., 11.}, kw::max_delta_t = 1e-2);
        CHECK(ta.get_propagate_res() == ta_copy.get_propagate_res());

        // Callbacks handed over as rvalues: moved in, used in place, handed back.
        using cb_u = counting_cb<0>;
        using cb_f = counting_cb<1>;
        step_callback_batch<double> f_until(cb_u{});
        cb_u::n_copies_after = cb_u::n_copies;
        auto [c0, out_cb] = ta.propagate_until(20., kw::callback = std::move(f_until));
        (void)c0;
        CHECK(cb_u::n_calls > 0u);
        out_cb(ta);
        CHECK(value_isa<cb_u>(out_cb));
        step_callback_batch<double> f_for(cb_f{});
        cb_f::n_copies_after = cb_f::n_copies;
        auto ret_for = ta.propagate_for(10., kw::callback = std::move(f_for));
        std::get<1>(ret_for)(ta);
        CHECK(value_isa<cb_f>(std::get<1>(ret_for)));
        {
            // A range of callbacks becomes a callback set.
            std::vector<cb_f> cbs(2);
            cb_f::n_copies_after = cb_f::n_copies;
            auto ret = ta.propagate_for(
                10., kw::callback = cbs | std::views::transform([](cb_f &c) -> cb_f && { return std::move(c); }));
            std::get<1>(ret)(ta);
            CHECK(value_isa<step_callback_batch_set<double>>(std::get<1>(ret)));
        }
    }

    // test/step_callback.cpp "step_callback pre_hook" (:448-490), batch part: pre_hook() runs before the first step (the
    // pendulum's length is set there: same trajectory as an integrator constructed with it); a pre_hook() which moves
    // the time coordinate is reported like a callback doing so.
    {
        const auto dyn = model::pendulum(kw::length = par[0]);
        auto ta0 = tab{dyn, {1., 1.1, 0., 0.1}, 2u};
        auto ta1 = tab{dyn, {1., 1.1, 0., 0.1}, 2u, kw::pars = {1.5, 1.5}};
        CHECK((ta0.get_pars() == dvec{0., 0.}));
        ta0.propagate_until(3., kw::callback = sets_length_in_pre_hook{});
        ta1.propagate_until(3.);
        CHECK((ta0.get_pars() == dvec{1.5, 1.5}));
        CHECK(ta0.get_state() == ta1.get_state());
        THROWS_WITH(std::runtime_error, ta0.propagate_until(6., kw::callback = moves_time_in_pre_hook{}),
                    "The invocation of the callback passed to propagate_until() resulted in the alteration of the time "
                    "coordinate of the integrator - this is not supported");
        CHECK((ta0.get_time() == dvec{4., 4.}));
        ta0.set_time(0.);
        ta0.get_pars_data()[0] = 0.1;
        ta0.get_pars_data()[1] = 0.1;
        ta1.set_time(0.);
        auto [cb0, res0] = ta0.propagate_grid({0., 0., 1., 1., 2., 2.}, kw::callback = sets_length_in_pre_hook{});
        auto [cb1, res1] = ta1.propagate_grid({0., 0., 1., 1., 2., 2.});
        CHECK(static_cast<bool>(cb0) && !cb1);
        CHECK(res0 == res1);
        CHECK((ta0.get_pars() == dvec{1.5, 1.5}));
        THROWS_WITH(std::runtime_error,
                    ta0.propagate_grid({ta0.get_time()[0], ta0.get_time()[1], 4., 4.}, kw::callback = moves_time_in_pre_hook{}),
                    "The invocation of the callback passed to propagate_grid() resulted in the alteration of the time "
                    "coordinate of the integrator - this is not supported");
    }

    // "step_callback_set" (:546-658) and "step_callback range" (:725-790) in propagate_until(): every member runs at
    // every step, in order, whatever the others return; the pre-hooks run once each; ranges become sets.
    {
        using cb_t = step_callback_batch<double>;
        using set_t = step_callback_batch_set<double>;
        const auto fresh = [&]() { return tab{model::pendulum(), {1., 1.1, 0., 0.1}, 2u}; };
        {
            auto ta = fresh();
            ta.propagate_until(10., kw::callback = set_t{});
            CHECK(all_time_limit(ta));
        }
        for (const auto stop_first : {0, 1, 2}) {
            int c1 = 0, c2 = 0;
            auto ta = fresh();
            ta.propagate_until(10., kw::callback = set_t{[&, stop_first](const auto &) {
                                                             CHECK(c1 == c2);
                                                             ++c1;
                                                             return stop_first != 1;
                                                         },
                                                         [&, stop_first](const auto &) {
                                                             ++c2;
                                                             CHECK(c1 == c2);
                                                             return stop_first != 2;
                                                         }});
            CHECK(c1 == c2 && c1 > 0);
            CHECK(oc(ta, 0) == (stop_first == 0 ? taylor_outcome::time_limit : taylor_outcome::cb_stop));
            CHECK(stop_first == 0 || c1 == 1);
        }
        {
            struct counted {
                int *calls, *hooks;
                bool operator()(tab &)
                {
                    ++*calls;
                    return true;
                }
                void pre_hook(tab &)
                {
                    CHECK(*hooks == 0);
                    ++*hooks;
                }
            };
            int a = 0, b = 0, h1 = 0, h2 = 0;
            auto ta = fresh();
            ta.propagate_until(10., kw::callback = set_t{counted{&a, &h1}, counted{&b, &h2}});
            CHECK(all_time_limit(ta) && a == b && a > 0 && h1 == 1 && h2 == 1);
        }
        {
            auto ta = fresh();
            auto r = ta.propagate_until(10., kw::callback = std::initializer_list<cb_t>{});
            CHECK(all_time_limit(ta) && static_cast<bool>(std::get<1>(r)) && value_isa<set_t>(std::get<1>(r)));
            int c1 = 0, c2 = 0;
            auto r2 = ta.propagate_until(20., kw::callback = {cb_t{[&](const auto &) {
                                                                  CHECK(c1 == c2);
                                                                  ++c1;
                                                                  return true;
                                                              }},
                                                              cb_t{[&](const auto &) {
                                                                  ++c2;
                                                                  CHECK(c1 == c2);
                                                                  return true;
                                                              }}});
            CHECK(all_time_limit(ta) && c1 == c2 && c1 > 0 && value_isa<set_t>(std::get<1>(r2)));
            c1 = c2 = 0;
            auto r3 = ta.propagate_until(30., kw::callback = std::vector<cb_t>{[&](const auto &) {
                                                                                   ++c1;
                                                                                   return false;
                                                                               },
                                                                               [&](const auto &) {
                                                                                   ++c2;
                                                                                   return true;
                                                                               }});
            CHECK(oc(ta, 0) == taylor_outcome::cb_stop && c1 == 1 && c2 == 1 && value_isa<set_t>(std::get<1>(r3)));
        }
    }

    // "propagate grid" (:246-445): a non-finite lane, a grid of one point, the harmonic oscillator sampled on 1 000 points
    // per lane (regular and irregular, forwards and backwards) against sin / cos, callbacks moved in and handed back.
    {
        auto ta = tab{pend, {0.05, 0.025, 0.051, 0.0251, 0.052, 0.0252, 0.053, 0.0253}, 4u};
        ta.get_state_data()[0] = inf;
        auto [cb, ret] = ta.propagate_grid({.0, .0, .0, .0});
        CHECK(!cb && ret.size() == 8u);
        CHECK(oc(ta, 0) == taylor_outcome::err_nf_state);
        for (const auto i : {1, 2, 3}) {
            CHECK(oc(ta, i) == taylor_outcome::time_limit);
        }
        const dvec st{0.05, 0.025, 0.051, 0.0251, 0.052, 0.0252, 0.053, 0.0253};
        ta = tab{pend, st, 4u};
        std::tie(cb, ret) = ta.propagate_grid({0., 0., 0., 0.});
        CHECK(!cb && ret == st);
        for (auto i = 0u; i < 4u; ++i) {
            const auto [o, min_h, max_h, ns] = ta.get_propagate_res()[i];
            CHECK(o == taylor_outcome::time_limit && min_h == inf && max_h == 0 && ns == 0u);
        }
        // Deterministic stand-in for the reference's random increments in (0, 0.1).
        std::uint64_t lcg = 12345;
        const auto next_inc = [&lcg]() {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            return 0.1 * (static_cast<double>(lcg >> 11) + 1.) / 9007199254740994.;
        };
        for (const auto sign : {1., -1.}) {
            for (const auto irregular : {false, true}) {
                dvec grid(4000u, 0.);
                for (auto i = 1u; i < 1000u; ++i) {
                    for (auto j = 0u; j < 4u; ++j) {
                        grid[i * 4u + j] = irregular ? grid[(i - 1u) * 4u + j] + sign * next_inc()
                                                     : sign * (i / 100. + j / 10.);
                    }
                }
                ta = tab{osc, {0., 0., 0., 0., 1., 1.1, 1.2, 1.3}, 4u};
                std::tie(cb, ret) = ta.propagate_grid(grid);
                CHECK(!cb && ret.size() == 8000u && all_time_limit(ta));
                const auto tol = (irregular ? (sign > 0 ? 400000. : 800000.) : 10000.) * eps;
                for (auto j = 0u; j < 4u; ++j) {
                    CHECK(ta.get_time()[j] == grid[3996u + j]);
                    for (auto i = 0u; i < 1000u; ++i) {
                        CHECK(close_to(ret[8u * i + j], (1 + j / 10.) * std::sin(grid[i * 4u + j]), tol));
                        CHECK(close_to(ret[8u * i + j + 4u], (1 + j / 10.) * std::cos(grid[i * 4u + j]), tol));
                    }
                }
            }
        }
        using cb_g = counting_cb<2>;
        const dvec st2{0., 0.01, 0.02, 0.03, 1., 1.01, 1.02, 1.03};
        ta = tab{osc, st2, 4};
        step_callback_batch<double> f_grid(cb_g{});
        cb_g::n_copies_after = cb_g::n_copies;
        auto [out_cb, _] = ta.propagate_grid({0., 0., 0., 0., 10., 10., 10., 10., 100., 100., 100., 100.},
                                             kw::callback = std::move(f_grid));
        (void)_;
        out_cb(ta);
        CHECK(value_isa<cb_g>(out_cb) && cb_g::n_calls > 0u);
        std::vector<cb_g> cbs(2);
        cb_g::n_copies_after = cb_g::n_copies;
        auto r2 = ta.propagate_grid({100., 100., 100., 100., 101., 101., 101., 101., 102., 102., 102., 102.},
                                    kw::callback = cbs | std::views::transform([](cb_g &c) -> cb_g && { return std::move(c); }));
        std::get<0>(r2)(ta);
        CHECK(value_isa<step_callback_batch_set<double>>(std::get<0>(r2)));
        CHECK(value_isa<cb_g>(value_ref<step_callback_batch_set<double>>(std::get<0>(r2))[0]));
        const std::string msg = "The invocation of the callback passed to propagate_grid() resulted in the alteration of the "
                                "time coordinate of the integrator - this is not supported";
        ta = tab{osc, st2, 4};
        THROWS_WITH(std::runtime_error,
                    ta.propagate_grid({0., 0., 0., 0., 10., 10., 10., 10., 100., 100., 100., 100.}, kw::callback = [](auto &t) {
                        t.set_time(-100.);
                        return true;
                    }),
                    msg);
        ta = tab{osc, st2, 4};
        THROWS_WITH(std::runtime_error,
                    ta.propagate_grid({0., 0., 0., 0., 10., 10., 10., 10., 100., 100., 100., 100.}, kw::callback = [](auto &t) {
                        t.set_time({t.get_time()[0], -100., t.get_time()[2], t.get_time()[3]});
                        return true;
                    }),
                    msg);
    }

    // "propagate for_until write_tc" (:733-766): the Taylor coefficients are written only on request.
    {
        const auto all_zero = [](tab &t) { return std::ranges::all_of(t.get_tc(), [](double val) { return val == 0.; }); };
        for (const auto use_for : {false, true}) {
            auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2};
            const auto go = [&](const dvec &ts, bool wtc, bool expect_zero) {
                const auto cb = [&, expect_zero](tab &t) {
                    CHECK(all_zero(t) == expect_zero);
                    return true;
                };
                if (use_for) {
                    ta.propagate_for(ts, kw::write_tc = wtc, kw::callback = cb);
                } else {
                    ta.propagate_until(ts, kw::write_tc = wtc, kw::callback = cb);
                }
            };
            go({10., 11.}, false, true);
            go({20., 21.}, true, false);
        }
    }

    // "propagate grid 2" (:802-851) with max_delta_t 100x larger: one callback per iteration, exact grid end points
    // forwards and backwards, a scalar max_delta_t is the vector with equal entries.
    {
        auto ta = tab{pend, {0.05, 0.06, 0.025, 0.026}, 2u};
        // (:795-800; the check follows the propagation to the first grid point.)
        const auto lowest = std::numeric_limits<double>::lowest();
        ta.set_time({0., lowest});
        THROWS_WITH(std::invalid_argument,
                    ta.propagate_grid({0., lowest, 1., std::numeric_limits<double>::max()}, kw::max_delta_t = dvec{}),
                    "The final time passed to the propagate_grid() function of an adaptive Taylor integrator in batch mode "
                    "results in an overflow condition");
        ta.set_time({0., 0.});
        auto counter0 = 0ul, counter1 = 0ul;
        auto cb = [&](tab &t) {
            counter0 += t.get_last_h()[0] != 0;
            counter1 += t.get_last_h()[1] != 0;
            return true;
        };
        auto [cbo, out] = ta.propagate_grid({0., 0., 5., 5.6, 10., 11.}, kw::max_delta_t = dvec{1e-2, 5e-3}, kw::callback = cb);
        CHECK(static_cast<bool>(cbo) && (ta.get_time() == dvec{10., 11.}) && all_time_limit(ta));
        CHECK(counter0 == 1000ul && counter1 == 2200ul);
        std::tie(cbo, out) = ta.propagate_grid({10., 11., 5., 5.6, 1., 1.5}, kw::max_delta_t = dvec{1e-2, 5e-3}, kw::callback = cb);
        CHECK(static_cast<bool>(cbo) && (ta.get_time() == dvec{1., 1.5}) && all_time_limit(ta));
        CHECK(counter0 == 1900ul && counter1 == 4100ul);
        auto ta_copy = ta;
        ta.set_time(0.);
        ta_copy.set_time(0.);
        std::tie(cbo, out) = ta.propagate_grid({0., 0., 5., 5.6, 10., 11.}, kw::max_delta_t = dvec{1e-2, 1e-2});
        CHECK(!cbo);
        auto r = ta_copy.propagate_grid({0., 0., 5., 5.6, 10., 11.}, kw::max_delta_t = 1e-2);
        CHECK(out == std::get<1>(r));
    }

    // test/ensemble_propagate.cpp "batch propagate until / for / grid" (:355-700): an empty ensemble, one copy of the
    // callback per iteration, every iteration bitwise equal to the same propagation done serially, continuous output.
    {
        using ens_cb = counting_copies_cb;
        const auto n_iter = 16u;
        std::vector<dvec> ics(n_iter);
        std::uint64_t lcg = 99;
        const auto small = [&lcg]() {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            return (static_cast<double>(lcg >> 11) / 9007199254740992. - .5) * 200 * eps;
        };
        for (auto &ic : ics) {
            ic = {small(), small(), 1 + small(), 1 + small()};
        }
        const auto gen = [&ics](tab tint, std::size_t i) {
            std::copy(ics[i].begin(), ics[i].end(), tint.get_state_data());
            return tint;
        };
        auto ta = tab{osc, {0., 0., 1., 1.}, 2u};
        const auto serial = [&](std::size_t i) {
            ta.set_time(dvec(2u, 0.));
            std::copy(ics[i].begin(), ics[i].end(), ta.get_state_data());
        };
        auto copies0 = ens_cb::n_copies;
        CHECK(ensemble_propagate_until_batch<double>(ta, 20, 0, gen, kw::callback = ens_cb{}).empty());
        CHECK(ensemble_propagate_for_batch<double>(ta, 20, 0, gen, kw::callback = ens_cb{}).empty());
        CHECK(ens_cb::n_copies == copies0);
        for (const auto use_for : {false, true}) {
            // (The template integrator is at t = 0 whenever an ensemble starts from it.)
            ta = tab{osc, {0., 0., 1., 1.}, 2u};
            copies0 = ens_cb::n_copies;
            auto res = use_for ? ensemble_propagate_for_batch<double>(ta, 20, n_iter, gen, kw::callback = ens_cb{})
                               : ensemble_propagate_until_batch<double>(ta, 20, n_iter, gen, kw::callback = ens_cb{});
            CHECK(ens_cb::n_copies == copies0 + n_iter);
            CHECK(res.size() == n_iter);
            for (auto i = 0u; i < n_iter; ++i) {
                serial(i);
                auto [loc_c, loc_cb] = ta.propagate_until(20);
                CHECK(std::ranges::all_of(std::get<0>(res[i]).get_time(), [](double t) { return close_to(t, 20., 10 * eps); }));
                CHECK(std::get<0>(res[i]).get_state() == ta.get_state());
                CHECK(std::get<0>(res[i]).get_propagate_res() == ta.get_propagate_res());
                CHECK(std::get<1>(res[i]).has_value() == loc_c.has_value());
                CHECK(static_cast<bool>(std::get<2>(res[i])));
            }
            ta = tab{osc, {0., 0., 1., 1.}, 2u};
            auto res_c = use_for ? ensemble_propagate_for_batch<double>(ta, 20, n_iter, gen, kw::c_output = true)
                                 : ensemble_propagate_until_batch<double>(ta, 20, n_iter, gen, kw::c_output = true);
            for (auto i = 0u; i < n_iter; ++i) {
                serial(i);
                auto [loc_c, loc_cb] = ta.propagate_until(dvec(2u, 20.), kw::c_output = true);
                CHECK(std::get<0>(res_c[i]).get_state() == ta.get_state());
                CHECK((*std::get<1>(res_c[i]))(1.5) == (*loc_c)(1.5));
                CHECK(!std::get<2>(res_c[i]));
            }
        }
        dvec grid, grid_splat;
        for (auto i = 0; i <= 20; ++i) {
            grid.push_back(i);
            grid_splat.insert(grid_splat.end(), 2u, static_cast<double>(i));
        }
        ta = tab{osc, {0., 0., 1., 1.}, 2u};
        copies0 = ens_cb::n_copies;
        CHECK(ensemble_propagate_grid_batch<double>(ta, grid, 0, gen, kw::callback = ens_cb{}).empty());
        CHECK(ens_cb::n_copies == copies0);
        auto res_g = ensemble_propagate_grid_batch<double>(ta, grid, n_iter, gen, kw::callback = ens_cb{});
        CHECK(ens_cb::n_copies == copies0 + n_iter && res_g.size() == n_iter);
        for (auto i = 0u; i < n_iter; ++i) {
            serial(i);
            auto [loc_cb, loc_res] = ta.propagate_grid(grid_splat);
            CHECK(std::get<0>(res_g[i]).get_state() == ta.get_state());
            CHECK(std::get<0>(res_g[i]).get_propagate_res() == ta.get_propagate_res());
            CHECK(static_cast<bool>(std::get<1>(res_g[i])));
            CHECK(std::get<2>(res_g[i]) == loc_res);
        }
    }

    // "cb interrupt" (:853-953).
    {
        const dvec st{0.05, 0.06, 0.025, 0.026};
        const auto stopped_after = [](const tab &ta, std::size_t n) {
            return oc(ta, 0) == taylor_outcome::cb_stop && oc(ta, 1) == taylor_outcome::cb_stop && nsteps(ta, 0) == n
                   && nsteps(ta, 1) == n;
        };
        {
            auto ta = tab{pend, st, 2u};
            ta.propagate_until({1., 1.1}, kw::callback = [](auto &) { return false; });
            CHECK(stopped_after(ta, 1u));
            CHECK(ta.get_time()[0] < 1. && ta.get_time()[1] < 1.);
            auto counter = 0u;
            ta.propagate_for({10., 10.1}, kw::callback = [&counter](auto &) { return counter++ != 5u; });
            CHECK(stopped_after(ta, 6u));
            CHECK(ta.get_time()[0] < 10. && ta.get_time()[1] < 10.);
        }
        {
            auto ta = tab{pend, st, 2u};
            auto [cb, res] = ta.propagate_grid({0., 0., 11., 11.1, 12., 12.1}, kw::callback = [](auto &) { return false; });
            CHECK(static_cast<bool>(cb));
            CHECK(std::all_of(res.begin() + 4, res.end(), [](double val) { return std::isnan(val); }));
            CHECK(stopped_after(ta, 1u));
            CHECK(ta.get_time()[0] < 11. && ta.get_time()[1] < 11.);
            auto counter = 0u;
            std::tie(cb, res) = ta.propagate_grid({ta.get_time()[0], ta.get_time()[1], 21., 21.1, 32., 32.1},
                                                  kw::callback = [&counter](auto &) { return counter++ != 5u; });
            CHECK(static_cast<bool>(cb));
            CHECK(stopped_after(ta, 6u));
            CHECK(ta.get_time()[0] < 32. && ta.get_time()[1] < 32.);
        }
        {
            // The grid points of lane 0 are reached within the first step, those of lane 1 are not.
            auto ta = tab{pend, st, 2u};
            auto [cb, res] = ta.propagate_grid({0., 0., 1e-6, 21.1, 2e-6, 32.1}, kw::callback = [](auto &) { return false; });
            CHECK(static_cast<bool>(cb));
            CHECK(stopped_after(ta, 1u));
            CHECK(res.size() == 12u);
            for (const auto i : {0, 1, 2, 3, 4, 6, 8, 10}) {
                CHECK(!std::isnan(res[i]));
            }
            for (const auto i : {5, 7, 9, 11}) {
                CHECK(std::isnan(res[i]));
            }
        }
    }

    // "ev inf state" (:1456-1471): the non-finite lane reports err_nf_state, the others their stopping terminal event.
    {
        auto ta = tab{{prime(x) = 1_dbl}, {0., 0., 0., 0.}, 4, kw::t_events = {te_t(x - 5.)}};
        ta.get_state_data()[2] = inf;
        ta.step({10., 10., 10., 10.});
        for (const auto i : {0, 1, 3}) {
            CHECK(std::get<0>(ta.get_step_res()[i]) == taylor_outcome{-1});
        }
        CHECK(std::get<0>(ta.get_step_res()[2]) == taylor_outcome::err_nf_state);
    }

    // "ev exception callback" (:1473-1556): exceptions of the event callbacks of several lanes are collected into one
    // std::runtime_error naming the batch indices; a single one is rethrown as it is.
    {
        const dvec st{0, 0.01, 0.02, 0.03, .25, .26, .27, .28};
        const auto te_throws = te_t(
            v, kw::callback = [](auto &, int, std::uint32_t) -> bool { throw std::invalid_argument("hello world 1"); });
        const auto what_of = [&](tab ta) {
            try {
                ta.propagate_until({4., 4., 4., 4.});
            } catch (const std::runtime_error &re) {
                return std::string(re.what());
            }
            return std::string("<no std::runtime_error>");
        };
        const auto has = [](const std::string &s, const char *p) { return s.find(p) != std::string::npos; };
        auto w = what_of(tab{pend, st, 4,
                             kw::nt_events = {nte_t(v * v - 1e-10,
                                                    [](auto &, double, int, std::uint32_t) {
                                                        throw std::invalid_argument("hello world 0");
                                                    })},
                             kw::t_events = {te_throws}});
        CHECK(!has(w, "Batch index #0") && !has(w, "hello world 1") && has(w, "hello world 0"));
        CHECK(has(w, "Batch index #1") && has(w, "Batch index #2") && has(w, "Batch index #3"));
        w = what_of(tab{pend, st, 4, kw::nt_events = {nte_t(v * v - 1e-10, [](auto &, double, int, std::uint32_t) {})},
                        kw::t_events = {te_throws}});
        CHECK(!has(w, "Batch index #0") && has(w, "hello world 1"));
        CHECK(has(w, "Batch index #1") && has(w, "Batch index #2") && has(w, "Batch index #3"));
        auto single = tab{pend, {0, 0., 0., 0.03, .25, .25, .25, .28}, 4,
                          kw::nt_events = {nte_t(v * v - 1e-10, [](auto &, double, int, std::uint32_t) {})},
                          kw::t_events = {te_throws}};
        THROWS_WITH(std::invalid_argument, single.propagate_until({4., 4., 4., 4.}), "hello world 1");
    }

    // "event cb time" (:1560-1724): callbacks which move the time coordinate are detected after ALL the callbacks of
    // the step have run; a state made non-finite by a callback surfaces at the next step.
    {
        auto c0 = 0u, c1 = 0u;
        const auto bump = [](unsigned &c, double new_t0) {
            return [&c, new_t0](auto &ta, auto t, auto, auto) {
                CHECK(std::isfinite(t));
                ++c;
                ta.set_time({new_t0, ta.get_time()[1]});
            };
        };
        auto ta = tab(osc, dvec{0., 0., 1., -1.}, 2u,
                      kw::nt_events = {nte_t(x - 1e-5, bump(c0, -10.)), nte_t(x - 1e-5, bump(c1, -10.))});
        THROWS_WITH(std::runtime_error, ta.step(), ev_time_msg);
        CHECK(c0 == 1u && c1 == 1u);

        // Both lanes trigger; the copy shares nothing with the original.
        ta = tab(osc, dvec{0., 0., 1., 1.}, 2u,
                 kw::nt_events = {nte_t(x - 1e-5, bump(c0, -inf)),
                                  nte_t(x - 1e-5, bump(c1, std::numeric_limits<double>::quiet_NaN()))});
        auto ta2(ta);
        c0 = c1 = 0;
        THROWS_WITH(std::runtime_error, ta.step(), ev_time_msg);
        CHECK(c0 == 2u && c1 == 2u);
        THROWS_WITH(std::runtime_error, ta2.step(), ev_time_msg);
        CHECK(c0 == 4u && c1 == 4u);

        const auto poison = [](auto &tint) {
            auto *ptr = tint.get_state_data();
            std::fill(ptr, ptr + 4, inf);
        };
        ta = tab(osc, dvec{0., 0., 1., 1.}, 2u,
                 kw::nt_events = {nte_t(x - 1e-5, [&](auto &tint, auto, auto, auto) { poison(tint); }),
                                  nte_t(x - 1e-5, [&](auto &tint, auto, auto, auto) { poison(tint); })});
        ta.step();
        CHECK(std::get<0>(ta.get_step_res()[0]) == taylor_outcome::success);
        CHECK(std::get<0>(ta.get_step_res()[1]) == taylor_outcome::success);
        ta.step();
        CHECK(std::get<0>(ta.get_step_res()[0]) == taylor_outcome::err_nf_state);
        CHECK(std::get<0>(ta.get_step_res()[1]) == taylor_outcome::err_nf_state);

        // A non-terminal and a terminal callback in the same step.
        ta = tab(osc, dvec{0., 0., 1., -1.}, 2u, kw::nt_events = {nte_t(x - 1e-5, bump(c0, -inf))},
                 kw::t_events = {te_t(
                     x - 2e-5, kw::callback = [&](auto &tint, auto, auto) {
                         ++c1;
                         tint.set_time({-10., tint.get_time()[1]});
                         return true;
                     })});
        c0 = c1 = 0;
        THROWS_WITH(std::runtime_error, ta.step(), ev_time_msg);
        CHECK(c0 == 1u && c1 == 1u);

        ta = tab(osc, dvec{0., 0., 1., 1.}, 2u, kw::t_events = {te_t(
                                                    x - 2e-5, kw::callback = [&](auto &tint, auto, auto) {
                                                        poison(tint);
                                                        return true;
                                                    })});
        ta.step();
        CHECK(std::get<0>(ta.get_step_res()[0]) == taylor_outcome{0});
        CHECK(std::get<0>(ta.get_step_res()[1]) == taylor_outcome{0});
        ta.step();
        CHECK(std::get<0>(ta.get_step_res()[0]) == taylor_outcome::err_nf_state);
        CHECK(std::get<0>(ta.get_step_res()[1]) == taylor_outcome::err_nf_state);
    }

    // "reset cooldowns" (:1726-1751).
    {
        auto ta = tab{pend, {0, 0.01, 0.02, 0.03, .25, .26, .27, .28}, 4,
                      kw::t_events = {te_t(v, kw::callback = [](auto &, int, std::uint32_t) { return false; })}};
        ta.propagate_until({100., 100., 100., 100.});
        const auto any_active = [&]() {
            return std::ranges::any_of(ta.get_te_cooldowns(), [](const auto &lane) {
                return std::ranges::any_of(lane, [](const auto &cd) { return static_cast<bool>(cd); });
            });
        };
        CHECK(any_active());
        ta.reset_cooldowns();
        CHECK(!any_active());
    }

    // "copy semantics" (:1753-1817): a copy (construction and assignment over a default-constructed object) behaves like
    // the original, events, parameters and options included.
    {
        auto ta = tab{pend, {0., 0., 0.5, 0.5}, 2,
                      kw::t_events = {te_t(v, kw::callback = [](auto &, int, std::uint32_t) { return false; })},
                      kw::nt_events = {nte_t(v - par[0], [](auto &, double, int, std::uint32_t) {})},
                      kw::pars = dvec{-1e-4, -1e-4}, kw::high_accuracy = true, kw::compact_mode = true, kw::tol = 1e-11};
        const auto same_behaviour = [&](tab &cp, double t_probe) {
            CHECK(cp.get_nt_events().size() == 1u && cp.get_t_events().size() == 1u);
            CHECK(cp.get_tol() == ta.get_tol() && cp.get_high_accuracy() == ta.get_high_accuracy()
                  && cp.get_compact_mode() == ta.get_compact_mode());
            ta.step();
            cp.step();
            CHECK(ta.get_state() == cp.get_state());
            CHECK(ta.get_dtime() == cp.get_dtime());
            auto r0 = ta.propagate_for(10., kw::c_output = true);
            auto r1 = cp.propagate_for(10., kw::c_output = true);
            CHECK((*std::get<0>(r0))(t_probe) == (*std::get<0>(r1))(t_probe));
        };
        auto cp = ta;
        same_behaviour(cp, 4.1);
        cp = tab{};
        cp = ta;
        same_behaviour(cp, 14.1);
    }

    // "propagate step count te stop bug" (:1819-1845): the step which ends on a stopping terminal event counts.
    {
        const dvec st{0., 0., 0.5, 0.5001};
        auto ta = tab{pend, st, 2, kw::t_events = {te_t(x - 1e-6)}};
        ta.propagate_until({10., 10.});
        CHECK(nsteps(ta, 0) == 1u && nsteps(ta, 1) == 1u);
        ta = tab{pend, st, 2, kw::t_events = {te_t(x - 1e-6)}};
        ta.propagate_grid({0., 0., 1., 1., 2., 2.});
        CHECK(nsteps(ta, 0) == 1u && nsteps(ta, 1) == 1u);
    }

    // "get_set_dtime" (:1864-1942), the part with steps: the low half starts to matter only after many steps.
    {
        auto ta = tab{pend, {0, 0.01, 0.1, 0.11}, 2};
        ta.step();
        CHECK(ta.get_dtime().first[0] != 0 && ta.get_dtime().first[1] != 0);
        CHECK(ta.get_dtime().second[0] == 0 && ta.get_dtime().second[1] == 0);
        for (auto i = 0; i < 1000; ++i) {
            ta.step();
        }
        CHECK(ta.get_dtime().second[0] != 0 && ta.get_dtime().second[1] != 0);
        const auto dtm = ta.get_dtime();
        const auto hi = dtm.first, lo = dtm.second;
        ta.set_dtime(hi, lo);
        CHECK(ta.get_dtime().first == hi && ta.get_dtime().second == lo);
    }

    // "callback ste" (:1944-1983): a stopping terminal event in ONE lane ends propagate_until() for the batch after the
    // step callback has run once.
    {
        auto ta = tab{pend, {-1, -0.0001, -1, -1, 0.025, 0.026, 0.027, 0.028}, 4, kw::t_events = {te_t(x)}};
        int n_invoked = 0;
        ta.propagate_until(10, kw::callback = [&n_invoked](auto &) {
            ++n_invoked;
            return true;
        });
        for (const auto i : {0, 2, 3}) {
            CHECK(oc(ta, i) == taylor_outcome::success && nsteps(ta, i) == 1u);
        }
        CHECK(oc(ta, 1) == taylor_outcome{-1} && nsteps(ta, 1) == 1u);
        CHECK(n_invoked == 1);
    }

    // "propagate_grid tc issue" (:1985-2009): grid points of one lane very close to its current time.
    {
        auto ta = tab(osc, {0., 0., 1., 1.}, 2);
        ta.propagate_until({-.5, -.5});
        const dvec t_grid = {-.5, -.5, -.1, -.4999, .1, .1, .2, .2};
        auto [cb, out] = ta.propagate_grid(t_grid);
        CHECK(!cb);
        CHECK(oc(ta, 0) == taylor_outcome::time_limit && oc(ta, 1) == taylor_outcome::time_limit);
        for (auto i = 0u; i < 4u; ++i) {
            CHECK(close_to(out[i * 4u], std::sin(t_grid[2u * i])));
            CHECK(close_to(out[i * 4u + 1u], std::sin(t_grid[2u * i + 1u])));
            CHECK(close_to(out[i * 4u + 2u], std::cos(t_grid[2u * i])));
            CHECK(close_to(out[i * 4u + 3u], std::cos(t_grid[2u * i + 1u])));
        }
    }

    // "propagate_grid ste" (:2011-2046): a stopping terminal event at t = 0.1; the samples before it are filled in, lane by
    // lane, the rest stays nan.
    {
        auto ta = tab(osc, {0., 0., 1., 1.}, 2, kw::t_events = {te_t(heyoka::time - .1)});
        auto [cb, res] = ta.propagate_grid({0., 0., .1 - 2e-6, 10., .1 - 1e-6, 20., .1 + 1e-6, 30.});
        CHECK(!cb);
        CHECK(res.size() == 16u);
        CHECK(oc(ta, 0) == taylor_outcome{-1} && oc(ta, 1) == taylor_outcome{-1});
        for (const auto i : {0, 1, 2, 3, 4, 6, 8, 10}) {
            CHECK(!std::isnan(res[i]));
        }
        for (const auto i : {5, 7, 9, 11, 12, 13, 14, 15}) {
            CHECK(std::isnan(res[i]));
        }
    }

    // "bug prop_cb time" (:2141-2176): a step callback which moves the time coordinate.
    {
        const std::string msg = "The invocation of the callback passed to propagate_until() resulted in the alteration of "
                                "the time coordinate of the integrator - this is not supported";
        auto ta = tab(osc, dvec{0., 0.1, 1., 1.1}, 2u);
        THROWS_WITH(std::runtime_error, ta.propagate_until(10., kw::callback = [](auto &t) {
            t.set_time(100.);
            return true;
        }),
                    msg);
        ta = tab(osc, dvec{0., 0.1, 1., 1.1}, 2u);
        THROWS_WITH(std::runtime_error, ta.propagate_until(10., kw::callback = [](auto &t) {
            t.set_time({t.get_time()[0], 100.});
            return true;
        }),
                    msg);
    }
}

} // namespace

int main(int argc, char **argv)
{
    const bool with_gpu = argc > 1 && std::string(argv[1]) == "gpu";
    host_cases();
    std::printf("host cases OK (%d checks)\n", n_checks);
    if (with_gpu) {
        gpu_cases();
        std::printf("GPU cases OK (%d checks)\n", n_checks);
    }
    return 0;
}
