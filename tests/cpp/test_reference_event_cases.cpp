// Event detection in batch mode: the behaviours and known answers which the reference's own unit tests pin down
// (test/batch_event_detection.cpp), restated against the reference's include layout and namespace. Every case names
// the TEST_CASE (file:line) whose assertions it re-expresses. Needs a GPU except for the "host" part.
// usage: test_reference_event_cases [gpu]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include <heyoka/heyoka.hpp>
#include <heyoka/kw.hpp>
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

using tab = taylor_adaptive_batch<double>;
using te_t = tab::t_event_t;
using nte_t = tab::nt_event_t;
using dvec = std::vector<double>;
constexpr auto inf = std::numeric_limits<double>::infinity();
constexpr auto eps = std::numeric_limits<double>::epsilon();
constexpr auto bs = 4u;

// |a - b| <= tol * eps * max(1, |b|): the reference's approximately() comparison.
bool approx(double a, double b, double tol = 100.)
{
    return std::abs(a - b) <= tol * eps * std::max(1., std::abs(b));
}
std::int64_t code(const tab &ta, std::uint32_t i)
{
    return static_cast<std::int64_t>(std::get<0>(ta.get_step_res()[i]));
}
template <typename P>
bool all_step_outcomes(const tab &ta, P pred)
{
    return std::ranges::all_of(ta.get_step_res(), [&](const auto &r) { return pred(std::get<0>(r)); });
}
bool all_step_outcomes_are(const tab &ta, taylor_outcome oc)
{
    return all_step_outcomes(ta, [oc](taylor_outcome o) { return o == oc; });
}

// Steps with per-lane limits until every lane has reported one terminal event; a lane which has reported is frozen with
// a zero limit. Returns the outcome codes of the events. (The loop shared by "te basic", "te close", "te retrigger",
// "te custom cooldown": :866-882, :1146-1166.)
std::vector<std::int64_t> step_until_all_triggered(tab &ta, double limit)
{
    dvec mdt(bs, limit);
    std::vector<std::int64_t> trig(bs, 0);
    for (auto n_trig = 0u; n_trig < bs;) {
        ta.step(mdt);
        for (std::uint32_t i = 0; i < bs; ++i) {
            const auto o = std::get<0>(ta.get_step_res()[i]);
            if (o > taylor_outcome::success) {
                ++n_trig;
                mdt[i] = 0;
                trig[i] = static_cast<std::int64_t>(o);
            } else {
                CHECK(o == taylor_outcome::success || o == taylor_outcome::time_limit);
            }
        }
    }
    return trig;
}
bool all_equal_to(const std::vector<std::int64_t> &v, std::int64_t x)
{
    return std::ranges::all_of(v, [x](auto y) { return y == x; });
}

void host_cases()
{
    // "nte def ctor" (:1020-1027), "te def ctor" (:1818-1826).
    nte_t nte;
    CHECK(nte.get_expression() == 0_dbl);
    CHECK(static_cast<bool>(nte.get_callback()));
    CHECK(nte.get_direction() == event_direction::any);
    te_t te;
    CHECK(te.get_expression() == 0_dbl);
    CHECK(!te.get_callback());
    CHECK(te.get_direction() == event_direction::any);
    CHECK(te.get_cooldown() == -1.);
    // "nte copy semantics" / "te copy semantics" (:44-98): copies carry the same event equation.
    auto v = make_vars("v");
    const auto ex = v + 3_dbl;
    nte_t ev(ex, [](auto &, double, int, std::uint32_t) {});
    auto ev2 = ev;
    CHECK(ev2.get_expression() == ex && ev.get_expression() == ex);
    ev2 = *&ev2;
    ev2 = ev;
    CHECK(ev2.get_expression() == ex);
    // test/taylor_nt_event.cpp "taylor nte" (:175-235) and test/taylor_t_event.cpp "taylor te" (:117-226), batch classes:
    // summaries and constructor checks.
    {
        const auto text = [](const auto &e) {
            std::ostringstream oss;
            oss << e;
            return oss.str();
        };
        const auto has = [](const std::string &str, const char *p) { return str.find(p) != std::string::npos; };
        const auto noop = [](auto &, double, int, std::uint32_t) {};
        const auto eq = v * v - 1e-10;
        auto str = text(nte_t(eq, noop));
        CHECK(has(str, "direction::any") && has(str, "non-terminal"));
        str = text(nte_t(eq, noop, kw::direction = event_direction::positive));
        CHECK(has(str, "event_direction::positive") && has(str, "non-terminal"));
        nte_t e0(eq, noop), e1(eq, noop, kw::direction = event_direction::negative);
        e0 = e1;
        CHECK(has(text(e0), "event_direction::negative"));
        const auto throws_with = [](auto &&f, const std::string &msg) {
            try {
                f();
            } catch (const std::invalid_argument &e) {
                return msg == e.what();
            }
            return false;
        };
        CHECK(throws_with([&]() { nte_t(eq, nte_t::callback_t{}); },
                          "Cannot construct a non-terminal event with an empty callback"));
        CHECK(throws_with([&]() { nte_t(eq, noop, kw::direction = event_direction{50}); },
                          "Invalid value selected for the direction of a non-terminal event"));
        str = text(te_t(eq));
        CHECK(has(str, " event_direction::any") && has(str, " terminal") && has(str, " auto") && has(str, " no"));
        str = text(te_t(eq, kw::direction = event_direction::negative,
                        kw::callback = [](auto &, int, std::uint32_t) { return true; }, kw::cooldown = 1));
        CHECK(has(str, " event_direction::negative") && has(str, " terminal") && has(str, " 1") && has(str, " yes"));
        CHECK(throws_with([&]() { te_t(eq, kw::cooldown = std::numeric_limits<double>::quiet_NaN()); },
                          "Cannot set a non-finite cooldown value for a terminal event"));
        CHECK(throws_with([&]() { te_t(eq, kw::direction = event_direction{50}); },
                          "Invalid value selected for the direction of a terminal event"));
    }
    te_t tev(ex);
    auto tev2 = tev;
    tev2 = *&tev2;
    tev2 = tev;
    CHECK(tev2.get_expression() == ex && tev.get_expression() == ex);
}

void gpu_cases()
{
    auto [x, v] = make_vars("x", "v");
    const auto pend = std::vector{prime(x) = v, prime(v) = -9.8 * sin(x)};
    const dvec st_a{0, 0.01, 0.02, 0.03, .25, .26, .27, .28};

    // "nte single step" / "te single step" (:100-260): forced damped pendulum, event x = -0.1 crossed downwards. The
    // reference compares the batch with its scalar integrator; here: the batch of four against four batches of one
    // (event times to 1000 eps, velocities at the events to 10 000 eps) and the pinned number of terminal triggers.
    {
        const dvec ic{0.00, 0.01, 0.02, 0.03, 1.85, 1.86, 1.87, 1.88}, pv{0.10, 0.11, 0.12, 0.13};
        const auto dyn = std::vector{prime(x) = v, prime(v) = cos(heyoka::time) - par[0] * v - sin(x)};
        const auto eq = x + .1;
        const auto run_nt = [&](const dvec &state, const dvec &pars, std::uint32_t n) {
            std::vector<dvec> times(n), vels(n);
            auto ta = tab{dyn, state, n, kw::pars = pars,
                          kw::nt_events = {nte_t(
                              eq,
                              [&, n](auto &tint, double tm, int, std::uint32_t idx) {
                                  times[idx].push_back(tm);
                                  tint.update_d_output(dvec(n, tm));
                                  vels[idx].push_back(tint.get_d_output()[n + idx]);
                              },
                              kw::direction = event_direction::negative)}};
            while (std::ranges::any_of(ta.get_time(), [](double tm) { return tm < 20; })) {
                ta.step();
                CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
            }
            return std::pair{times, vels};
        };
        const auto run_t = [&](const dvec &state, const dvec &pars, std::uint32_t n) {
            std::vector<dvec> times(n), vels(n);
            auto ta = tab{dyn, state, n, kw::pars = pars,
                          kw::t_events = {te_t(
                              eq,
                              kw::callback =
                                  [&, n](auto &tint, int, std::uint32_t idx) {
                                      times[idx].push_back(tint.get_time()[idx]);
                                      vels[idx].push_back(tint.get_state()[n + idx]);
                                      return true;
                                  },
                              kw::direction = event_direction::negative)}};
            while (std::ranges::any_of(ta.get_time(), [](double tm) { return tm < 20; })) {
                ta.step();
                CHECK(all_step_outcomes(
                    ta, [](taylor_outcome o) { return o == taylor_outcome::success || o == taylor_outcome{0}; }));
            }
            return std::pair{times, vels};
        };
        const auto [nt_times, nt_vels] = run_nt(ic, pv, bs);
        const auto [t_times, t_vels] = run_t(ic, pv, bs);
        const unsigned expected_triggers[] = {2, 1, 1, 1};
        for (std::uint32_t i = 0; i < bs; ++i) {
            const auto [nt1, nv1] = run_nt({ic[i], ic[bs + i]}, {pv[i]}, 1u);
            const auto [tt1, tv1] = run_t({ic[i], ic[bs + i]}, {pv[i]}, 1u);
            CHECK(nt1[0].size() == nt_times[i].size() && !nt1[0].empty());
            CHECK(tt1[0].size() == t_times[i].size() && tt1[0].size() == expected_triggers[i]);
            for (std::size_t j = 0; j < nt1[0].size(); ++j) {
                CHECK(approx(nt1[0][j], nt_times[i][j], 1000.) && approx(nv1[0][j], nt_vels[i][j], 10000.));
            }
            for (std::size_t j = 0; j < tt1[0].size(); ++j) {
                CHECK(approx(tt1[0][j], t_times[i][j], 1000.) && approx(tv1[0][j], t_vels[i][j], 10000.));
            }
        }
    }

    // "nte linear box" / "te linear box" (:262-327): x' = par, event at x = 1; the step limits end exactly AT the event
    // time, which belongs to the next step.
    {
        const dvec lims{1., 1 / 2., 1 / 4., 1 / 8.};
        auto counter = 0u;
        auto ta = tab{{prime(x) = par[0]}, {0., 0., 0., 0.}, 4,
                      kw::nt_events = {nte_t(x - 1.,
                                             [&counter](auto &tint, double tm, int, std::uint32_t idx) {
                                                 CHECK(approx(tm, 1 / tint.get_pars()[idx]));
                                                 ++counter;
                                             })},
                      kw::pars = {1., 2., 4., 8.}};
        ta.step(lims);
        CHECK(counter == 0u && all_step_outcomes_are(ta, taylor_outcome::time_limit));
        ta.step(lims);
        CHECK(counter == 4u && all_step_outcomes_are(ta, taylor_outcome::time_limit));

        counter = 0;
        ta = tab{{prime(x) = par[0]}, {0., 0., 0., 0.}, 4,
                 kw::t_events = {te_t(x - 1., kw::callback = [&counter](auto &, int, std::uint32_t) {
                     ++counter;
                     return true;
                 })},
                 kw::pars = {1., 2., 4., 8.}};
        ta.step(lims);
        CHECK(counter == 0u && all_step_outcomes_are(ta, taylor_outcome::time_limit));
        ta.step(lims);
        CHECK(counter == 4u && all_step_outcomes_are(ta, taylor_outcome{0}));
        CHECK(std::ranges::all_of(ta.get_step_res(), [](const auto &r) { return approx(std::get<1>(r), 0.); }));
    }

    // "nte glancing blow test" (:329-409): two discs in uniform motion, only lane 1 grazes (distance = sum of radii at
    // t = 10: a double root); it may be seen at most twice, the other lanes never.
    {
        auto [x0, vx0, x1, vx1] = make_vars("x0", "vx0", "x1", "vx1");
        auto [y0, vy0, y1, vy1] = make_vars("y0", "vy0", "y1", "vy1");
        const dvec st{0., 0.,  0., 0., 0., 0., 0., 0., -10., -10., -10., -10., 6., 2,  7., 8.,
                      0., 0.,  0., 0., 0., 0., 0., 0., 1.,   1.,   1.,   1.,   0., 0., 0., 0.};
        const auto dist = (x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1) - 4.;
        for (const auto acc : {0., .1}) {
            auto counter = 0u;
            auto ta = tab{{prime(x0) = vx0, prime(y0) = vy0, prime(x1) = vx1, prime(y1) = vy1, prime(vx0) = 0_dbl,
                           prime(vy0) = 0_dbl, prime(vx1) = expression(acc), prime(vy1) = 0_dbl},
                          st, 4,
                          kw::nt_events = {nte_t(dist, [&counter, acc](auto &, double t, int, std::uint32_t idx) {
                              if (acc == 0.) {
                                  CHECK((t - 10.) * (t - 10.) <= eps);
                              }
                              CHECK(idx == 1u);
                              ++counter;
                          })}};
            for (auto i = 0; i < 20; ++i) {
                ta.step({1.3, 1.3, 1.3, 1.3});
                CHECK(all_step_outcomes_are(ta, taylor_outcome::time_limit));
            }
            CHECK(counter <= 2u);
        }
    }

    // "nte multizero" (:411-690) and "nte multizero negative timestep" (:692-768): the zeros of v and the two zeros of
    // v^2 - 1e-10 around each of them (1e-5 apart in v) - chronological order within a step, times inside the step,
    // dense output at the event time reproduces the event equation; forwards, backwards, tighter tolerance, direction.
    {
        struct log_t {
            std::vector<unsigned> counter = std::vector<unsigned>(bs, 0u);
            dvec cur_time = dvec(bs, 0.);
        };
        const auto make = [&](log_t &lg, bool fwd, double tol, bool only_negative) {
            const auto common = [&lg, fwd](tab &ta_, double t, std::uint32_t idx) {
                CHECK(fwd ? (t > lg.cur_time[idx]) : (t < lg.cur_time[idx]));
                CHECK(fwd ? (ta_.get_time()[idx] > t) : (ta_.get_time()[idx] < t));
                ta_.update_d_output({t, t, t, t});
                const auto vel = ta_.get_d_output()[4u + idx];
                ++lg.counter[idx];
                lg.cur_time[idx] = t;
                return vel;
            };
            auto sq = nte_t(v * v - 1e-10, [&lg, common, only_negative](auto &ta_, double t, int, std::uint32_t idx) {
                const auto c = lg.counter[idx];
                if (only_negative) {
                    CHECK(c == 0u || (c >= 2u && c <= 6u) || (c >= 7u && c <= 9u));
                } else {
                    CHECK(c % 3u == 0u || c % 3u == 2u);
                }
                const auto vel = common(ta_, t, idx);
                CHECK(std::abs(vel * vel - 1e-10) < eps);
            });
            const auto lin_cb = [&lg, common, only_negative](auto &ta_, double t, int, std::uint32_t idx) {
                const auto c = lg.counter[idx];
                CHECK(only_negative ? (c == 1u || c == 6u) : (c % 3u == 1u));
                CHECK(std::abs(common(ta_, t, idx)) <= eps * 100);
            };
            auto lin = only_negative ? nte_t(v, lin_cb, kw::direction = event_direction::negative) : nte_t(v, lin_cb);
            return tab{pend, st_a, 4, kw::tol = tol, kw::nt_events = {sq, lin}};
        };
        for (const auto tol : {eps, eps / 100}) {
            for (const auto only_negative : {false, true}) {
                log_t lg;
                auto ta = make(lg, true, tol, only_negative);
                ta.propagate_until({4., 4., 4., 4.});
                for (auto i = 0u; i < bs; ++i) {
                    CHECK(std::get<0>(ta.get_propagate_res()[i]) == taylor_outcome::time_limit);
                    CHECK(lg.counter[i] == (only_negative ? 10u : 12u));
                }
            }
        }
        log_t lg;
        auto ta = make(lg, false, eps, false);
        ta.propagate_until({-4., -4., -4., -4.});
        for (auto i = 0u; i < bs; ++i) {
            CHECK(std::get<0>(ta.get_propagate_res()[i]) == taylor_outcome::time_limit);
            CHECK(lg.counter[i] == 12u);
        }
    }

    // "nte basic" (:770-813): pendulum released at rest: v = 0 at t = 0 exactly, then every half period; the third
    // zero is the period (known to 30+ digits for each amplitude).
    {
        const dvec periods{2.0149583072955119566777324135479727911105583481363,
                           2.015602866455777600694040810649276304933055944554756,
                           2.0162731039077591887007722648120652760856018525920970125217,
                           2.01696906642817313582861191326257261662145101139954930969969};
        std::vector<unsigned> counter(bs, 0u);
        auto ta = tab{pend, {-0.25, -0.26, -0.27, -0.28, 0., 0., 0., 0.}, 4,
                      kw::nt_events = {nte_t(v, [&](auto &, double t, int, std::uint32_t idx) {
                          if (counter[idx] == 0u) {
                              CHECK(t == 0);
                          }
                          if (counter[idx] == 2u) {
                              CHECK(approx(t, periods[idx], 1000.));
                          }
                          ++counter[idx];
                      })}};
        for (auto i = 0; i < 20; ++i) {
            ta.step();
            CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
        }
        CHECK(std::ranges::all_of(counter, [](unsigned c) { return c == 3u; }));
    }

    // "te basic" (:815-980): a terminal event at v = 0 between the two non-terminal zeros of v^2 - 1e-10; the
    // non-terminal event after the terminal one belongs to the next step. Forwards twice, backwards twice.
    for (const auto tol : {eps, eps / 100}) {
        std::vector<unsigned> counter_nt(bs, 0u), counter_t(bs, 0u);
        dvec cur_time(bs, 0.);
        bool fwd = true;
        const auto ordered = [&](double t, std::uint32_t idx) { return fwd ? (t > cur_time[idx]) : (t < cur_time[idx]); };
        auto ta = tab{pend, st_a, 4, kw::tol = tol,
                      kw::nt_events = {nte_t(v * v - 1e-10,
                                             [&](auto &ta_, double t, int, std::uint32_t idx) {
                                                 CHECK(ordered(t, idx));
                                                 ta_.update_d_output({t, t, t, t});
                                                 const auto vel = ta_.get_d_output()[4u + idx];
                                                 CHECK(std::abs(vel * vel - 1e-10) < eps);
                                                 ++counter_nt[idx];
                                                 cur_time[idx] = t;
                                             })},
                      kw::t_events = {te_t(v, kw::callback = [&](auto &ta_, int, std::uint32_t idx) {
                          const auto t = ta_.get_time()[idx];
                          CHECK(ordered(t, idx));
                          CHECK(std::abs(ta_.get_state()[4u + idx]) < eps * 100);
                          ++counter_t[idx];
                          cur_time[idx] = t;
                          return true;
                      })}};
        const unsigned expected[4][2] = {{1, 1}, {3, 2}, {5, 3}, {7, 4}};
        for (auto leg = 0; leg < 4; ++leg) {
            fwd = leg < 2;
            const auto trig = step_until_all_triggered(ta, fwd ? inf : -inf);
            CHECK(std::ranges::all_of(trig, [](auto c) { return c >= 0; }));
            for (std::uint32_t i = 0; i < bs; ++i) {
                CHECK(counter_nt[i] == expected[leg][0] && counter_t[i] == expected[leg][1]);
            }
        }
    }

    // "nte dir test" (:982-1018): only the zeros of v with positive derivative; going back the same times come up in
    // reverse order.
    {
        bool fwd = true;
        std::vector<dvec> tlist(bs);
        std::vector<dvec::reverse_iterator> rit(bs);
        auto ta = tab{pend, {-0.25, -0.26, -0.27, -0.28, 0., 0., 0., 0.}, 4,
                      kw::nt_events = {nte_t(
                          v,
                          [&](auto &, double t, int d_sgn, std::uint32_t idx) {
                              CHECK(d_sgn == 1);
                              if (fwd) {
                                  tlist[idx].push_back(t);
                              } else if (rit[idx] != tlist[idx].rend()) {
                                  CHECK(approx(*rit[idx], t));
                                  ++rit[idx];
                              }
                          },
                          kw::direction = event_direction::positive)}};
        ta.propagate_until({20, 20, 20, 20});
        CHECK(std::ranges::all_of(tlist, [](const auto &l) { return l.size() >= 9u; }));
        fwd = false;
        for (auto i = 0u; i < bs; ++i) {
            rit[i] = tlist[i].rbegin();
        }
        ta.propagate_until({0, 0, 0, 0});
        for (auto i = 0u; i < bs; ++i) {
            // (The zero at t = 0 itself, where the backward propagation ends, may or may not be seen again.)
            CHECK(rit[i] == tlist[i].rend() || rit[i] + 1 == tlist[i].rend());
        }
    }

    // "te identical" (:1074-1123): the same terminal event twice: one of the two is reported, then - if anything - the
    // other one.
    {
        te_t ev(v);
        auto ta = tab{pend, st_a, 4, kw::t_events = {ev, ev}};
        do {
            ta.step();
        } while (!std::ranges::any_of(ta.get_step_res(),
                                      [](const auto &r) { return std::get<0>(r) > taylor_outcome::success; }));
        std::vector<std::int64_t> first(bs);
        for (std::uint32_t i = 0; i < bs; ++i) {
            CHECK(std::get<0>(ta.get_step_res()[i]) > taylor_outcome::success);
            first[i] = -code(ta, i) - 1;
            CHECK(first[i] == 0 || first[i] == 1);
        }
        ta.step();
        for (std::uint32_t i = 0; i < bs; ++i) {
            if (std::get<0>(ta.get_step_res()[i]) > taylor_outcome::success) {
                const auto second = -code(ta, i) - 1;
                CHECK((second == 0 || second == 1) && second != first[i]);
            } else {
                CHECK(std::get<0>(ta.get_step_res()[i]) == taylor_outcome::success);
            }
        }
    }

    // "te close" (:1125-1264): two terminal events two ulps apart: going down through x = 0 the one with the callback
    // (index 1, continuing: outcome 1) comes first, then the stopping one (index 0: outcome -1); backwards the other way
    // round.
    {
        te_t ev1(x);
        te_t ev2(x - eps * 2, kw::callback = [](auto &, int, std::uint32_t) { return true; });
        auto ta = tab{pend, {0.1, 0.11, 0.12, 0.13, .25, .26, .27, .28}, 4, kw::t_events = {ev1, ev2}};
        CHECK(all_equal_to(step_until_all_triggered(ta, inf), 1));
        CHECK(all_equal_to(step_until_all_triggered(ta, inf), -1));
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
        CHECK(all_equal_to(step_until_all_triggered(ta, -inf), -1));
        CHECK(all_equal_to(step_until_all_triggered(ta, -inf), 1));
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
    }

    // "te retrigger" (:1266-1315): the event equation starts 6 ulps past its zero: detected in the first step and again
    // - after the cooldown - at every return.
    {
        te_t ev(x - (par[0] - eps * 6));
        auto ta = tab{pend, {1., 1.01, 1.02, 1.03, 0., 0.01, 0.02, 0.03}, 4, kw::t_events = {ev},
                      kw::pars = dvec{1., 1.01, 1.02, 1.03}};
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
        CHECK(std::ranges::all_of(ta.get_time(), [](double t) { return t != 0; }));
        CHECK(all_equal_to(step_until_all_triggered(ta, inf), -1));
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
    }

    // "te dir" (:1317-1399): direction filters on a terminal event; the state at the event is the mirror image / the
    // initial position.
    {
        const dvec st{1., 1.01, 1.02, 1.03, 0., 0., 0., 0.};
        const auto until_all_event0 = [](tab &ta) {
            do {
                ta.step();
            } while (!all_step_outcomes_are(ta, taylor_outcome{0}));
        };
        auto ta = tab{pend, st, 4,
                      kw::t_events = {te_t(
                          v,
                          kw::callback =
                              [](auto &, int d_sgn, std::uint32_t) {
                                  CHECK(d_sgn == 1);
                                  return true;
                              },
                          kw::direction = event_direction::positive)}};
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
        until_all_event0(ta);
        for (auto i = 0u; i < bs; ++i) {
            CHECK(approx(ta.get_state()[i], -st[i]));
        }
        ta = tab{pend, st, 4,
                 kw::t_events = {te_t(
                     v,
                     kw::callback =
                         [](auto &, int d_sgn, std::uint32_t) {
                             CHECK(d_sgn == -1);
                             return true;
                         },
                     kw::direction = event_direction::negative)}};
        // A step of length zero sees nothing; the next one finds the zero at the very beginning of the step.
        ta.step({0., 0., 0., 0.});
        CHECK(all_step_outcomes_are(ta, taylor_outcome::time_limit));
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome{0}));
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
        until_all_event0(ta);
        for (auto i = 0u; i < bs; ++i) {
            CHECK(approx(ta.get_state()[i], st[i]));
        }
    }

    // "te custom cooldown" (:1401-1438): a double zero would retrigger at once without the user's cooldown.
    {
        te_t ev(
            v * v - eps * 4, kw::callback = [](auto &, int, std::uint32_t) { return true; }, kw::cooldown = 1e-1);
        auto ta = tab{pend, st_a, 4, kw::t_events = {ev}};
        CHECK(all_equal_to(step_until_all_triggered(ta, inf), 0));
    }

    // "te propagate_for" (:1440-1477), "te propagate_grid" (:1479-1534), "te propagate_grid first step bug"
    // (:1536-1578).
    {
        std::vector<unsigned> counter(bs, 0u);
        te_t ev(
            v, kw::callback = [&counter](auto &, int, std::uint32_t idx) {
                ++counter[idx];
                return true;
            });
        const auto all_res = [](const tab &ta, taylor_outcome oc) {
            return std::ranges::all_of(ta.get_propagate_res(), [oc](const auto &r) { return std::get<0>(r) == oc; });
        };
        auto ta = tab{pend, st_a, 4, kw::t_events = {ev}};
        ta.propagate_for({100, 100, 100, 100});
        CHECK(all_res(ta, taylor_outcome::time_limit));
        CHECK(std::ranges::all_of(ta.get_time(), [](double t) { return t == 100.; }));
        CHECK(std::ranges::all_of(counter, [](unsigned c) { return c == 100u; }));
        ta = tab{pend, st_a, 4, kw::t_events = {te_t(v)}};
        ta.propagate_for({100, 100, 100, 100});
        CHECK(all_res(ta, taylor_outcome{-1}));

        std::ranges::fill(counter, 0u);
        dvec grid;
        for (auto i = 0; i < 101; ++i) {
            grid.insert(grid.end(), bs, static_cast<double>(i));
        }
        ta = tab{pend, st_a, 4, kw::t_events = {ev}};
        auto [cb, out] = ta.propagate_grid(grid);
        CHECK(!cb && out.size() == 202u * 4u);
        CHECK(std::all_of(out.begin() + 1, out.end(), [](double val) { return val != 0; }));
        CHECK(std::ranges::all_of(counter, [](unsigned c) { return c == 100u; }) && all_res(ta, taylor_outcome::time_limit));
        ta = tab{pend, st_a, 4, kw::t_events = {te_t(v)}};
        std::tie(cb, out) = ta.propagate_grid(grid);
        CHECK(!cb);
        CHECK(std::all_of(out.begin() + 8, out.end(), [](double val) { return std::isnan(val); }));
        CHECK(all_res(ta, taylor_outcome{-1}));

        // Several grid points inside the very first step, which ends on a terminal event.
        grid.clear();
        for (auto i = 0; i < 100; ++i) {
            grid.insert(grid.end(), bs, 5 / 100. * i);
        }
        const dvec st{0.05, 0.051, 0.052, 0.053, 0.025, 0.0251, 0.0252, 0.0253};
        ta = tab{pend, st, 4, kw::t_events = {te_t(v, kw::callback = [](auto &, int, std::uint32_t) { return true; })}};
        std::tie(cb, out) = ta.propagate_grid(grid);
        CHECK(!cb && out.size() == 200u * 4u);
        CHECK(std::ranges::all_of(out, [](double val) { return val != 0; }));
        ta = tab{pend, st, 4, kw::t_events = {te_t(v)}};
        std::tie(cb, out) = ta.propagate_grid(grid);
        CHECK(!cb && out.size() == 200u * 4u);
        CHECK(std::all_of(out.begin() + 32, out.end(), [](double val) { return std::isnan(val); }));
    }

    // "te damped pendulum" (:1580-1645): the callback of the terminal event switches the damping on and off through the
    // parameter array; 99 zeros of the velocity up to t = 100, the 100th in the step after.
    {
        std::vector<dvec> zero_vel_times(bs);
        te_t ev(v, kw::callback = [&zero_vel_times](auto &ta, int, std::uint32_t idx) {
            const auto tm = ta.get_time()[idx];
            ta.get_pars_data()[idx] = (ta.get_pars()[idx] == 0) ? 1 : 0;
            zero_vel_times[idx].push_back(tm);
            return true;
        });
        auto ta = tab{{prime(x) = v, prime(v) = -9.8 * sin(x) - par[0] * v},
                      {0.05, 0.051, 0.052, 0.053, 0.025, 0.0251, 0.0252, 0.0253}, 4, kw::t_events = {ev}};
        const auto counts = [&](std::size_t n) {
            return std::ranges::all_of(zero_vel_times, [n](const auto &l) { return l.size() == n; });
        };
        ta.propagate_until({100, 100, 100, 100});
        CHECK(counts(99u));
        ta.step();
        CHECK(counts(100u));
        ta.set_time({0, 0, 0, 0});
        for (auto i = 0u; i < bs; ++i) {
            zero_vel_times[i].clear();
            ta.get_state_data()[i] = 0.05 + i * 0.001;
            ta.get_state_data()[4u + i] = 0.025 + i * 0.0001;
        }
        do {
            ta.step();
        } while (all_step_outcomes_are(ta, taylor_outcome::success));
        ta.propagate_until({100, 100, 100, 100});
        CHECK(counts(99u));
        ta.step();
        CHECK(counts(100u));
    }

    // "te boolean callback" (:1647-1721): the callback asks to stop at its fifth invocation, forwards and backwards.
    {
        std::vector<unsigned> counter_t(bs, 0u);
        dvec cur_time(bs);
        bool fwd = true;
        auto ta = tab{pend, st_a, 4, kw::t_events = {te_t(v, kw::callback = [&](auto &ta_, int, std::uint32_t idx) {
                          const auto t = ta_.get_time()[idx];
                          CHECK(fwd ? (t > cur_time[idx]) : (t < cur_time[idx]));
                          CHECK(std::abs(ta_.get_state()[4u + idx]) < eps * 100);
                          ++counter_t[idx];
                          cur_time[idx] = t;
                          return counter_t[idx] != 5u;
                      })}};
        do {
            ta.step();
        } while (!all_step_outcomes_are(ta, taylor_outcome{0}));
        ta.propagate_until({1000., 1000., 1000., 1000.});
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
        std::ranges::fill(counter_t, 0u);
        fwd = false;
        do {
            ta.step_backward();
        } while (!all_step_outcomes_are(ta, taylor_outcome{0}));
        ta.propagate_until({-1000., -1000., -1000., -1000.});
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
    }

    // "te step end" (:1723-1747): an event on the time variable which coincides with the end of a (limited) step is
    // seen exactly once, at exactly t = 1.
    {
        std::vector<unsigned> counter(bs);
        auto ta = tab{pend, st_a, 4, kw::t_events = {te_t(heyoka::time - 1., kw::callback = [&](auto &ta_, int, std::uint32_t idx) {
                          ++counter[idx];
                          CHECK(ta_.get_time()[idx] == 1.);
                          return true;
                      })}};
        ta.propagate_until({10., 10., 10., 10.}, kw::max_delta_t = {0.005, 0.005, 0.005, 0.005});
        CHECK(std::ranges::all_of(counter, [](unsigned c) { return c == 1u; }));
    }

    // test/taylor_t_event.cpp "te open range" (:1142-1181) in batch form: a step covers [0, h): an event exactly at the
    // end of a limited step belongs to the next step, which then reports it with a step of length zero.
    {
        const auto t_ev = 97 / 100000.;
        const auto t_next = std::nextafter(t_ev, 1.);
        const dvec st0{-0.25, -0.25, -0.25, -0.25, 0., 0., 0., 0.};
        auto ta = tab{pend, st0, 4, kw::t_events = {te_t(heyoka::time - t_ev)}};
        const auto restart_at = [&](double t) {
            ta.set_time(t);
            std::copy(st0.begin(), st0.end(), ta.get_state_data());
            ta.reset_cooldowns();
        };
        ta.step(dvec(bs, t_ev));
        CHECK(all_step_outcomes_are(ta, taylor_outcome::time_limit));
        restart_at(0.);
        ta.step(dvec(bs, t_next));
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
        restart_at(t_ev);
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
        CHECK(std::ranges::all_of(ta.get_step_res(), [](const auto &r) { return std::get<1>(r) == 0; }));
        restart_at(t_next);
        ta.step();
        CHECK(all_step_outcomes_are(ta, taylor_outcome::success));
        CHECK(std::ranges::all_of(ta.get_step_res(), [](const auto &r) { return std::get<1>(r) > 0; }));
    }

    // "te zero cd mr bug" (:1749-1785): zero cooldown and a callback which stops.
    {
        auto ta = tab{pend, st_a, 4,
                      kw::t_events = {te_t(
                          v, kw::callback = [](auto &, int, std::uint32_t) { return false; }, kw::cooldown = 0)}};
        ta.propagate_until({10., 10., 10., 10.});
        CHECK(all_step_outcomes_are(ta, taylor_outcome{-1}));
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
