// Drop-in check of the C++ interface: this file is written against the reference's public API
// (include/heyoka/taylor.hpp:781-1121, tutorial/batch_mode.cpp:26-141, tutorial/ensemble.cpp:25-57,
// benchmark/two_body_step_batch.cpp:37-80) with the namespace aliased; it must compile unchanged and
// reproduce the published outputs of doc/tut_batch_mode.rst:160-360 on the GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <tuple>
#include <vector>

#include <sstream>

#include "../../heyoka_amd/csrc/cfunc.hpp"
#include "../../heyoka_amd/csrc/ensemble.hpp"
#include "../../heyoka_amd/csrc/model.hpp"
#include "../../heyoka_amd/csrc/taylor_adaptive_batch.hpp"

namespace heyoka = heyoka_amd;
using namespace heyoka;
namespace hy = heyoka;

#define REQUIRE(cond)                                                                                                  \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            std::fprintf(stderr, "REQUIRE failed at line %d: %s\n", __LINE__, #cond);                                \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

static bool close6(double a, double b)
{
    return std::abs(a - b) <= 6e-6 * std::max(1e-300, std::abs(b));
}

int main(int argc, char **argv)
{
    const bool with_gpu = argc > 1 && std::string(argv[1]) == "gpu";

    auto [x, v] = make_vars("x", "v");
    const auto batch_size = 4u;

    // ---- construction + error paths (no GPU needed) ----
    {
        bool thrown = false;
        try {
            taylor_adaptive_batch<double> bad{{prime(x) = v, prime(v) = -x}, std::vector<double>{0., 1., 2.}, 2u};
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what()).find("which is not a multiple of the batch size (2)") != std::string::npos;
        }
        REQUIRE(thrown);
    }

    std::vector<double> state{0.01, 0.02, 0.03, 0.04, 1.85, 1.86, 1.87, 1.88}, pars{0.10, 0.11, 0.12, 0.13};
    auto ta = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = cos(hy::time) - par[0] * v - sin(x)},
                                            std::move(state),
                                            batch_size,
                                            kw::pars = std::move(pars)};
    REQUIRE(ta.get_order() == 20u && ta.get_dim() == 2u && ta.get_batch_size() == 4u);
    REQUIRE(ta.get_tol() == std::numeric_limits<double>::epsilon());
    REQUIRE(ta.get_decomposition().size() == 12u);

    // propagate_grid() argument validation, verbatim messages (test/taylor_adaptive_batch.cpp:174-190).
    {
        const auto expect_msg = [&](std::vector<double> grid, const std::string &msg) {
            bool ok = false;
            try {
                ta.propagate_grid(std::move(grid));
            } catch (const std::invalid_argument &e) {
                ok = msg == e.what();
            }
            REQUIRE(ok);
        };
        expect_msg({}, "Cannot invoke propagate_grid() in an adaptive Taylor integrator in batch mode if the time grid is empty");
        for (const auto n : {1u, 2u, 5u}) {
            expect_msg(std::vector<double>(n, 1.),
                       "Invalid grid size detected in propagate_grid() for an adaptive Taylor integrator in batch mode: "
                       "the grid has a size of "
                           + std::to_string(n) + ", which is not a multiple of the batch size (4)");
        }
    }

    // Zero iterations: empty results, the generator is never called (test/ensemble_propagate.cpp:90-99, :307-315).
    {
        const auto gen = [](taylor_adaptive_batch<double> tint, std::size_t) {
            std::abort();
            return tint;
        };
        REQUIRE(ensemble_propagate_until_batch(ta, 20., 0u, gen).empty());
        REQUIRE(ensemble_propagate_for_batch(ta, 20., 0u, gen, kw::max_steps = 10u).empty());
        REQUIRE(ensemble_propagate_grid_batch(ta, std::vector<double>{0., 1.}, 0u, gen).empty());
    }

    // kwargs accepted like the reference's (LLVM-only ones are ignored).
    auto sys2 = model::nbody(2, kw::masses = {1., 0.});
    auto tad = taylor_adaptive_batch<double>{sys2, std::vector<double>(12u * 8u, 0.), 8u, kw::high_accuracy = true,
                                             kw::tol = 1e-12, kw::compact_mode = false, kw::fast_math = false};
    REQUIRE(tad.get_order() == 15u && tad.get_high_accuracy());
    // MI355X extensions of the keyword arguments: code generator, cluster kernel, exact quotients, outcome semantics
    // (tab_core::config; the same fields as hy_tab_config) - validated at construction.
    {
        auto oss = model::nbody(3, kw::masses = {1., 1e-3, 1e-4});
        auto t5 = taylor_adaptive_batch<double>{oss, std::vector<double>(18u * 4u, 0.), 4u};
        auto t3 = taylor_adaptive_batch<double>{oss, std::vector<double>(18u * 4u, 0.), 4u, kw::cluster_kernel = 3,
                                                kw::exact_division = true, kw::batch_semantics = 2};
        auto tt = taylor_adaptive_batch<double>{oss, std::vector<double>(18u * 4u, 0.), 4u, kw::emitter = 3};
        REQUIRE(t3.core().get_codegen_info().find("v3") != std::string::npos);
        REQUIRE(tt.core().get_codegen_info().rfind("table", 0) == 0u);
        REQUIRE(t5.core().get_codegen_info() != t3.core().get_codegen_info());
        bool thrown = false;
        try {
            taylor_adaptive_batch<double>{oss, std::vector<double>(18u * 4u, 0.), 4u, kw::batch_semantics = 7};
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what()).find("Invalid batch semantics") != std::string::npos;
        }
        REQUIRE(thrown);
    }

    // Default-constructed continuous output (test/c_output.cpp:306-331).
    {
        continuous_output_batch<double> co;
        REQUIRE(co.get_output().empty() && co.get_times().empty() && co.get_tcs().empty());
        REQUIRE(co.get_batch_size() == 0u);
        int n_thrown = 0;
        const auto expect = [&](auto &&f) {
            try {
                f();
            } catch (const std::invalid_argument &e) {
                n_thrown += std::string(e.what()) == "Cannot use a default-constructed continuous_output_batch object";
            }
        };
        expect([&]() { co(0.); });
        expect([&]() { co(std::vector<double>{0., 0.}); });
        expect([&]() { co.get_bounds(); });
        expect([&]() { co.get_n_steps(); });
        REQUIRE(n_thrown == 4);
        std::ostringstream oss;
        oss << co;
        REQUIRE(oss.str().find("Default-constructed continuous_output_batch") != std::string::npos);
    }

    // Events: construction as in tutorial/event_basic.cpp (batch flavour of the classes, include/heyoka/events.hpp).
    std::vector<double> zero_vel_times;
    nt_event_batch<double> ev(v, [&zero_vel_times](taylor_adaptive_batch<double> &, double tm, int, std::uint32_t) {
        zero_vel_times.push_back(tm);
    });
    t_event_batch<double> tev(
        x - 0.04,
        kw::callback = [](taylor_adaptive_batch<double> &t, int, std::uint32_t i) { return t.get_time()[i] < 2.; },
        kw::direction = event_direction::positive, kw::cooldown = 0.01);
    auto tae = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = -9.8 * sin(x)},
                                             std::vector<double>{-0.05, 0.},
                                             1u,
                                             kw::nt_events = {ev},
                                             kw::t_events = {tev}};
    REQUIRE(tae.with_events() && tae.get_te_cooldowns().size() == 1u && !tae.get_te_cooldowns()[0][0].has_value());
    {
        bool thrown = false;
        try {
            nt_event_batch<double> bad(v, nt_event_batch<double>::callback_t{});
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what()) == "Cannot construct a non-terminal event with an empty callback";
        }
        REQUIRE(thrown);
    }

    // operator<< (test/taylor_adaptive_batch.cpp:1268-1345).
    {
        std::ostringstream oss;
        oss << ta;
        const auto out = oss.str();
        for (const char *field : {"Tolerance", "Dimension", "Batch size", "Parameters", "High accuracy", "Compact mode", "Taylor order"}) {
            REQUIRE(out.find(field) != std::string::npos);
        }
        REQUIRE(out.find("events") == std::string::npos);
        std::ostringstream oss2;
        oss2 << tae;
        REQUIRE(oss2.str().find("N of terminal events    : 1") != std::string::npos);
        REQUIRE(oss2.str().find("N of non-terminal events: 1") != std::string::npos);
        REQUIRE(oss2.str().find("Parameters") == std::string::npos);
        REQUIRE(tae.get_t_events().size() == 1u && tae.get_nt_events().size() == 1u);
        {
            // Without events the getters throw (test/taylor_adaptive_batch.cpp:1441-1452).
            bool thrown = false;
            try {
                (void)ta.get_t_events();
            } catch (const std::invalid_argument &e) {
                thrown = std::string(e.what()) == "No events were defined for this integrator";
            }
            REQUIRE(thrown);
        }
        REQUIRE(tae.get_t_events()[0].get_direction() == event_direction::positive
                && tae.get_t_events()[0].get_cooldown() == 0.01);
    }

    // The other point-mass models, with the call syntax and the decomposition sizes of the reference's tests
    // (test/model_cr3bp.cpp:41-48, model_rotating.cpp:65-73, :96-107, model_fixed_centres.cpp:70, model_mascon.cpp:287-299,
    // model_nbody.cpp:492-497).
    {
        auto mk = [](auto dyn, std::size_t n_pars = 0) {
            return taylor_adaptive_batch<double>{dyn, std::vector<double>(dyn.size() * 2u, .5), 2u,
                                                 kw::pars = std::vector<double>(n_pars * 2u, .1)};
        };
        REQUIRE(mk(model::cr3bp()).get_decomposition().size() == 35u);
        REQUIRE(mk(model::cr3bp(kw::mu = 1e-2)).get_decomposition().size() == 35u);
        REQUIRE(mk(model::rotating(kw::omega = {.1, .2, .3})).get_decomposition().size() == 49u);
        REQUIRE(mk(model::rotating(kw::omega = {par[0], par[1], par[2]}), 3).get_decomposition().size() == 52u);
        REQUIRE(model::rotating_potential() == expression{0.});
        const std::vector<double> masses{-3e-3}, pos{1e-3, 2e-3, -4e-3}, omega{.1, .11, .12};
        auto mdyn = model::mascon(kw::masses = masses, kw::positions = pos, kw::Gconst = 1.01, kw::omega = omega);
        REQUIRE(mk(mdyn).get_decomposition().size() == 64u);
        auto fdyn = model::fixed_centres(kw::Gconst = 1.02, kw::masses = {1.01}, kw::positions = {1., 2., 3.});
        REQUIRE(fdyn.size() == 6u && fdyn[3].first == expression{"vx"});
        const std::vector<double> mss{1.00000597682, 1 / 1047.355, 1 / 3501.6, 1 / 22869., 1 / 19314., 7.4074074e-09};
        const auto G = 0.01720209895 * 0.01720209895 * 365 * 365;
        REQUIRE(mk(model::np1body(6, kw::masses = mss, kw::Gconst = G)).get_decomposition().size() == 294u);
        cfunc<double> cf({model::cr3bp_jacobi()}, {"x"_var, "y"_var, "z"_var, "px"_var, "py"_var, "pz"_var});
        REQUIRE(cf.get_dc().size() == 28u);
        bool thrown = false;
        try {
            model::cr3bp(kw::mu = -1.);
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what())
                     == "The 'mu' parameter in a CR3BP must be in the range (0, 0.5), but a value of -1 was provided instead";
        }
        REQUIRE(thrown);
    }

    // Two-argument and piecewise functions with the reference's spelling; decomposition sizes pinned by
    // test/kepE.cpp:183-192, test/atan2.cpp:159-167, test/taylor_atan2.cpp:62-73.
    {
        auto [xx, yy] = make_vars("x", "y");
        auto mk2 = [](std::vector<std::pair<expression, expression>> dyn) {
            return taylor_adaptive_batch<double>{std::move(dyn), std::vector<double>{.1, .2, .3, .4}, 2u, kw::tol = 1.};
        };
        REQUIRE(mk2({prime(xx) = cos(kepE(xx, yy)) + sin(kepE(xx, yy)) + kepE(xx, yy), prime(yy) = xx}).get_decomposition().size()
                == 10u);
        REQUIRE(mk2({prime(xx) = atan2(yy, xx) + (pow(yy, 2_dbl) + pow(xx, 2_dbl)), prime(yy) = xx}).get_decomposition().size() == 7u);
        REQUIRE(mk2({prime(xx) = atan2(xx, yy), prime(yy) = pow(xx, 2_dbl) + pow(yy, 2_dbl)}).get_decomposition().size() == 6u);
        REQUIRE(kepE(0_dbl, xx) == xx);
        auto pw = mk2({prime(xx) = relu(yy) - leaky_relu(.01)(xx) + select(gt(xx, yy), xx, yy * relup(xx, .1)),
                       prime(yy) = logical_and({lt(xx, .5_dbl), gte(yy, -.5_dbl)}) - logical_or({eq(xx, yy), neq(xx, 1_dbl), lte(yy, xx)})});
        REQUIRE(pw.get_decomposition().size() > 10u);
        bool thrown = false;
        try {
            relu(xx, -1.);
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what())
                     == "The slope parameter for a leaky ReLU must be finite and non-negative, but the value -1 was provided instead";
        }
        REQUIRE(thrown);
    }

    if (!with_gpu) {
        std::puts("CPU-only checks OK");
        return 0;
    }

    // ---- tutorial/batch_mode.cpp session ----
    ta.step();
    const double h_exp[] = {0.205801, 0.20587, 0.204791, 0.203963};
    for (auto i = 0u; i < batch_size; ++i) {
        auto [oc, h] = ta.get_step_res()[i];
        REQUIRE(oc == taylor_outcome::success && close6(h, h_exp[i]));
    }
    ta.step({0.010, 0.011, 0.012, 0.013});
    for (auto i = 0u; i < batch_size; ++i) {
        REQUIRE(std::get<0>(ta.get_step_res()[i]) == taylor_outcome::time_limit);
    }
    ta.propagate_for(std::vector<double>{10., 11., 12., 13.});
    const std::size_t n1[] = {34, 38, 41, 44};
    for (auto i = 0u; i < batch_size; ++i) {
        auto [oc, min_h, max_h, nsteps] = ta.get_propagate_res()[i];
        REQUIRE(oc == taylor_outcome::time_limit && nsteps == n1[i]);
    }
    ta.propagate_until(std::vector<double>{20., 21., 22., 23.});
    const std::size_t n2[] = {40, 38, 35, 34};
    const double x_exp[] = {1.801537, 2.833631, 3.399033, 6.072237};
    for (auto i = 0u; i < batch_size; ++i) {
        REQUIRE(std::get<3>(ta.get_propagate_res()[i]) == n2[i]);
        REQUIRE(close6(ta.get_state()[i], x_exp[i]) && ta.get_time()[i] == 20. + i);
    }
    ta.step(true);
    REQUIRE(close6(ta.get_tc()[0 * 21 * 4 + 2 * 4 + 0], -3.508356e-01));
    const auto &d_out = ta.update_d_output({20.1, 21.1, 22.1, 23.1});
    REQUIRE(close6(d_out[0], 1.934202) && close6(d_out[4 + 3], 0.776195));

    // Writing through get_state_data() (reference call sites do this), callback, max_steps.
    ta.get_state_data()[0] = 0.5;
    ta.set_time(0.);
    int n_cb = 0;
    ta.propagate_until(1., kw::callback = [&n_cb](taylor_adaptive_batch<double> &t) {
        ++n_cb;
        return t.get_time()[0] < 10.;
    });
    REQUIRE(n_cb > 0 && ta.get_time()[0] == 1.);
    // Step callbacks with pre_hook(), ranges of callbacks, and the callback handed back with its state
    // (include/heyoka/step_callback.hpp:46-62; src/taylor_adaptive_batch.cpp:1356-1365).
    {
        struct counting_cb {
            int n_pre = 0, n_call = 0;
            bool operator()(taylor_adaptive_batch<double> &)
            {
                ++n_call;
                return true;
            }
            void pre_hook(taylor_adaptive_batch<double> &t)
            {
                ++n_pre;
                t.get_state_data()[1] = 0.25; // a pre_hook may set the state up ...
            }
        };
        ta.set_time(0.);
        auto [co1, cb1] = ta.propagate_until(0.5, kw::callback = counting_cb{});
        const auto *p1 = cb1.extract<counting_cb>();
        REQUIRE(p1 != nullptr && p1->n_pre == 1 && p1->n_call > 0 && !co1);
        // A reference wrapper: the caller's own object is updated.
        counting_cb mine;
        ta.propagate_for(0.5, kw::callback = std::ref(mine));
        REQUIRE(mine.n_pre == 1 && mine.n_call > 0);
        // A range of callbacks -> a callback set: every member runs at every step, the results are and-ed.
        int n_a = 0, n_b = 0;
        std::vector<step_callback_batch<double>> cbs;
        cbs.emplace_back([&n_a](taylor_adaptive_batch<double> &) { return ++n_a < 3; });
        cbs.emplace_back([&n_b](taylor_adaptive_batch<double> &) {
            ++n_b;
            return true;
        });
        cbs.emplace_back(counting_cb{});
        auto [co2, cb2] = ta.propagate_for(50., kw::callback = cbs);
        REQUIRE(n_a == 3 && n_b == 3 && std::get<0>(ta.get_propagate_res()[0]) == taylor_outcome::cb_stop);
        auto *set2 = cb2.extract<step_callback_batch_set<double>>();
        REQUIRE(set2 != nullptr && set2->size() == 3u && (*set2)[2].extract<counting_cb>()->n_pre == 1);
        REQUIRE((*set2)[2].extract<counting_cb>()->n_call == 3);
        // propagate_grid() runs the hook too; a pre_hook which moves the time is rejected.
        ta.set_time(0.);
        counting_cb gcb;
        ta.propagate_grid(std::vector<double>{0., 0., 0., 0., .1, .1, .1, .1, .2, .2, .2, .2}, kw::callback = std::ref(gcb));
        REQUIRE(gcb.n_pre == 1 && gcb.n_call >= 1);
        struct bad_hook {
            bool operator()(taylor_adaptive_batch<double> &)
            {
                return true;
            }
            void pre_hook(taylor_adaptive_batch<double> &t)
            {
                t.set_time(42.);
            }
        };
        bool thrown = false;
        try {
            ta.propagate_for(1., kw::callback = bad_hook{});
        } catch (const std::runtime_error &e) {
            thrown = std::string(e.what()).find("alteration of the time coordinate") != std::string::npos;
        }
        REQUIRE(thrown);
        // Empty callbacks: a set cannot hold them; an empty std::function is "no callback".
        bool thrown2 = false;
        try {
            step_callback_batch_set<double>{step_callback_batch<double>{}};
        } catch (const std::invalid_argument &e) {
            thrown2 = std::string(e.what()) == "Cannot construct a callback set containing one or more empty callbacks";
        }
        REQUIRE(thrown2);
        ta.set_time(0.);
        auto [co3, cb3] = ta.propagate_for(.1, kw::callback = std::function<bool(taylor_adaptive_batch<double> &)>{});
        REQUIRE(!cb3);
        ta.get_state_data()[1] = 0.01;
        ta.set_time(1.);
    }
    ta.propagate_until(100., kw::max_steps = 2u);
    REQUIRE(std::get<0>(ta.get_propagate_res()[0]) == taylor_outcome::step_limit);

    // ---- ensemble (tutorial/ensemble.cpp with the batch integrator) ----
    auto tp = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = -9.8 * sin(x)}, std::vector<double>(2u * 2u, 0.), 2u};
    auto gen = [](taylor_adaptive_batch<double> ta_copy, std::size_t i) {
        for (auto j = 0u; j < 2u; ++j) {
            ta_copy.get_state_data()[j] = 0.05 + (2 * i + j) / 100.;
            ta_copy.get_state_data()[2 + j] = 0.025 + (2 * i + j) / 100.;
        }
        return ta_copy;
    };
    auto ret = ensemble_propagate_until_batch(tp, 20., 5, gen);
    REQUIRE(ret.size() == 5u);
    const auto &last = std::get<0>(ret[4]);
    // Member 9 of doc/tut_ensemble.rst:120-139.
    REQUIRE(std::abs(last.get_state()[1] - 0.12257736827306077) < 1e-12);
    REQUIRE(std::abs(last.get_state()[3] - 0.24068377640981869) < 1e-12);
    REQUIRE(std::get<3>(last.get_propagate_res()[1]) == 124u);

    // Several host threads (kw::device = -3: three workers spread round-robin over the visible devices, here one): the
    // threaded path of the ensemble driver gives bitwise the results of the serial one, with and without callbacks.
    {
        auto ret_t = ensemble_propagate_until_batch(tp, 20., 5, gen, kw::device = -3);
        REQUIRE(ret_t.size() == 5u);
        for (auto i = 0u; i < 5u; ++i) {
            REQUIRE(std::get<0>(ret_t[i]).get_state() == std::get<0>(ret[i]).get_state());
            REQUIRE(std::get<0>(ret_t[i]).get_propagate_res() == std::get<0>(ret[i]).get_propagate_res());
        }
        auto ret_c = ensemble_propagate_until_batch(
            tp, 20., 5, gen, kw::device = -2, kw::callback = [](taylor_adaptive_batch<double> &) { return true; });
        for (auto i = 0u; i < 5u; ++i) {
            REQUIRE(std::get<0>(ret_c[i]).get_state() == std::get<0>(ret[i]).get_state());
        }
    }

    // kw::gather: the final states of all the iterations in one buffer on device 0 (device-to-device copies on one GPU;
    // RCCL send / receive between devices - forced on the single device here through HEYOKA_AMD_GATHER_RCCL=1).
    for (const char *route : {"0", "1"}) {
        setenv("HEYOKA_AMD_GATHER_RCCL", route, 1);
        ensemble_gathered g;
        auto ret_g = ensemble_propagate_until_batch(tp, 20., 5, gen, kw::gather = &g);
        unsetenv("HEYOKA_AMD_GATHER_RCCL");
        REQUIRE(g.dim() == 2u && g.n_total() == 5u * tp.get_batch_size());
        REQUIRE(g.used_rccl() == (route[0] == '1'));
        const auto h = g.to_host();
        for (auto i = 0u; i < 5u; ++i) {
            const auto &st_i = std::get<0>(ret_g[i]).get_state();
            const auto bs = tp.get_batch_size();
            for (auto r = 0u; r < 2u; ++r) {
                for (auto l = 0u; l < bs; ++l) {
                    REQUIRE(h[r * g.n_total() + g.offset(i) + l] == st_i[r * bs + l]);
                }
            }
            REQUIRE(st_i == std::get<0>(ret[i]).get_state());
        }
        // The records which travel with the states: times (hi, lo), outcomes, step counts, min / max |h|.
        const auto thi = g.times_hi(), tlo = g.times_lo();
        const auto pres = g.propagate_res();
        REQUIRE(thi.size() == g.n_total());
        for (auto i = 0u; i < 5u; ++i) {
            const auto &ta_i = std::get<0>(ret_g[i]);
            for (auto l = 0u; l < tp.get_batch_size(); ++l) {
                REQUIRE(thi[g.offset(i) + l] == ta_i.get_dtime().first[l]);
                REQUIRE(tlo[g.offset(i) + l] == ta_i.get_dtime().second[l]);
                REQUIRE(pres[g.offset(i) + l] == ta_i.get_propagate_res()[l]);
            }
        }
    }

    // Every kwarg of the reference is forwarded; the continuous output slot is filled on request.
    auto ret2 = ensemble_propagate_for_batch(tp, 2., 3, gen, kw::c_output = true, kw::max_delta_t = 0.5,
                                             kw::write_tc = true);
    REQUIRE(ret2.size() == 3u && std::get<1>(ret2[2]).has_value());
    REQUIRE(std::get<1>(ret2[2])->get_n_steps() >= 4u);
    auto retg = ensemble_propagate_grid_batch(tp, std::vector<double>{0., 0.4, 0.8}, 3, gen);
    REQUIRE(retg.size() == 3u && std::get<2>(retg[0]).size() == 3u * 2u * 2u);
    {
        // doc/tut_adaptive.rst:324-325: x(0.4) = 0.0232578, v(0.4) = -0.14078 for the IC (0.05, 0.025).
        const auto &g = std::get<2>(retg[0]);
        REQUIRE(close6(g[(1u * 2u + 0u) * 2u + 0u], 0.0232578) && close6(g[(1u * 2u + 1u) * 2u + 0u], -0.14078));
    }

    // ---- continuous output (test/c_output.cpp:289-560): harmonic oscillator, analytical solution, ----
    // ---- agreement with propagate_grid(), bounds, error messages, backward integration.          ----
    for (const bool ha : {false, true}) {
        const auto bs = 5u;
        std::vector<double> ic(2u * bs), final_tm(bs);
        for (auto i = 0u; i < bs; ++i) {
            ic[i] = 0.01 + i / 100.;
            ic[bs + i] = 1.02 + i / 100.;
            final_tm[i] = 10. + i / 10.;
        }
        auto tc = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = -x}, ic, bs, kw::high_accuracy = ha};
        auto [d_out, cbk] = tc.propagate_until(final_tm, kw::c_output = true);
        REQUIRE(d_out.has_value());
        REQUIRE(d_out->get_batch_size() == bs && d_out->get_output().size() == 2u * bs);
        REQUIRE(d_out->get_times().size() == (d_out->get_n_steps() + 2u) * bs);
        REQUIRE(d_out->get_tcs().size() == d_out->get_n_steps() * 2u * 21u * bs);
        const auto [lb, ub] = d_out->get_bounds();
        for (auto j = 0u; j < bs; ++j) {
            REQUIRE(lb[j] == 0. && ub[j] == final_tm[j]);
        }
        // Analytical solution at interior, boundary and (slightly) extrapolated times.
        for (const double frac : {0., 0.113, 0.5, 0.977, 1., 1.001, -0.001}) {
            std::vector<double> tm(bs);
            for (auto j = 0u; j < bs; ++j) {
                tm[j] = frac * final_tm[j];
            }
            (*d_out)(tm);
            for (auto j = 0u; j < bs; ++j) {
                const auto ex = ic[j] * std::cos(tm[j]) + ic[bs + j] * std::sin(tm[j]);
                const auto ev = -ic[j] * std::sin(tm[j]) + ic[bs + j] * std::cos(tm[j]);
                REQUIRE(std::abs(d_out->get_output()[j] - ex) < 1e-13);
                REQUIRE(std::abs(d_out->get_output()[bs + j] - ev) < 1e-13);
            }
        }
        // Scalar-time overload.
        (*d_out)(3.3);
        for (auto j = 0u; j < bs; ++j) {
            REQUIRE(std::abs(d_out->get_output()[j] - (ic[j] * std::cos(3.3) + ic[bs + j] * std::sin(3.3))) < 1e-13);
        }
        // Agreement with propagate_grid() on a batch grid.
        auto tg = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = -x}, ic, bs, kw::high_accuracy = ha};
        const auto n_points = 7u;
        std::vector<double> grid(n_points * bs);
        for (auto i = 0u; i < n_points; ++i) {
            for (auto j = 0u; j < bs; ++j) {
                grid[i * bs + j] = final_tm[j] * i / (n_points - 1u);
            }
        }
        auto [gcb, grid_out] = tg.propagate_grid(grid);
        for (auto i = 0u; i < n_points; ++i) {
            (*d_out)(grid.data() + i * bs);
            for (auto j = 0u; j < 2u * bs; ++j) {
                const auto a = d_out->get_output()[j], b = grid_out[2u * i * bs + j];
                REQUIRE(std::abs(a - b) <= 100. * 2.3e-16 * std::max(1., std::abs(b)));
            }
        }
        // Copies share the device data and evaluate independently.
        auto co3 = *d_out;
        co3(1.5);
        REQUIRE(std::abs(co3.get_output()[0] - (ic[0] * std::cos(1.5) + ic[bs] * std::sin(1.5))) < 1e-13);
        // Error messages.
        bool thrown = false;
        try {
            (*d_out)(std::vector<double>(bs + 1u, 0.));
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what()).find("the vector size is 6, but a size of 5 was expected instead")
                     != std::string::npos;
        }
        REQUIRE(thrown);
        thrown = false;
        try {
            (*d_out)(std::numeric_limits<double>::infinity());
        } catch (const std::invalid_argument &e) {
            thrown = std::string(e.what())
                     == "Cannot compute the continuous output in batch mode at the non-finite time inf";
        }
        REQUIRE(thrown);
        std::ostringstream oss;
        oss << *d_out;
        REQUIRE(oss.str().find("forward") != std::string::npos && oss.str().find("N of steps") != std::string::npos);

        // Backward in time from the final state.
        auto [d_back, cb2] = tc.propagate_until(0., kw::c_output = true);
        REQUIRE(d_back.has_value());
        std::ostringstream oss2;
        oss2 << *d_back;
        REQUIRE(oss2.str().find("backward") != std::string::npos);
        (*d_back)(2.5);
        for (auto j = 0u; j < bs; ++j) {
            REQUIRE(std::abs(d_back->get_output()[j] - (ic[j] * std::cos(2.5) + ic[bs + j] * std::sin(2.5))) < 1e-12);
        }
        // No step taken -> empty optional (src/taylor_adaptive_batch.cpp:1278-1282).
        tc.get_state_data()[bs] = std::numeric_limits<double>::infinity();
        auto [d_none, cb3] = tc.propagate_until(10., kw::c_output = true);
        REQUIRE(!d_none.has_value());
    }

    // ---- events (doc/tut_events.rst:156-162: event times to machine precision) ----
    tae.propagate_until(1.5);
    REQUIRE(zero_vel_times.size() == 2u && zero_vel_times[0] == 0.
            && std::abs(zero_vel_times[1] - 1.003701787940065) < 1e-15);
    // The terminal event x = 0.04 (upwards) continues until t >= 2, then stops the propagation.
    tae.propagate_until(10.);
    {
        const auto oc = std::get<0>(tae.get_propagate_res()[0]);
        REQUIRE(static_cast<std::int64_t>(oc) == -1);
        REQUIRE(tae.get_time()[0] > 2. && tae.get_time()[0] < 3.1);
        REQUIRE(std::abs(tae.get_state()[0] - 0.04) < 1e-15);
        REQUIRE(tae.get_te_cooldowns()[0][0].has_value());
        tae.reset_cooldowns();
        REQUIRE(!tae.get_te_cooldowns()[0][0].has_value());
    }

    std::puts("GPU checks OK");
    return 0;
}
