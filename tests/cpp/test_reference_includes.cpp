// Written against the reference's own include layout and namespace, UNCHANGED call sites of tutorial/batch_mode.cpp:26-141
// (minus the xtensor pretty-printing) and benchmark/two_body_step_batch.cpp:37-80: compiled with only
//   g++ -std=c++20 -I include this_file.cpp -L heyoka_amd -lheyoka_amd
#include <cstddef>
#include <iostream>
#include <utility>
#include <vector>

#include <heyoka/callable.hpp>
#include <heyoka/exceptions.hpp>
#include <heyoka/func.hpp>
#include <heyoka/heyoka.hpp>
#include <heyoka/kw.hpp>
#include <heyoka/logging.hpp>
#include <heyoka/model/nbody.hpp>
#include <heyoka/math.hpp>
#include <heyoka/models.hpp>
#include <heyoka/number.hpp>
#include <heyoka/param.hpp>
#include <heyoka/step_callback.hpp>
#include <heyoka/taylor.hpp>
#include <heyoka/variable.hpp>

using namespace heyoka;
namespace hy = heyoka;

int main(int argc, char **)
{
    // include/heyoka/logging.hpp:19-24 (the reference's default level is warn: restored right away).
    heyoka::set_logger_level_err();
    heyoka::set_logger_level_warn();
    // tutorial/batch_mode.cpp
    auto [x, v] = make_vars("x", "v");
    const auto batch_size = 4u;
    auto ta = taylor_adaptive_batch<double>{{prime(x) = v, prime(v) = -9.8 * sin(x)},
                                            {0.01, 0.02, 0.03, 0.04, 1.85, 1.86, 1.87, 1.88},
                                            batch_size};
    // benchmark/two_body_step_batch.cpp
    const auto masses = std::vector{1.989e30, 1.989e30 / 333000};
    auto sys = model::nbody(2, kw::masses = masses, kw::Gconst = 6.674e-11);
    std::vector<double> init_state(12u * batch_size, 0.);
    for (unsigned i = 0; i < batch_size; ++i) {
        init_state[6u * batch_size + i] = 1.5e11;
        init_state[10u * batch_size + i] = 29800. + i;
    }
    auto tb = taylor_adaptive_batch<double>{sys, std::move(init_state), batch_size, kw::tol = 1e-18, kw::high_accuracy = true};
    // <heyoka/math/constants.hpp>, <heyoka/math/kepF.hpp>, <heyoka/math/kepDE.hpp>: heyoka::pi is a function without arguments
    // with its own u variable; kepF / kepDE bring their hidden dependencies (all three are registered node rules here).
    auto tk = taylor_adaptive_batch<double>{{prime(x) = kepF(0.1 * x, 0.2_dbl, v + hy::pi), prime(v) = kepDE(0.1_dbl, 0.2 * v, x) - hy::pi * x},
                                            {0.01, 0.02, 0.03, 0.04, 1.85, 1.86, 1.87, 1.88},
                                            batch_size};
    if (tk.get_decomposition().size() != 20u) {
        std::cout << "unexpected decomposition size " << tk.get_decomposition().size() << '\n';
        return 1;
    }
    std::cout << "order " << ta.get_order() << ' ' << tb.get_order() << ' ' << tb.get_decomposition().size() << '\n';
    if (argc > 1) {
        // On a GPU: the tutorial's calls.
        ta.step();
        for (auto i = 0u; i < batch_size; ++i) {
            auto [res, h] = ta.get_step_res()[i];
            std::cout << "Batch index " << i << ": (" << res << ", " << h << ")\n";
        }
        ta.propagate_for({10., 11., 12., 13.});
        ta.propagate_until({20., 21., 22., 23.});
        for (auto i = 0u; i < batch_size; ++i) {
            auto [res, min_h, max_h, nsteps] = ta.get_propagate_res()[i];
            std::cout << "Batch index " << i << ": (" << res << ", " << min_h << ", " << max_h << ", " << nsteps << ")\n";
        }
        for (int k = 0; k < 10; ++k) {
            tb.step();
        }
        std::cout << "GPU OK\n";
    }
    std::cout << "reference include layout OK\n";
    return 0;
}
