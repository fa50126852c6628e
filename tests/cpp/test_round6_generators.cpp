// Host-side construction (decomposition, planners, code generators up to the hiprtc call) of the round-6 paths - several classes
// of clusters, the staged table stepper, compact mode, the logger - against the reference's include layout; also run under
// AddressSanitizer / UndefinedBehaviorSanitizer by tests/run_sanitized_host_tests.sh. No GPU needed.
#include <heyoka/heyoka.hpp>
#include <heyoka/logging.hpp>
#include <cstdlib>
#include <iostream>
using namespace heyoka;
int main()
{
    set_logger_level_err();
    // A chain of pendula with cubic bonds (two classes of clusters), forced table (staged), compact mode.
    const unsigned ns = 16;
    std::vector<expression> th, om;
    for (unsigned i = 0; i < ns; ++i) {
        th.push_back(expression{"th_" + std::to_string(i)});
        om.push_back(expression{"om_" + std::to_string(i)});
    }
    std::vector<expression> f;
    for (unsigned i = 0; i < ns; ++i) {
        f.push_back(-1. * sin(th[i]) - 0.01 * om[i]);
    }
    for (unsigned i = 0; i + 1 < ns; ++i) {
        auto d = th[i + 1] - th[i];
        auto g = 0.7 * d + 0.4 * ((d * d) * d);
        f[i] = f[i] + g;
        f[i + 1] = f[i + 1] - g;
    }
    std::vector<std::pair<expression, expression>> sys;
    for (unsigned i = 0; i < ns; ++i) sys.emplace_back(th[i], om[i]);
    for (unsigned i = 0; i < ns; ++i) sys.emplace_back(om[i], f[i]);
    for (int variant = 0; variant < 3; ++variant) {
        if (variant == 1) setenv("HEYOKA_AMD_MULTI_CLASS", "0", 1);
        if (variant == 2) setenv("HEYOKA_AMD_TABLE_LDS", "0", 1);
        taylor_adaptive_batch<double> ta{sys, std::vector<double>(2u * ns * 64u, 0.1), 64u};
        std::cout << ta.core().get_codegen_info().substr(0, 160) << "\n";
    }
    unsetenv("HEYOKA_AMD_MULTI_CLASS");
    unsetenv("HEYOKA_AMD_TABLE_LDS");
    auto oss = model::nbody(6, kw::masses = std::vector<double>{1.00000597682, 1 / 1047.355, 1 / 3501.6, 1 / 22869., 1 / 19314., 7.4074074e-09});
    setenv("HEYOKA_AMD_EMIT_MODE", "table", 1);
    for (bool compact : {false, true}) {
        taylor_adaptive_batch<double> tb{oss, std::vector<double>(36u * 64u, 0.1), 64u, kw::high_accuracy = true, kw::compact_mode = compact};
        std::cout << tb.core().get_codegen_info().substr(0, 160) << "\n";
    }
    std::cout << "round-6 generators under the sanitizers OK\n";
}
