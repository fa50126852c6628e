#!/bin/bash
# Host side of the library (expression system, decomposition, planner passes, code generators up to the hiprtc call,
# integrator class) under AddressSanitizer + UndefinedBehaviorSanitizer: builds an instrumented copy of libheyoka_amd.so
# under /tmp and runs the host parts of the C++ test programs against it. No GPU needed. usage: bash tests/run_sanitized_host_tests.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/heyoka_amd_asan
mkdir -p "$O"
FLAGS="-O1 -g -std=c++20 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
cd "$R/heyoka_amd/csrc"
for f in *.cpp; do echo "g++ $FLAGS -c $f -o $O/${f%.cpp}.o"; done | xargs -P "$(nproc)" -I{} sh -c "{}"
g++ "$O"/*.o -shared -fsanitize=address,undefined -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lhiprtc -pthread -o "$O/libheyoka_amd.so"
cd "$R"
for t in test_reference_cases test_reference_event_cases test_reference_includes test_round6_generators; do
    g++ -std=c++20 -O1 -g -fsanitize=address,undefined -Iinclude tests/cpp/$t.cpp -o "$O/$t" -L"$O" -lheyoka_amd -Wl,-rpath,"$O" -Wl,-rpath,/opt/rocm/lib
    ASAN_OPTIONS=detect_leaks=0 "$O/$t"
done
g++ -std=c++20 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/test_dropin_api.cpp -o "$O/test_dropin_api" -L"$O" -lheyoka_amd -Wl,-rpath,"$O" -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=0 "$O/test_dropin_api"
echo "sanitized host tests OK"
