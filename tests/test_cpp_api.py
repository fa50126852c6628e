"""The C++ drop-in interface (taylor_adaptive_batch<double>, kw:: named arguments, ensemble_propagate_*):
tests/cpp/test_dropin_api.cpp is written against the reference's API and must compile unchanged."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "heyoka_amd", "csrc", "_build", "test_dropin_api")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_dropin_api.cpp")
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(
        ["g++", "-std=c++20", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", EXE,
         "-L" + os.path.join(ROOT, "heyoka_amd"), "-lheyoka_amd", "-Wl,-rpath," + os.path.join(ROOT, "heyoka_amd"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_api_compiles_and_validates_on_cpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "CPU-only checks OK" in out.stdout


@pytest.mark.gpu
def test_cpp_api_tutorial_sessions_on_gpu():
    _build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "GPU checks OK" in out.stdout


EXE2 = os.path.join(ROOT, "heyoka_amd", "csrc", "_build", "test_reference_includes")


def _build_ref_includes():
    """tests/cpp/test_reference_includes.cpp uses the reference's include layout (<heyoka/heyoka.hpp>, <heyoka/taylor.hpp>,
    <heyoka/kw.hpp>, <heyoka/model/nbody.hpp>) and namespace (heyoka::) unchanged: only -I include -lheyoka_amd."""
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_includes.cpp")
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    if os.path.exists(EXE2) and os.path.getmtime(EXE2) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    os.makedirs(os.path.dirname(EXE2), exist_ok=True)
    subprocess.check_call(
        ["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE2,
         "-L" + os.path.join(ROOT, "heyoka_amd"), "-lheyoka_amd", "-Wl,-rpath," + os.path.join(ROOT, "heyoka_amd"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_reference_include_layout_and_namespace_compile_unchanged():
    _build_ref_includes()
    out = subprocess.run([EXE2], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "reference include layout OK" in out.stdout and "order 20 22 36" in out.stdout


@pytest.mark.gpu
def test_reference_include_layout_runs_the_tutorial_calls_on_gpu():
    _build_ref_includes()
    out = subprocess.run([EXE2, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    # The first step of the four pendulums succeeds (outcomes printed in the reference's format).
    assert "GPU OK" in out.stdout and "Batch index 0: (taylor_outcome::success, 0.1" in out.stdout
    assert "taylor_outcome::time_limit" in out.stdout


EXE3 = os.path.join(ROOT, "heyoka_amd", "csrc", "_build", "test_reference_cases")


def _build_ref_cases():
    """tests/cpp/test_reference_cases.cpp: the behaviours pinned by the reference's own unit tests of the batch integrator
    (test/taylor_adaptive_batch.cpp), restated against the reference's include layout."""
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_cases.cpp")
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    if os.path.exists(EXE3) and os.path.getmtime(EXE3) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    os.makedirs(os.path.dirname(EXE3), exist_ok=True)
    subprocess.check_call(
        ["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE3,
         "-L" + os.path.join(ROOT, "heyoka_amd"), "-lheyoka_amd", "-Wl,-rpath," + os.path.join(ROOT, "heyoka_amd"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_reference_unit_test_cases_host_side():
    _build_ref_cases()
    out = subprocess.run([EXE3], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "host cases OK" in out.stdout


@pytest.mark.gpu
def test_reference_unit_test_cases_on_gpu():
    _build_ref_cases()
    out = subprocess.run([EXE3, "gpu"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "GPU cases OK" in out.stdout


EXE4 = os.path.join(ROOT, "heyoka_amd", "csrc", "_build", "test_reference_event_cases")


def _build_ref_event_cases():
    """tests/cpp/test_reference_event_cases.cpp: behaviours and known answers of test/batch_event_detection.cpp."""
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_event_cases.cpp")
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    if os.path.exists(EXE4) and os.path.getmtime(EXE4) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    os.makedirs(os.path.dirname(EXE4), exist_ok=True)
    subprocess.check_call(
        ["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE4,
         "-L" + os.path.join(ROOT, "heyoka_amd"), "-lheyoka_amd", "-Wl,-rpath," + os.path.join(ROOT, "heyoka_amd"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_reference_event_detection_cases_host_side():
    _build_ref_event_cases()
    out = subprocess.run([EXE4], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "host cases OK" in out.stdout


@pytest.mark.gpu
def test_reference_event_detection_cases_on_gpu():
    _build_ref_event_cases()
    out = subprocess.run([EXE4, "gpu"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "GPU cases OK" in out.stdout


EXE5 = os.path.join(ROOT, "heyoka_amd", "csrc", "_build", "test_round6_generators")


def test_round6_generators_through_the_cpp_api():
    """tests/cpp/test_round6_generators.cpp: a system with two classes of clusters on the multi-class wave-cluster stepper,
    the same on the staged / HBM-tape table steppers (developer switches), the outer Solar System forced onto the staged
    stepper with and without kw::compact_mode - constructed through the C++ drop-in class (also the program which
    tests/run_sanitized_host_tests.sh runs under ASan / UBSan)."""
    src = os.path.join(ROOT, "tests", "cpp", "test_round6_generators.cpp")
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    if not (os.path.exists(EXE5) and os.path.getmtime(EXE5) > max(os.path.getmtime(src), os.path.getmtime(lib))):
        os.makedirs(os.path.dirname(EXE5), exist_ok=True)
        subprocess.check_call(
            ["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE5,
             "-L" + os.path.join(ROOT, "heyoka_amd"), "-lheyoka_amd", "-Wl,-rpath," + os.path.join(ROOT, "heyoka_amd"),
             "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([EXE5], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert "2 classes of clusters" in lines[0] and "table mode (staged)" in lines[1] and "tape in HBM" in out.stdout
    assert lines[-1] == "round-6 generators under the sanitizers OK"
