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
