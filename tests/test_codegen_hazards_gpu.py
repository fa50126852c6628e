"""The exec-mask / live-range-split scan of tests/test_codegen_hazards.py on the toolchain which actually compiles the
kernels the MI355X executes: the CPU test runs under the authoring container's hiprtc (ROCm 7.2), the driver's GPU boxes
carry another one (ROCm 7.0.x at the time of writing) and every kernel is re-compiled there at construction. Here every
stepper variant is compiled by THIS box's hiprtc and its code object scanned (heyoka_amd/codegen_check.py); the toolchain
inventory goes to the test's output. Without llvm-objdump on the box the gap is reported as a skip with that inventory."""
import glob
import os
import subprocess

import pytest

import heyoka_amd as hy
from heyoka_amd import codegen_check

from test_codegen_hazards import _cases


def _inventory():
    rocm = sorted(glob.glob("/opt/rocm*/.info/version*"))
    ver = {p: open(p).read().strip() for p in rocm[:3]}
    od = codegen_check.find_objdump()
    odv = subprocess.run([od, "--version"], capture_output=True, text=True).stdout.split("\n")[:2] if od else None
    return {"heyoka_amd": hy.version(), "rocm_version_files": ver, "llvm_objdump": od, "llvm_objdump_version": odv}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_cases()))
def test_code_objects_of_this_box_are_free_of_the_exec_mask_split_hazard(name, monkeypatch, record_property):
    inv = _inventory()
    record_property("toolchain", str(inv))
    print("toolchain:", inv)
    if inv["llvm_objdump"] is None:
        pytest.skip("no llvm-objdump on this box: the code objects of its hiprtc cannot be scanned here (%s)" % inv)
    sys_f, kw, env, expect = _cases()[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ta = hy.taylor_adaptive_batch(sys_f(), None, 64, **kw)
    assert expect in ta.hip_source_mode, ta.hip_source_mode
    co = ta.code_object
    assert co[:4] == b"\x7fELF"
    assert codegen_check.scan_code_object(co) == [], inv
