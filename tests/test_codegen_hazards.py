"""Static check of the generated code objects (CPU: hiprtc cross-compiles for gfx950 without a GPU) for the exec-mask /
live-range-split hazard described in heyoka_amd/codegen_check.py and DESIGN.md ("Toolchain notes"): every stepper variant
on representative systems must be free of the pattern, and the detector must fire on the configuration that exposed it
(the math-library calls inlined into a 3500-statement kernel)."""
import os

import numpy as np
import pytest

import heyoka_amd as hy
from heyoka_amd import codegen_check, configs

pytestmark = pytest.mark.skipif(codegen_check.find_objdump() is None, reason="llvm-objdump not available")


def tan_system():
    x, y = hy.make_vars("x", "y")
    return [(x, 0.3 * hy.tan(y) - 0.4 * x), (y, -0.3 * hy.tan(x) * hy.cos(y) - 0.4 * y)]


def _cases():
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    x, y = hy.make_vars("x", "y")
    pw = [(x, hy.atan2(y, 1.5 + x * x) - hy.relu(x, 0.1) + hy.sin(hy.kepE(0.3, y)) + hy.erf(x * y)),
          (y, hy.select(hy.gt(x, y), hy.tanh(x), -y) - 0.5 * y + hy.asin(0.3 * hy.sin(x)))]
    ev = [hy.nt_event(y, lambda *a: None)]
    h_, k_, lam_ = hy.make_vars("h", "k", "lam")
    F_ = hy.kepF(h_, k_, lam_)
    kep = [(h_, -0.01 * hy.sin(F_)), (k_, 0.02 * hy.cos(F_) + 0.01 * hy.kepDE(0.1 * h_, 0.2 + k_, 0.3 * lam_)), (lam_, 1.0 + 0.1 * h_ * k_)]
    x1, x2 = hy.make_vars("x_1", "x_2")
    ev_ss = [hy.nt_event((x1 - x2) * (x1 - x2) - 4.0, lambda *a: None)]
    y1_, y2_, z1_, z2_ = hy.make_vars("y_1", "y_2", "z_1", "z_2")
    def _pos(b_):
        return hy.make_vars("x_%d" % b_, "y_%d" % b_, "z_%d" % b_)
    ev_pairs = []
    for a_ in range(6):
        for b_ in range(a_ + 1, 6):
            pa_, pb_ = _pos(a_), _pos(b_)
            ev_pairs.append(hy.nt_event((pa_[0] - pb_[0]) * (pa_[0] - pb_[0]) + (pa_[1] - pb_[1]) * (pa_[1] - pb_[1])
                                        + (pa_[2] - pb_[2]) * (pa_[2] - pb_[2]) - 1.0, lambda *a: None))
    ev_ss3 = [hy.nt_event(y2_, lambda *a: None), hy.nt_event(x1 - x2, lambda *a: None),
              hy.nt_event((x1 - x2) * (x1 - x2) + (y1_ - y2_) * (y1_ - y2_) + (z1_ - z2_) * (z1_ - z2_) - 81.0, lambda *a: None)]
    return {
        "outer_ss_cluster_event_stepper": (lambda: hy.model.nbody(6, masses=M, Gconst=G),
                                           {"high_accuracy": True, "nt_events": ev_ss}, {}, "events:"),
        "outer_ss_cluster_event_stepper_v3": (lambda: hy.model.nbody(6, masses=M, Gconst=G),
                                              {"high_accuracy": True, "nt_events": ev_ss}, {"HEYOKA_AMD_V5_EVENTS": "0"},
                                              "events:"),
        # Event equations evaluated inside the stepper (three events, three nonlinear nodes: the budget), exclusion test,
        # conditional store of the Taylor coefficients.
        "outer_ss_event_equations_inside_the_stepper": (lambda: hy.model.nbody(6, masses=M, Gconst=G),
                                                        {"high_accuracy": True, "nt_events": ev_ss3,
                                                         "t_events": [hy.t_event(x1 - 3.0)]}, {}, "inside the stepper"),
        # Close-encounter events taken from the lanes of their pairs (all 15 pair distances) next to a generic one.
        "outer_ss_pair_distance_events_on_the_lanes": (lambda: hy.model.nbody(6, masses=M, Gconst=G),
                                                       {"high_accuracy": True, "nt_events": ev_pairs + ev_ss3[:1]}, {}, "inside the stepper"),
        "outer_ss_cluster_v5": (lambda: hy.model.nbody(6, masses=M, Gconst=G), {"high_accuracy": True}, {}, "v5"),
        "outer_ss_cluster_v3": (lambda: hy.model.nbody(6, masses=M, Gconst=G), {"high_accuracy": True},
                                {"HEYOKA_AMD_ONE_LANE": "0"}, "v3"),
        "outer_ss_cluster_v2": (lambda: hy.model.nbody(6, masses=M, Gconst=G), {"high_accuracy": True},
                                {"HEYOKA_AMD_ONE_LANE": "0", "HEYOKA_AMD_PAIR_SPLIT": "0"}, "v2"),
        "nbody8_v5_two_systems_per_wavefront": (lambda: hy.model.nbody(8, masses=[1.0, 1e-3, 2e-3, 3e-3, 4e-3, 5e-3, 6e-3, 7e-3]),
                                                {}, {}, "v5"),
        "outer_ss_cluster_v1": (lambda: hy.model.nbody(6, masses=M, Gconst=G), {}, {"HEYOKA_AMD_CLUSTER_V1": "1"}, "cluster"),
        "np1body_aliased": (lambda: hy.model.np1body(6, masses=M, Gconst=G), {}, {}, "cluster"),
        "two_body_register_jets": (lambda: hy.model.nbody(2, masses=[1.0, 0.0]), {}, {}, "unrolled"),
        "nbody12_block": (lambda: hy.model.nbody(12), {}, {}, "block"),
        "nbody12_block_generic_cluster_phase": (lambda: hy.model.nbody(12), {}, {"HEYOKA_AMD_BLOCK_V2": "0"}, "block"),
        "nbody64_block_v2": (lambda: hy.model.nbody(64), {}, {}, "v2 cluster phase"),
        "tan_3500_statements": (tan_system, {}, {}, "unrolled"),
        "cr3bp_unrolled": (lambda: hy.model.cr3bp(), {}, {}, "unrolled"),
        "functions_unrolled": (lambda: pw, {}, {}, "unrolled"),
        "functions_table_wave_level": (lambda: pw, {}, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "1"}, "staged"),
        "functions_table_hbm": (lambda: pw, {}, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "0"}, "tape in HBM"),
        # Functions defined through the registry of node rules: their order-0 functions (Newton iterations, library calls)
        # sit behind out-of-line frames (node_rule.cpp).
        "node_rules_unrolled": (lambda: kep, {}, {}, "unrolled"),
        "node_rules_table_hbm": (lambda: kep, {}, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "0"}, "tape in HBM"),
        "node_rules_table_wave_level": (lambda: kep, {}, {"HEYOKA_AMD_EMIT_MODE": "table", "HEYOKA_AMD_TABLE_LDS": "1"}, "staged"),
        "events_unrolled": (lambda: pw, {"nt_events": ev}, {}, "unrolled"),
        "events_table": (lambda: pw, {"nt_events": ev}, {"HEYOKA_AMD_EMIT_MODE": "table"}, "table"),
    }


@pytest.mark.parametrize("name", sorted(_cases()))
def test_generated_kernels_are_free_of_the_exec_mask_split_hazard(name, monkeypatch):
    sys_f, kw, env, expect = _cases()[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ta = hy.taylor_adaptive_batch(sys_f(), None, 64, **kw)
    assert expect in ta.hip_source_mode, ta.hip_source_mode
    co = ta.code_object
    assert co[:4] == b"\x7fELF"
    assert codegen_check.scan_code_object(co) == []


def test_detector_fires_on_the_configuration_that_exposed_the_hazard():
    """Positive control: the same 3500-statement tan() kernel with the math-library calls inlined (as the generators
    emitted them before the out-of-line wrappers) still draws the pattern from this toolchain."""
    ta = hy.taylor_adaptive_batch(tan_system(), None, 64)
    src = ta.hip_source
    assert "hy_tan(" in src
    inlined = src.replace("= hy_tan(", "= tan(").replace("= hy_sin(", "= sin(").replace("= hy_cos(", "= cos(")
    hz = codegen_check.scan_code_object(hy.hiprtc_compile(inlined))
    if not hz:
        pytest.skip("this toolchain no longer produces the pattern for the inlined variant")
    assert any(h[0] == "hy_taylor" and "v_accvgpr_write" in h[3] or "scratch_store" in h[3] for h in hz)


def test_detector_on_synthetic_disassembly():
    """The pattern matcher itself, on hand-written listings: copies ahead of the exec restore at the target of an
    s_cbranch_execz are reported; the same copies after the restore, or at the target of an s_cbranch_execnz (an outlined
    first side, entered with its own mask), are not."""
    def listing(body):
        lines = ["0000000000001000 <hy_taylor>:"]
        addr = 0x1000
        for ins in body:
            tgt = ""
            if isinstance(ins, tuple):
                ins, off = ins
                tgt = " <hy_taylor+0x%x>" % off
            lines.append("\t%-58s // %012X: 00000000%s" % (ins, addr, tgt))
            addr += 4
        return "\n".join(lines) + "\n"

    bad = listing([
        "s_and_saveexec_b64 s[2:3], vcc",
        ("s_cbranch_execz 3", 0x10),         # skips the first side
        "v_add_f64 v[0:1], v[0:1], v[2:3]",  # first side
        "v_mul_f64 v[0:1], v[0:1], v[2:3]",
        "s_waitcnt vmcnt(0)",                # +0x10: flow block
        "v_accvgpr_write_b32 a3, v9",        # live-range split under the stale mask
        "s_mov_b32 s4, s5",
        "s_andn2_saveexec_b64 s[2:3], s[4:5]",
        "s_endpgm",
    ])
    hz = codegen_check.scan_disassembly(bad)
    assert hz == [("hy_taylor", "0x1010", 1, "v_accvgpr_write_b32 a3, v9")]

    good = bad.replace("v_accvgpr_write_b32 a3, v9", "s_nop 0")
    assert codegen_check.scan_disassembly(good) == []
    after = listing([
        "s_and_saveexec_b64 s[2:3], vcc",
        ("s_cbranch_execz 2", 0xc),
        "v_add_f64 v[0:1], v[0:1], v[2:3]",
        "s_or_b64 exec, exec, s[2:3]",       # +0xc: exec restored first
        "v_accvgpr_write_b32 a3, v9",
        "s_endpgm",
    ])
    assert codegen_check.scan_disassembly(after) == []
    outlined = bad.replace("s_cbranch_execz 3", "s_cbranch_execnz 3")
    assert codegen_check.scan_disassembly(outlined) == []


def test_round6_generator_choices_and_register_budgets():
    """The choices of the code generators which round 6 changed, checked on the compiled code objects (no GPU needed):
    - the test-particle two-body problem (config 3): histories of the velocities re-derived, two wavefronts per SIMD, the series
      of a pair x' = v evaluated in one pass;
    - cr3bp and two massive bodies: jets of the state variables in registers (scaled copies of coefficient histories folded into
      their consumers), no mass spilling - before: 526 spilled VGPRs / 651 spilled SGPRs;
    - model::nbody(3) / (4): the one-lane-per-pair kernel with more lanes per system than pairs, jets in LDS, no spills - before:
      the lane-pair kernel with the jets in global scratch, 176 / 116 spilled registers;
    - the staged table stepper never asks for more wavefronts per SIMD than its table registers leave room for."""
    import re

    ta = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, 64)
    res = codegen_check.kernel_resources(ta.code_object)
    assert "two wavefronts per SIMD" in ta.hip_source_mode and res["waves_per_simd_by_registers"] == 2
    assert res["vgpr_spill"] <= 64 and "der = res + der * h;" in ta.hip_source
    # (Strict arithmetic: none of the re-derivations.)
    ts = hy.taylor_adaptive_batch(hy.model.nbody(2, masses=[1.0, 0.0]), None, 64, exact_division=True)
    assert "two wavefronts per SIMD" not in ts.hip_source_mode and "der = res + der * h;" not in ts.hip_source

    for sys_ in (hy.model.cr3bp(mu=0.01), hy.model.nbody(2, masses=[1.0, 0.5])):
        ta = hy.taylor_adaptive_batch(sys_, None, 64)
        res = codegen_check.kernel_resources(ta.code_object)
        assert "register-resident state jets" in ta.hip_source_mode, ta.hip_source_mode
        assert res["vgpr_spill"] <= 32 and res["sgpr_spill"] <= 128, res

    for nb, lanes in ((3, 8), (4, 16)):
        ta = hy.taylor_adaptive_batch(hy.model.nbody(nb, masses=[1.0, 1e-3, 3e-4, 2e-4][:nb]), None, 64, high_accuracy=True)
        res = codegen_check.kernel_resources(ta.code_object)
        assert "cluster mode v5" in ta.hip_source_mode and "jets in LDS" in ta.hip_source_mode, ta.hip_source_mode
        assert ("lanes per system: %d" % lanes) in ta.hip_source_mode and res["vgpr_spill"] == 0, (ta.hip_source_mode, res)

    os.environ["HEYOKA_AMD_EMIT_MODE"] = "table"
    try:
        x, y, z = hy.make_vars("x", "y", "z")
        ta = hy.taylor_adaptive_batch([(x, hy.erf(y) - 0.2 * x + hy.atan2(0.7, y)), (y, x * z), (z, 0.3 - hy.tanh(y))], None, 64)
    finally:
        os.environ.pop("HEYOKA_AMD_EMIT_MODE", None)
    assert "staged" in ta.hip_source_mode
    w = int(re.search(r"amdgpu_waves_per_eu\((\d+)\)", ta.hip_source).group(1))
    t_regs = int(re.search(r"(\d+) table registers per lane", ta.hip_source_mode).group(1))
    assert w == 1 or w * (t_regs + 56 + 64) <= 512 + 7 * w
