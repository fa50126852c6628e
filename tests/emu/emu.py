"""TEST INFRASTRUCTURE (see wave_emu.hpp): compiles the generated HIP source of an integrator for the host and runs its
stepper kernel under the wavefront emulator. Nothing here is imported by the product or by bench.py."""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


class KArgs(ctypes.Structure):
    # (struct hy_kargs of the generated sources: heyoka_amd/csrc/hip_emit.cpp, prelude.)
    _fields_ = [
        ("state", ctypes.c_void_p), ("pars", ctypes.c_void_p), ("time_hi", ctypes.c_void_p), ("time_lo", ctypes.c_void_p),
        ("lim", ctypes.c_void_p), ("tfin_hi", ctypes.c_void_p), ("tfin_lo", ctypes.c_void_p), ("last_h", ctypes.c_void_p),
        ("outcome", ctypes.c_void_p), ("min_h", ctypes.c_void_p), ("max_h", ctypes.c_void_p), ("n_steps", ctypes.c_void_p),
        ("tc", ctypes.c_void_p), ("N", ctypes.c_uint64), ("max_steps", ctypes.c_uint64), ("mode", ctypes.c_int),
        ("pad", ctypes.c_int), ("counters", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("tfin_s_hi", ctypes.c_double),
        ("tfin_s_lo", ctypes.c_double), ("ev_tc", ctypes.c_void_p), ("max_abs_state", ctypes.c_void_p),
        ("sel_norms", ctypes.c_void_p), ("tc_thr", ctypes.c_void_p), ("grid_done", ctypes.c_void_p),
    ]


_TAIL = r"""
static void emu_tramp(const void *p) { hy_taylor(*static_cast<const hy_kargs *>(p)); }
extern "C" void emu_run(const hy_kargs *a, unsigned grid, unsigned block) { emu::launch(emu_tramp, a, grid, block); }
extern "C" unsigned emu_sizeof_kargs() { return (unsigned)sizeof(hy_kargs); }
"""


_UNIFORM_TOKENS = re.compile(
    r"^(?:\s|[()!&|=<>+*]|\d+u?l*|nullptr|HY_\w+|hy_static|hy_queue_empty|hy_tc_only|hy_wg_tc|hy_it|a\.\w+|SPW|N|"
    r"__builtin_amdgcn_ballot_w64\([^()]*(?:\([^()]*\))?[^()]*\)\s*[!=]=\s*0ull|gridDim\.x|blockIdx\.x|\(u64\))*$")


def _lockstep(src):
    """The lanes of a wavefront execute an instruction together: the single-buffered LDS exchanges of the generated
    kernels rely on every lane having READ a slot before any lane overwrites it. The fibres of the emulator run one after
    the other between rendezvous points, so a rendezvous goes in front of every LDS store - except inside blocks whose
    condition is not wave-uniform (the pickup of a new system by the lanes of a finished one: no exchange in there)."""
    lds = set(re.findall(r"__shared__\s+(?:__attribute__\(\([^)]*\)\)\)\s+)?[\w ]+?\s+(\w+)\[", src))
    decl = re.compile(r"^\s*(?:const\s+)?double\s*\*\s*(?:const\s+)?(\w+)\s*=\s*([^;]*);", re.M)
    grew = True
    while grew:
        grew = False
        for m in decl.finditer(src):
            if m.group(1) not in lds and any(re.search(r"\b" + re.escape(n) + r"\b", m.group(2)) for n in lds):
                lds.add(m.group(1))
                grew = True
    store = re.compile(r"^\s*(?:\*\s*)?(" + "|".join(sorted(lds)) + r")\s*(?:\[[^=]*\])?\s*=[^=]")
    out, stack, in_kernel = [], [], False
    for line in src.split("\n"):
        if "hy_taylor(const hy_kargs a)" in line:
            in_kernel = True
        s = line.strip()
        divergent = any(stack)
        if in_kernel and not divergent and store.match(line) and not s.startswith("const") and not s.startswith("double"):
            out.append("emu::wave_sync();")
        out.append(line)
        if not in_kernel:
            continue
        m = re.match(r"^if \((.*)\) \{$", s)
        if m:
            stack.append(_UNIFORM_TOKENS.match(m.group(1)) is None)
            continue
        if s == "} else {":
            continue  # (same flag as the block it continues)
        for ch in s:
            if ch == "{":
                stack.append(False)
            elif ch == "}" and stack:
                stack.pop()
    return "\n".join(out)


def host_source(hip_source):
    src = hip_source
    # Register-class constraints of the amdgcn inline asm ("v": a VGPR) -> an SSE register of the host.
    src = src.replace('"+v"(', '"+x"(')
    return '#define HY_NO_NMAX 1\n#define HY_HOST_EMU 1\n#include "wave_emu.hpp"\n' + _lockstep(src) + _TAIL


class EmulatedKernel:
    def __init__(self, hip_source, contract=False):
        os.makedirs(BUILD, exist_ok=True)
        text = host_source(hip_source)
        tag = hashlib.sha256((text + str(contract) + open(os.path.join(HERE, "wave_emu.hpp")).read()).encode()).hexdigest()[:16]
        cpp, so = os.path.join(BUILD, f"k_{tag}.cpp"), os.path.join(BUILD, f"k_{tag}.so")
        if not os.path.exists(so):
            with open(cpp, "w") as f:
                f.write(text)
            cmd = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-w", "-mfma", "-I", HERE,
                   "-ffp-contract=" + ("fast" if contract else "off"), "-o", so + ".tmp", cpp]
            subprocess.check_call(cmd)
            os.replace(so + ".tmp", so)
        self.lib = ctypes.CDLL(so)
        assert self.lib.emu_sizeof_kargs() == ctypes.sizeof(KArgs)
        self.lib.emu_run.argtypes = [ctypes.POINTER(KArgs), ctypes.c_uint, ctypes.c_uint]
        m = re.search(r"__launch_bounds__\((\d+)\) hy_taylor", hip_source)
        self.block = int(m.group(1))
        m = re.search(r"#define SPW (\d+)u", hip_source)
        self.lanes_per_system = 64 // int(m.group(1)) if m else 1
        self.order = None

    def _grid(self, n, max_grid):
        threads = n * self.lanes_per_system
        return max(1, min((threads + self.block - 1) // self.block, max_grid))

    def run(self, state, time_hi, time_lo, *, mode, lim=None, tfin=None, max_steps=0, pars=None, want_tc_rows=0, max_grid=2, pad=0,
            scratch_per_wave=0, tc_thr=None):
        """One launch of hy_taylor. state: (n_eq, n) array, modified in place like the device buffer. Returns a dict of the
        per-system outputs."""
        n = state.shape[1]
        f8 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
        st = f8(state).copy()
        thi, tlo = f8(time_hi).copy(), f8(time_lo).copy()
        out = {k: np.zeros(n) for k in ("last_h", "min_h", "max_h")}
        outcome = np.zeros(n, dtype=np.int64)
        n_steps = np.zeros(n, dtype=np.uint64)
        counters = np.zeros(16, dtype=np.uint32)
        a = KArgs()
        ptr = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
        a.state, a.time_hi, a.time_lo = ptr(st), ptr(thi), ptr(tlo)
        keep = [st, thi, tlo, outcome, n_steps, counters]
        if pars is not None:
            pa = f8(pars)
            keep.append(pa)
            a.pars = ptr(pa)
        if lim is not None:
            la = f8(lim).copy()
            keep.append(la)
            a.lim = ptr(la)
        if tfin is not None:
            tf = f8(np.broadcast_to(tfin, (n,))).copy()
            tfl = np.zeros(n)
            keep += [tf, tfl]
            a.tfin_hi, a.tfin_lo = ptr(tf), ptr(tfl)
        a.last_h, a.min_h, a.max_h = ptr(out["last_h"]), ptr(out["min_h"]), ptr(out["max_h"])
        a.outcome, a.n_steps, a.counters = ptr(outcome), ptr(n_steps), ptr(counters)
        tc = None
        if want_tc_rows:
            tc = np.zeros((want_tc_rows, n))
            a.tc = ptr(tc)
        if tc_thr is not None:
            th = f8(np.broadcast_to(tc_thr, (n,))).copy()
            out["grid_done"] = np.full(n, -1.0)
            keep.append(th)
            a.tc_thr, a.grid_done = ptr(th), ptr(out["grid_done"])
        if scratch_per_wave:
            # (Jet scratch of the steppers which keep the jets of the state variables in global memory: per resident wave.)
            sc = np.zeros(self._grid(n, max_grid) * (self.block // 64) * int(scratch_per_wave))
            keep.append(sc)
            a.scratch = ptr(sc)
        a.N, a.max_steps, a.mode, a.pad = n, max_steps, mode, pad
        self.lib.emu_run(ctypes.byref(a), self._grid(n, max_grid), self.block)
        out.update(state=st, time_hi=thi, time_lo=tlo, outcome=outcome, n_steps=n_steps, counters=counters, tc=tc)
        if lim is not None:
            out["lim"] = la
        return out
