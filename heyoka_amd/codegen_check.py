"""Static check of generated gfx950 code objects for a code-generation hazard of the toolchain.

With several hundred live registers the register allocator splits long live ranges; with ROCm 7.2 such a split (a
VGPR -> AGPR copy or a scratch spill) can be placed at the top of the exec-masked *flow block* between the two sides of a
divergent if/else, ahead of the instruction that restores the exec mask. The flow block is the target of the
`s_cbranch_execz` that skips the first side: when no lane takes that side the copy runs with exec = 0 and the value is
lost (seen as wild addresses in a 3500-statement kernel with tan(), DESIGN.md "Toolchain notes"). The generators avoid
divergent if/else regions in the steppers; this module looks for the pattern itself in the disassembly:

    target of s_cbranch_execz:
        [v_accvgpr_write | scratch_store ...]      <- vector copies / spills executed under the stale exec mask
        s_or_saveexec / s_andn2_saveexec / s_or_b64 exec, ... <- exec restored only here

Needs llvm-objdump (ROCm: /opt/rocm/lib/llvm/bin)."""
import os
import re
import shutil
import subprocess
import tempfile

_RESTORE = re.compile(r"^(s_or_saveexec_b64|s_andn2_saveexec_b64|s_or_b64 exec, exec|s_xor_b64 exec, exec|s_mov_b64 exec)")
_COPY = re.compile(r"^(v_accvgpr_write|scratch_store)")
_STOP = ("s_cbranch", "s_branch", "s_endpgm", "s_and_saveexec", "s_setpc", "s_swappc")


def find_objdump():
    for cand in (os.environ.get("HEYOKA_AMD_OBJDUMP"), "/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")):
        if cand and os.path.exists(cand):
            return cand
    return None


def disassemble(code_object):
    objdump = find_objdump()
    if objdump is None:
        raise RuntimeError("llvm-objdump not found")
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object)
        f.flush()
        return subprocess.run([objdump, "-d", f.name], check=True, capture_output=True, text=True).stdout


def scan_disassembly(text):
    """List of (function, address of the flow block, number of copies / spills ahead of the exec restore, first one)."""
    ins = []
    func = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):(.*)$", line)
        if m and func:
            ins.append((int(m.group(2), 16), m.group(1), func, m.group(3)))
    base = {}
    for a, _, f, _ in ins:
        base.setdefault(f, a)
    targets = set()
    for _, t, _, rest in ins:
        # NOTE: execz only - the target of an execnz branch is an (outlined) first side, entered with its own mask.
        if t.startswith("s_cbranch_execz"):
            m = re.search(r"<(.+?)\+0x([0-9a-f]+)>", rest)
            if m and m.group(1) in base:
                targets.add(base[m.group(1)] + int(m.group(2), 16))
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    hazards = []
    for tg in sorted(targets):
        i = index.get(tg)
        if i is None:
            continue
        seen = []
        for _, t, f, _ in ins[i:i + 400]:
            if _RESTORE.match(t):
                if seen:
                    hazards.append((f, hex(tg), len(seen), seen[0]))
                break
            if t.startswith(_STOP):
                break
            if _COPY.match(t):
                seen.append(t)
    return hazards


def scan_code_object(code_object):
    return scan_disassembly(disassemble(code_object))


def kernel_resources(code_object, kernel="hy_taylor"):
    """Register / LDS / scratch usage of a kernel from the metadata notes of its code object (llvm-readelf --notes):
    {"vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes_per_lane", "lds_bytes", "waves_per_simd"}.
    The register file of a gfx950 SIMD holds 512 registers per lane (VGPR + AGPR, allocation granule 8)."""
    objdump = find_objdump()
    if objdump is None:
        return None
    readelf = os.path.join(os.path.dirname(objdump), "llvm-readelf")
    if not os.path.exists(readelf):
        return None
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object)
        f.flush()
        text = subprocess.run([readelf, "--notes", f.name], check=True, capture_output=True, text=True).stdout
    # One YAML-ish block per kernel, fields sorted alphabetically; .name sits in the middle of its block.
    blocks = re.split(r"\n\s+- \.agpr_count:", "\n" + text)
    for b in blocks[1:]:
        b = ".agpr_count:" + b
        m = re.search(r"\.name:\s+(\S+)", b)
        if not m or m.group(1) != kernel:
            continue

        def field(name):
            mm = re.search(r"\." + name + r":\s+(\d+)", b)
            return int(mm.group(1)) if mm else None

        vg, ag = field("vgpr_count"), field("agpr_count")
        # NOTE: on gfx90a+ .vgpr_count is the unified total (ArchVGPR + AGPR).
        alloc = ((vg + 7) // 8) * 8 if vg else None
        return {
            "vgpr_total": vg,
            "agpr": ag,
            "arch_vgpr": (vg - ag) if (vg is not None and ag is not None) else None,
            "sgpr": field("sgpr_count"),
            "vgpr_spill": field("vgpr_spill_count"),
            "sgpr_spill": field("sgpr_spill_count"),
            "scratch_bytes_per_lane": field("private_segment_fixed_size"),
            "lds_bytes": field("group_segment_fixed_size"),
            "waves_per_simd_by_registers": min(8, 512 // alloc) if alloc else None,
        }
    return None
