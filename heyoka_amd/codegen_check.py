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
