#!/usr/bin/env python
"""Static instruction accounting of the step loop of the outer-SS stepper (no GPU needed: hiprtc cross-compiles): the
integrator is built under the given environment variants, its code object disassembled (llvm-objdump) and the
instructions between the head and the back edge of the step loop - the second-largest backward branch span of hy_taylor,
the largest one being the work loop - are counted by class. usage: isa_count.py ['K=V,K2=V2' ...]"""
import os, re, subprocess, sys, tempfile, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs, codegen_check

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def step_loop_counts(code_object):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(bytes(code_object)); f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout
    lines, on = [], False
    for ln in txt.split("\n"):
        if re.match(r"^[0-9a-f]+ <hy_taylor>:", ln):
            on = True
            continue
        if on and re.match(r"^[0-9a-f]+ <", ln):
            break
        if on and "//" in ln:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            lines.append((int(m.group(1), 16), ln.split("//")[0].strip()))
    addr0 = lines[0][0]
    back = []
    for i, (ad, ins) in enumerate(lines):
        m = re.search(r"<hy_taylor\+0x([0-9a-f]+)>", txt.split("\n")[0]) if False else None
    # (branch targets are printed as <hy_taylor+0xNNN> in the comment part: re-scan the raw text)
    tgt = {}
    for ln in txt.split("\n"):
        m = re.search(r"(s_c?branch\w*)\s.*//\s*([0-9A-Fa-f]+):.*<hy_taylor\+0x([0-9a-f]+)>", ln)
        if m:
            tgt[int(m.group(2), 16)] = addr0 + int(m.group(3), 16)
    spans = sorted(((ad - t, t, ad) for ad, t in tgt.items() if t < ad), reverse=True)
    (_, o_lo, o_hi) = spans[0]
    inner = [s for s in spans[1:] if s[1] > o_lo and s[2] < o_hi]
    (_, lo, hi) = inner[0]
    cnt = collections.Counter()
    for ad, ins in lines:
        if lo <= ad <= hi:
            op = ins.split()[0]
            cls = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop")
                   else "wait" if op.startswith("s_waitcnt") else "vmem" if op.startswith(("global_", "scratch_", "buffer_", "flat_")) else "other")
            cnt[cls] += 1
            if cls == "lds":
                cnt["lds_write" if "write" in op else "lds_read"] += 1
            if cls == "valu":
                cnt["valu_fp64" if "_f64" in op else "valu_other"] += 1
    return dict(cnt)


if __name__ == "__main__":
    sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
    for v in (sys.argv[1:] or [""]):
        kv = dict(x.split("=", 1) for x in v.split(",") if x)
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        ta = hy.taylor_adaptive_batch(sys_, None, 64, high_accuracy=True)
        for k, o in old.items():
            if o is None:
                del os.environ[k]
            else:
                os.environ[k] = o
        res = codegen_check.kernel_resources(ta.code_object)
        print(json.dumps({"variant": v, "step_loop": step_loop_counts(ta.code_object),
                          "res": res}))
