#!/usr/bin/env python3
"""Static opcode table of a circuit's generated eval_check kernels (no GPU: hipcc -S cross-compiles gfx950).

The generated kernels are straight-line code, one lane per domain point, so the instructions of a kernel ARE what a point costs in
that part; the sum over the parts is the cost of a point.  Prints, per part and in total: VALU instructions by class, loads, s_nop /
s_waitcnt fillers, VGPRs.  What the round-5 work on the instruction count is measured with before a GPU is asked.

    python tools/static_valu.py syn_heavy [KNOB=value ...]        # knobs: ZKH_CODEGEN_<KNOB> (REGS, PART, FACTOR, ...)
"""
import collections
import importlib
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = [
    ("mad64", ("v_mad_u64_u32", "v_mad_i64_i32")),
    ("mul32", ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24")),
    ("add_sub", ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc", "v_sub_co", "v_subb", "v_subrev_co", "v_subbrev", "v_add3", "v_lshl_add", "v_add_lshl")),
    ("min_select", ("v_min_u32", "v_cndmask", "v_max_u32", "v_cmp")),
    ("mov_logic", ("v_mov", "v_and", "v_or", "v_xor", "v_lshlrev", "v_lshrrev", "v_ashrrev", "v_bfe", "v_readfirstlane", "v_readlane", "v_writelane", "v_accvgpr", "v_perm", "v_alignbit")),
]


def classify(op: str) -> str:
    for name, pre in CLASSES:
        if op.startswith(pre):
            return name
    return "other_valu"


def count(asm: str, kernel: str):
    m = re.search(rf"^{re.escape(kernel)}:\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)
    body = m.group(1) if m else asm
    c = collections.Counter()
    for ln in body.split("\n"):
        t = ln.strip().split()
        if not t or t[0].startswith((";", ".", "/")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_"):
            c["valu"] += 1
            c[classify(op)] += 1
        elif op.startswith("global_load"):
            c["global_load"] += 1
        elif op.startswith("s_load"):
            c["s_load"] += 1
        elif op.startswith("s_nop"):
            c["s_nop"] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    v = re.search(r"\.vgpr_count:\s*(\d+)", asm) or re.search(r"; NumVgprs: (\d+)", asm)
    c["vgprs"] = int(v.group(1)) if v else 0
    sp = re.search(r"; ScratchSize: (\d+)", asm)
    c["scratch"] = int(sp.group(1)) if sp else 0
    return c


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "syn_heavy"
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1)
        os.environ["ZKH_CODEGEN_" + k] = v
    from zeth_amd.circuits import codegen, jit
    importlib.reload(codegen)
    importlib.reload(jit)
    desc = codegen.shipped()[name] if name in codegen.shipped() else None
    if desc is None:
        from zeth_amd.circuits import syn_heavy
        desc = {"syn_heavy_small": syn_heavy.syn_heavy_small, "syn_huge": syn_heavy.syn_huge}[name]()
    srcs = jit.eval_check_sources(desc)
    tmp = tempfile.mkdtemp(prefix="zkh_static_")

    def one(item):
        kname, src = item
        p = os.path.join(tmp, kname + ".hip")
        open(p, "w").write(src)
        out = os.path.join(tmp, kname + ".s")
        subprocess.run([jit.hipcc_path(), *jit.FLAGS, "--cuda-device-only", "-S", "-I", jit.CSRC, "-I", jit.INCLUDE, p, "-o", out], check=True, capture_output=True)
        return kname, count(open(out).read(), kname)
    import time
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        rows = list(ex.map(one, srcs))
    compile_s = time.perf_counter() - t0
    tot = collections.Counter()
    keys = ["valu", "mad64", "mul32", "add_sub", "min_select", "mov_logic", "other_valu", "global_load", "s_load", "s_nop", "s_waitcnt", "salu"]
    print(f"# {name}: {len(rows)} parts; knobs " + " ".join(sys.argv[2:]))
    print("# part " + " ".join(f"{k:>10s}" for k in keys) + "   vgprs scratch")
    for kname, c in rows:
        print(f"{kname[-4:]:>6s} " + " ".join(f"{c[k]:10d}" for k in keys) + f"   {c['vgprs']:5d} {c['scratch']:7d}")
        for k in keys:
            tot[k] += c[k]
        tot["vgprs"] = max(tot["vgprs"], c["vgprs"])
    print(" total " + " ".join(f"{tot[k]:10d}" for k in keys) + f"   {tot['vgprs']:5d}")
    print(json.dumps({"circuit": name, "parts": len(rows), **{k: tot[k] for k in keys}, "vgprs_max": tot["vgprs"],
                      "scratch_max": max(c["scratch"] for _, c in rows), "hipcc_S_wall_s": round(compile_s, 1), "jobs": min(os.cpu_count() or 8, len(srcs))}))


if __name__ == "__main__":
    main()
