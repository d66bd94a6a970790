#!/usr/bin/env python3
"""Instruction histogram of k_hash_rows' steady-state absorb block (one Poseidon2 permutation per lane), by opcode class and
modelled issue cycles, against the multiply floor — the evidence behind "Poseidon2 is at its instruction floor".

    python tools/isa_histogram.py > profiles/r03_hash_rows_isa_histogram.txt

Compiles zeth_amd/csrc/hash.hip for gfx950 with the build's flags (`-S`, device only), takes the first copy of the block
loop of k_hash_rows (interior blocks of the sponge: 16 absorbed columns, one permutation), finds its inner loops from the
backward branches and weights them by their trip counts (4 full rounds, 7 groups of three partial rounds, 4 full rounds).
Issue cycles per wave64 instruction from tools/ubench_valu.hip (profiles/r01_ubench_valu.txt): 32-bit multiplies,
v_mad_*64*, fp64 and v_cvt_f64 4.0; plain add / sub / logic / shift / mov 2.5 (measured 2.46); v_min / v_add3 / v_lshl_add 4.0;
compare-select pairs 2.1 + 2.1.
"""
from __future__ import annotations

import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOUR = ("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64", "v_add_f64", "v_fma_f64", "v_mul_f64", "v_cvt_f64", "v_min_u32", "v_max",
        "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or")
TRIPS = {"full": 4, "partial": 7}
CLASSES = [
    ("s-box + reduction multiplies (v_mad_i64_i32, v_mad_u64_u32)", ("v_mad_i64", "v_mad_u64")),
    ("low-word multiplies (v_mul_lo_u32: the Montgomery m)", ("v_mul_lo", "v_mul_hi")),
    ("M_ext on doubles (v_add_f64, v_fma_f64)", ("v_add_f64", "v_fma_f64", "v_mul_f64")),
    ("int -> double (v_cvt_f64_*)", ("v_cvt_f64",)),
    ("round-constant / plain adds, subs (v_add_u32, v_sub_u32, v_add_co ...)", ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc", "v_sub_co", "v_subrev_co", "v_subb")),
    ("select / min (modular corrections, canonicalisation)", ("v_cndmask", "v_min_u32", "v_max")),
    ("64-bit shift-adds (v_lshl_add_u64: unreduced sum folds)", ("v_lshl_add", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")),
    ("moves, logic, shifts (v_mov, v_and, v_or, v_xor, v_ashrrev, v_lshl ...)", ("v_mov", "v_and", "v_or", "v_xor", "v_ashrrev", "v_lshlrev", "v_lshrrev", "v_bfe", "v_perm", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr")),
]


def cycles(op: str) -> float:
    return 4.0 if op.startswith(FOUR) else 2.5


def main():
    sys.path.insert(0, ROOT)
    from zeth_amd import build as B
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "hash.s")
        cmd = [B.HIPCC, *B.FLAGS, *B.EXTRA_FLAGS.get("hash.hip", []), "--cuda-device-only", "-S", os.path.join(B.CSRC, "hash.hip"), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*k_hash_rows\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    back = []                                           # (target index, branch index) of backward branches = loops
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            back.append((labels[m.group(1)], i))
    back.sort()
    outer = back[0] if back else None                   # first loop nest = the interior-block loop
    outer = max((b for b in back if b[0] <= back[0][0] + 400 and b[1] > back[0][1]), key=lambda b: b[1], default=back[0])
    inner = [b for b in back if outer[0] < b[0] and b[1] < outer[1]]
    # merge loops that share a region (rotated loops produce two backward branches into the same body)
    merged = []
    for b in inner:
        if merged and b[0] <= merged[-1][1]:
            merged[-1] = (min(merged[-1][0], b[0]), max(merged[-1][1], b[1]))
        else:
            merged.append(b)
    kinds = ["full", "partial", "full"] if len(merged) == 3 else ["?"] * len(merged)
    weight = [1.0] * len(body)
    for (a, b), kind in zip(merged, kinds):
        for i in range(a, b + 1):
            weight[i] = TRIPS.get(kind, 1)
    hist = collections.Counter()
    static = collections.Counter()
    other = collections.Counter()
    for i in range(outer[0], outer[1] + 1):
        l = body[i]
        if not l.startswith("\t") or not l.strip() or l.strip().startswith((".", ";")):
            continue
        op = l.strip().split()[0]
        if op.startswith("v_"):
            hist[op] += weight[i]
            static[op] += 1
        else:
            other[op.split("_")[0] + "_" + op.split("_")[1] if "_" in op else op] += weight[i]
    total = sum(hist.values())
    tot_cyc = sum(cycles(op) * n for op, n in hist.items())
    print("# k_hash_rows, steady-state absorb block = ONE Poseidon2 permutation per lane (gfx950, hipcc -O3 -enable-misched=0)")
    print(f"# block loop at asm lines {outer[0]}..{outer[1]} of the kernel; inner loops " +
          ", ".join(f"{k} x{TRIPS.get(k, 1)} [{a}..{b}]" for (a, b), k in zip(merged, kinds)))
    print(f"# dynamic VALU instructions per wave-permutation (static count x trip counts): {total:.0f}"
          f"   (hardware: SQ_INSTS_VALU / (leaves x blocks / 64) = 6.44 k, profiles/r02_sq_counters.txt)")
    print(f"# modelled issue cycles per wave-permutation: {tot_cyc:.0f}  (1024 SIMDs: {tot_cyc / 1024:.1f} SIMD-cycles per 64 permutations)")
    print("#")
    print(f"# {'class':82s} {'instr':>7s} {'share':>6s} {'cycles':>8s} {'share':>6s}")
    seen = set()
    for name, prefixes in CLASSES:
        ops = [op for op in hist if op.startswith(prefixes)]
        seen.update(ops)
        n = sum(hist[o] for o in ops)
        cy = sum(cycles(o) * hist[o] for o in ops)
        print(f"  {name:82s} {n:7.0f} {100 * n / total:5.1f}% {cy:8.0f} {100 * cy / tot_cyc:5.1f}%")
    rest = [op for op in hist if op not in seen]
    n = sum(hist[o] for o in rest)
    cy = sum(cycles(o) * hist[o] for o in rest)
    print(f"  {'other VALU (' + ', '.join(sorted(rest)[:6]) + ')':82s} {n:7.0f} {100 * n / total:5.1f}% {cy:8.0f} {100 * cy / tot_cyc:5.1f}%")
    print("#")
    print("# per opcode (dynamic):")
    for op, n in hist.most_common():
        print(f"  {op:28s} {n:7.0f}  x {cycles(op):.1f} cycles")
    print("# non-VALU in the same block (dynamic): " + ", ".join(f"{k} {v:.0f}" for k, v in other.most_common(8)))
    # the floor
    prods = (8 * 24 + 21) * 4 + 21 * 24                        # s-box products (x^7 = 4 products) + one diagonal product per cell per partial round
    reds = 8 * 24                                              # one Montgomery reduction per cell after each full round's M_ext
    floor = 3 * prods + 2 * reds
    print("#")
    print(f"# FLOOR: {8 * 24 + 21} s-boxes x 4 products = {(8 * 24 + 21) * 4}, plus one diagonal product per cell per partial round ({21 * 24}):")
    print(f"#   {prods} Montgomery products x 3 instructions (multiply-add, low multiply, multiply-add) = {3 * prods}, plus the {reds} reductions")
    print(f"#   that bring every full round's M_ext output back to a word (low multiply + multiply-add) = {2 * reds}:  {floor} instructions, {4 * floor} issue cycles.")
    mult = sum(hist[o] for o in hist if o.startswith(("v_mad_i64", "v_mad_u64", "v_mul_lo", "v_mul_hi")))
    print(f"#   This build spends {mult:.0f} instructions in those opcodes ({100 * mult / total:.0f} % of all VALU) = {mult / floor:.2f} x that floor.")
    print(f"#   Everything that is not a multiply — M_ext on doubles ({sum(hist[o] for o in hist if 'f64' in o and 'cvt' not in o):.0f}), int->double conversions "
          f"({sum(hist[o] for o in hist if 'cvt' in o):.0f}), round-constant adds, the corrections at the partial-round borders —")
    print(f"#   is {total - mult:.0f} instructions ({100 * (total - mult) / total:.0f} %).  The whole block is {total / floor:.2f} x the multiply floor; in issue cycles "
          f"{tot_cyc / (4 * floor):.2f} x.")
    print("# What is left to take: the per-block scale fixes / canonicalisation at the loop borders (the segments outside the three")
    out_loops = sum(n for op, n in static.items()) - 0
    print("#   inner loops) are ~%d instructions per block (%.1f %%); v_cvt_f64 (3.3 %%) has no cheaper form; nothing else is not a multiply or an M_ext add." %
          (sum(1 for i in range(outer[0], outer[1] + 1) if weight[i] == 1.0 and body[i].startswith("\tv_")), 100.0 * sum(1 for i in range(outer[0], outer[1] + 1) if weight[i] == 1.0 and body[i].startswith("\tv_")) / total))


if __name__ == "__main__":
    main()
