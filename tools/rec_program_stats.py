#!/usr/bin/env python3
"""What a lift / lift2 / join program is made of: permutations, gates by kind, witness ops by kind, dependency levels
(python tools/rec_program_stats.py > profiles/r03_recursion_program_stats.txt).  Pure host work (no GPU)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeth_amd.circuits import rec_verify as V, recursion as R, syn_air  # noqa: E402

OPS = {R.OP_INPUT: "input", R.OP_GEN: "gen", R.OP_MUX: "mux", R.OP_PACK: "pack", R.OP_UNPACK: "unpack", R.OP_INV: "inv", R.OP_BITS: "bits",
       R.OP_P2: "p2", R.OP_EQ: "eq", R.OP_ISZ: "isz"}
N_IN = {R.OP_GEN: 3, R.OP_MUX: 3, R.OP_PACK: 4, R.OP_UNPACK: 1, R.OP_INV: 1, R.OP_BITS: 1, R.OP_P2: 6, R.OP_EQ: 2, R.OP_ISZ: 1}
N_OUT = {R.OP_UNPACK: 4, R.OP_BITS: 31, R.OP_P2: 6, R.OP_EQ: 0}


def stats(name, pr):
    gates = collections.Counter()
    for g in pr.gates:
        f = g.flags
        kind = ("mux" if f & R.G_MUX else "pub" if f & R.G_PUB else "pack/unpack" if f & (15 * R.G_PACK0) else
                "bool" if f & R.G_BOOL and not any(g.q) else "bit-sum (gen + bool)" if f & R.G_BOOL else
                "assert (gen, no output)" if g.q[4] == 0 else "const" if g.pos[0] < 0 else "mul / muladd" if g.q[0] else "linear")
        gates[kind] += 1
    ops = collections.Counter(OPS[o[0] & 0xff] for o in pr.ops)
    lvl = [0] * pr.n_vars
    width = collections.Counter()
    p2w = collections.Counter()
    for o in pr.ops:
        op, out, ins = o[0] & 0xff, o[1], o[2:]
        lv = 0 if op == R.OP_INPUT else 1 + max((lvl[v] for v in ins[:N_IN[op]] if not (op == R.OP_GEN and v == out)), default=-1)
        for t in range(N_OUT.get(op, 1)):
            lvl[out + t] = lv
        width[lv] += 1
        p2w[lv] += op == R.OP_P2
    n_levels = max(width) + 1
    narrow = sum(1 for lv in range(n_levels) if ((width[lv] - p2w[lv] + 63) & ~63) + 8 * p2w[lv] <= 1024)
    po2 = pr.min_po2()
    A = (1 << po2) - R.ZK_CYCLES
    print(f"{name}: po2 {po2}  ({len(pr.p2s)} of {A // R.BLOCK} Poseidon2 blocks, {len(pr.gates)} of {A - 2 * (A // R.BLOCK)} gate rows), "
          f"{pr.n_vars} wires, {pr.n_inputs} input words")
    print("   gates: " + ", ".join(f"{k} {v} ({100 * v / len(pr.gates):.0f} %)" for k, v in gates.most_common()))
    print("   witness ops: " + ", ".join(f"{k} {v}" for k, v in ops.most_common()))
    print(f"   dependency levels: {n_levels} ({sum(1 for lv in p2w if p2w[lv])} with permutations; widest {max(width.values())} ops, "
          f"{max(p2w.values())} permutations); {narrow} fit a 1024-lane persistent run")


if __name__ == "__main__":
    print("# RECURSION programs of a SYN-A block (tools/rec_program_stats.py); control roots do not change the shape")
    root = list(range(8))
    stats("lift(20)", V.build_lift(syn_air.syn_a(), 20, root))
    stats("lift2(20, 20)", V.build_lift2(syn_air.syn_a(), 20, root, 20, root))
    stats("join(18, 18)", V.build_join(R.recursion_circuit(), 18, 18))
