#!/usr/bin/env python3
"""Emit this repository's data in the Rust syntax of the upstream files it stands in for — the fixtures that keep
tools/import_upstream_consts.py and tools/import_upstream_circuit.py honest (round trips in tests/test_importers.py), and a
readable dump of a circuit for anyone comparing it with a Zirgen-generated one.

    python tools/export_rust_syntax.py circuit <syn_a|syn_heavy|syn_join|keccak_f|desc.npy> <outdir>   -> taps.rs poly_ext.rs info.rs
    python tools/export_rust_syntax.py consts <outdir> [--montgomery] [--compact]                      -> consts.rs

The layout follows risc0-zkp 3.0.2 `taps.rs` (TapSet / TapData), `adapter.rs` (PolyExtStepDef / PolyExtStep),
`core/hash/poseidon2/consts.rs` (un-vendored: /root/reference/Cargo.lock:5393) as recalled in SURVEY.md Appendix A.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeth_amd.circuits import desc as D  # noqa: E402

P = D.P
_STEP = {D.OP_CONST: ("Const", 1), D.OP_CONST_EXT: ("ConstExt", 4), D.OP_GET: ("Get", 1), D.OP_GET_GLOBAL: ("GetGlobal", 2),
         D.OP_ADD: ("Add", 2), D.OP_SUB: ("Sub", 2), D.OP_MUL: ("Mul", 2), D.OP_TRUE: ("True", 0), D.OP_AND_EQZ: ("AndEqz", 2),
         D.OP_AND_COND: ("AndCond", 3)}


def circuit_rust(desc) -> dict:
    c = D.Circuit.parse(desc)
    regs = c.regs
    taps, group_begin = [], []
    for (g, off, backs, combo) in regs:
        while len(group_begin) <= g:
            group_begin.append(len(taps))
        for k, b in enumerate(backs):
            taps.append(f"        TapData {{ offset: {off}, back: {b}, group: {g}, combo: {combo}, skip: {len(backs) if k == 0 else len(backs)} }},")
    while len(group_begin) < 3:
        group_begin.append(len(taps))
    group_begin.append(len(taps))
    combo_taps = [b for cb in c.combos for b in cb]
    combo_begin = [0]
    for cb in c.combos:
        combo_begin.append(combo_begin[-1] + len(cb))
    taps_rs = ("// taps.rs — emitted by tools/export_rust_syntax.py (TapSet syntax of risc0-zkp src/taps.rs)\n"
               "use risc0_zkp::taps::{TapData, TapSet};\n\n"
               "pub const TAPSET: &TapSet = &TapSet::<'static> {\n    taps: &[\n" + "\n".join(taps) + "\n    ],\n"
               f"    combo_taps: &[{', '.join(map(str, combo_taps))}],\n    combo_begin: &[{', '.join(map(str, combo_begin))}],\n"
               f"    group_begin: &[{', '.join(map(str, group_begin))}],\n    combos_count: {len(c.combos)},\n    reg_count: {len(regs)},\n"
               f"    tot_combo_backs: {len(combo_taps)},\n    group_names: &[\"accum\", \"code\", \"data\"],\n}};\n")
    lines = []
    for (op, a, b, cc, d) in c.steps:
        name, arity = _STEP[op]
        args = (a, b, cc, d)[:arity]
        lines.append(f"        PolyExtStep::{name}" + (f"({', '.join(map(str, args))})" if arity else "") + ",")
    poly_rs = ("// poly_ext.rs — emitted by tools/export_rust_syntax.py (PolyExtStepDef syntax of risc0-zkp src/adapter.rs)\n"
               "use risc0_zkp::adapter::{PolyExtStep, PolyExtStepDef};\n\n"
               "pub const DEF: PolyExtStepDef = PolyExtStepDef {\n    block: &[\n" + "\n".join(lines) + f"\n    ],\n    ret: {c.ret},\n}};\n")
    info_rs = ("// info.rs — emitted by tools/export_rust_syntax.py (CircuitInfo constants)\n"
               "impl CircuitInfo for CircuitImpl {\n"
               f"    const OUTPUT_SIZE: usize = {c.global_sizes[0]};\n    const MIX_SIZE: usize = {c.global_sizes[1]};\n}}\n")
    return {"taps.rs": taps_rs, "poly_ext.rs": poly_rs, "info.rs": info_rs}


def consts_rust(montgomery: bool = False, compact: bool = False) -> str:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from import_upstream_consts import shipped_header
    rc, diag, _ = shipped_header()
    conv = (lambda v: (v << 32) % P) if montgomery else (lambda v: v)
    ctor = "Elem::from_raw" if montgomery else "Elem::new"       # hint only; the importer decides by the known-answer vector
    if compact:
        vals = rc[:4 * 24] + [rc[(4 + r) * 24] for r in range(21)] + rc[25 * 24:]
    else:
        vals = rc
    body = ",\n".join("    " + ", ".join(f"{ctor}(0x{conv(v):08x}_u32)" for v in vals[i:i + 4]) for i in range(0, len(vals), 4))
    dg = ", ".join(f"{ctor}({conv(d)})" for d in diag)
    return ("// consts.rs — emitted by tools/export_rust_syntax.py from include/zkh_poseidon2_consts.h\n"
            "use risc0_core::field::baby_bear::Elem;\n\n"
            f"/// round constants, {'compact (full, partial, full)' if compact else '[round][cell]'}\n"
            f"pub const ROUND_CONSTANTS: [Elem; {len(vals)}] = [\n{body},\n];\n\n"
            f"pub const M_INT_DIAG_HZN: [Elem; 24] = [{dg}];\n")


def named_circuit(name: str):
    from zeth_amd.circuits import syn_air, syn_heavy
    table = {"syn_a": syn_air.syn_a, "syn_small": syn_air.syn_small, "syn_join": syn_air.syn_join, "syn_heavy": syn_heavy.syn_heavy,
             "syn_heavy_small": syn_heavy.syn_heavy_small}
    if name in table:
        return table[name]()
    return np.load(name)


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    if sys.argv[1] == "circuit":
        files = circuit_rust(named_circuit(sys.argv[2]))
        out = sys.argv[3]
    elif sys.argv[1] == "consts":
        out = sys.argv[2]
        files = {"consts.rs": consts_rust("--montgomery" in sys.argv, "--compact" in sys.argv)}
    else:
        sys.exit(__doc__)
    os.makedirs(out, exist_ok=True)
    for fn, txt in files.items():
        with open(os.path.join(out, fn), "w") as fh:
            fh.write(txt)
        print("wrote", os.path.join(out, fn))


if __name__ == "__main__":
    main()
