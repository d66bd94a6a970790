#!/usr/bin/env python3
"""Turn a Zirgen-generated circuit's Rust tables into this repository's circuit description blob.

    python tools/import_upstream_circuit.py <taps.rs> <poly_ext.rs> [info.rs] [--out-size N --mix-size N] [--kind K] -o circuit.desc.npy

What it stands in for: `risc0-circuit-rv32im 4.0.2 src/zirgen/{taps.rs, poly_ext.rs, info.rs}` (and the same three files of
risc0-circuit-recursion / risc0-circuit-keccak; un-vendored: /root/reference/Cargo.lock:5320, :5305, :5289) — the tables
`risc0_zkp::adapter::{TapsProvider, PolyExtStepDef, CircuitInfo}` expose to the prover that
/root/reference/crates/host/src/lib.rs:137 ends up in.  Here a circuit is DATA (zeth_amd/circuits/desc.py): the blob this
tool writes goes through `zkh_circuit_load`, the eval_check generator (circuits/codegen.py / jit.py), the prover and the
verifier unchanged.  The crate sources are not in this image; this is the command a maintainer runs when they are.

Accepted syntax (whitespace, comments, field order and trailing commas free):

    taps.rs      TapSet { taps: &[ TapData { offset: 0, back: 0, group: 0, combo: 0, skip: 1 }, ... ],
                          combo_taps: &[...], combo_begin: &[...], group_begin: &[...], combos_count: N, reg_count: N,
                          tot_combo_backs: N, ... }
    poly_ext.rs  PolyExtStepDef { block: &[ PolyExtStep::Const(1), PolyExtStep::ConstExt(a, b, c, d), PolyExtStep::Get(t),
                          PolyExtStep::GetGlobal(base, off), PolyExtStep::Add(a, b), ::Sub, ::Mul, PolyExtStep::True,
                          PolyExtStep::AndEqz(x, v), PolyExtStep::AndCond(x, cond, inner) ], ret: R }
    info.rs      `const OUTPUT_SIZE: usize = N;`  `const MIX_SIZE: usize = N;`   (or pass --out-size / --mix-size)

Everything the tables state redundantly is cross-checked (tap order, `skip`, `combo` ids against combo_taps / combo_begin,
group_begin, reg_count, tot_combo_backs, operand indices of every step) so that a mis-parse fails here, not inside a seal.
"""
from __future__ import annotations

import argparse
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeth_amd.circuits import desc as D  # noqa: E402

P = D.P


class ImportError_(ValueError):
    pass


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _int(tok: str) -> int:
    t = tok.strip().replace("_", "")
    t = re.sub(r"(?i)(u8|u16|u32|u64|usize)$", "", t)
    return int(t, 16) if t.lower().startswith("0x") else int(t)


def _matching(src: str, open_at: int) -> int:
    """index of the bracket that closes the one at `open_at`"""
    pairs = {"(": ")", "[": "]", "{": "}"}
    close = pairs[src[open_at]]
    depth = 0
    for j in range(open_at, len(src)):
        if src[j] == src[open_at]:
            depth += 1
        elif src[j] == close:
            depth -= 1
            if depth == 0:
                return j
    raise ImportError_("unbalanced brackets")


def _field_array(body: str, name: str):
    m = re.search(r"\b" + name + r"\s*:\s*&?\s*\[", body)
    if not m:
        return None
    a = m.end() - 1
    b = _matching(body, a)
    return body[a + 1:b]


def _field_int(body: str, name: str):
    m = re.search(r"\b" + name + r"\s*:\s*([0-9xXa-fA-F_]+)", body)
    return None if not m else _int(m.group(1))


def parse_taps_rs(src: str) -> dict:
    src = strip_comments(src)
    start = src.index("=") if "=" in src else 0          # the struct literal after `const TAPSET: &TapSet = ...`
    k = src.index("{", src.index("TapSet", start))
    body = src[k + 1:_matching(src, k)]
    taps_txt = _field_array(body, "taps")
    if taps_txt is None:
        raise ImportError_("taps.rs: no `taps: &[...]` field")
    taps = []
    for t in re.finditer(r"TapData\s*\{([^}]*)\}", taps_txt):
        f = {k: _int(v) for k, v in re.findall(r"(\w+)\s*:\s*([0-9xXa-fA-F_]+\w*)", t.group(1))}
        missing = {"offset", "back", "group", "combo", "skip"} - set(f)
        if missing:
            raise ImportError_(f"taps.rs: TapData without {sorted(missing)}: {t.group(0)[:80]}")
        taps.append(f)
    if not taps:
        raise ImportError_("taps.rs: no TapData entries")

    def ints(name, required=True):
        txt = _field_array(body, name)
        if txt is None:
            if required:
                raise ImportError_(f"taps.rs: no `{name}` field")
            return None
        return [_int(x) for x in re.findall(r"[0-9][0-9xXa-fA-F_]*\w*", txt)]
    return {"taps": taps, "combo_taps": ints("combo_taps"), "combo_begin": ints("combo_begin"),
            "group_begin": ints("group_begin", False), "combos_count": _field_int(body, "combos_count"),
            "reg_count": _field_int(body, "reg_count"), "tot_combo_backs": _field_int(body, "tot_combo_backs")}


_STEP_OPS = {"Const": (D.OP_CONST, 1), "ConstExt": (D.OP_CONST_EXT, 4), "Get": (D.OP_GET, 1), "GetGlobal": (D.OP_GET_GLOBAL, 2),
             "Add": (D.OP_ADD, 2), "Sub": (D.OP_SUB, 2), "Mul": (D.OP_MUL, 2), "True": (D.OP_TRUE, 0),
             "AndEqz": (D.OP_AND_EQZ, 2), "AndCond": (D.OP_AND_COND, 3)}


def parse_poly_ext_rs(src: str):
    src = strip_comments(src)
    k = src.index("{", src.index("PolyExtStepDef", src.index("=") if "=" in src else 0))
    body = src[k + 1:_matching(src, k)]
    block = _field_array(body, "block")
    if block is None:
        raise ImportError_("poly_ext.rs: no `block: &[...]` field")
    steps = []
    for m in re.finditer(r"PolyExtStep\s*::\s*(\w+)\s*(?:\(([^)]*)\))?", block):
        name, args = m.group(1), m.group(2)
        if name not in _STEP_OPS:
            raise ImportError_(f"poly_ext.rs: unknown step PolyExtStep::{name}")
        op, arity = _STEP_OPS[name]
        vals = [_int(x) for x in args.split(",") if x.strip()] if args else []
        if len(vals) != arity:
            raise ImportError_(f"poly_ext.rs: PolyExtStep::{name} takes {arity} operands, got {len(vals)}")
        steps.append((op, *vals, *([0] * (4 - len(vals)))))
    ret = _field_int(body.replace(block, " "), "ret")
    if ret is None:
        raise ImportError_("poly_ext.rs: no `ret` field")
    return steps, ret


def parse_info_rs(src: str) -> dict:
    src = strip_comments(src)
    out = {}
    for key in ("OUTPUT_SIZE", "MIX_SIZE"):
        m = re.search(r"\b" + key + r"\s*:\s*usize\s*=\s*([0-9xXa-fA-F_]+)", src)
        if m:
            out[key] = _int(m.group(1))
    return out


def build_desc(tapset: dict, steps, ret: int, out_size: int, mix_size: int, kind: int = 0) -> np.ndarray:
    taps = tapset["taps"]
    triples = [(t["group"], t["offset"], t["back"]) for t in taps]
    if triples != sorted(triples) or len(set(triples)) != len(triples):
        raise ImportError_("taps.rs: taps are not strictly sorted by (group, offset, back)")
    if any(g > 2 for g, _, _ in triples):
        raise ImportError_("taps.rs: more than three register groups")
    # combos, in upstream's own order (the order fixes every register's combo id, hence the seal)
    cb, ct = tapset["combo_begin"], tapset["combo_taps"]
    if cb[0] != 0 or cb[-1] != len(ct) or any(a >= b for a, b in zip(cb, cb[1:])):
        raise ImportError_("taps.rs: combo_begin is not an increasing partition of combo_taps")
    combos = [tuple(ct[a:b]) for a, b in zip(cb, cb[1:])]
    if len(set(combos)) != len(combos):
        raise ImportError_("taps.rs: a combo appears twice")
    if tapset["combos_count"] is not None and tapset["combos_count"] != len(combos):
        raise ImportError_(f"taps.rs: combos_count {tapset['combos_count']} but combo_begin describes {len(combos)}")
    if tapset["tot_combo_backs"] is not None and tapset["tot_combo_backs"] != len(ct):
        raise ImportError_("taps.rs: tot_combo_backs does not match combo_taps")
    # registers: runs of equal (group, offset); skip of the first tap = run length; combo id = position of the run's backs
    regs, i = 0, 0
    group_sizes = [0, 0, 0]
    group_first = {}
    while i < len(taps):
        j = i
        while j < len(taps) and triples[j][:2] == triples[i][:2]:
            j += 1
        backs = tuple(t[2] for t in triples[i:j])
        if taps[i]["skip"] != j - i:
            raise ImportError_(f"taps.rs: tap {i} has skip {taps[i]['skip']} but its register has {j - i} taps")
        if backs not in combos:
            raise ImportError_(f"taps.rs: register (group {triples[i][0]}, offset {triples[i][1]}) reads backs {backs}, which is no combo")
        for t in taps[i:j]:
            if t["combo"] != combos.index(backs):
                raise ImportError_(f"taps.rs: register (group {triples[i][0]}, offset {triples[i][1]}) says combo {t['combo']}, "
                                   f"its backs {backs} are combo {combos.index(backs)}")
        g, off = triples[i][:2]
        group_first.setdefault(g, i)
        if off != group_sizes[g]:
            raise ImportError_(f"taps.rs: group {g} skips from offset {group_sizes[g] - 1} to {off}: every register must be tapped")
        group_sizes[g] = off + 1
        regs += 1
        i = j
    if tapset["reg_count"] is not None and tapset["reg_count"] != regs:
        raise ImportError_(f"taps.rs: reg_count {tapset['reg_count']} but the taps describe {regs} registers")
    gb = tapset["group_begin"]
    if gb is not None:
        want = [group_first.get(g, len(taps)) for g in range(3)] + [len(taps)]
        for g in (1, 0):                       # an empty group begins where the next one does
            if g not in group_first:
                want[g] = want[g + 1]
        if gb != want:
            raise ImportError_(f"taps.rs: group_begin {gb} does not match the taps ({want})")
    # steps: operand indices must refer backwards
    n_fp = n_mix = 0
    for k, (op, a, b, c, d) in enumerate(steps):
        def need(idx, limit, what):
            if not (0 <= idx < limit):
                raise ImportError_(f"poly_ext.rs: step {k} refers to {what} {idx}, only {limit} defined so far")
        if op == D.OP_GET:
            need(a, len(taps), "tap")
        elif op == D.OP_GET_GLOBAL:
            if a not in (0, 1) or b >= (out_size, mix_size)[a]:
                raise ImportError_(f"poly_ext.rs: step {k} reads global ({a}, {b}) outside out[{out_size}] / mix[{mix_size}]")
        elif op in (D.OP_ADD, D.OP_SUB, D.OP_MUL):
            need(a, n_fp, "value"); need(b, n_fp, "value")
        elif op == D.OP_AND_EQZ:
            need(a, n_mix, "mix state"); need(b, n_fp, "value")
        elif op == D.OP_AND_COND:
            need(a, n_mix, "mix state"); need(b, n_fp, "value"); need(c, n_mix, "mix state")
        elif op in (D.OP_CONST, D.OP_CONST_EXT):
            if any(v >= P for v in (a, b, c, d)):
                raise ImportError_(f"poly_ext.rs: step {k} holds a constant >= P")
        if op in (D.OP_TRUE, D.OP_AND_EQZ, D.OP_AND_COND):
            n_mix += 1
        else:
            n_fp += 1
    if not (0 <= ret < n_mix):
        raise ImportError_(f"poly_ext.rs: ret {ret} is not a mix state ({n_mix} defined)")
    words = [D.MAGIC, 1, 3, *group_sizes, 2, out_size, mix_size, len(taps), len(combos), len(steps), ret, kind, 0, 0]
    for t in triples:
        words.extend(t)
    for c in combos:
        words.append(len(c))
        words.extend(c)
    for s in steps:
        words.extend(s)
    return np.asarray(words, dtype=np.uint32)


def import_circuit(taps_rs: str, poly_ext_rs: str, info_rs: str = None, out_size: int = None, mix_size: int = None, kind: int = 0):
    tapset = parse_taps_rs(open(taps_rs).read())
    steps, ret = parse_poly_ext_rs(open(poly_ext_rs).read())
    info = parse_info_rs(open(info_rs).read()) if info_rs else {}
    out_size = out_size if out_size is not None else info.get("OUTPUT_SIZE")
    mix_size = mix_size if mix_size is not None else info.get("MIX_SIZE")
    if out_size is None or mix_size is None:
        raise ImportError_("OUTPUT_SIZE / MIX_SIZE unknown: pass info.rs or --out-size / --mix-size")
    return build_desc(tapset, steps, ret, out_size, mix_size, kind)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("taps_rs"); ap.add_argument("poly_ext_rs"); ap.add_argument("info_rs", nargs="?")
    ap.add_argument("--out-size", type=int); ap.add_argument("--mix-size", type=int)
    ap.add_argument("--kind", type=int, default=0, help="0 = no built-in witness generator (seal through zkh_prove_begin / zkh_prove_finish)")
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args()
    try:
        blob = import_circuit(a.taps_rs, a.poly_ext_rs, a.info_rs, a.out_size, a.mix_size, a.kind)
    except (ImportError_, OSError) as e:
        sys.exit(f"import failed: {e}")
    np.save(a.output, blob)
    c = D.Circuit.parse(blob)
    from zeth_amd.circuits.codegen import desc_hash64
    print(f"{a.output}: groups accum/code/data = {c.group_sizes}, globals out/mix = {c.global_sizes}, {len(c.taps)} taps, "
          f"{len(c.combos)} combos, {len(c.steps)} steps, desc hash {desc_hash64(blob):016x}")


if __name__ == "__main__":
    main()
