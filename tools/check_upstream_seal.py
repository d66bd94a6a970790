#!/usr/bin/env python3
"""Walk a seal against a circuit description and report the FIRST place where it disagrees with the layout and the
Fiat-Shamir order this repository implements — the layout cross-check for the day an upstream seal is available.

    python tools/check_upstream_seal.py <seal.bin|seal.npy> <circuit.desc.npy|syn_a|...> [--control-root 8 hex words | self]

`seal.bin` = the `Vec<u32>` of a `SegmentReceipt.seal` as little-endian bytes (what `receipt.verify(image_id)`,
/root/reference/crates/host/src/bin/cli.rs:103, checks; produced by risc0-zkp 3.0.2's prover, un-vendored:
/root/reference/Cargo.lock:5393).  The circuit comes from tools/import_upstream_circuit.py.  No GPU is needed: the walk uses
the host verifier `zkh_verify_segment` (csrc/verifier.hip), whose error carries the seal position it had reached.

It prints the section table the seal must have for this (circuit, po2) — header `out ‖ po2`; the 32 top digests of the
code, data, accum and check trees; coeff_u; per FRI round 32 top digests; the final coefficients; 50 queries x (column +
path per tree) — checks the total length, checks that field-element sections hold reduced words, runs the verifier, and
maps its first failure to a section together with the recalled protocol detail that section depends on.

exit 0: the seal is accepted   exit 1: first disagreement reported   exit 2: unreadable input
"""
from __future__ import annotations

import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P = 2013265921
RINV = pow(1 << 32, -1, P)
INV_RATE, QUERIES, FRI_FOLD, FRI_MIN_DEGREE, CHECK_SIZE, EXT = 4, 50, 16, 256, 16, 4

HINTS = {
    "header": "seal header = OUTPUT_SIZE `out` words then po2, all as Elems, hashed and committed once (SegmentProver / verify/mod.rs)",
    "top": "MerkleTreeParams: top_layer = largest i < log2(rows) with 2^i <= QUERIES (32 digests), written as nodes[32..64), then commit(root); "
           "tree order code, data, (mix draw), accum, (poly_mix draw), check",
    "code top": "verify/mod.rs check_code: the code tree's root must equal the control root of this (circuit, po2)",
    "coeff_u": "coeff_u = per register poly_interpolate of the tap evaluations, in tap order (group, offset, back), then the 16 check "
               "coefficients evaluated at z^4; hash_ext_elem_slice, one commit; z drawn after the check commit",
    "fri top": "FRI round: commit(top of the rows = domain/16, cols = 64 tree), then the fold mix; rounds while degree > 256",
    "final": "final polynomial: 4 planes of `degree` Elems (plane-major), hashed and committed",
    "query": "query index = random_bits(log2 domain) — FOUR element draws per call, first non-zero wins; per query: accum, code, data, "
             "check openings (column then siblings up to the top layer), then every FRI round at pos mod (domain/16)",
}


def log2(x: int) -> int:
    return x.bit_length() - 1


def tree_shape(rows: int):
    layers = log2(rows)
    top_layer = 0
    for i in range(1, layers):
        if (1 << i) > QUERIES:
            break
        top_layer = i
    return layers, top_layer, 1 << top_layer


def layout(circ, po2: int):
    """[(name, kind, start, end)] for a seal of this circuit at this po2."""
    n = 1 << po2
    dom = n * INV_RATE
    wa, wc, wd = circ.group_sizes
    out = []
    pos = 0

    def add(name, kind, words):
        nonlocal pos
        out.append((name, kind, pos, pos + words))
        pos += words
    add("header (out ‖ po2)", "header", circ.global_sizes[0] + 1)
    _, _, top = tree_shape(dom)
    add("code tree top layer", "code top", 8 * top)
    add("data tree top layer", "top", 8 * top)
    add("accum tree top layer", "top", 8 * top)
    add("check tree top layer", "top", 8 * top)
    add("coeff_u", "coeff_u", EXT * (len(circ.taps) + CHECK_SIZE))
    degree, d = n, dom
    fri = []
    while degree > FRI_MIN_DEGREE:
        rows = d // FRI_FOLD
        _, _, t = tree_shape(rows)
        add(f"FRI round {len(fri)} top layer (rows {rows})", "fri top", 8 * t)
        fri.append(rows)
        degree //= FRI_FOLD
        d //= FRI_FOLD
    add(f"final coefficients (degree {degree})", "final", EXT * degree)
    trees = [("accum", dom, wa), ("code", dom, wc), ("data", dom, wd), ("check", dom, CHECK_SIZE)] + \
            [(f"FRI {k}", rows, FRI_FOLD * EXT) for k, rows in enumerate(fri)]
    for q in range(QUERIES):
        for name, rows, cols in trees:
            layers, top_layer, _ = tree_shape(rows)
            add(f"query {q}: {name} opening", "query", cols + 8 * (layers - top_layer))
    return out


def load_words(path: str) -> np.ndarray:
    if path.endswith(".npy"):
        return np.load(path).astype(np.uint32).reshape(-1)
    raw = open(path, "rb").read()
    if len(raw) % 4:
        raise ValueError(f"{path}: {len(raw)} bytes is not a whole number of u32 words")
    return np.frombuffer(raw, dtype="<u4").astype(np.uint32)


def load_desc(arg: str) -> np.ndarray:
    if os.path.exists(arg):
        return np.load(arg).astype(np.uint32)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_rust_syntax import named_circuit
    return named_circuit(arg)


def self_root(seal: np.ndarray, lay) -> np.ndarray:
    """Root of the code tree as the seal itself states it (folds the 32 top digests): skips check_code on purpose."""
    from zeth_amd import hal as H
    H.load_library()
    _, _, a, b = next(s for s in lay if s[1] == "code top")
    layer = [seal[a + 8 * i:a + 8 * i + 8] for i in range((b - a) // 8)]
    while len(layer) > 1:
        nxt = []
        for i in range(0, len(layer), 2):
            st = np.zeros(24, dtype=np.uint32)
            st[:8], st[8:16] = layer[i], layer[i + 1]
            H._check(H._lib.zkh_poseidon2_mix_host(None, None, st.ctypes.data_as(C.POINTER(C.c_uint32)), 1))
            nxt.append(st[:8].copy())
        layer = nxt
    return layer[0]


def check(seal: np.ndarray, desc: np.ndarray, control_root=None, out=print) -> int:
    from zeth_amd.circuits.desc import Circuit
    from zeth_amd import hal as H
    circ = Circuit.parse(desc)
    out_size = circ.global_sizes[0]
    if seal.size <= out_size:
        out(f"seal has {seal.size} words: shorter than the header ({out_size + 1}) — {HINTS['header']}")
        return 1
    po2_word = int(seal[out_size])
    po2 = po2_word * RINV % P
    if po2_word >= P or not (1 <= po2 <= 24):
        out(f"header word {out_size} = 0x{po2_word:08x} decodes to po2 = {po2}: not a segment size.  {HINTS['header']}")
        return 1
    lay = layout(circ, po2)
    out(f"circuit: groups accum/code/data = {circ.group_sizes}, {len(circ.taps)} taps, {len(circ.combos)} combos; seal says po2 = {po2}")
    out("expected layout:")
    per_query = sum(1 for s in lay if s[1] == "query") // QUERIES
    for name, kind, a, b in [s for s in lay if s[1] != "query"] + [s for s in lay if s[1] == "query"][:per_query]:
        out(f"  [{a:>8}, {b:>8})  {name}")
    out(f"  ... {QUERIES} queries in all; total {lay[-1][3]} words")
    if lay[-1][3] != seal.size:
        out(f"LENGTH: the seal has {seal.size} words, this layout has {lay[-1][3]} (difference {seal.size - lay[-1][3]:+d}): "
            "tap count / group widths / FRI parameters / top-layer rule of the circuit or protocol differ")
    # reduced-word checks of the field-element sections (digests are 8 Elems each as well)
    for name, kind, a, b in lay:
        part = seal[a:min(b, seal.size)]
        bad = np.nonzero(part >= P)[0]
        if bad.size:
            out(f"UNREDUCED word 0x{int(part[bad[0]]):08x} at seal word {a + int(bad[0])} in section `{name}`: every seal word is a Montgomery-form "
                f"Elem < P — a different encoding of this section?  {HINTS.get(kind, '')}")
            return 1
    if control_root is None:
        control_root = self_root(seal, lay)
        out("control root: taken from the seal itself (check_code skipped; pass --control-root to enforce it)")
    try:
        H.HostCircuit(desc).verify_segment(seal, np.asarray(control_root, dtype=np.uint32))
    except H.HalError as e:
        msg = str(e)
        m = re.search(r"\(seal word (\d+) of (\d+)\)", msg)
        at = int(m.group(1)) if m else None
        sect = None
        if at is not None:
            # the verifier has CONSUMED everything before `at`: the failing check concerns the section that ends there, or
            # (for checks that follow a draw) the one that starts there
            sect = next((s for s in lay if s[2] < at <= s[3]), None) or next((s for s in lay if s[2] <= at < s[3]), None)
        out(f"FIRST DISAGREEMENT: {msg}")
        if sect:
            out(f"  in section `{sect[0]}` = seal words [{sect[2]}, {sect[3]})")
            out(f"  what that section depends on: {HINTS.get(sect[1], '')}")
        return 1
    out("seal accepted: header, commitments, constraint check at z, DEEP quotients, FRI and all 50 queries agree with this layout")
    return 0


def main():
    args = [a for a in sys.argv[1:]]
    root = None
    if "--control-root" in args:
        i = args.index("--control-root")
        vals = args[i + 1:i + 9]
        if vals and vals[0] == "self":
            del args[i:i + 2]
        else:
            root = [int(v, 16) for v in vals]
            del args[i:i + 9]
    if len(args) != 2:
        sys.exit(__doc__)
    try:
        seal, desc = load_words(args[0]), load_desc(args[1])
    except (OSError, ValueError) as e:
        print(f"cannot read input: {e}")
        sys.exit(2)
    sys.exit(check(seal, desc, root))


if __name__ == "__main__":
    main()
