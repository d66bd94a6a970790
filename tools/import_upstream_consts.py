#!/usr/bin/env python3
"""Diff upstream's Poseidon2 tables against the ones this repository ships — the one-command check that turns
"poseidon2_consts=derived" into "=upstream".

    python tools/import_upstream_consts.py <path/to/risc0-zkp/src/core/hash/poseidon2/consts.rs> [--write-header]

What it stands in for: `ROUND_CONSTANTS` / `M_INT_DIAG_HZN` of risc0-zkp 3.0.2 (un-vendored: /root/reference/Cargo.lock:5393),
the tables behind every digest of every seal `default_prover().prove` returns (/root/reference/crates/host/src/lib.rs:137).
The crate source is not in this image; this tool is what a maintainer runs the moment it is.

It parses the two arrays out of Rust syntax (integer literals, hex or decimal, `_` separators and `u32` suffixes, wrapped
in `Elem::new(..)`, `Elem::from_raw(..)`, `baby_bear_array![..]` or bare), decides how they are ENCODED by running the
permutation on the published known-answer input (tests/golden/poseidon2_kat.json) under every candidate reading —
canonical residues or Montgomery words; 24 x 29 words laid out [round][cell] or the compact 4*24 + 21 + 4*24 = 213 — and
then compares them word for word with include/zkh_poseidon2_consts.h.

exit 0: the upstream tables reproduce the KAT and are identical to the shipped header (with --write-header the header is
        rewritten with ZKH_P2_CONSTS_ARE_DERIVED 0, i.e. zkh_version() reports poseidon2_consts=upstream)
exit 1: they reproduce the KAT but DIFFER from the header (differences are listed; --write-header adopts upstream's)
exit 2: the file could not be parsed, or no reading of it reproduces the KAT
"""
from __future__ import annotations

import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
T, HALF, RP = 24, 4, 21
ROUNDS = 2 * HALF + RP
RINV = pow(1 << 32, -1, P)
M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]


# ---------------------------------------------------------------- a literal Poseidon2 permutation (canonical residues)
def m_ext(s):
    out = [0] * T
    for c in range(0, T, 4):
        for i in range(4):
            out[c + i] = sum(M4[i][j] * s[c + j] for j in range(4)) % P
    sums = [sum(out[c + i] for c in range(0, T, 4)) % P for i in range(4)]
    return [(out[k] + sums[k % 4]) % P for k in range(T)]


def permute(state, rc, diag):
    """rc: 24 x 29 canonical words [round][cell]; diag: 24 canonical words (mu - 1)."""
    s = m_ext([x % P for x in state])
    for r in range(ROUNDS):
        if HALF <= r < HALF + RP:
            s[0] = pow((s[0] + rc[r * T]) % P, 7, P)
            tot = sum(s) % P
            s = [(tot + diag[i] * s[i]) % P for i in range(T)]
        else:
            s = m_ext([pow((s[i] + rc[r * T + i]) % P, 7, P) for i in range(T)])
    return s


# ---------------------------------------------------------------- Rust-syntax array extraction
_LIT = re.compile(r"0[xX][0-9a-fA-F_]+|\d[\d_]*")


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def rust_array(src: str, name_pattern: str):
    """Integer literals of `const <NAME>: <type> = <initialiser>;` -> (name, [ints]) or None.  The type annotation
    (`[Elem; 24 * 29]`) is skipped: literals are collected after the `=` only, up to the `;` at bracket depth 0."""
    m = re.search(r"\b(?:pub(?:\([a-z]+\))?\s+)?(?:const|static)\s+(" + name_pattern + r")\s*:", src)
    if not m:
        return None
    i = src.index("=", _type_end(src, m.end()))
    depth, j = 0, i + 1
    while j < len(src):
        ch = src[j]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == ";" and depth == 0:
            break
        j += 1
    body = src[i + 1:j]
    body = re.sub(r"\b(?:u32|u64|usize)\b", " ", body)            # `0x1234u32` is one token for the regex below; `as u32` casts
    vals = []
    for tok in _LIT.finditer(body):
        # literals that are part of an identifier (Elem2, x86) are not numbers
        if tok.start() > 0 and (body[tok.start() - 1].isalpha() or body[tok.start() - 1] == "_"):
            continue
        t = tok.group(0).replace("_", "")
        t = re.sub(r"(?i)u(32|64|size)$", "", t) if not t.lower().startswith("0x") else t
        vals.append(int(t, 16) if t.lower().startswith("0x") else int(t))
    return m.group(1), vals


def _type_end(src: str, start: int) -> int:
    """Position of the `=` that ends the type annotation starting at `start` (skips `;` inside `[Elem; N]`)."""
    depth, j = 0, start
    while j < len(src):
        ch = src[j]
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        elif ch == "=" and depth == 0:
            return j
        j += 1
    raise ValueError("no initialiser")


def parse_consts(src: str):
    src = strip_comments(src)
    rc = rust_array(src, r"ROUND_CONSTANTS\w*")
    diag = rust_array(src, r"M_INT_DIAG\w*")
    if rc is None or diag is None:
        raise ValueError("ROUND_CONSTANTS / M_INT_DIAG_* not found")
    return rc, diag


# ---------------------------------------------------------------- readings
def layouts(vals):
    """Candidate [round][cell] tables (696 words, unused cells zero) for a list of round-constant words."""
    out = []
    if len(vals) == T * ROUNDS:
        rc = list(vals)
        for r in range(HALF, HALF + RP):
            for c in range(1, T):
                rc[r * T + c] = 0                      # upstream's partial rounds only read cell 0
        out.append(("24 x 29 words, [round][cell]", rc))
    if len(vals) == 2 * HALF * T + RP:
        rc = [0] * (T * ROUNDS)
        rc[:HALF * T] = vals[:HALF * T]
        for r in range(RP):
            rc[(HALF + r) * T] = vals[HALF * T + r]
        rc[(HALF + RP) * T:] = vals[HALF * T + RP:]
        out.append(("compact 4*24 + 21 + 4*24", rc))
    return out


def shipped_header():
    path = os.path.join(ROOT, "include", "zkh_poseidon2_consts.h")
    src = open(path).read()

    def arr(name):
        body = src[src.index(name):]
        body = body[body.index("{") + 1:body.index("}")]
        return [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-fA-F]+u?", body)]
    return arr("ZKH_P2_ROUND_CONSTANTS"), arr("ZKH_P2_M_INT_DIAG"), path


def load_kat():
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_kat.json")))
    return k["input"], [int(x, 16) for x in k["output_hex"]]


def identify(rc_vals, diag_vals):
    """-> (description, rc[696] canonical, diag[24] canonical) of the first reading that reproduces the KAT, or None."""
    kin, kout = load_kat()
    if len(diag_vals) != T:
        return None
    for enc_name, conv in (("canonical residues", lambda v: v % P), ("Montgomery words", lambda v: v * RINV % P)):
        if any(v >= (1 << 32) for v in rc_vals + diag_vals):
            return None
        diag = [conv(v) for v in diag_vals]
        for lay_name, rc in layouts([conv(v) for v in rc_vals]):
            for diag_name, dg in (("diagonal stored as mu - 1", diag), ("diagonal stored as mu", [(d - 1) % P for d in diag])):
                if permute(kin, rc, dg) == kout:
                    return f"{enc_name}; {lay_name}; {diag_name}", rc, dg
    return None


def write_header(rc, diag, path, source):
    with open(path, "w") as f:
        f.write("/* WRITTEN by tools/import_upstream_consts.py from " + source + " — Poseidon2 (BabyBear, t = 24, x^7, R_F = 8,\n"
                " * R_P = 21) tables of risc0-zkp src/core/hash/poseidon2/consts.rs, checked against the published known-answer vector\n"
                " * (tests/golden/poseidon2_kat.json).  Canonical (non-Montgomery) residues mod P = 2013265921; RC[round * 24 + cell],\n"
                " * partial rounds (4..24) only have cell 0. */\n"
                "#ifndef ZKH_POSEIDON2_CONSTS_H\n#define ZKH_POSEIDON2_CONSTS_H\n#include <stdint.h>\n"
                "#define ZKH_P2_CELLS 24\n#define ZKH_P2_RATE 16\n#define ZKH_P2_OUT 8\n"
                "#define ZKH_P2_ROUNDS_HALF_FULL 4\n#define ZKH_P2_ROUNDS_PARTIAL 21\n#define ZKH_P2_ROUNDS 29\n"
                "#define ZKH_P2_CONSTS_ARE_PLACEHOLDER 0\n"
                "#define ZKH_P2_CONSTS_ARE_DERIVED 0      /* compared word for word with upstream's consts.rs */\n")
        f.write("static const uint32_t ZKH_P2_M_INT_DIAG[24] = {\n")
        for i in range(0, 24, 8):
            f.write("    " + ", ".join("0x%08xu" % d for d in diag[i:i + 8]) + ",\n")
        f.write("};\nstatic const uint32_t ZKH_P2_ROUND_CONSTANTS[24 * 29] = {\n")
        for i in range(0, 24 * 29, 8):
            f.write("    " + ", ".join("0x%08xu" % d for d in rc[i:i + 8]) + ",\n")
        f.write("};\n#endif\n")


def compare(path: str, write: bool = False, header_path: str = None, out=print) -> int:
    try:
        (rc_name, rc_vals), (diag_name, diag_vals) = parse_consts(open(path).read())
    except (OSError, ValueError) as e:
        out(f"cannot parse {path}: {e}")
        return 2
    out(f"{path}: {rc_name} has {len(rc_vals)} words, {diag_name} has {len(diag_vals)}")
    found = identify(rc_vals, diag_vals)
    if found is None:
        out("no reading of these tables (canonical / Montgomery, 696 / 213 words, mu / mu - 1) reproduces the published known-answer vector")
        return 2
    how, rc, diag = found
    out(f"reading that reproduces the published known-answer vector: {how}")
    h_rc, h_diag, h_path = shipped_header()
    if header_path:
        h_path = header_path
    diffs = [("ROUND_CONSTANTS", i // T, i % T, h_rc[i], rc[i]) for i in range(T * ROUNDS) if h_rc[i] != rc[i]]
    diffs += [("M_INT_DIAG", 0, i, h_diag[i], diag[i]) for i in range(T) if h_diag[i] != diag[i]]
    if not diffs:
        out(f"identical to include/zkh_poseidon2_consts.h: all {2 * HALF * T + RP} round constants and {T} diagonal entries")
        if write:
            write_header(rc, diag, h_path, os.path.basename(path))
            out(f"rewrote {h_path}: poseidon2_consts=upstream after the next build")
        return 0
    out(f"{len(diffs)} words differ from include/zkh_poseidon2_consts.h; first ones:")
    for name, r, c, ours, theirs in diffs[:12]:
        out(f"  {name}[round {r}][cell {c}]: shipped 0x{ours:08x}, upstream 0x{theirs:08x}")
    if write:
        write_header(rc, diag, h_path, os.path.basename(path))
        out(f"rewrote {h_path} with upstream's tables (regenerate the goldens: tests/golden/make_golden*.py)")
    return 1


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 1:
        sys.exit(__doc__)
    sys.exit(compare(args[0], write="--write-header" in sys.argv))
