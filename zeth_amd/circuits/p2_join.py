"""P2-JOIN: the join circuit of the succinct-receipt tree — Poseidon2 INSIDE the AIR.

Upstream's recursion circuit (risc0-circuit-recursion 4.0.2, un-vendored: /root/reference/Cargo.lock:5305; BASELINE.json
config 5: "lift/join recursion to a single succinct receipt") verifies two child seals in-circuit; its dominant workload is
the Poseidon2 permutation evaluated as trace rows (Merkle paths and the Fiat-Shamir sponge of the child proofs).  That
Zirgen-generated circuit cannot be obtained offline.  P2-JOIN keeps the part that can be stated from public material: it
CONSTRAINS   parent = Poseidon2-hash_pair(claim_left, claim_right)   with the permutation unrolled over trace rows, so a
join tree is a Merkle tree of claims that a verifier follows from the root receipt's `out` alone (zeth_amd/host.py
SuccinctReceipt) — and fills the rest of the segment with further permutations of the same kind (inputs bound to the
parent and to public sibling words in the code group: the shape of Merkle-path verification), ~8.4 k permutations at po2 18.
It does NOT verify the child seals (declared).  The permutation, its tables and hash_pair are the ones every Merkle tree of
the prover uses (risc0-zkp src/core/hash/poseidon2/mod.rs; pinned here by the published known-answer vector,
tests/golden/poseidon2_kat.json).

Trace (A = n - zk_cycles active rows, K = A // 31 blocks of 31 rows; row k of a block):
  k = 0      S = the permutation's input (block 0: left ‖ right ‖ 0^8; block p >= 1: parent ‖ sib_p ‖ 0^8)
  k = 1..4   S = state before full round k-1 (k = 1: M_ext of the input),  Q_j = (S_j + rc_j)^3
  k = 5..25  partial rounds 0..20: Q_0 = (S_0 + rc_0)^3
  k = 26..29 full rounds 4..7;   k = 30: S = the output state
  data (48 columns): S[24], Q[24].   x^7 = Q^2 (S + rc): every constraint has degree <= 3 (+ its selector).
  code (43 columns, a function of (po2, zk_cycles)): 0 active 1 first 2 body (accum argument)  3 in0  4 inp  5 lin
      6 fullr 7 partr (this row performs a full / partial round)  8 lf 9 lp (this row follows a full / partial round)
      10 bind0 (row 30)  11..34 the round constants of this row's round  35..42 sib_p on the input rows of blocks p >= 1
  accum: one Fp4 running product of (mix + data column 0) — SYN-AIR's argument and kernel.
Globals: out = parent (8) ‖ left (8) ‖ right (8);  mix = 4 words.
"""
from __future__ import annotations

import os
import re

import numpy as np

from .desc import GLOBAL_MIX, GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, P, CircuitBuilder

KIND_P2_JOIN = 3
T, HALF, RP = 24, 4, 21
ROUNDS = 2 * HALF + RP
BLOCK_ROWS = ROUNDS + 2                     # input row, 29 round rows, output row
WC, WD, WA, OUT_WORDS = 43, 48, 4, 24
C_IN0, C_INP, C_LIN, C_FULLR, C_PARTR, C_LF, C_LP, C_BIND0, C_RC, C_SIB = 3, 4, 5, 6, 7, 8, 9, 10, 11, 35
M4 = ((5, 7, 1, 3), (4, 6, 1, 1), (1, 3, 5, 7), (1, 1, 4, 6))
NBETA = P - 11
R = (1 << 32) % P
RINV = pow(R, -1, P)


def shipped_tables():
    """(rc[24 * 29], diag[24]) canonical residues: circuits/poseidon2_consts.py, generated in the same run as
    include/zkh_poseidon2_consts.h (tests/test_abi_and_host.py compares the two word for word)."""
    from . import poseidon2_consts as pc
    return list(pc.ROUND_CONSTANTS), list(pc.M_INT_DIAG)


RC, DIAG = shipped_tables()


def round_kind(k: int) -> str:
    """what row k of a block DOES: 'in', 'full', 'partial' or 'out'"""
    if k == 0:
        return "in"
    if k <= HALF or HALF + RP < k <= ROUNDS:
        return "full"
    if k <= HALF + RP:
        return "partial"
    return "out"


def m_ext(s):
    out = [0] * T
    for c in range(0, T, 4):
        for i in range(4):
            out[c + i] = sum(M4[i][j] * s[c + j] for j in range(4)) % P
    sums = [sum(out[c + i] for c in range(0, T, 4)) % P for i in range(4)]
    return [(out[k] + sums[k % 4]) % P for k in range(T)]


def block_rows(inp):
    """One permutation as trace rows over canonical residues: [(S[24], Q[24])] for k = 0..30.  The plain-Python statement
    of what the witness generators (oracle/p2join.c, circuit.hip k_p2join_*) must produce; rows[30][0] is the output."""
    rows = [([x % P for x in inp], [0] * T)]
    s = m_ext(rows[0][0])
    for rnd in range(ROUNDS):
        rc = RC[rnd * T:(rnd + 1) * T]
        if HALF <= rnd < HALF + RP:
            q = [pow((s[0] + rc[0]) % P, 3, P)] + [0] * (T - 1)
            rows.append((s, q))
            x7 = q[0] * q[0] % P * ((s[0] + rc[0]) % P) % P
            tot = (x7 + sum(s[1:])) % P
            s = [(tot + DIAG[0] * x7) % P] + [(tot + DIAG[i] * s[i]) % P for i in range(1, T)]
        else:
            q = [pow((s[i] + rc[i]) % P, 3, P) for i in range(T)]
            rows.append((s, q))
            s = m_ext([q[i] * q[i] % P * ((s[i] + rc[i]) % P) % P for i in range(T)])
    rows.append((s, [0] * T))
    return rows


def permute(state):
    return block_rows(state)[-1][0]


def hash_pair_words(left, right):
    """`hash_pair` on raw Montgomery words (what Merkle nodes and claims are): 8 + 8 words -> 8 words."""
    st = [int(w) * RINV % P for w in list(left) + list(right)] + [0] * 8
    return [v * R % P for v in permute(st)[:8]]


def build_p2_join() -> np.ndarray:
    b = CircuitBuilder((WA, WC, WD), (OUT_WORDS, WA), kind=KIND_P2_JOIN)
    code = lambda c, back=0: b.get(GROUP_CODE, c, back)
    S = lambda j, back=0: b.get(GROUP_DATA, j, back)
    Q = lambda j, back=0: b.get(GROUP_DATA, T + j, back)
    acc = lambda c, back=0: b.get(GROUP_ACCUM, c, back)
    out = lambda i: b.get_global(GLOBAL_OUT, i)
    one = b.const(1)
    active, first, body = code(0), code(1), code(2)
    cst = {v: b.const(v) for v in (2, 3, 4, 5, 6, 7)}

    def times(c, x):
        return x if c == 1 else b.mul(cst[c], x)

    def lin_m_ext(x):                                  # the 24 x 24 external matrix as expressions over 24 values
        y = []
        for blk in range(0, T, 4):
            for i in range(4):
                e = times(M4[i][0], x[blk])
                for j in range(1, 4):
                    e = b.add(e, times(M4[i][j], x[blk + j]))
                y.append(e)
        tot = []
        for i in range(4):
            e = y[i]
            for blk in range(4, T, 4):
                e = b.add(e, y[blk + i])
            tot.append(e)
        return [b.add(y[k], tot[k % 4]) for k in range(T)]

    def gated(chain, sel, constraints):
        inner = b.true()
        for c in constraints:
            inner = b.and_eqz(inner, c)
        return b.and_cond(chain, sel, inner)

    chain = b.true()
    # inputs
    chain = gated(chain, code(C_IN0), [b.sub(S(i), out(8 + i)) for i in range(16)] + [S(16 + i) for i in range(8)])
    chain = gated(chain, code(C_INP), [b.sub(S(i), out(i)) for i in range(8)] + [b.sub(S(8 + i), code(C_SIB + i)) for i in range(8)]
                  + [S(16 + i) for i in range(8)])
    # the cubes of this row's round
    u = [b.add(S(j), code(C_RC + j)) for j in range(T)]
    cube = lambda x: b.mul(b.mul(x, x), x)
    chain = gated(chain, code(C_FULLR), [b.sub(Q(j), cube(u[j])) for j in range(T)])
    chain = gated(chain, code(C_PARTR), [b.sub(Q(0), cube(u[0]))])
    # links: this row's state from the previous row
    prev = [S(j, 1) for j in range(T)]
    chain = gated(chain, code(C_LIN), [b.sub(S(k), e) for k, e in enumerate(lin_m_ext(prev))])
    x7 = [b.mul(b.mul(Q(j, 1), Q(j, 1)), b.add(S(j, 1), code(C_RC + j, 1))) for j in range(T)]
    chain = gated(chain, code(C_LF), [b.sub(S(k), e) for k, e in enumerate(lin_m_ext(x7))])
    tot = x7[0]
    for j in range(1, T):
        tot = b.add(tot, prev[j])
    diag = [b.const(d) for d in DIAG]
    chain = gated(chain, code(C_LP), [b.sub(S(0), b.add(tot, b.mul(diag[0], x7[0])))]
                  + [b.sub(S(j), b.add(tot, b.mul(diag[j], prev[j]))) for j in range(1, T)])
    # block 0's output is the parent claim
    chain = gated(chain, code(C_BIND0), [b.sub(S(i), out(i)) for i in range(8)])

    # accum: one Fp4 running product of (mix + data column 0), SYN-AIR's argument
    nbeta = b.const(NBETA)
    m = [b.get_global(GLOBAL_MIX, i) for i in range(4)]
    term = [b.add(m[0], S(0)), m[1], m[2], m[3]]
    chain = gated(chain, first, [b.sub(acc(i), term[i]) for i in range(4)])
    pa = [acc(i, 1) for i in range(4)]
    mm = lambda i, j: b.mul(pa[i], term[j])
    pr = [b.add(mm(0, 0), b.mul(nbeta, b.add(b.add(mm(1, 3), mm(2, 2)), mm(3, 1)))),
          b.add(b.add(mm(0, 1), mm(1, 0)), b.mul(nbeta, b.add(mm(2, 3), mm(3, 2)))),
          b.add(b.add(b.add(mm(0, 2), mm(1, 1)), mm(2, 0)), b.mul(nbeta, mm(3, 3))),
          b.add(b.add(mm(0, 3), mm(1, 2)), b.add(mm(2, 1), mm(3, 0)))]
    chain = gated(chain, body, [b.sub(acc(i), pr[i]) for i in range(4)])
    # selector sanity (ungated)
    chain = b.and_eqz(chain, b.mul(active, b.sub(one, active)))
    chain = b.and_eqz(chain, b.mul(first, b.sub(one, first)))
    chain = b.and_eqz(chain, b.sub(b.sub(active, first), body))
    return b.finish(chain)


_cached = None


def p2_join_circuit() -> np.ndarray:
    global _cached
    if _cached is None:
        _cached = build_p2_join()
    return _cached.copy()


if __name__ == "__main__":      # python -m zeth_amd.circuits.p2_join out.desc   (blob for non-Python hosts, e.g. examples/seal_segments)
    import sys
    blob = p2_join_circuit()
    np.asarray(blob, dtype="<u4").tofile(sys.argv[1])
    print(f"{sys.argv[1]}: {blob.size} words")
