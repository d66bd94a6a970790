"""SYN-AIR: the declared-synthetic stand-in circuit for the (un-obtainable) rv32im constraint system.

The real rv32im-v2 constraint polynomial, tap set and witness generator are Zirgen-generated artefacts of
risc0-circuit-rv32im 4.0.2 (/root/reference/Cargo.lock:5320) and are not available offline (SURVEY.md §7
hard part 1).  SYN-AIR has the same *shape* — three register groups (accum, code, data), back-0/back-1 taps,
degree-5 constraints gated by code selectors, an accum group that is a grand product over Fp4 driven by
Fiat-Shamir `mix` globals — so every HAL op and the whole DEEP-ALI + FRI protocol is exercised, and a
random trace is made satisfying by construction.  The witness definition lives in DESIGN.md §SYN-AIR and is
implemented twice: oracle/circuit.c (CPU) and zeth_amd/csrc/circuit.hip (HIP, `k_syn_*`).

Columns (n rows, A = n - zk_cycles active rows):
  code : c0 active, c1 first, c2 body (active & !first), c3 row index, c4 last (row A-1), c5.. public noise
  data : triples (3j, 3j+1, 3j+2 = product) for j < T = (wd-2)//3; col wd-2 = d0*d1*d3*d4; col wd-1 = running
         sum s: s[0] = d0[0], s[r] = s[r-1] + d0[r] + c3[r]*d1[r]
  accum: k = wa/4 Fp4 columns, a_e[r] = prod_{r' <= r} (mix_e + d_{e mod wd}[r'])
Globals: out = (s[A-1], 0, 0, 0, pub_0 .. pub_{n_pub-1}); mix = k Fp4 challenges (4k words).
Public inputs (n_pub > 0, used by the join circuit SYN-J): word k is placed in row 0 of data column 3k (the x cell of
triple k; column 0 also feeds the running sum, so out[0] depends on pub_0) and bound to out[4 + k] by a `first`-gated
constraint — the verifier, which reads `out` from the seal header, thereby learns which inputs the witness used.
The code group depends only on (shape, po2, zk_cycles): its committed Merkle root is the control root (the control-ID
analogue) that the verifier compares against.
"""
from __future__ import annotations

import numpy as np

from .desc import GLOBAL_MIX, GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, P, CircuitBuilder

NBETA = P - 11


def build_syn_air(wc: int = 16, wd: int = 208, wa: int = 32, n_pub: int = 0) -> np.ndarray:
    assert wc >= 5 and wd >= 8 and wa >= 4 and wa % 4 == 0
    assert 0 <= n_pub <= (wd - 2) // 3, "one triple per public input word"
    b = CircuitBuilder((wa, wc, wd), (4 + n_pub, wa))
    code = lambda c, back=0: b.get(GROUP_CODE, c, back)
    data = lambda c, back=0: b.get(GROUP_DATA, c, back)
    acc = lambda c, back=0: b.get(GROUP_ACCUM, c, back)
    one = b.const(1)
    nbeta = b.const(NBETA)
    active, first, body, rowidx, last = (code(i) for i in range(5))
    T = (wd - 2) // 3
    s_col = wd - 1

    # (A) multiplicative triples + the degree-4 product, gated by `active` (max degree 5)
    inner = b.true()
    for j in range(T):
        inner = b.and_eqz(inner, b.sub(b.mul(data(3 * j), data(3 * j + 1)), data(3 * j + 2)))
    prod4 = b.mul(b.mul(data(0), data(1)), b.mul(data(3), data(4)))
    inner = b.and_eqz(inner, b.sub(prod4, data(wd - 2)))
    chain = b.and_cond(b.true(), active, inner)

    def term(e):          # mix_e + d (as Fp4 components)
        m = [b.get_global(GLOBAL_MIX, 4 * e + i) for i in range(4)]
        return [b.add(m[0], data(e % wd)), m[1], m[2], m[3]]

    def ext_mul(x, y):    # Fp4 product written out in Fp steps (x^4 = -11)
        m = lambda i, j: b.mul(x[i], y[j])
        c0 = b.add(m(0, 0), b.mul(nbeta, b.add(b.add(m(1, 3), m(2, 2)), m(3, 1))))
        c1 = b.add(b.add(m(0, 1), m(1, 0)), b.mul(nbeta, b.add(m(2, 3), m(3, 2))))
        c2 = b.add(b.add(b.add(m(0, 2), m(1, 1)), m(2, 0)), b.mul(nbeta, m(3, 3)))
        c3 = b.add(b.add(m(0, 3), m(1, 2)), b.add(m(2, 1), m(3, 0)))
        return [c0, c1, c2, c3]

    # (first) s = d0 ; a_e = term_e
    inner = b.and_eqz(b.true(), b.sub(data(s_col), data(0)))
    for e in range(wa // 4):
        t = term(e)
        for i in range(4):
            inner = b.and_eqz(inner, b.sub(acc(4 * e + i), t[i]))
    # (first) public inputs: d_{3k}[0] = out[4 + k]
    for k in range(n_pub):
        inner = b.and_eqz(inner, b.sub(data(3 * k), b.get_global(GLOBAL_OUT, 4 + k)))
    chain = b.and_cond(chain, first, inner)

    # (body) s = s@1 + d0 + row*d1 ; a_e = a_e@1 * term_e
    rhs = b.add(b.add(data(s_col, 1), data(0)), b.mul(rowidx, data(1)))
    inner = b.and_eqz(b.true(), b.sub(data(s_col), rhs))
    for e in range(wa // 4):
        prev = [acc(4 * e + i, 1) for i in range(4)]
        pr = ext_mul(prev, term(e))
        for i in range(4):
            inner = b.and_eqz(inner, b.sub(acc(4 * e + i), pr[i]))
    chain = b.and_cond(chain, body, inner)

    # (last) s = out[0]
    inner = b.and_eqz(b.true(), b.sub(data(s_col), b.get_global(GLOBAL_OUT, 0)))
    chain = b.and_cond(chain, last, inner)

    # selector sanity (ungated)
    chain = b.and_eqz(chain, b.mul(active, b.sub(one, active)))
    chain = b.and_eqz(chain, b.mul(first, b.sub(one, first)))
    chain = b.and_eqz(chain, b.sub(b.sub(active, first), body))
    return b.finish(chain)


# named shapes
def syn_a() -> np.ndarray:
    """SYN-A: W_code 16, W_data 208, W_accum 32."""
    return build_syn_air(16, 208, 32)


def syn_tiny() -> np.ndarray:
    return build_syn_air(6, 11, 4)


def syn_small() -> np.ndarray:
    return build_syn_air(8, 20, 8)


def syn_chain() -> np.ndarray:
    """SYN-C: SYN-A with ONE public input — the segment's PRE-STATE.  out = (post, 0, 0, 0, pre): `pre` sits in row 0 of data
    column 0 and is bound to out[4] by the `first`-gated public-input constraint; the running sum starts there (s[0] = d0[0] =
    pre) and ends in out[0] = post = pre + (the segment's own contribution).  Two consecutive segments of a session are
    CONTINUOUS when post(i) = pre(i + 1): what upstream's `ReceiptClaim{pre, post}` carries and `CompositeReceipt::verify_integrity`
    checks (risc0-zkvm 3.0.3, un-vendored; reached from /root/reference/crates/host/src/bin/cli.rs:103).  Same code group as
    SYN-A (the control columns do not depend on the number of public inputs), hence the same control roots."""
    return build_syn_air(16, 208, 32, n_pub=1)


def syn_chain_small() -> np.ndarray:
    """the same at SYN-small's widths (tests)"""
    return build_syn_air(8, 20, 8, n_pub=1)


CHAIN_PRE, CHAIN_POST = 4, 0      # positions of the pre / post state words in a SYN-C seal's `out` header

# SYN-S: what a segment of a SESSION publishes, bound word for word by the public-input constraints — what upstream's
# `ReceiptClaim` holds per segment (risc0-zkvm 3.0.3 receipt_claim.rs, recalled): pre / post state, the exit code as its
# (system, user) pair, and the digest of the output (journal) as sixteen 16-bit limbs (a SHA-256 word is not a field element).
SESSION_EXIT_SYS, SESSION_EXIT_USER, SESSION_JOURNAL, SESSION_JOURNAL_LIMBS = 5, 6, 7, 16
SESSION_PUB_WORDS = 1 + 2 + SESSION_JOURNAL_LIMBS                  # pre, exit (sys, user), journal digest limbs
SESSION_OUT_WORDS = 4 + SESSION_PUB_WORDS


def syn_session() -> np.ndarray:
    """SYN-S: SYN-C whose segments also publish their EXIT CODE and OUTPUT DIGEST: out = (post, 0, 0, 0, pre, exit_sys, exit_user,
    j_0 .. j_15).  Every word after the first four is a public input bound to a witness cell by a `first`-gated constraint, so a
    holder of the receipt cannot change it without breaking the seal: a session cut short ends in a segment that says SystemSplit
    (2, 0), not Halted(0) (0, 0), and a rewritten journal no longer hashes to the limbs the last seal carries.  What
    `receipt.verify(image_id)` + the journal comparison check (/root/reference/crates/host/src/bin/cli.rs:103-107) through
    upstream's `CompositeReceipt::verify_integrity` + exit-code check.  The executor fixes these words before any segment is proven
    (as upstream's does); the circuit has no notion of halting of its own (declared: SYN-AIR's computation is a stand-in).  Same
    code group as SYN-A, hence the same control roots."""
    return build_syn_air(16, 208, 32, n_pub=SESSION_PUB_WORDS)


def syn_session_small() -> np.ndarray:
    """the same at small widths (tests): 20 triples hold the 19 public words"""
    return build_syn_air(8, 62, 8, n_pub=SESSION_PUB_WORDS)
JOIN_PUB_WORDS = 16      # two child claim digests (8 words each)


def syn_join() -> np.ndarray:
    """SYN-J: recursion-like widths (W ~ 16/128/16) with 16 public input words = the claim
    digests of the two receipts a join combines, bound to its `out` globals by constraints.  Declared synthetic: the real
    join additionally VERIFIES its children in-circuit (risc0-circuit-recursion 4.0.2, un-vendored)."""
    return build_syn_air(16, 128, 16, n_pub=JOIN_PUB_WORDS)


if __name__ == "__main__":      # python -m zeth_amd.circuits.syn_air syn_a out.desc  (blob for non-Python hosts)
    import sys
    shape, path = sys.argv[1], sys.argv[2]
    blob = {"syn_a": syn_a, "syn_tiny": syn_tiny, "syn_small": syn_small, "syn_join": syn_join, "syn_chain": syn_chain, "syn_session": syn_session}[shape]()
    np.asarray(blob, dtype="<u4").tofile(path)
    print(f"{path}: {blob.size} words")
