"""KECCAK-F: a constraint system that proves keccak-f[1600] permutations — the third circuit of SURVEY.md §8 row f4.

zeth's guest hashes through the keccak accelerator (the `Keccak calls` statistic /root/reference/run-parallel.sh:70 scrapes;
patched `tiny-keccak`, /root/reference/guests/stateless-client/Cargo.toml:39), whose batches upstream proves with
risc0-circuit-keccak 4.0.2 (un-vendored: /root/reference/Cargo.lock:5289) and attaches to the composite receipt as
assumption receipts.  That Zirgen-generated circuit cannot be obtained offline, but the FUNCTION it proves is public
(FIPS 202), so unlike SYN-AIR this circuit is not a stand-in for the computation: every active block of 25 trace rows is one
real keccak-f[1600] permutation, constrained bit by bit, and the witness is checked against `hashlib.sha3_256`
(tests/test_keccak_circuit.py).  Its layout is this repository's own, not upstream's (declared).

Trace (n rows, A = n - zk_cycles active, K = A // 25 permutations, row r of permutation p = 25 p + r):
  rows 0..23 of a block hold the state BEFORE round r and that round's intermediates, row 24 holds the output state.
  data (3840 bit columns; column = 64 * lane + z):
      A[x, y]  lanes  0..24 (lane = x + 5 y)   state
      T[x]     lanes 25..29                    A[x,0] ^ A[x,1] ^ A[x,2]
      C[x]     lanes 30..34                    T[x] ^ A[x,3] ^ A[x,4]                       (theta column parities)
      B[X, Y]  lanes 35..59                    rho / pi of theta's output: B[y, 2x+3y] = rotl(A[x,y] ^ D[x], r[x,y]),
                                               D[x] = C[x-1] ^ rotl(C[x+1], 1)
  code (15 columns, a function of (po2, zk_cycles) alone -> control root):
      0 active  1 first  2 body (the accum argument's selectors, as in SYN-AIR)
      3 round   (rows 0..23 of a block)        4 link  (rows 1..24: A follows from the previous row's B by chi + iota)
      5 input   (row 0 of a block: A is boolean)       6 final (row 24 of the LAST block: A is bound to `out`)
      7..13     the round constant of the PREVIOUS row's round at its 7 possible bit positions (0,1,3,7,15,31,63), on link rows
      14 bind   (row 0 of the LAST block: that permutation's INPUT state is bound to `out` too)
  accum: one Fp4 running product of (mix + data column 0), the same argument (and kernel) as SYN-AIR's.
Constraints (degree <= 5 with the selector): xor(a, b) = a + b - 2ab;
  round: T, C (3-way xors), B = xor(A', D) through the rho / pi wiring;
  link : A = B@1 ^ (~B@1[x+1] & B@1[x+2])  [^ rc on lane 0];   input: a (1 - a) = 0;   final: 100 16-bit limbs = out[0..100);
  bind : the 100 16-bit limbs of the last permutation's input state = out[100..200).
Globals: out = 200 words — the OUTPUT state of the last permutation (lane l, limb j at 4 l + j), then its INPUT state at 100 + 4 l + j —
mix = 4 words.
What a receipt says: "keccak-f(input) = output" for the (input, output) pair in `out`, i.e. in the claim.  Binding the output alone
(as this circuit did before round 4) said nothing: keccak-f is a bijection, every output has a preimage, so any `out` was provable.
The other K - 1 permutations of a segment are filler the claim does not mention (upstream's circuit folds all of them into a
transcript digest; here one receipt = one (input, output) claim — declared).
"""
from __future__ import annotations

import numpy as np

from .desc import GLOBAL_MIX, GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, P, CircuitBuilder

KIND_KECCAK_F = 2
ROUNDS, BLOCK_ROWS = 24, 25
WC, WD, WA, OUT_WORDS = 15, 60 * 64, 4, 200
LANE_A, LANE_T, LANE_C, LANE_B = 0, 25, 30, 35
RC_POS = (0, 1, 3, 7, 15, 31, 63)
NBETA = P - 11
M64 = (1 << 64) - 1


def rho_offsets():
    """r[x][y] by the FIPS 202 walk: (x, y) = (1, 0); for t in 0..23: r[x][y] = (t+1)(t+2)/2; (x, y) = (y, 2x + 3y)."""
    r = [[0] * 5 for _ in range(5)]
    x, y = 1, 0
    for t in range(24):
        r[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return r


def round_constants():
    """RC[i] from the degree-8 LFSR x^8 + x^6 + x^5 + x^4 + 1 (FIPS 202 algorithm 5)."""
    def rc_bit(t):
        if t % 255 == 0:
            return 1
        reg = 1
        for _ in range(t % 255):
            reg <<= 1
            if reg & 0x100:
                reg ^= 0x171
        return reg & 1
    out = []
    for i in range(ROUNDS):
        v = 0
        for j in range(7):
            if rc_bit(j + 7 * i):
                v |= 1 << ((1 << j) - 1)
        out.append(v)
    return out


RHO = rho_offsets()
RC = round_constants()


def rotl(v, k):
    k %= 64
    return ((v << k) | (v >> (64 - k))) & M64 if k else v


def keccak_round_rows(state):
    """One permutation, row by row: [(A[25], T[5], C[5], B[25])] for rounds 0..23 plus the output state (lanes as ints,
    lane = x + 5 y).  The plain-Python statement of what the witness generators (circuit.hip, oracle/keccak.c) must produce."""
    rows = []
    a = list(state)
    for rnd in range(ROUNDS):
        t = [a[x] ^ a[x + 5] ^ a[x + 10] for x in range(5)]
        c = [t[x] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rotl(c[(x + 1) % 5], 1) for x in range(5)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y] ^ d[x], RHO[x][y])
        rows.append((a, t, c, b))
        a = [b[x + 5 * y] ^ ((~b[(x + 1) % 5 + 5 * y]) & b[(x + 2) % 5 + 5 * y] & M64) for y in range(5) for x in range(5)]
        a[0] ^= RC[rnd]
    return rows, a


def keccak_f(state):
    return keccak_round_rows(state)[1]


def sha3_256_block(msg: bytes):
    """The padded single-block sponge input of SHA3-256 (rate 136 bytes) as 25 lanes; len(msg) < 136."""
    assert len(msg) < 136
    blk = bytearray(msg) + bytearray(136 - len(msg))
    blk[len(msg)] ^= 0x06
    blk[135] ^= 0x80
    blk += bytearray(64)
    return [int.from_bytes(blk[8 * i:8 * i + 8], "little") for i in range(25)]


def digest_of_state(state) -> bytes:
    return b"".join(int(v).to_bytes(8, "little") for v in state[:4])


def out_words(state, input_state=None):
    """The `out` globals (canonical values): the final state's 100 16-bit limbs (lane l, limb j at 4 l + j), then — given the
    last permutation's input state — its 100 limbs (the full 200-word `out` of the circuit)."""
    limbs = lambda st: [(int(st[l]) >> (16 * j)) & 0xFFFF for l in range(25) for j in range(4)]
    return limbs(state) + (limbs(input_state) if input_state is not None else [])


def build_keccak_f() -> np.ndarray:
    b = CircuitBuilder((WA, WC, WD), (OUT_WORDS, WA), kind=KIND_KECCAK_F)
    code = lambda c: b.get(GROUP_CODE, c, 0)
    bit = lambda lane, z, back=0: b.get(GROUP_DATA, 64 * lane + (z % 64), back)
    acc = lambda c, back=0: b.get(GROUP_ACCUM, c, back)
    one, two = b.const(1), b.const(2)
    active, first, body, rnd, link, inp, final = (code(i) for i in range(7))
    bind = code(14)
    rcb = {pos: code(7 + j) for j, pos in enumerate(RC_POS)}

    def xor(p, q):                                   # p + q - 2 p q
        return b.sub(b.add(p, q), b.mul(two, b.mul(p, q)))

    # ---- round rows: theta's parities and the rho / pi wiring ----
    inner = b.true()
    for x in range(5):
        for z in range(64):
            inner = b.and_eqz(inner, b.sub(bit(LANE_T + x, z), xor(xor(bit(x, z), bit(x + 5, z)), bit(x + 10, z))))
    for x in range(5):
        for z in range(64):
            inner = b.and_eqz(inner, b.sub(bit(LANE_C + x, z), xor(xor(bit(LANE_T + x, z), bit(x + 15, z)), bit(x + 20, z))))
    d_cache = {}

    def d_bit(x, z):
        key = (x, z % 64)
        if key not in d_cache:
            d_cache[key] = xor(bit(LANE_C + (x - 1) % 5, z), bit(LANE_C + (x + 1) % 5, z - 1))
        return d_cache[key]
    for x in range(5):
        for y in range(5):
            dst = LANE_B + y + 5 * ((2 * x + 3 * y) % 5)
            for z in range(64):
                src = z - RHO[x][y]                  # B[..][z] = A'[x, y][z - r]
                inner = b.and_eqz(inner, b.sub(bit(dst, z), xor(bit(x + 5 * y, src), d_bit(x, src))))
    chain = b.and_cond(b.true(), rnd, inner)

    # ---- link rows: chi + iota from the previous row's B ----
    inner = b.true()
    for y in range(5):
        for x in range(5):
            for z in range(64):
                b0 = bit(LANE_B + x + 5 * y, z, 1)
                b1 = bit(LANE_B + (x + 1) % 5 + 5 * y, z, 1)
                b2 = bit(LANE_B + (x + 2) % 5 + 5 * y, z, 1)
                v = xor(b0, b.mul(b.sub(one, b1), b2))
                if x == 0 and y == 0 and z in rcb:
                    v = xor(v, rcb[z])
                inner = b.and_eqz(inner, b.sub(bit(x + 5 * y, z), v))
    chain = b.and_cond(chain, link, inner)

    # ---- input rows: the state bits are bits (everything downstream is then boolean by construction) ----
    inner = b.true()
    for lane in range(25):
        for z in range(64):
            a = bit(lane, z)
            inner = b.and_eqz(inner, b.mul(a, b.sub(one, a)))
    chain = b.and_cond(chain, inp, inner)

    # ---- final row: the output state, as 16-bit limbs, is the public output ----
    inner = b.true()
    pow2 = [b.const(1 << i) for i in range(16)]
    for lane in range(25):
        for j in range(4):
            s = bit(lane, 16 * j)
            for i in range(1, 16):
                s = b.add(s, b.mul(pow2[i], bit(lane, 16 * j + i)))
            inner = b.and_eqz(inner, b.sub(s, b.get_global(GLOBAL_OUT, 4 * lane + j)))
    chain = b.and_cond(chain, final, inner)

    # ---- bind row (row 0 of the last block): that permutation's INPUT state, as 16-bit limbs, is public too ----
    inner = b.true()
    for lane in range(25):
        for j in range(4):
            s = bit(lane, 16 * j)
            for i in range(1, 16):
                s = b.add(s, b.mul(pow2[i], bit(lane, 16 * j + i)))
            inner = b.and_eqz(inner, b.sub(s, b.get_global(GLOBAL_OUT, 100 + 4 * lane + j)))
    chain = b.and_cond(chain, bind, inner)

    # ---- accum: one Fp4 running product of (mix + data column 0), SYN-AIR's argument ----
    nbeta = b.const(NBETA)
    m = [b.get_global(GLOBAL_MIX, i) for i in range(4)]
    term = [b.add(m[0], bit(0, 0)), m[1], m[2], m[3]]
    inner = b.true()
    for i in range(4):
        inner = b.and_eqz(inner, b.sub(acc(i), term[i]))
    chain = b.and_cond(chain, first, inner)
    prev = [acc(i, 1) for i in range(4)]
    mm = lambda i, j: b.mul(prev[i], term[j])
    pr = [b.add(mm(0, 0), b.mul(nbeta, b.add(b.add(mm(1, 3), mm(2, 2)), mm(3, 1)))),
          b.add(b.add(mm(0, 1), mm(1, 0)), b.mul(nbeta, b.add(mm(2, 3), mm(3, 2)))),
          b.add(b.add(b.add(mm(0, 2), mm(1, 1)), mm(2, 0)), b.mul(nbeta, mm(3, 3))),
          b.add(b.add(mm(0, 3), mm(1, 2)), b.add(mm(2, 1), mm(3, 0)))]
    inner = b.true()
    for i in range(4):
        inner = b.and_eqz(inner, b.sub(acc(i), pr[i]))
    chain = b.and_cond(chain, body, inner)

    # selector sanity (ungated)
    chain = b.and_eqz(chain, b.mul(active, b.sub(one, active)))
    chain = b.and_eqz(chain, b.mul(first, b.sub(one, first)))
    chain = b.and_eqz(chain, b.sub(b.sub(active, first), body))
    return b.finish(chain)


_cached = None


def keccak_f_circuit() -> np.ndarray:
    """The KECCAK-F circuit description (built once per process: ~60 k steps)."""
    global _cached
    if _cached is None:
        _cached = build_keccak_f()
    return _cached.copy()


if __name__ == "__main__":      # python -m zeth_amd.circuits.keccak_f out.desc   (blob for non-Python hosts, e.g. examples/seal_segments)
    import sys
    blob = keccak_f_circuit()
    np.asarray(blob, dtype="<u4").tofile(sys.argv[1])
    print(f"{sys.argv[1]}: {blob.size} words")
