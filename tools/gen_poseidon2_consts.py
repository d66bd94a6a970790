#!/usr/bin/env python3
"""Generate include/zkh_poseidon2_consts.h — the Poseidon2 (BabyBear, t = 24, x^7, R_F = 8, R_P = 21) constant tables —
by running the PUBLISHED parameter-generation procedure, not by copying a table.

The tables upstream ships (risc0-zkp 3.0.2 src/core/hash/poseidon2/consts.rs: ROUND_CONSTANTS[24*29],
M_INT_DIAG_HZN[24]; un-vendored, /root/reference/Cargo.lock:5393, not fetchable here) are, by that file's own account, the
output of the Horizen Labs Poseidon2 parameter script for this instance.  That procedure is public (Poseidon paper,
appendix on the Grain LFSR; Poseidon2 paper, section 5.3 for the internal matrix) and is restated here from its description:

  * an 80-bit Grain LFSR seeded with (field = 1, sbox = 0, n = 31, t = 24, R_F = 8, R_P = 21, thirty ones), 160 warm-up
    steps, then the self-shrinking output rule (a bit is kept only when the bit before it is 1);
  * round constants: R_F * t + R_P = 213 field elements, 31 bits each, most significant bit first, resampled while >= P,
    in the order: 4 x 24 for the first full rounds, 21 for the partial rounds (cell 0 only), 4 x 24 for the last full rounds;
  * internal matrix M_I = (J - I) + diag(mu): candidates mu (24 x 31 bits from the same stream, reduced mod P, no
    resampling) are drawn until, for every i in 1..2t, the characteristic polynomial of M_I^i is irreducible of degree t
    (the script's check_minpoly_condition).  M_I x = sum(x) + (mu - 1) * x, so the stored diagonal is mu - 1.

What pins this to upstream, short of the crate itself: the values this procedure produces agree with every value of the
published instance that was on record here BEFORE the procedure was written — the first eight external round constants
(0x0fa20c37, 0x0795bb97, 0x12c60b9c, 0x0eabd88e, ...), the first four internal ones (0x1da78ec2, 0x730b0924, ...) and all
24 diagonal entries (round 1 carried them as "recalled, unverified"; they turn out to be the FIFTH candidate of the search:
the first four fail the irreducibility test).  Those anchors are asserted below.  Status therefore: DERIVED — not a
placeholder any more, not yet compared word for word with consts.rs.  Constants stay data: zkh_poseidon2_set_constants /
zko_poseidon2_set_constants swap them at run time.

    python tools/gen_poseidon2_consts.py          # ~20 s (the irreducibility tests are plain Python)
"""
import os

P = 2013265921
T, RF, RP, NBITS = 24, 8, 21, 31


# ---------------------------------------------------------------- Grain LFSR
def grain_stream():
    bits = [int(c) for c in (bin(1)[2:].zfill(2) + bin(0)[2:].zfill(4) + bin(NBITS)[2:].zfill(12) + bin(T)[2:].zfill(12)
                             + bin(RF)[2:].zfill(10) + bin(RP)[2:].zfill(10))] + [1] * 30
    assert len(bits) == 80

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb
    for _ in range(160):
        step()

    def next_bit():
        nb = step()
        while nb == 0:          # self-shrinking: a 0 discards the bit that follows it
            step()
            nb = step()
        return step()

    def random_bits(n):
        v = 0
        for _ in range(n):
            v = (v << 1) | next_bit()
        return v
    return random_bits


# ---------------------------------------------------------------- polynomials / matrices over F_P (plain lists)
def mat_mul(a, b):
    n = len(a)
    bt = list(zip(*b))
    return [[sum(x * y for x, y in zip(row, col)) % P for col in bt] for row in a]


def charpoly(m):
    """Monic characteristic polynomial (coefficients low -> high) through the Hessenberg form."""
    n = len(m)
    a = [row[:] for row in m]
    for j in range(n - 2):
        piv = next((i for i in range(j + 1, n) if a[i][j]), None)
        if piv is None:
            continue
        if piv != j + 1:
            a[piv], a[j + 1] = a[j + 1], a[piv]
            for row in a:
                row[piv], row[j + 1] = row[j + 1], row[piv]
        inv = pow(a[j + 1][j], P - 2, P)
        for i in range(j + 2, n):
            if a[i][j]:
                f = a[i][j] * inv % P
                a[i] = [(x - f * y) % P for x, y in zip(a[i], a[j + 1])]
                for row in a:
                    row[j + 1] = (row[j + 1] + f * row[i]) % P
    polys = [[1]]
    for k in range(1, n + 1):
        prev = polys[k - 1]
        cur = [0] + prev                                            # x * p_{k-1}
        h = a[k - 1][k - 1]
        for i, c in enumerate(prev):
            cur[i] = (cur[i] - h * c) % P
        prod = 1
        for j in range(k - 2, -1, -1):                              # subdiagonal products h_{j+1,j} ... h_{k-1,k-2}
            prod = prod * a[j + 1][j] % P
            f = a[j][k - 1] * prod % P
            if f:
                for i, c in enumerate(polys[j]):
                    cur[i] = (cur[i] - f * c) % P
        polys.append(cur)
    return polys[n]


def poly_mulmod(a, b, f):
    n = len(f) - 1
    r = [0] * (2 * n - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % P
    for d in range(2 * n - 2, n - 1, -1):                           # f is monic
        c = r[d]
        if c:
            for j in range(n + 1):
                r[d - n + j] = (r[d - n + j] - c * f[j]) % P
    return r[:n]


def poly_gcd_is_one(a, f):
    a, b = f[:], a[:]
    while b and any(b):
        while b and b[-1] == 0:
            b.pop()
        if not b:
            break
        inv = pow(b[-1], P - 2, P)
        while len(a) >= len(b) and any(a):
            while a and a[-1] == 0:
                a.pop()
            if len(a) < len(b):
                break
            c = a[-1] * inv % P
            s = len(a) - len(b)
            for j, y in enumerate(b):
                a[s + j] = (a[s + j] - c * y) % P
        a, b = b, a
    while a and a[-1] == 0:
        a.pop()
    return len(a) == 1


def irreducible(f):
    """Rabin's test for a monic polynomial of degree n over F_P."""
    n = len(f) - 1
    x = [0, 1] + [0] * (n - 2)
    xp, base, e = [1] + [0] * (n - 1), x[:], P                       # x^P mod f
    while e:
        if e & 1:
            xp = poly_mulmod(xp, base, f)
        base = poly_mulmod(base, base, f)
        e >>= 1
    frob = [[1] + [0] * (n - 1)]                                    # images (x^j)^P of the basis: Frobenius as a matrix
    for _ in range(1, n):
        frob.append(poly_mulmod(frob[-1], xp, f))

    def apply(v):
        out = [0] * n
        for c, img in zip(v, frob):
            if c:
                for i, y in enumerate(img):
                    out[i] = (out[i] + c * y) % P
        return out
    primes = [q for q in range(2, n + 1) if n % q == 0 and all(q % r for r in range(2, q))]
    want = {n // q for q in primes}
    h = x[:]
    for k in range(1, n + 1):
        h = apply(h)                                                # x^(P^k) mod f
        if k in want:
            d = h[:]
            d[1] = (d[1] - 1) % P
            if not poly_gcd_is_one(d, f):
                return False
    return h == x


def minpoly_condition(mu):
    m = [[(mu[i] if i == j else 1) % P for j in range(T)] for i in range(T)]      # (J - I) + diag(mu)
    cur = m
    for i in range(1, 2 * T + 1):
        if not irreducible(charpoly(cur)):
            return False, i
        cur = mat_mul(m, cur)
    return True, 2 * T


# ---------------------------------------------------------------- run the procedure
def generate(verbose=True):
    rnd = grain_stream()
    consts = []
    while len(consts) < RF * T + RP:
        v = rnd(NBITS)
        while v >= P:
            v = rnd(NBITS)
        consts.append(v)
    attempt = 0
    while True:
        attempt += 1
        mu = [rnd(NBITS) % P for _ in range(T)]
        ok, at = minpoly_condition(mu)
        if verbose:
            print(f"internal matrix candidate {attempt}: " + ("accepted" if ok else f"rejected (M^{at} has a reducible characteristic polynomial)"))
        if ok:
            break
    diag = [(m - 1) % P for m in mu]
    rc = [0] * (T * (RF + RP))
    half = RF // 2
    for r in range(half):
        rc[r * T:(r + 1) * T] = consts[r * T:(r + 1) * T]
    for r in range(RP):
        rc[(half + r) * T] = consts[half * T + r]
    for r in range(half):
        rc[(half + RP + r) * T:(half + RP + r + 1) * T] = consts[half * T + RP + r * T: half * T + RP + (r + 1) * T]
    return rc, diag, attempt


ANCHOR_EXTERNAL = [0x0fa20c37, 0x0795bb97, 0x12c60b9c, 0x0eabd88e, 0x096485ca, 0x07093527, 0x1b1d4e50, 0x30a01ace]
ANCHOR_INTERNAL = [0x1da78ec2, 0x730b0924, 0x3eb56cf3, 0x5bd93073]
ANCHOR_DIAG = [0x409133f0, 0x1667a8a1, 0x06a6c7b6, 0x6f53160e, 0x273b11d1, 0x03176c5d, 0x72f9bbf9, 0x73ceba91,
               0x5cdef81d, 0x01393285, 0x46daee06, 0x065d7ba6, 0x52d72d6f, 0x05dd05e0, 0x3bab4b63, 0x6ada3842,
               0x2fc5fbec, 0x770d61b0, 0x5715aae9, 0x03ef0e90, 0x75b6c770, 0x242adf5f, 0x00d0ca4c, 0x36c0e388]


def main():
    rc, diag, attempt = generate()
    assert rc[:8] == ANCHOR_EXTERNAL, "first external round constants differ from the published instance"
    assert [rc[(4 + r) * T] for r in range(4)] == ANCHOR_INTERNAL, "first internal round constants differ from the published instance"
    assert diag == ANCHOR_DIAG and attempt == 5, "internal diagonal differs from the published instance"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "zkh_poseidon2_consts.h")
    with open(out, "w") as f:
        f.write("/* GENERATED by tools/gen_poseidon2_consts.py — Poseidon2 (BabyBear, t = 24, x^7, R_F = 8, R_P = 21) tables DERIVED by\n"
                " * running the published parameter-generation procedure (Grain LFSR + the internal matrix's irreducibility search;\n"
                " * see that file), which reproduces every value of the published instance on record here.  Not yet compared word for\n"
                " * word with risc0-zkp 3.0.2 src/core/hash/poseidon2/consts.rs (un-vendored; Cargo.lock:5393), which they stand in\n"
                " * for as data.  Canonical (non-Montgomery) residues mod P = 2013265921; RC[round * 24 + cell], partial rounds\n"
                " * (4..24) only have cell 0. */\n"
                "#ifndef ZKH_POSEIDON2_CONSTS_H\n#define ZKH_POSEIDON2_CONSTS_H\n#include <stdint.h>\n"
                "#define ZKH_P2_CELLS 24\n#define ZKH_P2_RATE 16\n#define ZKH_P2_OUT 8\n"
                "#define ZKH_P2_ROUNDS_HALF_FULL 4\n#define ZKH_P2_ROUNDS_PARTIAL 21\n#define ZKH_P2_ROUNDS 29\n"
                "#define ZKH_P2_CONSTS_ARE_PLACEHOLDER 0\n"
                "#define ZKH_P2_CONSTS_ARE_DERIVED 1      /* from the published procedure; 0 once checked against consts.rs itself */\n")
        f.write("static const uint32_t ZKH_P2_M_INT_DIAG[24] = {\n")
        for i in range(0, 24, 8):
            f.write("    " + ", ".join("0x%08xu" % d for d in diag[i:i + 8]) + ",\n")
        f.write("};\nstatic const uint32_t ZKH_P2_ROUND_CONSTANTS[24 * 29] = {\n")
        for i in range(0, 24 * 29, 8):
            f.write("    " + ", ".join("0x%08xu" % d for d in rc[i:i + 8]) + ",\n")
        f.write("};\n#endif\n")
    print("wrote", os.path.normpath(out))
    # the same tables as a Python module INSIDE the package (circuits/p2_join.py, circuits/recursion.py put the constants into
    # their code groups): an installed zeth_amd must not depend on the repository's include/ directory being next to it
    py = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zeth_amd", "circuits", "poseidon2_consts.py")
    with open(py, "w") as f:
        f.write('"""GENERATED by tools/gen_poseidon2_consts.py together with include/zkh_poseidon2_consts.h (same run, same values; a CPU test\n'
                'compares the two word for word): Poseidon2 (BabyBear, t = 24, x^7, R_F = 8, R_P = 21) tables, canonical residues.\n'
                'ROUND_CONSTANTS[round * 24 + cell]; partial rounds (4..24) only have cell 0.  Provenance: DERIVED (see the header)."""\n')
        f.write("M_INT_DIAG = [\n")
        for i in range(0, 24, 8):
            f.write("    " + ", ".join("0x%08x" % d for d in diag[i:i + 8]) + ",\n")
        f.write("]\nROUND_CONSTANTS = [\n")
        for i in range(0, 24 * 29, 8):
            f.write("    " + ", ".join("0x%08x" % d for d in rc[i:i + 8]) + ",\n")
        f.write("]\n")
    print("wrote", os.path.normpath(py))


if __name__ == "__main__":
    main()
