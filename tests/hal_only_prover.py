"""A segment prover that touches the GPU ONLY through the 1:1 `trait Hal` / `CircuitHal` methods, in upstream's literal order.

What a Rust `risc0_zkp::prove::Prover<HipHal>` (the generic prover behind `default_prover().prove`,
/root/reference/crates/host/src/lib.rs:137; risc0-zkp 3.0.2 src/prove/{prover,poly_group,merkle,fri}.rs, un-vendored:
/root/reference/Cargo.lock:5393) would do with an `impl Hal for HipHal` bound to include/zkhal.h: no fused entry point of the
library is used —

  commit_group : eltwise_copy_elem, batch_interpolate_ntt, zk_shift                       (three calls, not one)
  PolyGroup    : batch_expand_into_evaluate_ntt, batch_bit_reverse of the COEFFICIENTS, hash_rows, one hash_fold PER LAYER
  finalize     : eval_check, batch_interpolate_ntt, natural-order batch_evaluate_any, mix_poly_coeffs,
                 combos_prepare with upstream's argument list, combos_divide PER COMBO, eltwise_sum_extelem
  fri_prove    : per round batch_expand_into_evaluate_ntt, hash_rows / hash_fold, fri_fold
  queries      : per query and per tree gather_sample + the sibling digests read one by one (MerkleTreeProver::prove)

while the library's own prover (csrc/prover.hip) fuses, batches and reorders.  tests/test_seal_stages_gpu.py requires the two
seals to be byte-identical.  Host-side arithmetic (Fiat-Shamir sponge, poly_interpolate, challenge powers) is plain Python
over canonical residues; the Poseidon2 permutation is the library's HOST entry point zkh_poseidon2_mix_host.  The oracle is
not imported here.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from zeth_amd import hal as H
from zeth_amd.circuits.desc import Circuit as Desc

P = H.P
R = (1 << 32) % P
RINV = pow(R, -1, P)
INV_RATE, QUERIES, FRI_FOLD, FRI_MIN_DEGREE, CHECK_SIZE, EXT = 4, 50, 16, 256, 16, 4
ROU_REV = [pow(pow(137, 1 << (27 - k), P), -1, P) if k else 1 for k in range(28)]     # canonical w_{2^k}^-1
NBETA = P - 11


def enc(x: int) -> int:
    return (x % P) * R % P


def dec(w: int) -> int:
    return int(w) * RINV % P


# ---- Fp4 = Fp[x] / (x^4 + 11) over canonical residues ----
def e_mul(a, b):
    c = [0] * 7
    for i in range(4):
        for j in range(4):
            c[i + j] += a[i] * b[j]
    return tuple((c[k] + NBETA * c[k + 4]) % P if k < 3 else c[k] % P for k in range(4))


def e_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def e_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def e_scale(a, s):
    return tuple(x * s % P for x in a)


def e_pow(a, e):
    r = (1, 0, 0, 0)
    while e:
        if e & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        e >>= 1
    return r


def e_inv(a):
    a0, a1, a2, a3 = a
    b0 = (a0 * a0 + 11 * (2 * a1 * a3 - a2 * a2)) % P
    b2 = (2 * a0 * a2 - a1 * a1 + 11 * a3 * a3) % P
    ic = pow((b0 * b0 + 11 * b2 * b2) % P, -1, P)
    return e_mul((a0, (-a1) % P, a2, (-a3) % P), (b0 * ic % P, 0, (-b2 * ic) % P, 0))


def e_words(a) -> List[int]:
    return [enc(x) for x in a]


def poly_interpolate(xs, fx):
    """core/poly.rs poly_interpolate: coefficients of the degree < len polynomial through (xs, fx)."""
    size = len(xs)
    if size == 1:
        return [fx[0]]
    out = [(0, 0, 0, 0)] * size
    for i in range(size):
        poly = [(1, 0, 0, 0)]
        for j in range(size):
            if j == i:
                continue
            poly = poly + [(0, 0, 0, 0)]
            for k in range(len(poly) - 1, 0, -1):
                poly[k] = e_sub(poly[k - 1], e_mul(poly[k], xs[j]))
            poly[0] = e_sub((0, 0, 0, 0), e_mul(poly[0], xs[j]))
        d = (0, 0, 0, 0)
        for k in range(len(poly) - 1, -1, -1):
            d = e_add(e_mul(d, xs[i]), poly[k])
        m = e_mul(fx[i], e_inv(d))
        out = [e_add(o, e_mul(pk, m)) for o, pk in zip(out, poly)]
    return out


# ---- Poseidon2 sponge + Fiat-Shamir RNG on the host (core/hash/poseidon2/{mod,rng}.rs) ----
def _mix(state: np.ndarray) -> None:
    H._check(H._lib.zkh_poseidon2_mix_host(None, None, state.ctypes.data_as(C.POINTER(C.c_uint32)), 1))


def hash_elems(words: Sequence[int]) -> np.ndarray:
    s = np.zeros(24, dtype=np.uint32)
    w = np.asarray(words, dtype=np.uint32)
    used = 0
    for i in range(w.size):
        s[used] = w[i]
        used += 1
        if used == 16:
            _mix(s)
            used = 0
    if used or w.size == 0:
        s[used:16] = 0
        _mix(s)
    return s[:8].copy()


class Rng:
    def __init__(self):
        self.cells = np.zeros(24, dtype=np.uint32)
        self.used = 0

    def mix(self, digest) -> None:
        if self.used:
            _mix(self.cells)
            self.used = 0
        for i in range(8):
            self.cells[i] = (int(self.cells[i]) + int(digest[i])) % P
        _mix(self.cells)

    def elem(self) -> int:
        if self.used == 16:
            _mix(self.cells)
            self.used = 0
        v = int(self.cells[self.used])
        self.used += 1
        return v

    def bits(self, bits: int) -> int:
        val = dec(self.elem())
        for _ in range(3):
            nv = dec(self.elem())
            if val == 0:
                val = nv
        return val & ((1 << bits) - 1)

    def ext(self):
        return tuple(dec(self.elem()) for _ in range(4))


class Iop:
    def __init__(self):
        self.proof: List[int] = []
        self.rng = Rng()

    def write(self, words) -> None:
        self.proof.extend(int(x) for x in np.asarray(words, dtype=np.uint32).reshape(-1))

    def commit(self, digest) -> None:
        self.rng.mix(digest)


def log2(x: int) -> int:
    return x.bit_length() - 1


class MerkleTreeProver:
    """prove/merkle.rs: hash_rows, then ONE hash_fold per layer; prove(idx) = gather_sample + sibling digests one by one."""

    def __init__(self, hal: "H.HipHal", matrix: "H.Buffer", rows: int, cols: int):
        self.hal, self.matrix, self.rows, self.cols = hal, matrix, rows, cols
        self.layers = log2(rows)
        self.top_layer = 0
        for i in range(1, self.layers):
            if (1 << i) > QUERIES:
                break
            self.top_layer = i
        self.top_size = 1 << self.top_layer
        self.nodes = hal.alloc_digest("nodes", rows * 2)
        hal.hash_rows(self.nodes.slice(rows * 8, rows * 8), matrix)
        for i in range(self.layers - 1, -1, -1):
            size = 1 << i
            hal.hash_fold(self.nodes, size * 2, size)
        self.root = self.nodes.slice(8, 8).to_vec()

    def commit(self, iop: Iop) -> None:
        iop.write(self.nodes.slice(self.top_size * 8, self.top_size * 8).to_vec())
        iop.commit(self.root)

    def prove(self, iop: Iop, idx: int) -> None:
        col = self.hal.alloc_elem("merkle column", self.cols)
        self.hal.gather_sample(col, self.matrix, idx, self.cols, self.rows)
        iop.write(col.to_vec())
        j = idx + self.rows
        while j >= 2 * self.top_size:
            low = j & 1
            j >>= 1
            other = 2 * j + (1 - low)
            iop.write(self.nodes.slice(other * 8, 8).to_vec())


class PolyGroup:
    """prove/poly_group.rs: expand + evaluate, bit-reverse the coefficients, Merkle tree over the evaluations."""

    def __init__(self, hal, coeffs, count: int, n: int):
        self.coeffs, self.count, self.n = coeffs, count, n
        dom = n * INV_RATE
        self.evaluated = hal.alloc_elem("evaluated", count * dom)
        hal.batch_expand_into_evaluate_ntt(self.evaluated, coeffs, count, 2)
        hal.batch_bit_reverse(coeffs, count)                     # coefficients in natural order from here on
        self.merkle = MerkleTreeProver(hal, self.evaluated, dom, count)


def prove_segment_trait_only(hal: "H.HipHal", circuit: "H.Circuit", po2: int, code, data, out_global,
                             accumulate) -> np.ndarray:
    """`SegmentProver::prove_segment` through trait methods only.  code / data: W x 2^po2 traces on the device;
    accumulate(mix_global words) -> accum trace buffer (CircuitHal::accumulate, circuit-specific)."""
    desc = Desc.parse(circuit.desc)
    wa, wc, wd = desc.group_sizes
    out_size, mix_size = desc.global_sizes
    n = 1 << po2
    dom = n * INV_RATE
    iop = Iop()
    out_global = np.asarray(out_global, dtype=np.uint32)
    hdr = list(int(x) for x in out_global) + [enc(po2)]
    iop.write(hdr)
    iop.commit(hash_elems(hdr))

    def commit_group(trace, count):
        coeffs = hal.alloc_elem("coeffs", count * n)
        hal.eltwise_copy_elem(coeffs, trace)
        hal.batch_interpolate_ntt(coeffs, count)
        hal.zk_shift(coeffs, count)
        pg = PolyGroup(hal, coeffs, count, n)
        pg.merkle.commit(iop)
        return pg

    groups = [None, None, None]
    groups[1] = commit_group(code, wc)
    groups[2] = commit_group(data, wd)
    mix_global = [iop.rng.elem() for _ in range(mix_size)]
    accum = accumulate(np.asarray(mix_global, dtype=np.uint32))
    groups[0] = commit_group(accum, wa)

    # ---- finalize ----
    poly_mix = iop.rng.ext()
    check = hal.alloc_elem("check_poly", EXT * dom)
    g_out = hal.copy_from("out", out_global)
    g_mix = hal.copy_from("mix", np.asarray(mix_global if mix_global else [0], dtype=np.uint32))
    circuit.eval_check(check, [g.evaluated for g in groups], [g_out, g_mix], e_words(poly_mix), po2)
    hal.batch_interpolate_ntt(check, EXT)
    check_group = PolyGroup(hal, check, CHECK_SIZE, n)
    check_group.merkle.commit(iop)

    z = iop.rng.ext()
    back_one = ROU_REV[po2]
    taps = desc.taps
    regs = desc.regs
    all_xs, eval_u = [], []
    for g in range(3):
        which = [off for (gg, off, back) in taps if gg == g]
        xs = [e_scale(z, pow(back_one, back, P)) for (gg, off, back) in taps if gg == g]
        all_xs.extend(xs)
        if not which:
            continue
        dw = hal.copy_from("which", np.asarray(which, dtype=np.uint32))
        dx = hal.copy_from("xs", np.asarray([w for x in xs for w in e_words(x)], dtype=np.uint32))
        dout = hal.alloc_extelem("out", len(which))
        hal.batch_evaluate_any(groups[g].coeffs, groups[g].count, dw, dx, dout)
        o = dout.to_vec()
        eval_u.extend(tuple(dec(o[4 * k + i]) for i in range(4)) for k in range(len(which)))
    z_pow = e_pow(z, EXT)
    dw = hal.copy_from("which", np.arange(CHECK_SIZE, dtype=np.uint32))
    dx = hal.copy_from("xs", np.asarray(e_words(z_pow) * CHECK_SIZE, dtype=np.uint32))
    dout = hal.alloc_extelem("out", CHECK_SIZE)
    hal.batch_evaluate_any(check_group.coeffs, CHECK_SIZE, dw, dx, dout)
    o = dout.to_vec()
    check_u = [tuple(dec(o[4 * k + i]) for i in range(4)) for k in range(CHECK_SIZE)]
    coeff_u, pos = [], 0
    for (g, off, backs, combo_id) in regs:
        k = len(backs)
        coeff_u.extend(poly_interpolate(all_xs[pos:pos + k], eval_u[pos:pos + k]))
        pos += k
    coeff_u.extend(check_u)
    cu_words = [w for c in coeff_u for w in e_words(c)]
    iop.write(cu_words)
    iop.commit(hash_elems(cu_words))

    mix = iop.rng.ext()
    combo_count = len(desc.combos)
    combos = hal.alloc("combos", n * (combo_count + 1) * EXT, zero=True)
    cur_mix = (1, 0, 0, 0)
    for g in range(3):
        which = [combo_id for (gg, off, backs, combo_id) in regs if gg == g]
        dw = hal.copy_from("which", np.asarray(which, dtype=np.uint32))
        hal.mix_poly_coeffs(combos, e_words(cur_mix), e_words(mix), groups[g].coeffs, dw, len(which), n)
        cur_mix = e_mul(cur_mix, e_pow(mix, len(which)))
    dw = hal.copy_from("which", np.full(CHECK_SIZE, combo_count, dtype=np.uint32))
    hal.mix_poly_coeffs(combos, e_words(cur_mix), e_words(mix), check_group.coeffs, dw, CHECK_SIZE, n)

    d_cu = hal.copy_from("coeff_u", np.asarray(cu_words, dtype=np.uint32))
    d_sizes = hal.copy_from("reg_sizes", np.asarray([len(r[2]) for r in regs], dtype=np.uint32))
    d_ids = hal.copy_from("reg_combo_ids", np.asarray([r[3] for r in regs], dtype=np.uint32))
    hal.combos_prepare_regs(combos, d_cu, combo_count, n, d_sizes, d_ids, e_words(mix))

    for i in range(combo_count + 1):
        pts = [z_pow] if i == combo_count else [e_scale(z, pow(back_one, b, P)) for b in desc.combos[i]]
        rem = hal.alloc("rems", 4 * len(pts), zero=True)
        hal.combos_divide(combos, i, n, np.asarray([w for p in pts for w in e_words(p)], dtype=np.uint32), rem)
        assert not rem.to_vec().any(), "DEEP quotient has a non-zero remainder"

    final_coeffs = hal.alloc_elem("final_poly_coeffs", n * EXT)
    hal.eltwise_sum_extelem(final_coeffs, combos)
    hal.batch_bit_reverse(final_coeffs, EXT)

    # ---- fri_prove ----
    rounds = []
    cur = final_coeffs
    while cur.size() // EXT > FRI_MIN_DEGREE:
        size = cur.size() // EXT
        domain = size * INV_RATE
        evaluated = hal.alloc_elem("evaluated", domain * EXT)
        hal.batch_expand_into_evaluate_ntt(evaluated, cur, EXT, 2)
        merkle = MerkleTreeProver(hal, evaluated, domain // FRI_FOLD, FRI_FOLD * EXT)
        merkle.commit(iop)
        fold_mix = iop.rng.ext()
        out_coeffs = hal.alloc_elem("out_coeffs", size // FRI_FOLD * EXT)
        hal.fri_fold(out_coeffs, cur, e_words(fold_mix))
        rounds.append((domain, merkle))
        cur = out_coeffs
    fin = hal.alloc_elem("final_coeffs", cur.size())
    hal.eltwise_copy_elem(fin, cur)
    hal.batch_bit_reverse(fin, EXT)
    fw = fin.to_vec()
    iop.write(fw)
    iop.commit(hash_elems(fw))

    # ---- queries ----
    for _ in range(QUERIES):
        rng_idx = iop.rng.bits(log2(dom))
        pos = rng_idx % dom
        for g in groups:
            g.merkle.prove(iop, pos)
        check_group.merkle.prove(iop, pos)
        for domain, merkle in rounds:
            pos %= domain // FRI_FOLD
            merkle.prove(iop, pos)
    return np.asarray(iop.proof, dtype=np.uint32)
