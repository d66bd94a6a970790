"""Seeded random shape sweep of the HAL ops against the CPU oracle (bit-exact), complementing the fixed shapes of
test_hal_parity_gpu.py: ragged sizes, column counts around the sponge rate, sizes around the 256-wide scan blocks and
the kernels' path switches (2^12 / 2^18 / 2^20 transforms, 2^7 / 2^15 Merkle layers).  Inputs mix uniform elements
with runs of 0 and P-1.  The seeds are fixed, so a failure reproduces.  ZKH_FUZZ_SEED_OFFSET=N shifts every seed by N: a soak
over fresh shapes on spare GPU time (`tools/gpu.sh fuzzsoak`); unset (the suite) = offset 0."""
import os

import numpy as np
import pytest

import zko
from conftest import P, rand_fp

pytestmark = pytest.mark.gpu
OFF = int(os.environ.get("ZKH_FUZZ_SEED_OFFSET", "0"))


def eq(a, b):
    assert a.shape == b.shape
    bad = np.flatnonzero(a != b)
    assert bad.size == 0, f"{bad.size} mismatches, first at {bad[:5]}"


def spicy(rng, n):
    """Uniform field elements with stretches of the extreme values 0 and P-1."""
    x = rand_fp(rng, n)
    if n:
        for _ in range(3):
            a = int(rng.integers(0, n))
            b = min(n, a + int(rng.integers(1, 40)))
            x[a:b] = P - 1 if rng.integers(0, 2) else 0
    return x


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_ntt(hal, oracle, seed):
    rng = np.random.default_rng(1000 + seed + OFF)
    log_n = int(rng.choice([1, 2, 3, 5, 7, 9, 11, 12, 13, 14, 16, 17, 18, 19, 20]))
    count = int(rng.integers(1, 4)) if log_n >= 18 else int(rng.integers(1, 7))
    n = 1 << log_n
    x = spicy(rng, n * count)
    want = x.copy()
    oracle.zko_batch_interpolate_ntt(want, want.size, count)
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(buf, count)
    eq(buf.to_vec(), want)
    bits = int(rng.integers(0, min(4, log_n) + 1))
    if bits < log_n + 2 and log_n + bits <= 22:
        big = np.zeros(count * (n << bits), dtype=np.uint32)
        oracle.zko_batch_expand_into_evaluate_ntt(big, big.size, want, want.size, count, bits)
        out = hal.alloc_elem("out", big.size)
        hal.batch_expand_into_evaluate_ntt(out, buf, count, bits)
        eq(out.to_vec(), big)


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_hash_rows_and_tree(hal, oracle, seed):
    rng = np.random.default_rng(2000 + seed + OFF)
    rows = 1 << int(rng.integers(1, 17))
    cols = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 64, 100, 208]))
    m = spicy(rng, rows * cols)
    nodes = np.zeros(rows * 16, dtype=np.uint32)
    oracle.zko_hash_rows(nodes[rows * 8:], rows, m, rows * cols)
    buf = hal.alloc_digest("nodes", 2 * rows)
    hal.hash_rows(buf.slice(8 * rows, 8 * rows), hal.copy_from("m", m))
    layer = rows
    while layer > 1:
        oracle.zko_hash_fold(nodes, layer, layer // 2)
        layer //= 2
    hal.merkle_fold_all(buf, rows)
    eq(buf.to_vec()[8:], nodes[8:])


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_evaluate_and_mix(hal, oracle, seed):
    rng = np.random.default_rng(3000 + seed + OFF)
    po = int(rng.choice([1, 3, 255, 256, 257, 1000, 4096, 16383, 16384, 16385, 40000]))
    polys, evals = int(rng.integers(1, 6)), int(rng.integers(1, 9))
    coeffs = spicy(rng, po * polys)
    which = rng.integers(0, polys, size=evals).astype(np.uint32)
    xs = rand_fp(rng, 4 * evals)
    want = np.zeros(4 * evals, dtype=np.uint32)
    oracle.zko_batch_evaluate_any(coeffs, coeffs.size, polys, which, xs, evals, want)
    out = hal.alloc_extelem("out", evals)
    dc = hal.copy_from("c", coeffs)
    hal.batch_evaluate_any(dc, polys, hal.copy_from("w", which), hal.copy_from("x", xs), out)
    eq(out.to_vec(), want)
    # mix_poly_coeffs over the same matrix: rows = po, columns = polys, random (unsorted) combo ids
    ncombo = int(rng.integers(1, 4))
    combos = rng.integers(0, ncombo, size=polys).astype(np.uint32)
    out0 = rand_fp(rng, 4 * ncombo * po)
    mix_start, mix = rand_fp(rng, 4), rand_fp(rng, 4)
    want = out0.copy()
    oracle.zko_mix_poly_coeffs(want, mix_start, mix, coeffs, combos, polys, po)
    dout = hal.copy_from("out", out0)
    hal.mix_poly_coeffs(dout, mix_start, mix, dc, hal.copy_from("cb", combos), polys, po)
    eq(dout.to_vec(), want)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_scans_and_fold(hal, oracle, seed):
    rng = np.random.default_rng(4000 + seed + OFF)
    n = int(rng.choice([1, 2, 3, 255, 256, 257, 511, 65535, 65536, 65537, 100000]))
    x = spicy(rng, 4 * n)
    want = x.copy()
    oracle.zko_prefix_products(want, n)
    buf = hal.copy_from("io", x)
    hal.prefix_products(buf)
    eq(buf.to_vec(), want)
    # synthetic division of one polynomial by two points
    cycles = max(2, n)
    poly = spicy(rng, 4 * cycles)
    pts = rand_fp(rng, 8)
    wantp, rems = poly.copy(), []
    for k in range(2):
        rem = np.zeros(4, dtype=np.uint32)
        oracle.zko_poly_divide(wantp, cycles, pts[4 * k: 4 * k + 4].copy(), rem)
        rems.append(rem)
    dp = hal.copy_from("p", poly)
    rem_out = hal.alloc_extelem("r", 2)
    hal.combos_divide(dp, 0, cycles, pts, rem_out)
    eq(dp.to_vec(), wantp)
    eq(rem_out.to_vec(), np.concatenate(rems))
    # fri_fold
    count = int(rng.choice([1, 2, 15, 16, 17, 255, 256, 1000]))
    inp = spicy(rng, 4 * 16 * count)
    mixv = rand_fp(rng, 4)
    wantf = np.zeros(4 * count, dtype=np.uint32)
    oracle.zko_fri_fold(wantf, wantf.size, inp, mixv)
    outf = hal.alloc_elem("o", 4 * count)
    hal.fri_fold(outf, hal.copy_from("i", inp), mixv)
    eq(outf.to_vec(), wantf)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_whole_seals(hal, oracle, seed, tmp_path, monkeypatch):
    """Random circuit widths, segment sizes and padding lengths: the HIP prover's seal equals the oracle prover's byte
    for byte (eval_check compiled at load time for each shape), and the host verifier accepts it."""
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.prover import Segment, SegmentProver
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    rng = np.random.default_rng(5000 + seed + OFF)
    wc, wd, wa = int(rng.integers(5, 20)), int(rng.integers(8, 120)), 4 * int(rng.integers(1, 6))
    po2 = int(rng.integers(9, 14))
    zk = int(rng.integers(50, min(1994, (1 << po2) - 64)))
    desc = syn_air.build_syn_air(wc, wd, wa)
    prover = SegmentProver(hal, desc)
    seg = Segment(index=seed, po2=po2, seed=int(rng.integers(1, 1 << 40)), noise_seed=int(rng.integers(1, 1 << 40)), zk_cycles=zk)
    receipt = prover.prove_segment(seg)
    want = zko.OracleCircuit(oracle, desc).prove(po2, zk, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want), f"shape ({wc},{wd},{wa}) po2 {po2} zk {zk}"
    receipt.verify(desc, prover.control_root(po2, zk))


@pytest.mark.parametrize("po2,zk", [(7, 40), (8, 60), (8, 200)])
def test_smallest_segments(hal, oracle, po2, zk):
    """Below FRI_MIN_DEGREE * 2 there is no FRI folding round at all (the final coefficients are sent directly)."""
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_tiny()
    seg = Segment(index=0, po2=po2, seed=3, noise_seed=4, zk_cycles=zk)
    prover = SegmentProver(hal, desc)
    receipt = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(po2, zk, 3, 4)
    assert np.array_equal(receipt.seal, want)
    assert np.array_equal(prover.control_root(po2, zk), oc.control_root(po2, zk))
    receipt.verify(desc, prover.control_root(po2, zk))


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_eval_check_random_circuits(hal, oracle, seed, tmp_path, monkeypatch):
    """Random constraint systems (circuits/syn_random.py: squares, +/- chains, values used as factor and addend, Fp4
    operands on either side, nested AndCond with base and Fp4 conditions) through all three evaluators: the kernels
    generated and compiled at load time, the on-device step interpreter, the oracle's literal interpreter."""
    import ctypes as C
    from zeth_amd.circuits import syn_random
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    rng = np.random.default_rng(7000 + seed + OFF)
    groups = [(4, 6, 12), (8, 5, 20), (4, 16, 33)][seed % 3]
    desc = syn_random.random_circuit(seed + OFF, groups=groups, n_values=160 + 40 * (seed % 4), n_constraints=30 + 15 * (seed % 3),
                                     max_back=1 + seed % 4)
    circ = hal.load_circuit(desc, jit=True)
    assert circ.kernel_kind() == "attached"
    oc = zko.OracleCircuit(oracle, desc)
    po2 = 5 + seed % 3
    dom = 4 << po2
    edge = np.array([0, 1, P - 1, (P - 1) // 2, (P + 1) // 2], np.uint32)
    gs = []
    for w in (int(x) for x in desc[3:6]):
        g = rand_fp(rng, w * dom)
        hit = rng.integers(0, 8, size=g.size) == 0              # edge words sprinkled over the evaluations
        g[hit] = edge[rng.integers(0, edge.size, size=int(hit.sum()))]
        gs.append(g)
    out, mix, poly_mix = rand_fp(rng, 4), rand_fp(rng, int(desc[3])), rand_fp(rng, 4)
    want = np.zeros(4 * dom, np.uint32)
    gp = (C.c_void_p * 3)(*[a.ctypes.data for a in gs])
    glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
    oracle.zko_eval_check(oc.h, want, gp, glp, poly_mix, po2)
    dev = [hal.copy_from("g", g) for g in gs]
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    for interp in (False, True):
        check = hal.alloc_elem("check", 4 * dom)
        circ.eval_check(check, dev, [g_out, g_mix], poly_mix, po2, use_interpreter=interp)
        assert np.array_equal(check.to_vec(), want), f"eval_check mismatch (seed={seed}, interpreter={interp})"
