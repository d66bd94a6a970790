"""HIP HAL vs CPU oracle, op by op, bit-exact on seeded inputs (mirrors upstream's cpu-vs-gpu hal tests in
risc0-zkp src/hal/mod.rs `mod testutil`).  All calls go through the C ABI (zeth_amd.hal -> libzkhal_mi355x.so)."""
import ctypes as C

import numpy as np
import pytest

from conftest import P, rand_fp

pytestmark = pytest.mark.gpu


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    if not np.array_equal(a, b):
        bad = np.flatnonzero(a.reshape(-1) != b.reshape(-1))
        raise AssertionError(f"{bad.size} mismatches, first at {bad[:5]}: {a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}")


@pytest.mark.parametrize("log_n,count", [(3, 1), (8, 3), (12, 2), (13, 5), (16, 2), (17, 3), (18, 2), (19, 1), (20, 1), (21, 1), (22, 1), (23, 1)])
def test_batch_interpolate_ntt(hal, oracle, log_n, count):
    rng = np.random.default_rng(log_n * 100 + count)
    n = 1 << log_n
    x = rand_fp(rng, count * n)
    want = x.copy()
    oracle.zko_batch_interpolate_ntt(want, want.size, count)
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(buf, count)
    eq(buf.to_vec(), want)
    # fused zk_shift variant == interpolate then zk_shift
    oracle.zko_zk_shift(want, want.size, count)
    buf2 = hal.copy_from("io", x)
    hal.batch_interpolate_ntt_zk_shift(buf2, count)
    eq(buf2.to_vec(), want)
    out = hal.alloc_elem("out", x.size)
    src = hal.copy_from("in", x)
    hal.batch_interpolate_ntt_from(out, src, count, True)
    eq(out.to_vec(), want)
    eq(src.to_vec(), x)
    hal.zk_shift(buf, count)
    eq(buf.to_vec(), want)


@pytest.mark.parametrize("log_n,count,bits", [(4, 2, 2), (10, 3, 2), (14, 4, 2), (15, 2, 0), (18, 3, 2), (20, 2, 2), (22, 1, 2), (12, 2, 1), (24, 1, 2), (13, 1, 3),
                                              (20, 2, 0), (20, 1, 3), (22, 1, 4), (22, 1, 1), (21, 1, 2), (19, 2, 2), (23, 2, 2), (21, 3, 0), (24, 1, 0), (23, 1, 4)])
def test_batch_expand_into_evaluate_ntt(hal, oracle, log_n, count, bits):
    rng = np.random.default_rng(log_n * 7 + count)
    n_out = 1 << log_n
    n_in = n_out >> bits
    x = rand_fp(rng, count * n_in)
    want = np.zeros(count * n_out, dtype=np.uint32)
    oracle.zko_batch_expand_into_evaluate_ntt(want, want.size, x, x.size, count, bits)
    out = hal.alloc_elem("out", count * n_out)
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from("in", x), count, bits)
    eq(out.to_vec(), want)


@pytest.mark.parametrize("log_n,bits", [(20, 2), (22, 2), (20, 0)])
def test_expand_ntt_extreme_inputs(hal, oracle, log_n, bits):
    """The 2^20 / 2^22 forward transforms run lazy signed butterflies (values in (-P, P), |x + w y| < P 2^31): drive
    them with all-(P-1), all-zero and alternating columns, where every intermediate sits at the edge of its range."""
    n_out = 1 << log_n
    n_in = n_out >> bits
    cols = [np.full(n_in, P - 1, np.uint32), np.zeros(n_in, np.uint32),
            np.where(np.arange(n_in) % 2 == 0, P - 1, 0).astype(np.uint32),
            np.where(np.arange(n_in) % 3 == 0, 1, P - 1).astype(np.uint32)]
    x = np.concatenate(cols)
    want = np.zeros(len(cols) * n_out, dtype=np.uint32)
    oracle.zko_batch_expand_into_evaluate_ntt(want, want.size, x, x.size, len(cols), bits)
    out = hal.alloc_elem("out", len(cols) * n_out)
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from("in", x), len(cols), bits)
    eq(out.to_vec(), want)


def test_ntt_roundtrip_full_size(hal):
    """Size-independent property at BASELINE size: evaluate(expand 0) o interpolate == id on 2^20 x 4 columns."""
    rng = np.random.default_rng(5)
    n, count = 1 << 20, 4
    x = rand_fp(rng, count * n)
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(buf, count)
    out = hal.alloc_elem("out", count * n)
    hal.batch_expand_into_evaluate_ntt(out, buf, count, 0)
    eq(out.to_vec(), x)


@pytest.mark.parametrize("log_n,count", [(1, 3), (5, 2), (10, 2), (11, 3), (16, 2), (20, 1)])
def test_batch_bit_reverse(hal, oracle, log_n, count):
    rng = np.random.default_rng(log_n)
    x = rand_fp(rng, count << log_n)
    want = x.copy()
    oracle.zko_batch_bit_reverse(want, want.size, count)
    buf = hal.copy_from("io", x)
    hal.batch_bit_reverse(buf, count)
    eq(buf.to_vec(), want)
    hal.batch_bit_reverse(buf, count)      # involution
    eq(buf.to_vec(), x)


@pytest.mark.parametrize("rows,cols", [(64, 1), (256, 16), (1000, 17), (4096, 33), (512, 0), (1 << 14, 208), (300, 64)])
def test_hash_rows(hal, oracle, rows, cols):
    rng = np.random.default_rng(rows + cols)
    m = rand_fp(rng, rows * cols)
    want = np.zeros(rows * 8, dtype=np.uint32)
    oracle.zko_hash_rows(want, rows, m if cols else np.zeros(1, np.uint32), rows * cols)
    out = hal.alloc_digest("out", rows)
    hal.hash_rows(out, hal.copy_from("m", m) if cols else hal.alloc("m", 0))
    eq(out.to_vec(), want)


@pytest.mark.parametrize("rows", [2, 32, 512, 1024, 4096, 1 << 15])
def test_merkle_fold(hal, oracle, rows):
    rng = np.random.default_rng(rows)
    nodes = np.zeros(rows * 16, dtype=np.uint32)
    nodes[rows * 8:] = rand_fp(rng, rows * 8)
    want = nodes.copy()
    layer = rows
    while layer > 1:
        oracle.zko_hash_fold(want, layer, layer // 2)
        layer //= 2
    buf = hal.copy_from("nodes", nodes)
    hal.merkle_fold_all(buf, rows)
    got = buf.to_vec()
    eq(got[8:], want[8:])
    # layer-by-layer Hal::hash_fold gives the same tree
    buf2 = hal.copy_from("nodes", nodes)
    layer = rows
    while layer > 1:
        hal.hash_fold(buf2, layer, layer // 2)
        layer //= 2
    eq(buf2.to_vec()[8:], want[8:])


@pytest.mark.parametrize("po,polys,evals", [(16, 3, 5), (1 << 12, 4, 7), (1 << 16, 3, 4), (70000, 2, 3)])
def test_batch_evaluate_any(hal, oracle, po, polys, evals):
    rng = np.random.default_rng(po)
    coeffs = rand_fp(rng, po * polys)
    which = rng.integers(0, polys, size=evals).astype(np.uint32)
    xs = rand_fp(rng, 4 * evals)
    want = np.zeros(4 * evals, dtype=np.uint32)
    oracle.zko_batch_evaluate_any(coeffs, coeffs.size, polys, which, xs, evals, want)
    out = hal.alloc_extelem("out", evals)
    hal.batch_evaluate_any(hal.copy_from("c", coeffs), polys, hal.copy_from("w", which), hal.copy_from("x", xs), out)
    eq(out.to_vec(), want)


@pytest.mark.parametrize("count,input_size,ncombo", [(100, 5, 2), (4096, 17, 3), (1 << 14, 40, 4)])
def test_mix_poly_coeffs(hal, oracle, count, input_size, ncombo):
    rng = np.random.default_rng(count)
    inp = rand_fp(rng, input_size * count)
    combos = np.sort(rng.integers(0, ncombo, size=input_size)).astype(np.uint32)
    rng.shuffle(combos[: input_size // 2])          # unsorted prefix: flush logic must still be right
    out0 = rand_fp(rng, 4 * ncombo * count)
    mix_start, mix = rand_fp(rng, 4), rand_fp(rng, 4)
    want = out0.copy()
    oracle.zko_mix_poly_coeffs(want, mix_start, mix, inp, combos, input_size, count)
    out = hal.copy_from("out", out0)
    hal.mix_poly_coeffs(out, mix_start, mix, hal.copy_from("in", inp), hal.copy_from("cb", combos), input_size, count)
    eq(out.to_vec(), want)


def test_eltwise_ops(hal, oracle):
    rng = np.random.default_rng(11)
    n = 10007
    a, b = rand_fp(rng, n), rand_fp(rng, n)
    want = np.zeros(n, dtype=np.uint32)
    oracle.zko_eltwise_add_elem(want, a, b, n)
    out = hal.alloc_elem("o", n)
    hal.eltwise_add_elem(out, hal.copy_from("a", a), hal.copy_from("b", b))
    eq(out.to_vec(), want)
    hal.eltwise_copy_elem(out, hal.copy_from("a", a))
    eq(out.to_vec(), a)
    z = a.copy()
    z[::7] = 0xFFFFFFFF
    zb = hal.copy_from("z", z)
    hal.eltwise_zeroize_elem(zb)
    z[::7] = 0
    eq(zb.to_vec(), z)
    # sum_extelem
    count, k = 777, 5
    inp = rand_fp(rng, 4 * count * k)
    want = np.zeros(4 * count, dtype=np.uint32)
    oracle.zko_eltwise_sum_extelem(want, want.size, inp, count * k)
    out = hal.alloc_elem("s", 4 * count)
    hal.eltwise_sum_extelem(out, hal.copy_from("i", inp))
    eq(out.to_vec(), want)


@pytest.mark.parametrize("count", [1, 16, 4096, 1 << 16])
def test_fri_fold(hal, oracle, count):
    rng = np.random.default_rng(count)
    inp = rand_fp(rng, 4 * 16 * count)
    mix = rand_fp(rng, 4)
    want = np.zeros(4 * count, dtype=np.uint32)
    oracle.zko_fri_fold(want, want.size, inp, mix)
    out = hal.alloc_elem("o", 4 * count)
    hal.fri_fold(out, hal.copy_from("i", inp), mix)
    eq(out.to_vec(), want)


def test_gather_scatter(hal, oracle):
    rng = np.random.default_rng(3)
    src = rand_fp(rng, 64 * 1000)
    want = np.zeros(64, dtype=np.uint32)
    oracle.zko_gather_sample(want, src, 123, 64, 1000)
    dst = hal.alloc_elem("d", 64)
    hal.gather_sample(dst, hal.copy_from("s", src), 123, 64, 1000)
    eq(dst.to_vec(), want)
    into = np.zeros(500, dtype=np.uint32)
    index = rng.permutation(500)[:200].astype(np.uint32)
    values = rand_fp(rng, 200)
    offsets = np.array([0, 50, 50, 120, 200], dtype=np.uint32)
    b = hal.copy_from("into", into)
    hal.scatter(b, index, offsets, values)
    into[index] = values
    eq(b.to_vec(), into)


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 70000, 1 << 18])
def test_prefix_products(hal, oracle, n):
    rng = np.random.default_rng(n)
    x = rand_fp(rng, 4 * n)
    want = x.copy()
    oracle.zko_prefix_products(want, n)
    buf = hal.copy_from("io", x)
    hal.prefix_products(buf)
    eq(buf.to_vec(), want)


@pytest.mark.parametrize("cycles", [64, 256, 1000, 65536, 1 << 18])
def test_combos_divide(hal, oracle, cycles):
    rng = np.random.default_rng(cycles)
    ncombo = 3
    combos = rand_fp(rng, 4 * ncombo * cycles)
    pts = rand_fp(rng, 4 * 2)
    want = combos.copy()
    rems = []
    poly = want[4 * cycles: 8 * cycles].copy()
    for k in range(2):
        rem = np.zeros(4, dtype=np.uint32)
        oracle.zko_poly_divide(poly, cycles, pts[4 * k: 4 * k + 4].copy(), rem)
        rems.append(rem)
    want[4 * cycles: 8 * cycles] = poly
    buf = hal.copy_from("c", combos)
    rem_out = hal.alloc_extelem("r", 2)
    hal.combos_divide(buf, 1, cycles, pts, rem_out)
    eq(buf.to_vec(), want)
    eq(rem_out.to_vec(), np.concatenate(rems))


def test_combos_prepare(hal):
    rng = np.random.default_rng(8)
    combos = rand_fp(rng, 4 * 100)
    pos = np.array([0, 7, 99], dtype=np.uint32)
    vals = rand_fp(rng, 12)
    buf = hal.copy_from("c", combos)
    hal.combos_prepare(buf, pos, vals)
    want = combos.astype(np.int64)
    for k, p in enumerate(pos):
        want[4 * p: 4 * p + 4] = (want[4 * p: 4 * p + 4] - vals[4 * k: 4 * k + 4].astype(np.int64)) % P
    eq(buf.to_vec(), want.astype(np.uint32))


def test_merkle_open(hal, oracle):
    rng = np.random.default_rng(21)
    rows, cols = 1 << 12, 7
    m = rand_fp(rng, rows * cols)
    mat = hal.copy_from("m", m)
    nodes = hal.alloc_digest("n", 2 * rows)
    hal.hash_rows(nodes.slice(rows * 8, rows * 8), mat)
    hal.merkle_fold_all(nodes, rows)
    nd = nodes.to_vec().reshape(-1, 8)
    idx = rng.integers(0, rows, size=50).astype(np.uint32)
    wpq = cols + 8 * (12 - 5)
    out = hal.alloc("o", wpq * 50)
    hal.merkle_open(mat, nodes, rows, cols, idx, out)
    got = out.to_vec().reshape(50, wpq)
    for q, i in enumerate(idx):
        eq(got[q, :cols], m.reshape(cols, rows)[:, i])
        j, k = int(i) + rows, 0
        while j >= 64:
            eq(got[q, cols + 8 * k: cols + 8 * k + 8], nd[j ^ 1])
            j //= 2
            k += 1
        assert k == 7


@pytest.mark.parametrize("log_n,polys,evals", [(14, 3, 4), (16, 2, 5), (20, 2, 3)])
def test_batch_evaluate_any_bitrev(hal, oracle, log_n, polys, evals):
    """Coefficients left in the iNTT's bit-reversed order give the same evaluations as natural order + evaluate_any."""
    rng = np.random.default_rng(log_n)
    po = 1 << log_n
    nat = rand_fp(rng, po * polys)
    br = nat.copy()
    oracle.zko_batch_bit_reverse(br, br.size, polys)
    which = rng.integers(0, polys, size=evals).astype(np.uint32)
    xs = rand_fp(rng, 4 * evals)
    want = np.zeros(4 * evals, dtype=np.uint32)
    oracle.zko_batch_evaluate_any(nat, nat.size, polys, which, xs, evals, want)
    out = hal.alloc_extelem("out", evals)
    hal.batch_evaluate_any_bitrev(hal.copy_from("c", br), polys, hal.copy_from("w", which), hal.copy_from("x", xs), out)
    eq(out.to_vec(), want)
    from zeth_amd.hal import HalError
    with pytest.raises(HalError, match="2\\^14"):
        hal.batch_evaluate_any_bitrev(hal.copy_from("c", br[:1 << 10]), 1, hal.copy_from("w", which[:1] * 0), hal.copy_from("x", xs[:4]),
                                      hal.alloc_extelem("o", 1))


@pytest.mark.parametrize("log_n,count", [(1, 2), (8, 3), (15, 2)])
def test_batch_bit_reverse_extelem(hal, log_n, count):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    x = rand_fp(rng, 4 * n * count)
    buf = hal.copy_from("io", x)
    hal.batch_bit_reverse_extelem(buf, count)
    got = buf.to_vec().reshape(count, n, 4)
    src = x.reshape(count, n, 4)
    rev = np.array([int(format(i, f"0{log_n}b")[::-1], 2) for i in range(n)])
    eq(got, src[:, rev, :])


def test_poseidon2_permutation_known_answer_and_oracle(hal, oracle):
    """The bare permutation on the device: the published known-answer vector of the instance
    (tests/golden/poseidon2_kat.json), and the oracle's literal permutation on random and edge-valued states."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon2_kat.json")))
    want = np.array([int(x, 16) for x in kat["output_hex"]], np.uint32)
    rng = np.random.default_rng(24)
    states = rand_fp(rng, 1000 * 24)
    states[:24] = [oracle.zko_fp_encode(int(v)) for v in kat["input"]]
    states[24:48] = 0
    states[48:72] = P - 1
    states[72:96] = np.array([0, 1, P - 1, (P - 1) // 2, (P + 1) // 2, 268435454] * 4, np.uint32)
    ref = states.copy()
    for k in range(1000):
        oracle.zko_poseidon2_mix(ref[24 * k: 24 * k + 24])
    buf = hal.copy_from("states", states)
    hal.poseidon2_mix(buf)
    got = buf.to_vec()
    assert np.array_equal(got, ref)
    assert np.array_equal(np.array([oracle.zko_fp_decode(int(v)) for v in got[:24]], np.uint32), want)
