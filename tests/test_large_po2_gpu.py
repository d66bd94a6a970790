"""--segment-po2 above 22 (/root/reference/crates/host/src/bin/cli.rs:61-66 and lib.rs:132-135 pass any po2 up to upstream's
MAX_CYCLES_PO2 = 24): the evaluation domain of a po2-24 segment is 2^26 points, so every transform-side op must be right at
log_n 25 / 26 — a 14-bit high twiddle table, a 4-bit generic top pass above the two register-radix passes, 64-bit addressing of
buffers beyond 2^32 words.  Op parity against the CPU oracle on 1-2 columns, a po2-23 seal byte-identical to the oracle's on a
reduced-width SYN-AIR shape, full-width SYN-A seals at po2 23 and 24 accepted by the product's verifier AND the oracle's."""
import numpy as np
import pytest

from conftest import rand_fp

pytestmark = pytest.mark.gpu


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    if not np.array_equal(a, b):
        bad = np.flatnonzero(a.reshape(-1) != b.reshape(-1))
        raise AssertionError(f"{bad.size} mismatches, first at {bad[:5]}: {a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}")


@pytest.fixture(scope="module", autouse=True)
def _more_oracle_threads(oracle):
    oracle.zko_set_num_threads(64)          # the 2^25 / 2^26 oracle transforms and a po2-23 oracle seal are minutes at 16 threads
    yield
    oracle.zko_set_num_threads(16)


@pytest.mark.parametrize("log_n,count", [(24, 2), (25, 2), (26, 1)])
def test_interpolate_ntt_and_zk_shift_above_2_24(hal, oracle, log_n, count):
    rng = np.random.default_rng(log_n)
    x = rand_fp(rng, count << log_n)
    want = x.copy()
    oracle.zko_batch_interpolate_ntt(want, want.size, count)
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(buf, count)
    eq(buf.to_vec(), want)
    oracle.zko_zk_shift(want, want.size, count)
    fused = hal.copy_from("io", x)
    hal.batch_interpolate_ntt_zk_shift(fused, count)
    eq(fused.to_vec(), want)
    hal.zk_shift(buf, count)
    eq(buf.to_vec(), want)


@pytest.mark.parametrize("log_n,count,bits", [(25, 2, 2), (26, 1, 2), (26, 1, 0), (25, 1, 3)])
def test_expand_into_evaluate_ntt_above_2_24(hal, oracle, log_n, count, bits):
    rng = np.random.default_rng(log_n * 7 + bits)
    n_out = 1 << log_n
    x = rand_fp(rng, count * (n_out >> bits))
    want = np.zeros(count * n_out, dtype=np.uint32)
    oracle.zko_batch_expand_into_evaluate_ntt(want, want.size, x, x.size, count, bits)
    out = hal.alloc_elem("out", count * n_out)
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from("in", x), count, bits)
    eq(out.to_vec(), want)


def test_bit_reverse_and_roundtrip_at_2_26(hal, oracle):
    rng = np.random.default_rng(26)
    x = rand_fp(rng, 1 << 26)
    want = x.copy()
    oracle.zko_batch_bit_reverse(want, want.size, 1)
    buf = hal.copy_from("io", x)
    hal.batch_bit_reverse(buf, 1)
    eq(buf.to_vec(), want)
    # evaluate(expand 0) o interpolate == id (size-independent property, at the largest size)
    io = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(io, 1)
    out = hal.alloc_elem("out", 1 << 26)
    hal.batch_expand_into_evaluate_ntt(out, io, 1, 0)
    eq(out.to_vec(), x)


def test_merkle_tree_of_2_25_rows(hal, oracle):
    """hash_rows + every hash_fold layer over 2^25 leaves (a po2-23 group's tree) == the oracle's, root and top layer."""
    rng = np.random.default_rng(2)
    rows, cols = 1 << 25, 2
    m = rand_fp(rng, cols * rows)
    nodes = hal.alloc_digest("nodes", 2 * rows)
    hal.merkle_build(nodes, hal.copy_from("m", m), rows)
    want = np.zeros(2 * rows * 8, dtype=np.uint32)
    leaves = want[rows * 8:]
    oracle.zko_hash_rows(leaves, rows, m, rows * cols)
    layer = rows
    while layer >= 2:
        oracle.zko_hash_fold(want, layer, layer // 2)
        layer //= 2
    got = nodes.to_vec()
    eq(got[8:16 * 64], want[8:16 * 64])            # root .. the 2^5-wide layers
    eq(got[rows * 8:rows * 8 + 4096], want[rows * 8:rows * 8 + 4096])
    eq(got[-4096:], want[-4096:])


def test_po2_23_seal_equals_the_oracles_on_a_reduced_width_shape(hal, oracle):
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_small()                      # W_code 8, W_data 20, W_accum 8: the same constraint family as SYN-A
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=23, seed=0x5EED0023, noise_seed=0x2E80)
    rec = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed)
    eq(rec.seal, want)
    root = prover.control_root(23)
    eq(root, oc.control_root(23, seg.zk_cycles))
    assert oc.verify(rec.seal, root) is None
    rec.verify(desc, root)


@pytest.mark.parametrize("po2", [23, 24])
def test_full_width_syn_a_seal_is_accepted_by_both_verifiers(hal, oracle, po2):
    """po2 24 = upstream's MAX_CYCLES_PO2: 208 data columns x 2^26 evaluations = 56 GB in one buffer, ~110 GB resident."""
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_a()
    prover = SegmentProver(hal, desc)
    rec = prover.prove_segment(Segment(index=0, po2=po2, seed=0x5EED0000 + po2, noise_seed=0x2E80))
    root = prover.control_root(po2)
    rec.verify(desc, root)
    assert zko.OracleCircuit(oracle, desc).verify(rec.seal, root) is None
    tampered = rec.seal.copy()
    tampered[len(tampered) // 2] ^= 1
    with pytest.raises(Exception):
        type(rec)(seal=tampered, index=0, po2=po2).verify(desc, root)
    hal.trim()
