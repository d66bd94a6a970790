"""DEEP-side ops on the GPU against the oracle: batch_evaluate_any over runs of equal columns, combos_divide_all, combos_prepare in upstream's argument list."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import zko
from conftest import rand_fp
from zeth_amd.circuits import syn_air
from zeth_amd.circuits.desc import Circuit
from zeth_amd.circuits.desc import Circuit as Desc
from zeth_amd.hal import HalError, HipHal
from zeth_amd.prover import Segment, SegmentProver, shipped_control_root

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
P = 2013265921



@pytest.mark.parametrize("bitrev", [False, True])
def test_batch_evaluate_any_runs_of_equal_columns(hal, oracle, bitrev):
    """Taps of one register = consecutive entries with the same `which`: one block streams the column once for up to 8
    points.  Runs of every length around that limit, interleaved with singletons, against the oracle."""
    rng = np.random.default_rng(77)
    po2, count = 14, 12
    n = 1 << po2
    coeffs = rng.integers(0, 2013265921, size=count * n, dtype=np.uint64).astype(np.uint32)
    which = []
    for col, run in zip([3, 0, 7, 7, 1, 11, 5, 2, 9, 4, 4, 6], [1, 5, 8, 9, 2, 17, 1, 3, 16, 7, 1, 24]):
        which += [col] * run
    which = np.array(which, dtype=np.uint32)
    xs = rng.integers(0, 2013265921, size=4 * which.size, dtype=np.uint64).astype(np.uint32)
    want = np.zeros(4 * which.size, np.uint32)
    oracle.zko_batch_evaluate_any(coeffs, coeffs.size, count, which, xs, which.size, want)
    dev = coeffs.copy()
    if bitrev:
        oracle.zko_batch_bit_reverse(dev, dev.size, count)       # the layout batch_interpolate_ntt leaves behind
    out = hal.alloc_elem("out", 4 * which.size)
    fn = hal.batch_evaluate_any_bitrev if bitrev else hal.batch_evaluate_any
    fn(hal.copy_from("c", dev), count, hal.copy_from("w", which), hal.copy_from("x", xs), out)
    assert np.array_equal(out.to_vec(), want)


def test_combos_divide_all_matches_sequential_division(hal, oracle):
    """Seven combo polynomials with 1..5 division points each (SYN-HEAVY's tap combos): the batched rounds give the same
    quotients and remainders as dividing every polynomial by its points one after the other on the CPU."""
    rng = np.random.default_rng(91)
    cycles, counts = 1 << 13, [1, 2, 3, 4, 5, 3, 2, 1]
    P = 2013265921
    combos = rng.integers(0, P, size=4 * cycles * len(counts), dtype=np.uint64).astype(np.uint32)
    pts = rng.integers(0, P, size=4 * sum(counts), dtype=np.uint64).astype(np.uint32)
    begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    want, want_rem = combos.copy(), np.zeros(4 * sum(counts), np.uint32)
    for i, cnt in enumerate(counts):
        poly = want[4 * cycles * i: 4 * cycles * (i + 1)]
        for k in range(cnt):
            j = int(begin[i]) + k
            rem = np.zeros(4, np.uint32)
            oracle.zko_poly_divide(poly, cycles, pts[4 * j: 4 * j + 4].copy(), rem)
            want_rem[4 * j: 4 * j + 4] = rem
    dev = hal.copy_from("combos", combos)
    rem_out = hal.alloc("rem", 4 * sum(counts), zero=True)
    hal.combos_divide_all(dev, cycles, pts, begin, rem_out)
    assert np.array_equal(dev.to_vec(), want)
    assert np.array_equal(rem_out.to_vec(), want_rem)


def test_combos_prepare_with_upstreams_argument_list_equals_the_flattened_form(hal):
    """zkh_combos_prepare_regs (device operands, upstream's signature) against zkh_combos_prepare fed the host aggregation
    the in-library prover computes, on a register list where several registers hit the same combo position."""
    import hal_only_prover as hop
    rng = np.random.default_rng(7)
    cycles, combo_count = 64, 5
    sizes = rng.integers(1, 6, size=97).astype(np.uint32)
    ids = rng.integers(0, combo_count, size=97).astype(np.uint32)
    n_u = int(sizes.sum()) + hop.CHECK_SIZE
    coeff_u = rng.integers(0, P, size=4 * n_u, dtype=np.uint64).astype(np.uint32)
    start = rng.integers(0, P, size=4 * cycles * (combo_count + 1), dtype=np.uint64).astype(np.uint32)
    mix = tuple(int(x) for x in rng.integers(1, P, size=4))
    a = hal.copy_from("combos", start)
    hal.combos_prepare_regs(a, hal.copy_from("cu", coeff_u), combo_count, cycles, hal.copy_from("s", sizes), hal.copy_from("i", ids),
                            hop.e_words(mix))
    # the flattened form: aggregate per position on the host (what csrc/prover.hip does)
    sub, cur, pos = {}, (1, 0, 0, 0), 0
    cu = [tuple(hop.dec(coeff_u[4 * k + i]) for i in range(4)) for k in range(n_u)]
    for sz, cid in zip(sizes, ids):
        for i in range(int(sz)):
            key = cycles * int(cid) + i
            sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos + i]))
        cur = hop.e_mul(cur, mix)
        pos += int(sz)
    for _ in range(hop.CHECK_SIZE):
        key = cycles * combo_count
        sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos]))
        pos += 1
        cur = hop.e_mul(cur, mix)
    b = hal.copy_from("combos", start)
    hal.combos_prepare(b, np.asarray(list(sub), dtype=np.uint32),
                       np.asarray([w for v in sub.values() for w in hop.e_words(v)], dtype=np.uint32))
    assert np.array_equal(a.to_vec(), b.to_vec())
    assert not np.array_equal(a.to_vec(), start)


def test_combos_prepare_regs_takes_any_register_size_and_refuses_inconsistent_lists(hal):
    """Round-3 advisor finding: registers larger than 32 silently lost their terms and the sizes were never checked against coeff_u.
    Now: any size up to `cycles` (against the host aggregation), and an inconsistent register list is an ERROR before the launch."""
    import hal_only_prover as hop
    from zeth_amd.hal import HalError
    P = 2013265921
    rng = np.random.default_rng(11)
    cycles, combo_count = 128, 3
    sizes = np.array([1, 40, 3, 97, 32, 33, 128, 2], dtype=np.uint32)
    ids = rng.integers(0, combo_count, size=sizes.size).astype(np.uint32)
    n_u = int(sizes.sum()) + hop.CHECK_SIZE
    coeff_u = rng.integers(0, P, size=4 * n_u, dtype=np.uint64).astype(np.uint32)
    start = rng.integers(0, P, size=4 * cycles * (combo_count + 1), dtype=np.uint64).astype(np.uint32)
    mix = tuple(int(x) for x in rng.integers(1, P, size=4))
    a = hal.copy_from("combos", start)
    hal.combos_prepare_regs(a, hal.copy_from("cu", coeff_u), combo_count, cycles, hal.copy_from("s", sizes), hal.copy_from("i", ids), hop.e_words(mix))
    sub, cur, pos = {}, (1, 0, 0, 0), 0
    cu = [tuple(hop.dec(coeff_u[4 * k + i]) for i in range(4)) for k in range(n_u)]
    for sz, cid in zip(sizes, ids):
        for i in range(int(sz)):
            key = cycles * int(cid) + i
            sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos + i]))
        cur = hop.e_mul(cur, mix)
        pos += int(sz)
    for _ in range(hop.CHECK_SIZE):
        key = cycles * combo_count
        sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos]))
        pos += 1
        cur = hop.e_mul(cur, mix)
    b = hal.copy_from("combos", start)
    hal.combos_prepare(b, np.asarray(list(sub), dtype=np.uint32), np.asarray([w for v in sub.values() for w in hop.e_words(v)], dtype=np.uint32))
    assert np.array_equal(a.to_vec(), b.to_vec())

    def call(sz, idv, cu_words):
        # the list is checked ON THE DEVICE (no read-back inside the call); the refusal arrives with the next host-visible sync, and the
        # combos are untouched
        target = hal.copy_from("combos", start)
        hal.combos_prepare_regs(target, hal.copy_from("cu", coeff_u[:cu_words]), combo_count, cycles,
                                hal.copy_from("s", np.asarray(sz, dtype=np.uint32)), hal.copy_from("i", np.asarray(idv, dtype=np.uint32)), hop.e_words(mix))
        try:
            hal.sync()
        finally:
            assert np.array_equal(target.to_vec(), start)
    with pytest.raises(HalError, match="add up to"):                  # sizes claim more U coefficients than coeff_u holds
        call(sizes, ids, 4 * (n_u - 5))
    with pytest.raises(HalError, match="has size 0 or more"):         # a register larger than the polynomial
        call([cycles + 1] + list(sizes[1:]), ids, coeff_u.size)
    with pytest.raises(HalError, match="has size 0 or more"):
        call([0] + list(sizes[1:]), ids, coeff_u.size)
    with pytest.raises(HalError, match="names a combo beyond"):
        call(sizes, [combo_count] + list(ids[1:]), coeff_u.size)
