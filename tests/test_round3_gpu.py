"""Round-3 parity tests on the GPU: a prover that uses ONLY the 1:1 trait methods (tests/hal_only_prover.py) must produce the
same seal, byte for byte, as the library's fused prover; Hal::combos_prepare with upstream's argument list equals the
host-flattened form."""
import hashlib
import json
import os

import numpy as np
import pytest

from zeth_amd.circuits import syn_air
from zeth_amd.circuits.desc import Circuit as Desc
from zeth_amd.prover import Segment, SegmentProver

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
P = 2013265921


def _trait_only_seal(hal, prover, seg):
    import hal_only_prover as hop
    code, data, out = prover.witgen(seg)
    seal = hop.prove_segment_trait_only(hal, prover.circuit, seg.po2, code, data, out, prover.syn_accumulate(seg, data))
    return seal, prover.seal(seg, code, data, out).seal


@pytest.mark.parametrize("shape,po2", [("syn_small", 12), ("syn_a", 13), ("syn_a", 16), ("syn_heavy_small", 13), ("syn_a", 20)])
def test_trait_only_prover_seals_are_byte_identical_to_the_fused_prover(hal, shape, po2):
    """`Prover<HipHal>` as upstream Rust would drive it (separate zk_shift, bit-reversed coefficients, per-layer hash_fold,
    natural-order batch_evaluate_any, literal combos_prepare, per-combo combos_divide, 50 x gather_sample openings) against
    zkh_prove_segment (fused / batched / reordered).  po2 20 = BASELINE config 2's shape: that seal is also the CPU oracle's
    golden seal (tests/golden/large_digests.json), which closes the triangle trait-only == fused == oracle."""
    from zeth_amd.circuits import syn_heavy
    desc = {"syn_small": syn_air.syn_small, "syn_a": syn_air.syn_a,
            "syn_heavy_small": syn_heavy.syn_heavy_small}[shape]()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=po2, seed=0x5EED0000, noise_seed=0x2E80)
    got, want = _trait_only_seal(hal, prover, seg)
    assert got.size == want.size
    assert np.array_equal(got, want), f"first differing word: {int(np.argmax(got != want))}"
    if shape == "syn_a" and po2 == 20:
        with open(os.path.join(G, "large_digests.json")) as fh:
            case = next(c for c in json.load(fh)["cases"] if c["shape"] == "syn_a" and c["po2"] == 20)
        assert hashlib.sha256(got.astype("<u4").tobytes()).hexdigest() == case["seal_sha256"]


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
