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


def test_native_session_executor_equals_the_python_orchestration(hal):
    """zkh_session_prove (csrc/session.hip: C++ threads over lanes, one call per session) against the Python mirrors: the same
    seals (fixed noise), the same root receipt as prove_succinct with the same join noise, zkh_session_verify accepts, and a
    swapped leaf is rejected by the compact verification."""
    from zeth_amd.circuits import p2_join
    from zeth_amd.hal import HalError
    from zeth_amd.host import Session, SuccinctReceipt, node_claim, prove_succinct
    ldesc, jdesc = syn_air.syn_small(), p2_join.p2_join_circuit()
    segs = [Segment(index=i, po2=12 if i < 4 else 13, seed=700 + i, noise_seed=0x51) for i in range(5)]
    sess = Session(ldesc, devices=(0,), lanes_per_device=2, join_desc=jdesc)
    comp, root, stats = sess.prove(segs, join_tree=True, join_po2=13, join_noise_seed=0x77, verify=True)
    assert stats["n_joins"] == 4 and stats["verified"] and root is not None and stats["wall_s"] > 0
    lp, jp = SegmentProver(hal, ldesc), SegmentProver(hal, jdesc)
    want = [lp.prove_segment(s) for s in segs]
    for a, b in zip(comp.segments, want):
        assert np.array_equal(a.seal, b.seal)
    roots = {p: lp.control_root(p) for p in (12, 13)}
    jroot = jp.control_root(13)

    def claim_of(r, is_leaf):
        return node_claim(r, ldesc if is_leaf else jdesc, roots[r.po2] if is_leaf else jroot, is_leaf)
    ref = prove_succinct(want, jp.prove_segment, claim_of, join_po2=13, noise_seed=0x77)
    assert np.array_equal(root.seal, ref.root.seal)
    SuccinctReceipt(root, [], comp.segments).verify(ldesc, jdesc, roots, jroot)
    # a session without a join circuit refuses the join tree; a one-segment session has no root
    plain = Session(ldesc, lanes_per_device=1)
    with pytest.raises(HalError, match="without a join circuit"):
        plain.prove(segs[:2], join_tree=True)
    comp1, root1, _ = plain.prove(segs[:1], verify=True)
    assert root1 is None and np.array_equal(comp1.segments[0].seal, want[0].seal)
    sess.close(); plain.close()


def test_session_executor_seals_caller_produced_traces(hal):
    """The session's other input: traces the CALLER produced (upstream's flow — CPU preflight + witgen — here read back from the
    device generator), uploaded and sealed through prove_begin / accumulate / prove_finish inside the library.  Same traces, same
    noise: the same seals as the built-in path."""
    from zeth_amd.host import Session
    desc = syn_air.syn_small()
    prover = SegmentProver(hal, desc)
    segs = [Segment(index=i, po2=12, seed=40 + i, noise_seed=0x99) for i in range(3)]
    traces, want = [], []
    for seg in segs:
        code, data, out = prover.witgen(seg)
        traces.append((code.to_vec(), data.to_vec(), out))
        want.append(prover.seal(seg, code, data, out).seal)
    sess = Session(desc, lanes_per_device=2)
    comp, root, _ = sess.prove(segs, host_traces=[traces[0], None, traces[2]], verify=True)      # mixed: two uploaded, one generated
    for got, w in zip(comp.segments, want):
        assert np.array_equal(got.seal, w)
    sess.close()
