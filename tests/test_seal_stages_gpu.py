"""A whole seal on the GPU, stage by stage: the BASELINE shape (po2-20 SYN-A / SYN-HEAVY) op by op against the oracle's stage digests, a prover that
uses ONLY the 1:1 trait methods (tests/hal_only_prover.py) against the fused prover, the resident code group, the two-halves seal (zkh_prove_begin / _finish)."""
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


def _sha(buf) -> str:
    return hashlib.sha256(buf.to_vec().tobytes()).hexdigest()


def _large_cases():
    with open(os.path.join(G, "large_digests.json")) as fh:
        return json.load(fh)["cases"]


def _commit_group_stagewise(hal, trace, w, n, stages, name):
    """Prover::commit_group through the individual Hal ops, every buffer compared with the oracle's digest.
    Returns (bit-reversed coeffs, evaluated)."""
    dom = 4 * n
    coeffs = hal.alloc_elem("coeffs", w * n)
    hal.batch_interpolate_ntt_from(coeffs, trace, w, True)                    # iNTT + zk_shift, bit-reversed output
    natural = hal.alloc_elem("natural", w * n)
    hal.eltwise_copy_elem(natural, coeffs)
    hal.batch_bit_reverse(natural, w)
    assert _sha(natural) == stages[f"coeffs.{name}"]["sha256"], f"coefficients of group {name} ({w} x 2^{n.bit_length() - 1})"
    del natural
    ev = hal.alloc_elem("evaluated", w * dom)
    hal.batch_expand_into_evaluate_ntt(ev, coeffs, w, 2)
    assert _sha(ev) == stages[f"evaluated.{name}"]["sha256"], f"expand-NTT of group {name} ({w} x 2^{dom.bit_length() - 1})"
    nodes = hal.alloc_digest("nodes", 2 * dom)
    hal.hash_rows(nodes.slice(dom * 8, dom * 8), ev)
    hal.merkle_fold_all(nodes, dom)
    assert _sha(nodes.slice(8, (2 * dom - 1) * 8)) == stages[f"nodes.{name}"]["sha256"], f"hash_rows / hash_fold of group {name}"
    return coeffs, ev


def _trait_only_seal(hal, prover, seg):
    import hal_only_prover as hop
    code, data, out = prover.witgen(seg)
    seal = hop.prove_segment_trait_only(hal, prover.circuit, seg.po2, code, data, out, prover.syn_accumulate(seg, data))
    return seal, prover.seal(seg, code, data, out).seal


@pytest.mark.parametrize("case", _large_cases(), ids=lambda c: f"{c['shape']}-po2-{c['po2']}")
def test_baseline_shape_stage_by_stage_and_whole_seal(hal, case):
    """BASELINE config 2's exact shape, under SYN-A and under the heavy constraint system.  The oracle sealed each segment
    once on the CPU (tests/golden/make_golden_large.py, minutes) and recorded a SHA-256 of every intermediate buffer; here the
    same pipeline runs through the C ABI op by op:
    witgen, iNTT + zk_shift (208 x 2^20), expand-NTT (208 x 2^20 -> 2^22), hash_rows (208 cols x 2^22 rows), the full
    Merkle fold, accum, eval_check (2^22 points), the check group, mix_poly_coeffs — and finally the whole seal."""
    from zeth_amd.circuits import syn_heavy
    desc = {"syn_a": syn_air.syn_a, "syn_heavy": syn_heavy.syn_heavy}[case["shape"]]()
    po2, zk = case["po2"], case["zk_cycles"]
    st = case["stages"]
    n, dom = 1 << po2, 4 << po2
    wa, wc, wd = (int(x) for x in desc[3:6])
    prover = SegmentProver(hal, desc)
    circ = prover.circuit
    seg = Segment(index=0, po2=po2, seed=case["seed"], noise_seed=case["noise_seed"], zk_cycles=zk)
    code, data, out = prover.witgen(seg)
    assert _sha(code) == st["trace.code"]["sha256"] and _sha(data) == st["trace.data"]["sha256"]
    co_code, ev_code = _commit_group_stagewise(hal, code, wc, n, st, "code")
    co_data, ev_data = _commit_group_stagewise(hal, data, wd, n, st, "data")
    mix_global = np.array(st["global.mix"]["values"], dtype=np.uint32)
    accum = hal.alloc_elem("accum", wa * n)
    hal.syn_accum(circ, po2, zk, seg.noise_seed, data, mix_global, accum)
    assert _sha(accum) == st["trace.accum"]["sha256"]
    co_accum, ev_accum = _commit_group_stagewise(hal, accum, wa, n, st, "accum")
    # eval_check on the 4n coset, then the check group (4 polys of 4n read as 16 of n)
    poly_mix = np.array(st["poly_mix"]["values"], dtype=np.uint32)
    check = hal.alloc_elem("check", 4 * dom)
    circ.eval_check(check, [ev_accum, ev_code, ev_data], [hal.copy_from("out", out), hal.copy_from("mix", mix_global)], poly_mix, po2)
    assert _sha(check) == st["check.evaluated"]["sha256"], "eval_check at the BASELINE size"
    del ev_accum, ev_code, ev_data
    hal.batch_interpolate_ntt(check, 4)
    ev_check = hal.alloc_elem("evaluated", 16 * dom)
    hal.batch_expand_into_evaluate_ntt(ev_check, check, 16, 2)
    assert _sha(ev_check) == st["evaluated.check"]["sha256"]
    nodes = hal.alloc_digest("nodes", 2 * dom)
    hal.hash_rows(nodes.slice(dom * 8, dom * 8), ev_check)
    hal.merkle_fold_all(nodes, dom)
    assert _sha(nodes.slice(8, (2 * dom - 1) * 8)) == st["nodes.check"]["sha256"]
    del ev_check, nodes
    # mix_poly_coeffs over the (bit-reversed) coefficient columns, then combos back to natural order
    c = Circuit.parse(desc)
    mix = np.array(st["mix"]["values"], dtype=np.uint32)
    lib = zko.load()

    def fp4_mul(a, b):
        o = np.zeros(4, np.uint32)
        lib.zko_fp4_mul(np.ascontiguousarray(a, dtype=np.uint32), np.ascontiguousarray(b, dtype=np.uint32), o)
        return o

    def fp4_pow(a, e):
        r = np.array([lib.zko_fp_encode(1), 0, 0, 0], np.uint32)
        while e:
            if e & 1:
                r = fp4_mul(r, a)
            a, e = fp4_mul(a, a), e >> 1
        return r

    combos = hal.alloc("combos", 4 * n * (len(c.combos) + 1), zero=True)
    cur = np.array([lib.zko_fp_encode(1), 0, 0, 0], np.uint32)
    for g, co in ((0, co_accum), (1, co_code), (2, co_data)):
        which = np.array([r[3] for r in c.regs if r[0] == g], dtype=np.uint32)
        hal.mix_poly_coeffs(combos, cur, mix, co, hal.copy_from("which", which), which.size, n)
        cur = fp4_mul(cur, fp4_pow(mix, which.size))
    which = np.full(16, len(c.combos), dtype=np.uint32)
    hal.mix_poly_coeffs(combos, cur, mix, check, hal.copy_from("which", which), 16, n)
    hal.batch_bit_reverse_extelem(combos, len(c.combos) + 1)
    assert _sha(combos) == st["combos.mixed"]["sha256"], "mix_poly_coeffs over all four groups"
    del combos, co_accum, co_code, co_data, check
    # ... and the whole seal, byte for byte
    receipt = prover.seal(seg, code, data, out)
    assert receipt.seal.size == case["seal_words"] and receipt.seal[:8].tolist() == case["seal_head"]
    assert hashlib.sha256(receipt.seal.astype("<u4").tobytes()).hexdigest() == case["seal_sha256"], \
        f"po2-{po2} {case['shape']} seal differs from the CPU oracle's"
    receipt.verify(desc, prover.control_root(po2, zk))


def test_prove_begin_can_be_aborted_and_rejects_bad_shapes(hal):
    desc = syn_air.syn_tiny()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=9, seed=5, noise_seed=6, zk_cycles=100)
    code, data, out = prover.witgen(seg)

    def boom(mix):
        raise RuntimeError("accum witgen failed")
    with pytest.raises(RuntimeError, match="accum witgen failed"):
        prover.seal_with_accum(seg, code, data, out, boom)                  # job aborted, nothing leaks
    with pytest.raises(HalError, match="wrong shape"):
        prover.seal_with_accum(seg, code, data, out, lambda mix: hal.alloc_elem("a", 8))
    with pytest.raises(HalError, match="commit_group"):
        prover.seal_with_accum(seg, data, data, out, prover.syn_accumulate(seg, data))
    bad_out = out.copy()
    bad_out[0] = 0xFFFFFFFF
    with pytest.raises(HalError, match="reduced"):
        prover.seal_with_accum(seg, code, data, bad_out, prover.syn_accumulate(seg, data))
    ok = prover.seal_with_accum(seg, code, data, out, prover.syn_accumulate(seg, data))
    assert np.array_equal(ok.seal, prover.seal(seg, code, data, out).seal)


# ---------------------------------------------------------------------------------------------------------------
# the code (control) group kept resident per segment size (zkh_prover_cache_code)
# ---------------------------------------------------------------------------------------------------------------
def test_resident_code_group_gives_byte_identical_seals(hal, oracle):
    """The code group is a function of (circuit, po2, zk_cycles): a prover that keeps its committed form resident must
    produce the seals a recomputing prover produces — across segments, sizes, a change of zk_cycles, and both entry points
    (zkh_prove_segment and the zkh_prove_begin / zkh_prove_finish halves)."""
    desc = syn_air.syn_small()
    plain = SegmentProver(hal, desc)
    resident = SegmentProver(hal, desc, resident_code_group=True)
    oc = zko.OracleCircuit(oracle, desc)
    cases = [(10, 100, 1), (10, 100, 2), (12, 200, 3), (10, 100, 4), (10, 300, 5), (12, 200, 6)]
    for po2, zk, seed in cases:
        seg = Segment(index=seed, po2=po2, seed=seed, noise_seed=77 + seed, zk_cycles=zk)
        want = plain.prove_segment(seg).seal
        code, data, out = resident.witgen(seg)
        got = resident.seal(seg, code, data, out).seal
        assert np.array_equal(got, want), f"resident code group changed the seal (po2={po2}, zk={zk}, seed={seed})"
        got2 = resident.seal_with_accum(seg, code, data, out, resident.syn_accumulate(seg, data)).seal
        assert np.array_equal(got2, want)
        assert resident._resident[po2] == zk
    assert np.array_equal(want, oc.prove(12, 200, 6, 77 + 6))       # and both equal the oracle's


def test_prove_begin_without_code_needs_a_resident_group(hal):
    import ctypes as C
    from zeth_amd import hal as zhal
    desc = syn_air.syn_small()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=9, seed=1, noise_seed=2, zk_cycles=100)
    code, data, out = prover.witgen(seg)
    job = C.c_void_p()
    outp = out.ctypes.data_as(C.POINTER(C.c_uint32))
    with pytest.raises(HalError, match="resident code group"):
        zhal._check(zhal._lib.zkh_prove_begin(prover.h, 9, None, data.h, outp, C.byref(job), None))
    zhal._check(zhal._lib.zkh_prover_cache_code(prover.h, 9, code.h))
    zhal._check(zhal._lib.zkh_prove_begin(prover.h, 9, None, data.h, outp, C.byref(job), None))
    zhal._lib.zkh_prove_abort(job)
    zhal._lib.zkh_prover_drop_code_cache(prover.h)
    with pytest.raises(HalError, match="resident code group"):
        zhal._check(zhal._lib.zkh_prove_begin(prover.h, 9, None, data.h, outp, C.byref(job), None))


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
