"""The product's host-side verifier (zkh_verify_segment: the analogue of `receipt.verify`, cli.rs:103) needs no GPU.
It must accept what the oracle's prover produces, agree with the oracle's independent verifier on every tampered seal,
and fail with error strings (risc0-sys convention), never crash, on malformed input."""
import numpy as np
import pytest

import zko
from zeth_amd.circuits import syn_air
from zeth_amd.hal import HalError, HostCircuit


@pytest.mark.parametrize("shape,po2,zk", [("syn_tiny", 9, 100), ("syn_tiny", 12, 1994), ("syn_small", 13, 1994)])
def test_accepts_oracle_seals(oracle, shape, po2, zk):
    desc = getattr(syn_air, shape)()
    oc = zko.OracleCircuit(oracle, desc)
    seal = oc.prove(po2, zk, 7, 9)
    HostCircuit(desc).verify_segment(seal, oc.control_root(po2, zk))


def test_agrees_with_oracle_verifier_on_tampering(oracle):
    desc = syn_air.syn_tiny()
    oc = zko.OracleCircuit(oracle, desc)
    hc = HostCircuit(desc)
    seal = oc.prove(10, 300)
    root = oc.control_root(10, 300)
    hc.verify_segment(seal, root)
    rng = np.random.default_rng(1)
    for pos in [0, 3, 4, 5, 200, 300, 1100, seal.size // 2, seal.size - 1, *rng.integers(0, seal.size, size=40)]:
        bad = seal.copy()
        bad[pos] ^= 1 << int(rng.integers(0, 31))
        assert oc.verify(bad, root) is not None
        with pytest.raises(HalError, match="verify_segment"):
            hc.verify_segment(bad, root)
    for cut in (1, 8, 100, seal.size - 4):
        with pytest.raises(HalError, match="truncated|trailing|mismatch|range|po2"):
            hc.verify_segment(seal[:-cut], root)
    with pytest.raises(HalError, match="trailing"):
        hc.verify_segment(np.concatenate([seal, np.zeros(3, np.uint32)]), root)
    with pytest.raises(HalError):
        HostCircuit(syn_air.syn_small()).verify_segment(seal, root)        # wrong circuit
    with pytest.raises(HalError, match="po2|truncated"):
        hc.verify_segment(np.zeros(5, np.uint32), root)


def test_constraint_violation_is_rejected(oracle):
    """Same scenario as the oracle test: a circuit whose selector constraint is violated by the witness."""
    from zeth_amd.circuits.desc import Circuit
    desc = syn_air.syn_tiny().copy()
    c = Circuit.parse(desc)
    pos = 16 + 3 * len(c.taps) + sum(1 + len(cb) for cb in c.combos)
    desc[pos + 1] = 2
    oc = zko.OracleCircuit(oracle, desc)
    seal = oc.prove(9, 100)
    with pytest.raises(HalError, match="constraint check failed"):
        HostCircuit(desc).verify_segment(seal, oc.control_root(9, 100))


def test_custom_poseidon2_tables(oracle):
    """The verifier takes the hash tables as data too."""
    import re, os
    desc = syn_air.syn_tiny()
    rng = np.random.default_rng(5)
    P = 2013265921
    rc = rng.integers(0, P, size=24 * 29, dtype=np.uint64).astype(np.uint32)
    diag = rng.integers(1, P, size=24, dtype=np.uint64).astype(np.uint32)
    oc = zko.OracleCircuit(oracle, desc)
    base = oc.prove(9, 100)
    try:
        oracle.zko_poseidon2_set_constants(rc, diag)
        seal = oc.prove(9, 100)
        assert not np.array_equal(seal[5:40], base[5:40])
        hc = HostCircuit(desc)
        root = np.zeros(8, np.uint32)
        oracle.zko_control_root(oc.h, 9, 100, root)          # under the custom tables (not the cached one)
        hc.verify_segment(seal, root, rc, diag)
        with pytest.raises(HalError):
            hc.verify_segment(seal, root)               # shipped tables: different transcript
    finally:
        txt = open(os.path.join(os.path.dirname(__file__), "..", "include", "zkh_poseidon2_consts.h")).read()
        nums = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)]
        oracle.zko_poseidon2_set_constants(np.array(nums[24:], np.uint32), np.array(nums[:24], np.uint32))


def test_receipt_container_roundtrip_and_integrity(oracle):
    """The versioned word container around a seal (zkh_receipt_encode / zkh_receipt_decode): round trip, then every kind of
    damage is refused — flipped payload, truncated blob, wrong circuit, tampered envelope fields."""
    from zeth_amd.prover import SegmentReceipt
    desc = syn_air.syn_tiny()
    oc = zko.OracleCircuit(oracle, desc)
    seal, root = oc.prove(10, 300), oc.control_root(10, 300)
    rec = SegmentReceipt(seal=seal, index=7, po2=10, output=seal[:4].copy())
    blob = rec.to_words(desc, root)
    assert blob[0] == 0x31524B5A and blob.size == seal.size + 28
    back = SegmentReceipt.from_words(desc, blob)
    assert back.index == 7 and back.po2 == 10 and np.array_equal(back.seal, seal) and np.array_equal(back.control_root, root)
    back.verify(desc, root)
    hc = HostCircuit(desc)
    hdr, _ = hc.receipt_decode(blob)
    assert np.array_equal(hdr["claim"], hc.receipt_claim(seal, root)) and hdr["tables"] == "derived" and not hdr["placeholder_tables"]
    for pos in (0, 4, 9, 12, 20, 26, 500, blob.size - 1):
        bad = blob.copy()
        bad[pos] ^= 1
        with pytest.raises(HalError, match="receipt_decode"):
            hc.receipt_decode(bad)
    with pytest.raises(HalError, match="receipt_decode"):
        hc.receipt_decode(blob[:-3])
    with pytest.raises(HalError, match="another circuit"):
        HostCircuit(syn_air.syn_small()).receipt_decode(blob)
    # an envelope that lies about the control root is caught by the claim digest even when its checksum is recomputed
    forged = blob.copy()
    forged[10] ^= 1
    h = 0xCBF29CE484222325
    for b in forged[:-2].astype("<u4").tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    forged[-2], forged[-1] = h & 0xFFFFFFFF, h >> 32
    with pytest.raises(HalError, match="claim digest"):
        hc.receipt_decode(forged)


def test_chained_session_is_continuous_and_tampering_breaks_the_chain(oracle):
    """Claim continuity (SYN-C: SYN-A's family with the segment's pre-state as its one public input; out = (post, 0, 0, 0, pre)).
    The executor's pass fixes every segment's pre-state, the segments are proven independently (here by the oracle), and
    `CompositeReceipt.verify(chained=True)` — upstream's verify_integrity: pre == prev.post — accepts the session; a receipt
    taken out, two receipts swapped, a receipt of another session spliced in, or a wrong initial state are refused."""
    import zko
    from dataclasses import replace
    from zeth_amd.circuits import syn_air
    from zeth_amd.hal import P, fp_encode
    from zeth_amd.host import CompositeReceipt, chain_segments
    from zeth_amd.prover import Segment, SegmentReceipt
    desc = syn_air.syn_chain_small()
    oc = zko.OracleCircuit(oracle, desc)
    po2, zk = 12, 1994
    root = oc.control_root(po2, zk)

    def contribution(seg):
        return int(oc.witgen(seg.po2, zk, seg.seed, seg.noise_seed, pub=np.zeros(1, np.uint32))[2][0])

    def prove(seg):
        seal = oc.prove(seg.po2, zk, seg.seed, seg.noise_seed, pub=np.asarray(seg.pub, dtype=np.uint32))
        return SegmentReceipt(seal=seal, index=seg.index, po2=seg.po2, output=seal[:5].copy())
    base = [Segment(index=i, po2=po2, seed=700 + i, noise_seed=0x61) for i in range(4)]
    chained = chain_segments(base, contribution, initial_state=7)
    assert chained[0].pub == (fp_encode(7),) and all(len(s.pub) == 1 for s in chained)
    recs = [prove(s) for s in chained]
    comp = CompositeReceipt(recs)
    comp.verify(desc, root, chained=True, initial_state=7)
    # the chain really is the running sum: post(i) = pre(i) + contribution(i), and the last post is the session's final state
    for s, r in zip(chained, recs):
        assert int(r.seal[4]) == s.pub[0] and int(r.seal[0]) == (s.pub[0] + contribution(s)) % P
    assert comp.final_state() == int(recs[-1].seal[0])
    with pytest.raises(ValueError, match="not continuous"):
        comp.verify(desc, root, chained=True, initial_state=8)                      # wrong initial state
    swapped = [replace(recs[0]), replace(recs[2], index=1), replace(recs[1], index=2), replace(recs[3])]
    with pytest.raises(ValueError, match="not continuous"):
        CompositeReceipt(swapped).verify(desc, root, chained=True, initial_state=7)
    other = prove(chain_segments([Segment(index=0, po2=po2, seed=999, noise_seed=0x61)], contribution, initial_state=7)[0])
    spliced = [recs[0], replace(other, index=1), recs[2], recs[3]]
    with pytest.raises(ValueError, match="not continuous"):
        CompositeReceipt(spliced).verify(desc, root, chained=True, initial_state=7)
    with pytest.raises(ValueError, match="missing or unordered"):
        CompositeReceipt([recs[0], recs[2], recs[3]]).verify(desc, root, chained=True, initial_state=7)
    # the reference's flow (lib.rs:123-143 then cli.rs:103-107): prove -> (receipt, image id); receipt.verify(image_id); journal compare
    from zeth_amd.hal import HalError, fp_decode
    from zeth_amd.host import Receipt, image_id, prove_chained_block
    receipt, iid = prove_chained_block(prove, contribution, desc, base, initial_state=7)
    receipt.verify(iid, desc, initial_state=7, control_root=root, n_segments=4)
    with pytest.raises(HalError, match="does not bind its termination"):       # SYN-C seals carry no exit code: the verifier must say how long the session is
        receipt.verify(iid, desc, initial_state=7, control_root=root)
    # ... because a holder can cut the session short and rewrite the 4-byte journal: only the expected segment count refuses that
    cut = Receipt(CompositeReceipt(recs[:3]), int(fp_decode(int(recs[2].seal[0]))).to_bytes(4, "little"))
    with pytest.raises(HalError, match="holds 3 segments, the session has 4"):
        cut.verify(iid, desc, initial_state=7, control_root=root, n_segments=4)
    assert receipt.journal == int(fp_decode(comp.final_state())).to_bytes(4, "little") and np.array_equal(iid, image_id(desc, 7))
    with pytest.raises(HalError, match="image id"):
        receipt.verify(image_id(desc, 8), desc, initial_state=7, control_root=root, n_segments=4)           # another program / initial state
    with pytest.raises(HalError, match="journal"):
        Receipt(receipt.inner, b"\x00\x00\x00\x01").verify(iid, desc, initial_state=7, control_root=root, n_segments=4)
    with pytest.raises(ValueError, match="not continuous"):
        Receipt(CompositeReceipt(swapped), receipt.journal).verify(iid, desc, initial_state=7, control_root=root, n_segments=4)
    # without the chain check the same receipts are individually valid: continuity is a property of the SESSION
    CompositeReceipt(swapped).verify(desc, root)
    # and a forged pre-state inside a seal is refused by the seal verification itself (the state words are bound public inputs)
    forged = recs[1].seal.copy()
    forged[4] = recs[0].seal[4]
    with pytest.raises(Exception):
        SegmentReceipt(seal=forged, index=1, po2=po2, output=forged[:5].copy()).verify(desc, root)
