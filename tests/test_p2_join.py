"""P2-JOIN: joins that hash their children's claims IN-CIRCUIT — Poseidon2 unrolled over trace rows.

Upstream's recursion circuit (risc0-circuit-recursion 4.0.2, un-vendored: /root/reference/Cargo.lock:5305; BASELINE.json
config 5) cannot be obtained offline; what is public is the hash it evaluates.  Pinned OUTSIDE the repository: the row
form of the permutation every witness generator is compared with reproduces the PUBLISHED known-answer vector
(tests/golden/poseidon2_kat.json), and block 0 of every witness is `hash_pair` as the Merkle trees compute it."""
import json
import os

import numpy as np
import pytest

import zko
from zeth_amd.circuits import p2_join as J
from zeth_amd.circuits.desc import Circuit

PO2, ZK = 13, 1994
G = os.path.join(os.path.dirname(__file__), "golden")


def _children(seed=3):
    return np.random.default_rng(seed).integers(0, J.P, size=16, dtype=np.uint64).astype(np.uint32)


def test_row_form_of_the_permutation_reproduces_the_published_vector():
    with open(os.path.join(G, "poseidon2_kat.json")) as fh:
        k = json.load(fh)
    rows = J.block_rows(k["input"])
    assert len(rows) == 31 and rows[-1][0] == [int(x, 16) for x in k["output_hex"]]
    assert [J.round_kind(i) for i in (0, 1, 4, 5, 25, 26, 29, 30)] == ["in", "full", "full", "partial", "partial", "full", "full", "out"]
    c = Circuit.parse(J.p2_join_circuit())
    assert c.group_sizes == (4, 43, 48) and c.global_sizes == (24, 4) and c.kind == 3


def test_oracle_witness_rows_hash_pair_and_seal(oracle):
    from zeth_amd import host
    from zeth_amd.hal import HalError, HostCircuit
    desc = J.p2_join_circuit()
    oc = zko.OracleCircuit(oracle, desc)
    kids = _children()
    code, data, out = oc.witgen(PO2, ZK, noise_seed=0x2E80, pub=kids)
    n = 1 << PO2
    d, cg = data.reshape(48, n), code.reshape(43, n)
    enc = lambda v: [int(x) * J.R % J.P for x in v]
    rows = J.block_rows([int(w) * J.RINV % J.P for w in kids] + [0] * 8)
    for k in range(31):
        assert list(d[:24, k]) == enc(rows[k][0]) and list(d[24:, k]) == enc(rows[k][1]), f"row {k} of block 0"
    parent = np.zeros(8, np.uint32)
    oracle.zko_hash_pair(np.ascontiguousarray(kids[:8]), np.ascontiguousarray(kids[8:]), parent)     # the Merkle trees' hash_pair
    assert np.array_equal(out[:8], parent) and np.array_equal(out[8:], kids)
    assert np.array_equal(host.hash_pair(kids[:8], kids[8:]), parent)                                # the product's host path
    assert list(parent) == J.hash_pair_words(kids[:8], kids[8:])
    # block 7: parent ‖ the public sibling words of its input row ‖ 0
    r0 = 31 * 7
    sib = cg[35:43, r0]
    assert sib.any() and np.array_equal(d[:8, r0], parent) and np.array_equal(d[8:16, r0], sib) and not d[16:24, r0].any()
    want = np.zeros(8, np.uint32)
    oracle.zko_hash_pair(np.ascontiguousarray(parent), np.ascontiguousarray(sib), want)
    assert np.array_equal(d[:8, r0 + 30], want)
    # the seal: accepted by both verifiers, rejected when the parent claim is forged
    seal = oc.prove(PO2, ZK, 0, 0x2E80, pub=kids)
    root = oc.control_root(PO2, ZK)
    assert oc.verify(seal) is None
    HostCircuit(desc).verify_segment(seal, root)
    bad = seal.copy()
    bad[0] = (int(bad[0]) + 1) % J.P
    assert oc.verify(bad) is not None
    with pytest.raises(HalError):
        HostCircuit(desc).verify_segment(bad, root)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("po2", [13, 18])
def test_gpu_join_witness_and_seal_are_byte_identical_to_the_oracles(hal, oracle, po2):
    from zeth_amd.prover import Segment, SegmentProver
    desc = J.p2_join_circuit()
    prover = SegmentProver(hal, desc)
    assert prover.circuit.has_compiled_kernel()
    kids = _children(po2)
    seg = Segment(index=0, po2=po2, seed=0, noise_seed=0x2E80, pub=tuple(int(x) for x in kids))
    oc = zko.OracleCircuit(oracle, desc)
    gcode, gdata, gout = prover.witgen(seg)
    code, data, out = oc.witgen(po2, ZK, noise_seed=0x2E80, pub=kids)
    assert np.array_equal(gcode.to_vec(), code) and np.array_equal(gdata.to_vec(), data) and np.array_equal(gout, out)
    rec = prover.seal(seg, gcode, gdata, gout)
    assert np.array_equal(rec.seal, oc.prove(po2, ZK, 0, 0x2E80, pub=kids))
    root = prover.control_root(po2)
    assert np.array_equal(root, oc.control_root(po2, ZK))
    rec.verify(desc, root)
    rec.verify(desc)                                                   # the shipped control-root table
    assert oc.verify(rec.seal, root) is None


@pytest.mark.gpu
def test_succinct_receipt_needs_only_the_root_and_the_leaves(hal):
    """BASELINE config 5 restated: leaves sealed on the GPU, the P2-JOIN tree proven on the GPU, the joins below the root
    dropped; the verifier recomputes the claim tree from the leaves and follows it to the root receipt's output."""
    from zeth_amd.circuits import syn_air
    from zeth_amd.hal import HalError
    from zeth_amd.host import SuccinctReceipt, fold_claims, node_claim, prove_succinct, receipt_claim
    from zeth_amd.prover import Segment, SegmentProver, SegmentReceipt
    ldesc, jdesc = syn_air.syn_small(), J.p2_join_circuit()
    lp, jp = SegmentProver(hal, ldesc), SegmentProver(hal, jdesc)
    leaves = [lp.prove_segment(Segment(index=i, po2=13, seed=900 + i, noise_seed=5)) for i in range(7)]
    lroot, jroot = lp.control_root(13), jp.control_root(13)

    def claim_of(r, is_leaf):
        return node_claim(r, ldesc if is_leaf else jdesc, lroot if is_leaf else jroot, is_leaf)

    rec = prove_succinct(leaves, jp.prove_segment, claim_of, join_po2=13, noise_seed=11)
    assert [len(lvl) for lvl in rec.joins] == [3, 2, 1]
    small = rec.compact()
    small.verify(ldesc, jdesc, lroot, jroot)
    small.verify(ldesc, jdesc)                                         # shipped control roots
    assert np.array_equal(small.root.seal[:8], fold_claims([receipt_claim(r, ldesc, lroot) for r in leaves]))
    rec.verify(ldesc, jdesc, lroot, jroot)                             # kept joins are checked as well
    other = lp.prove_segment(Segment(index=3, po2=13, seed=1, noise_seed=5))
    with pytest.raises(ValueError, match="claim tree"):
        SuccinctReceipt(small.root, [], leaves[:3] + [other] + leaves[4:]).verify(ldesc, jdesc, lroot, jroot)
    forged = SegmentReceipt(seal=small.root.seal.copy(), index=small.root.index, po2=small.root.po2)
    forged.seal[9] = (int(forged.seal[9]) + 1) % J.P
    with pytest.raises((HalError, ValueError)):
        SuccinctReceipt(forged, [], leaves).verify(ldesc, jdesc, lroot, jroot)


@pytest.mark.gpu
def test_smallest_join_has_only_the_parent_block(hal, oracle):
    """Edge case: po2 11 leaves 54 active rows = ONE block (the parent hash, no sibling blocks) + 23 idle rows."""
    from zeth_amd.prover import Segment, SegmentProver
    desc = J.p2_join_circuit()
    prover = SegmentProver(hal, desc)
    kids = _children(11)
    rec = prover.prove_segment(Segment(index=0, po2=11, seed=0, noise_seed=9, pub=tuple(int(x) for x in kids)))
    oc = zko.OracleCircuit(oracle, desc)
    assert np.array_equal(rec.seal, oc.prove(11, ZK, 0, 9, pub=kids))
    assert list(rec.seal[:8]) == J.hash_pair_words(kids[:8], kids[8:])
    rec.verify(desc, prover.control_root(11))


@pytest.mark.gpu
def test_keccak_assumption_receipts_are_leaves_of_the_claim_tree(hal):
    """A block whose guest called the keccak accelerator: its segment receipts AND the KECCAK-F receipt of the accelerator batch
    fold into one root receipt (upstream resolves assumptions during lift / join); the compact receipt verifies each kind of
    leaf with its own circuit and control root and follows the claim tree to the root."""
    import hashlib
    from zeth_amd.circuits import keccak_f, syn_air
    from zeth_amd.hal import fp_decode
    from zeth_amd.host import SuccinctReceipt, node_claim, prove_succinct, receipt_claim
    from zeth_amd.prover import Segment, SegmentProver
    ldesc, kdesc, jdesc = syn_air.syn_small(), keccak_f.keccak_f_circuit(), J.p2_join_circuit()
    lp, kp, jp = SegmentProver(hal, ldesc), SegmentProver(hal, kdesc), SegmentProver(hal, jdesc)
    segs = [lp.prove_segment(Segment(index=i, po2=13, seed=60 + i, noise_seed=2)) for i in range(3)]
    msg = b"accelerator batch of block 19000000"
    pub = tuple(w for lane in keccak_f.sha3_256_block(msg) for w in (lane & 0xFFFFFFFF, lane >> 32))
    krec = kp.prove_segment(Segment(index=3, po2=13, seed=9, noise_seed=2, pub=pub))
    lroot, kroot, jroot = lp.control_root(13), kp.control_root(13), jp.control_root(13)
    leaves = segs + [krec]

    def claim_of(r, is_leaf):
        if not is_leaf:
            return node_claim(r, jdesc, jroot, False)
        return receipt_claim(r, kdesc, kroot) if r is krec else receipt_claim(r, ldesc, lroot)

    rec = prove_succinct(leaves, jp.prove_segment, claim_of, join_po2=13, noise_seed=4)
    small = SuccinctReceipt(rec.root, [], leaves, n_assumptions=1)
    small.verify(ldesc, jdesc, lroot, jroot, assumption_desc=kdesc, assumption_root=kroot)
    small.verify(ldesc, jdesc, assumption_desc=kdesc)                     # shipped control roots for all three circuits
    limbs = [fp_decode(int(w)) for w in krec.seal[:100]]
    assert keccak_f.digest_of_state([sum(limbs[4 * l + j] << (16 * j) for j in range(4)) for l in range(25)]) == hashlib.sha3_256(msg).digest()
    with pytest.raises(ValueError, match="assumption"):
        small.verify(ldesc, jdesc, lroot, jroot)
    # the assumption leaf treated as a segment of the main circuit is rejected (wrong circuit)
    with pytest.raises(Exception):
        SuccinctReceipt(rec.root, [], leaves, n_assumptions=0).verify(ldesc, jdesc, lroot, jroot)
