"""Join trees on the GPU: P2-JOIN joins commit to their children's claims; keccak assumption receipts ride in the composite."""
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



def test_join_tree_commits_to_children(hal, oracle):
    """BASELINE config 5 restated: 5 leaf segments -> 4 SYN-J joins in 3 dependent levels -> one root.  Every join takes
    the claim digests of its two children as public inputs (bound to its `out` globals by constraints); the succinct
    receipt verifies only if every seal is accepted AND every join commits to the receipts actually below it."""
    from zeth_amd.host import prove_succinct, receipt_claim
    leaf_desc, join_desc = syn_air.syn_small(), syn_air.build_syn_air(8, 64, 8, n_pub=16)
    leaf_prover, join_prover = SegmentProver(hal, leaf_desc), SegmentProver(hal, join_desc)
    lp, jp, zk = 11, 10, 500
    leaves = [leaf_prover.prove_segment(Segment(index=i, po2=lp, seed=0x5EED0000 + i, noise_seed=7, zk_cycles=zk)) for i in range(5)]
    leaf_root, join_root = leaf_prover.control_root(lp, zk), join_prover.control_root(jp, zk)

    def claim_of(rec, is_leaf):
        return receipt_claim(rec, leaf_desc if is_leaf else join_desc, leaf_root if is_leaf else join_root)

    calls = []

    def prove_join(seg):
        calls.append(seg)
        return join_prover.prove_segment(Segment(index=seg.index, po2=seg.po2, seed=seg.seed, noise_seed=8, zk_cycles=zk, pub=seg.pub))

    rec = prove_succinct(leaves, prove_join, claim_of, join_po2=jp)
    assert [len(lvl) for lvl in rec.joins] == [2, 1, 1] and len(calls) == 4
    assert list(calls[0].pub) == [*claim_of(leaves[0], True), *claim_of(leaves[1], True)]
    assert list(calls[3].pub[8:]) == list(claim_of(leaves[4], True))       # the odd leaf is carried up two levels
    assert rec.root is rec.joins[2][0]
    rec.verify(leaf_desc, join_desc, leaf_root, join_root)
    # the oracle seals the same join byte for byte (public inputs included), and the claim is the Poseidon2 of header + root
    oc = zko.OracleCircuit(oracle, join_desc)
    want = oc.prove(jp, zk, calls[0].seed, 8, pub=np.array(calls[0].pub, np.uint32))
    assert np.array_equal(rec.joins[0][0].seal, want)
    hdr = np.concatenate([leaves[0].seal[:5], leaf_root]).astype(np.uint32)
    dg = np.zeros(8, np.uint32)
    oracle.zko_hash_elem_slice(hdr, hdr.size, 1, dg)
    assert np.array_equal(claim_of(leaves[0], True), dg)
    # swapping two leaves keeps every seal valid but breaks the commitment chain
    rec.leaves[0], rec.leaves[1] = rec.leaves[1], rec.leaves[0]
    with pytest.raises(ValueError, match="claims of its children"):
        rec.verify(leaf_desc, join_desc, leaf_root, join_root)
    rec.leaves[0], rec.leaves[1] = rec.leaves[1], rec.leaves[0]
    rec.joins[1][0].seal[100] ^= 1
    with pytest.raises(HalError):
        rec.verify(leaf_desc, join_desc, leaf_root, join_root)


def test_keccak_assumption_receipts_ride_in_the_composite(hal):
    """Row f4: the guest's keccak accelerator calls are proven by a third circuit — KECCAK-F, real keccak-f[1600] permutations
    (tests/test_keccak_circuit.py holds its parity and SHA-3 known-answer tests) — whose receipts ride in the composite as
    assumption receipts and are verified with their own circuit + control root (upstream: `prove_keccak`,
    risc0-circuit-keccak 4.0.2, /root/reference/Cargo.lock:5289)."""
    import hashlib
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import fp_decode
    from zeth_amd.host import BlockProcessor
    kdesc, sdesc = keccak_f.keccak_f_circuit(), syn_air.syn_small()
    kprover, sprover = SegmentProver(hal, kdesc), SegmentProver(hal, sdesc)
    msg = b"assumption: one accelerator batch"
    pub = tuple(w for lane in keccak_f.sha3_256_block(msg) for w in (lane & 0xFFFFFFFF, lane >> 32))
    krec = kprover.prove_segment(Segment(index=0, po2=13, seed=0xCECC, noise_seed=3, pub=pub))
    limbs = [fp_decode(int(w)) for w in krec.seal[:100]]
    assert keccak_f.digest_of_state([sum(limbs[4 * l + j] << (16 * j) for j in range(4)) for l in range(25)]) == hashlib.sha3_256(msg).digest()
    segs = [Segment(index=i, po2=13, seed=40 + i, noise_seed=9) for i in range(2)]
    comp = BlockProcessor(sprover.prove_segment).prove(segs)
    comp.assumptions.append(krec)
    comp.verify(sdesc, sprover.control_root, kdesc, kprover.control_root(13))
    comp.verify(sdesc, sprover.control_root, kdesc)                    # ... and against the shipped control-root table
    with pytest.raises(ValueError, match="assumption"):
        comp.verify(sdesc, sprover.control_root)
    with pytest.raises(HalError):
        comp.verify(sdesc, sprover.control_root, kdesc, sprover.control_root(13))
