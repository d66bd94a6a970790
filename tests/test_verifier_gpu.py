"""The verifier's binding to the control root on the GPU box: product, oracle and the shipped table agree; a forged seal over another code group is refused."""
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



def test_control_roots_product_oracle_and_shipped_table_agree(hal, oracle):
    for shape, po2s in (("syn_tiny", (9, 13)), ("syn_small", (12, 14)), ("syn_a", (13, 16))):
        desc = getattr(syn_air, shape)()
        prover, oc = SegmentProver(hal, desc), zko.OracleCircuit(oracle, desc)
        for po2 in po2s:
            zk = 100 if po2 < 12 else 1994
            assert np.array_equal(prover.control_root(po2, zk), oc.control_root(po2, zk))
            shipped = shipped_control_root(desc, po2)
            if shipped is not None and zk == 1994:
                assert np.array_equal(shipped, prover.control_root(po2, zk)), "zeth_amd/circuits/control_roots.json is stale"


def test_forged_output_with_zeroed_code_is_rejected(hal, oracle):
    """The attack the round-1 verifier missed: with an all-zero code group every selector-gated constraint is switched
    off and the ungated sanity constraints hold trivially, so ANY `out` global can be 'proven'.  The seal is internally
    consistent — it is accepted against the code root the forger committed to — and must be rejected against the control root."""
    desc = syn_air.syn_small()
    po2, zk = 12, 1994
    wa, wc, wd = (int(x) for x in desc[3:6])
    n = 1 << po2
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=po2, seed=1, noise_seed=2, zk_cycles=zk)
    _, data, _ = prover.witgen(seg)
    zero_code = hal.alloc("code", wc * n, zero=True)
    forged_out = np.array([zko.load().zko_fp_encode(0xBADC0DE), 0, 0, 0], dtype=np.uint32)
    forged = prover.seal_with_accum(seg, zero_code, data, forged_out, prover.syn_accumulate(seg, data))
    assert np.array_equal(forged.seal[:4], forged_out)
    forger_root = prover.code_root(zero_code, po2)
    forged.verify(desc, forger_root)                                        # self-consistent ...
    assert zko.OracleCircuit(oracle, desc).verify(forged.seal, forger_root) is None
    with pytest.raises(HalError, match="control root"):                     # ... but not the registered program
        forged.verify(desc, prover.control_root(po2, zk))
    assert "control root" in zko.OracleCircuit(oracle, desc).verify(forged.seal, prover.control_root(po2, zk))
    with pytest.raises(HalError, match="no control root"):
        from zeth_amd.hal import HostCircuit
        HostCircuit(desc).verify_segment(forged.seal, None)


def test_version_names_the_provenance_of_the_poseidon2_tables():
    from zeth_amd import hal as zhal
    v = zhal.load_library().zkh_version().decode()
    assert "gfx950" in v and "poseidon2_consts=derived" in v
