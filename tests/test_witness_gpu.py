"""Witness ingress on the GPU: host-produced traces (pinned uploads + the two-halves seal), the trace-driven witness (host preflight records -> row fill -> scatter),
and the blinding rows (csrc/noise.h: ChaCha12 keyed stream) — all word for word against the oracle."""
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



@pytest.mark.parametrize("shape,po2,zk", [("syn_small", 12, 1994), ("syn_a", 16, 1994)])
def test_host_witness_ingress_bit_exact(hal, oracle, shape, po2, zk):
    """Upstream's flow: preflight + witgen on the CPU, traces uploaded, sealed.  The oracle's witness generator plays the
    CPU witgen; the traces go through pinned memory + zkh_write_async and the two-halves seal; result == oracle seal."""
    desc = getattr(syn_air, shape)()
    oc = zko.OracleCircuit(oracle, desc)
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=po2, seed=0x77 + po2, noise_seed=0x99, zk_cycles=zk)
    ocode, odata, oout = oc.witgen(po2, zk, seg.seed, seg.noise_seed)
    hcode, hdata = hal.host_alloc(ocode.size), hal.host_alloc(odata.size)
    hcode[:] = ocode
    hdata[:] = odata
    try:
        receipt = prover.seal_host_witness(seg, hcode, hdata, oout)
        hal.sync()
    finally:
        hal.host_free(hcode)
        hal.host_free(hdata)
    want = oc.prove(po2, zk, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want)
    # the device-witness entry point gives the same bytes
    assert np.array_equal(prover.prove_segment(seg).seal, want)
    with pytest.raises(HalError, match="zkh_host_alloc"):
        hal.write_async(hal.alloc_elem("x", 16), np.zeros(16, np.uint32))   # pageable memory is refused


@pytest.mark.parametrize("shape,po2", [("syn_a", 13), ("wd21", 12), ("syn_small", 12)])
def test_trace_driven_witness_equals_the_oracles(hal, oracle, shape, po2):
    """Row f1: host preflight (sequential machine, 16 bytes per cycle) -> upload -> k_syn_rowfill + scan + the preload through
    zkh_scatter gives the oracle's data group word for word (oracle/preflight.c), and the seal of those traces is the oracle's."""
    import zko
    from zeth_amd import hal as H
    from zeth_amd.prover import Segment, SegmentProver
    desc = {"syn_a": syn_air.syn_a, "syn_small": syn_air.syn_small, "wd21": lambda: syn_air.build_syn_air(8, 21, 8)}[shape]()
    oc = zko.OracleCircuit(oracle, desc)
    seed, noise = 0x5EED0000 + po2, 0x2E80
    rec, ram, secs = H.syn_preflight(seed, po2)
    orec, oram = oc.preflight(seed, po2)
    assert np.array_equal(rec, orec) and np.array_equal(ram, oram) and secs > 0 and (rec < 2013265921).all()
    sp = SegmentProver(hal, desc)
    wa, wc, wd = sp.group_sizes()
    n = 1 << po2
    pinned = hal.host_alloc(rec.size)                        # the ingress path proper: pinned memory + an enqueued upload
    pinned[:] = rec
    drec = hal.alloc("records", rec.size)
    hal.write_async(drec, pinned)
    code, data = hal.alloc_elem("code", wc * n), hal.alloc_elem("data", wd * n)
    out = hal.syn_witgen_trace(sp.circuit, po2, 1994, noise, drec, ram, code, data)
    ocode, odata, oout = oc.witgen_trace(po2, rec, ram, noise)
    assert np.array_equal(data.to_vec(), odata) and np.array_equal(code.to_vec(), ocode) and np.array_equal(out, oout)
    T = (wd - 2) // 3
    if wd - 2 > 3 * T:                                        # the preload landed: the first unconstrained column holds the RAM image
        assert np.array_equal(odata[3 * T * n: 3 * T * n + 1024], ram)
    seg = Segment(index=0, po2=po2, seed=seed, noise_seed=noise)
    got = sp.seal(seg, code, data, out)
    want = oc.prove_traces(po2, ocode, odata, oout, noise_seed=noise)
    assert np.array_equal(got.seal, want)
    got.verify(desc, sp.control_root(po2))
    hal.sync()
    hal.host_free(pinned)


def test_blinding_rows_are_the_keyed_chacha12_stream_and_the_default_key_is_fresh(hal, oracle):
    """csrc/noise.h on the device: rows >= A of the data and accum groups are ChaCha12(key; (row, column), (group, "ZKN1")) folded mod P —
    equal to the host twin (zkh_noise_cell_host, itself pinned by RFC 8439's vector and the oracle: tests/test_noise.py) cell for cell;
    nothing else of the witness depends on the key; and with no key given (the product default) every call draws a fresh one from the OS."""
    import ctypes as C
    from zeth_amd import hal as zhal
    lib = zhal.load_library()
    desc = syn_air.syn_small()
    prover = SegmentProver(hal, desc)
    po2, zk = 12, 300
    n, A = 1 << po2, (1 << po2) - zk
    big = int.from_bytes(bytes(range(3, 35)), "little")                       # a full 256-bit key
    seg = Segment(index=0, po2=po2, seed=0x5EED0099, noise_seed=big, zk_cycles=zk)
    code, data, out = prover.witgen(seg)
    wd = prover.group_sizes()[2]
    d = data.to_vec().reshape(wd, n)
    key = zhal.noise_key(big)
    kp = key.ctypes.data_as(C.POINTER(C.c_uint32))
    for c in (0, 1, wd // 2, wd - 1):
        want = np.array([lib.zkh_noise_cell_host(kp, 2, c, r) for r in range(A, n)], dtype=np.uint32)
        assert np.array_equal(d[c, A:], want), f"data column {c}"
    assert np.array_equal(d[:, A:], zko.OracleCircuit(oracle, desc).witgen(po2, zk, seg.seed, big)[1].reshape(wd, n)[:, A:])
    _, data2, out2 = prover.witgen(Segment(index=0, po2=po2, seed=seg.seed, noise_seed=big + 1, zk_cycles=zk))
    d2 = data2.to_vec().reshape(wd, n)
    assert np.array_equal(d[:, :A], d2[:, :A]) and np.array_equal(out, out2) and not np.array_equal(d[:, A:], d2[:, A:])
    # the product default: no key -> 256 fresh bits from getrandom per call; the seals differ, both verify
    root = prover.control_root(po2, zk)
    a = prover.prove_segment(Segment(index=0, po2=po2, seed=seg.seed, zk_cycles=zk))
    b = prover.prove_segment(Segment(index=0, po2=po2, seed=seg.seed, zk_cycles=zk))
    assert not np.array_equal(a.seal, b.seal) and np.array_equal(a.seal[:4], b.seal[:4])
    for r in (a, b):
        r.verify(desc, root)
    _, dn, _ = prover.witgen(Segment(index=0, po2=po2, seed=seg.seed, noise_seed=0, zk_cycles=zk))      # 0 / None = NULL at the ABI = OS key
    _, dm, _ = prover.witgen(Segment(index=0, po2=po2, seed=seg.seed, noise_seed=0, zk_cycles=zk))
    assert not np.array_equal(dn.to_vec().reshape(wd, n)[:, A:], dm.to_vec().reshape(wd, n)[:, A:])
