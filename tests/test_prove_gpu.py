"""Whole-seal parity: HIP prover (through the C ABI) vs the CPU oracle prover, byte for byte, plus the independent
verifier restatement as the acceptance test (the analogue of cli.rs:103 `receipt.verify`)."""
import ctypes as C

import numpy as np
import pytest

import zko
from conftest import rand_fp
from zeth_amd.circuits import syn_air
from zeth_amd.prover import Segment, SegmentProver

pytestmark = pytest.mark.gpu

SHAPES = {"tiny": syn_air.syn_tiny, "small": syn_air.syn_small, "syn_a": syn_air.syn_a}


def _witness_parity(hal, oracle, desc, po2, zk):
    circ = hal.load_circuit(desc)
    oc = zko.OracleCircuit(oracle, desc)
    wa, wc, wd = (int(x) for x in desc[3:6])
    n = 1 << po2
    code, data = hal.alloc_elem("code", wc * n), hal.alloc_elem("data", wd * n)
    out = hal.syn_witgen(circ, po2, zk, 77, 99, code, data)
    ocode, odata, oout = oc.witgen(po2, zk, 77, 99)
    assert np.array_equal(code.to_vec(), ocode)
    assert np.array_equal(data.to_vec(), odata)
    assert np.array_equal(out, oout)
    mix = rand_fp(np.random.default_rng(1), wa)
    accum = hal.alloc_elem("accum", wa * n)
    hal.syn_accum(circ, po2, zk, 99, data, mix, accum)
    oacc = np.zeros(wa * n, np.uint32)
    oracle.zko_syn_accum(oc.h, po2, zk, zko.key_words(99), odata, mix, oacc)
    assert np.array_equal(accum.to_vec(), oacc)
    return circ, oc, (code, data, accum), (ocode, odata, oacc), (out, mix)


@pytest.mark.parametrize("shape,po2,zk", [("tiny", 9, 100), ("small", 12, 1994)])
def test_witgen_and_eval_check_parity(hal, oracle, shape, po2, zk):
    desc = SHAPES[shape]()
    circ, oc, (code, data, accum), (ocode, odata, oacc), (out, mix) = _witness_parity(hal, oracle, desc, po2, zk)
    wa, wc, wd = (int(x) for x in desc[3:6])
    n, dom = 1 << po2, 4 << po2
    # evaluate every group on the 4n coset with both implementations, then eval_check
    ev, oev = [], []
    for buf, host, w in ((accum, oacc, wa), (code, ocode, wc), (data, odata, wd)):
        co = hal.alloc_elem("co", w * n)
        hal.batch_interpolate_ntt_from(co, buf, w, True)
        e = hal.alloc_elem("ev", w * dom)
        hal.batch_expand_into_evaluate_ntt(e, co, w, 2)
        ev.append(e)
        h = host.copy()
        oracle.zko_batch_interpolate_ntt(h, h.size, w)
        oracle.zko_zk_shift(h, h.size, w)
        oe = np.zeros(w * dom, np.uint32)
        oracle.zko_batch_expand_into_evaluate_ntt(oe, oe.size, h, h.size, w, 2)
        oev.append(oe)
        assert np.array_equal(e.to_vec(), oe)
    poly_mix = rand_fp(np.random.default_rng(2), 4)
    want = np.zeros(4 * dom, np.uint32)
    gp = (C.c_void_p * 3)(*[a.ctypes.data for a in oev])
    glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
    oracle.zko_eval_check(oc.h, want, gp, glp, poly_mix, po2)
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    for interp in (True, False):
        check = hal.alloc_elem("check", 4 * dom)
        circ.eval_check(check, ev, [g_out, g_mix], poly_mix, po2, use_interpreter=interp)
        assert np.array_equal(check.to_vec(), want), f"eval_check mismatch (interpreter={interp})"
    assert circ.has_compiled_kernel()


@pytest.mark.parametrize("shape,po2,zk", [("tiny", 9, 100), ("tiny", 13, 1994), ("small", 12, 1994), ("small", 14, 1994)])
def test_seal_bit_exact_vs_oracle(hal, oracle, shape, po2, zk):
    desc = SHAPES[shape]()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=po2, seed=0x5EED0000 + po2, noise_seed=0x2E80, zk_cycles=zk)
    receipt = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(po2, zk, seg.seed, seg.noise_seed)
    assert receipt.seal.size == want.size
    assert np.array_equal(receipt.seal, want), "HIP seal differs from the CPU oracle seal"
    assert oc.verify(receipt.seal, zk_cycles=zk) is None
    assert np.array_equal(prover.control_root(po2, zk), oc.control_root(po2, zk))
    receipt.verify(desc, prover.control_root(po2, zk))
    bad = receipt.seal.copy()
    bad[bad.size // 3] ^= 1
    assert oc.verify(bad, zk_cycles=zk) is not None


def test_seal_syn_a_po2_16_verifies(hal, oracle):
    """SYN-A column counts at 1/16 of the BASELINE segment size: oracle verifier must accept; seal equality too."""
    desc = syn_air.syn_a()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=3, po2=16, seed=0x5EED0003)
    receipt = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    assert oc.verify(receipt.seal) is None
    want = oc.prove(16, seg.zk_cycles, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want)


def test_seal_full_size_po2_20_verifies(hal, oracle):
    """BASELINE config 2: one 2^20-cycle SYN-A segment sealed on the GPU; accepted by the independent verifier."""
    desc = syn_air.syn_a()
    prover = SegmentProver(hal, desc)
    receipt = prover.prove_segment(Segment(index=0, po2=20, noise_seed=0x2E80))
    oc = zko.OracleCircuit(oracle, desc)
    assert oc.verify(receipt.seal, prover.control_root(20)) is None
    receipt.verify(desc, prover.control_root(20))
    # determinism: same witness + noise -> identical seal; fresh OS noise (the default) -> a different seal
    again = prover.prove_segment(Segment(index=0, po2=20, noise_seed=0x2E80))
    assert np.array_equal(receipt.seal, again.seal)
    fresh = prover.prove_segment(Segment(index=0, po2=20))
    assert not np.array_equal(receipt.seal, fresh.seal) and np.array_equal(receipt.seal[:5], fresh.seal[:5])


@pytest.mark.parametrize("po2", [21, 22])
def test_seal_maximum_sizes_verify(hal, oracle, po2):
    """Largest supported segments (po2 22 -> 2^24 evaluation domain, three-pass NTTs): the verifier must accept."""
    desc = syn_air.syn_small()
    prover = SegmentProver(hal, desc)
    receipt = prover.prove_segment(Segment(index=0, po2=po2, seed=0xABC0 + po2))
    oc = zko.OracleCircuit(oracle, desc)
    assert oc.verify(receipt.seal, prover.control_root(po2)) is None
    from zeth_amd.hal import fp_decode
    assert fp_decode(int(receipt.seal[4])) == po2


def test_po2_beyond_limit_is_an_error(hal):
    from zeth_amd.hal import HalError
    prover = SegmentProver(hal, syn_air.syn_tiny())
    with pytest.raises(HalError, match="too large|too small"):
        prover.prove_segment(Segment(index=0, po2=25))          # upstream's MAX_CYCLES_PO2 is 24: the evaluation domain 2^26 is the largest NTT
    with pytest.raises(HalError, match="too small"):
        prover.prove_segment(Segment(index=0, po2=10))          # n <= zk_cycles


def test_block_of_segments_prove_then_verify(hal, oracle):
    """The reference's flow for one block (lib.rs:123-143 then cli.rs:103): split the session into segments (short tail),
    seal each on the GPU, assemble the composite in index order, verify with the PRODUCT's host verifier (and the oracle's)."""
    from zeth_amd.hal import HalError
    from zeth_amd.host import BlockProcessor, session_segments
    desc = syn_air.syn_small()
    prover = SegmentProver(hal, desc)
    segs = session_segments(3 * (1 << 15) + 9000, segment_po2=15)       # three full 2^15 segments + a 2^14 tail
    assert [s.po2 for s in segs] == [15, 15, 15, 14]
    receipt = BlockProcessor(prover.prove_segment).prove(segs)
    receipt.verify(desc, prover.control_root)
    oc = zko.OracleCircuit(oracle, desc)
    for r in receipt.segments:
        assert oc.verify(r.seal) is None
    receipt.segments[1].output[0] ^= 1                  # metadata must match the seal's own output globals
    with pytest.raises(HalError, match="output"):
        receipt.verify(desc, prover.control_root)
    receipt.segments[1].output[0] ^= 1
    receipt.segments[2].seal[1000] ^= 4
    with pytest.raises(HalError, match="verify_segment"):
        receipt.verify(desc, prover.control_root)


def test_load_time_compiled_eval_check_matches_oracle_and_interpreter(hal, oracle, tmp_path, monkeypatch):
    """A circuit shape with no built-in kernel gets its eval_check kernel generated + compiled when the blob is loaded
    (circuits/jit.py -> zkh_circuit_attach_code_object); results must equal the oracle's and the interpreter's, and a
    whole seal through that kernel must equal the oracle prover's seal byte for byte."""
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    desc = syn_air.build_syn_air(7, 17, 8)              # not one of the shipped shapes
    po2, zk = 10, 300
    plain = hal.load_circuit(desc, jit=False)
    assert plain.kernel_kind() == "interpreter"
    circ, oc, (code, data, accum), (ocode, odata, oacc), (out, mix) = _witness_parity(hal, oracle, desc, po2, zk)
    assert circ.kernel_kind() == "attached"             # load_circuit compiled it (hipcc is part of the image)
    assert any(f.name.endswith(".hsaco") for f in tmp_path.iterdir())
    wa, wc, wd = (int(x) for x in desc[3:6])
    n, dom = 1 << po2, 4 << po2
    ev, oev = [], []
    for buf, host, w in ((accum, oacc, wa), (code, ocode, wc), (data, odata, wd)):
        co = hal.alloc_elem("co", w * n)
        hal.batch_interpolate_ntt_from(co, buf, w, True)
        e = hal.alloc_elem("ev", w * dom)
        hal.batch_expand_into_evaluate_ntt(e, co, w, 2)
        ev.append(e)
        oev.append(e.to_vec())
    poly_mix = rand_fp(np.random.default_rng(5), 4)
    want = np.zeros(4 * dom, np.uint32)
    gp = (C.c_void_p * 3)(*[a.ctypes.data for a in oev])
    glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
    oracle.zko_eval_check(oc.h, want, gp, glp, poly_mix, po2)
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    for interp in (False, True):
        check = hal.alloc_elem("check", 4 * dom)
        circ.eval_check(check, ev, [g_out, g_mix], poly_mix, po2, use_interpreter=interp)
        assert np.array_equal(check.to_vec(), want), f"eval_check mismatch (interpreter={interp})"
    # attaching over a built-in kernel is allowed too and must not change results
    small = hal.load_circuit(syn_air.syn_small(), jit=True)
    assert small.kernel_kind() == "attached"
    prover = SegmentProver(hal, desc)
    assert prover.circuit.kernel_kind() == "attached"
    seg = Segment(index=0, po2=po2, seed=0x5EED0000 + 3, noise_seed=0x2E80, zk_cycles=zk)
    receipt = prover.prove_segment(seg)
    want_seal = oc.prove(po2, zk, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want_seal)
    assert oc.verify(receipt.seal, zk_cycles=zk) is None


def test_attach_rejects_garbage(hal):
    from zeth_amd.hal import HalError
    circ = hal.load_circuit(syn_air.syn_tiny(), jit=False)
    with pytest.raises(HalError, match="not an ELF"):
        circ.attach_code_object(b"\0" * 256, "k")
    from zeth_amd.circuits import jit
    image, name = jit.compile_code_object(syn_air.syn_tiny())
    with pytest.raises(HalError, match="no kernel"):
        circ.attach_code_object(image, "k_does_not_exist")
    circ.attach_code_object(image, name)
    assert circ.kernel_kind() == "attached"


def test_wide_circuit_seal_bit_exact(hal, oracle, tmp_path, monkeypatch):
    """A circuit four times wider than SYN-A (W_code 16, W_data 800, W_accum 64: 945 taps, 2615 steps) goes through the
    same path end to end — eval_check compiled at load time — and its seal equals the oracle prover's byte for byte."""
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    desc = syn_air.build_syn_air(16, 800, 64)
    prover = SegmentProver(hal, desc)
    assert prover.circuit.kernel_kind() == "attached"
    seg = Segment(index=0, po2=11, seed=0x5EED0000 + 99, noise_seed=0x2E80, zk_cycles=700)
    receipt = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(11, 700, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want)
    receipt.verify(desc, prover.control_root(11, 700))
