"""eval_check on the GPU for a constraint system of realistic weight (SYN-HEAVY): generated kernels == interpreter == oracle, whole seals bit-exact,
gathered power tables inside the code objects."""
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


# ---------------------------------------------------------------------------------------------------------------
# SYN-HEAVY: a constraint system of realistic weight (value numbering, windows, split kernels, ConstExt, nested AndCond)
# ---------------------------------------------------------------------------------------------------------------
def _evaluated_groups(hal, oracle, prover, oc, seg):
    """Witness + accum + the three evaluated groups on the device, and the same on the host (oracle)."""
    import ctypes as C
    desc = prover.circuit.desc
    wa, wc, wd = (int(x) for x in desc[3:6])
    n, dom = 1 << seg.po2, 4 << seg.po2
    code, data, out = prover.witgen(seg)
    ocode, odata, oout = oc.witgen(seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed)
    assert np.array_equal(code.to_vec(), ocode) and np.array_equal(data.to_vec(), odata) and np.array_equal(out, oout)
    mix = np.random.default_rng(1).integers(0, 2013265921, size=wa, dtype=np.uint64).astype(np.uint32)
    accum = hal.alloc_elem("accum", wa * n)
    hal.syn_accum(prover.circuit, seg.po2, seg.zk_cycles, seg.noise_seed, data, mix, accum)
    ev, oev = [], []
    for buf, w in ((accum, wa), (code, wc), (data, wd)):
        co = hal.alloc_elem("co", w * n)
        hal.batch_interpolate_ntt_from(co, buf, w, True)
        e = hal.alloc_elem("ev", w * dom)
        hal.batch_expand_into_evaluate_ntt(e, co, w, 2)
        ev.append(e)
        oev.append(e.to_vec())               # the NTTs have their own parity tests: feed both sides the same evaluations
    return ev, oev, out, mix


def test_syn_heavy_eval_check_generated_interpreted_and_oracle_agree(hal, oracle, tmp_path, monkeypatch):
    """ConstExt operands, AndCond inside AndCond, thousands of constraints with shared sub-expressions: the generated
    kernels (value numbering + windows, TWO parts accumulated into `check`, compiled at load time), the on-device step
    interpreter (taps / constants as operands, Fp4-typed slots) and the oracle's literal interpreter give identical words."""
    import ctypes as C
    from zeth_amd.circuits import syn_heavy
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("ZKH_CODEGEN_PART", "1600")          # several parts (the shipped weight keeps this small circuit whole)
    desc = syn_heavy.syn_heavy_small()
    prover = SegmentProver(hal, desc)
    assert prover.circuit.kernel_kind() == "attached" and prover.circuit.compiled_parts() >= 2
    oc = zko.OracleCircuit(oracle, desc)
    seg = Segment(index=0, po2=10, seed=11, noise_seed=12, zk_cycles=300)
    ev, oev, out, mix = _evaluated_groups(hal, oracle, prover, oc, seg)
    dom = 4 << seg.po2
    poly_mix = np.random.default_rng(2).integers(0, 2013265921, size=4, dtype=np.uint64).astype(np.uint32)
    want = np.zeros(4 * dom, np.uint32)
    gp = (C.c_void_p * 3)(*[a.ctypes.data for a in oev])
    glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
    oracle.zko_eval_check(oc.h, want, gp, glp, poly_mix, seg.po2)
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    for interp in (False, True):
        check = hal.alloc_elem("check", 4 * dom)
        prover.circuit.eval_check(check, ev, [g_out, g_mix], poly_mix, seg.po2, use_interpreter=interp)
        assert np.array_equal(check.to_vec(), want), f"eval_check mismatch (interpreter={interp})"


@pytest.mark.parametrize("po2,zk", [(9, 100), (13, 1994)])
def test_syn_heavy_small_seal_bit_exact(hal, oracle, po2, zk, tmp_path, monkeypatch):
    from zeth_amd.circuits import syn_heavy
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    desc = syn_heavy.syn_heavy_small()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=po2, seed=21 + po2, noise_seed=22, zk_cycles=zk)
    receipt = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(po2, zk, seg.seed, seg.noise_seed)
    assert np.array_equal(receipt.seal, want)
    receipt.verify(desc, prover.control_root(po2, zk))


def test_syn_heavy_full_circuit_seal_bit_exact_and_po2_20_verifies(hal, oracle):
    """The bench's `--circuit syn_heavy` (54 k steps, 1061 taps, 7 built-in kernels): byte-identical to the oracle at
    po2 13, and a 2^20-cycle seal is accepted by the product's verifier and by the oracle's."""
    from zeth_amd.circuits import syn_heavy
    desc = syn_heavy.syn_heavy()
    prover = SegmentProver(hal, desc)
    assert prover.circuit.kernel_kind() == "builtin" and prover.circuit.compiled_parts() >= 4
    oc = zko.OracleCircuit(oracle, desc)
    seg = Segment(index=0, po2=13, seed=31, noise_seed=32)
    receipt = prover.prove_segment(seg)
    assert np.array_equal(receipt.seal, oc.prove(13, 1994, seg.seed, seg.noise_seed))
    big = prover.prove_segment(Segment(index=1, po2=20, seed=33, noise_seed=34))
    big.verify(desc, prover.control_root(20))
    assert oc.verify(big.seal, prover.control_root(20)) is None


def test_gathered_power_tables_travel_inside_the_code_objects(hal, oracle, tmp_path, monkeypatch):
    """Round 4's eval_check generator gives every kernel its own mix-power table in emission order and exports the exponent list
    as `<kernel>_exps` inside the code object: a host that attaches the parts needs to know nothing about it.  Kernels generated
    WITH the table and WITHOUT it give the interpreter's words; a set that mixes the two is refused, not launched."""
    from zeth_amd.circuits import codegen, jit, syn_heavy
    from zeth_amd.hal import HalError
    from zeth_amd.prover import Segment, SegmentProver
    import zko
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("ZKH_CODEGEN_PART", "1600")
    desc = syn_heavy.syn_heavy_small()
    prover = SegmentProver(hal, desc)
    circ = prover.circuit
    oc = zko.OracleCircuit(oracle, desc)
    seg = Segment(index=0, po2=10, seed=31, noise_seed=32, zk_cycles=300)
    ev, _, out, mix = _evaluated_groups(hal, oracle, prover, oc, seg)
    dom = 4 << seg.po2
    poly_mix = np.random.default_rng(4).integers(1, 2013265921, size=4, dtype=np.uint64).astype(np.uint32)
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    want = hal.alloc_elem("want", 4 * dom)
    circ.eval_check(want, ev, [g_out, g_mix], poly_mix, seg.po2, use_interpreter=True)
    want = want.to_vec()
    objs = {}
    for gather in (1, 0):
        monkeypatch.setattr(codegen, "GATHER", gather)
        objs[gather] = jit.compile_code_objects(desc, use_cache=False)
        assert len(objs[gather]) >= 2
    for gather in (1, 0, 1):                                   # gathered, plain, gathered again: a new set replaces the old one whole
        for i, (img, name) in enumerate(objs[gather]):
            circ.attach_code_object(img, name, i, len(objs[gather]))
        assert circ.kernel_kind() == "attached"
        got = hal.alloc_elem("check", 4 * dom)
        circ.eval_check(got, ev, [g_out, g_mix], poly_mix, seg.po2)
        assert np.array_equal(got.to_vec(), want), f"gather={gather}"
    img, name = objs[0][1]                                     # one plain part among gathered ones: such a set is not launched ...
    circ.attach_code_object(img, name, 1, len(objs[1]))
    with pytest.raises(HalError, match="gathered power table"):
        circ.eval_check(got, ev, [g_out, g_mix], poly_mix, seg.po2)
    img, name = objs[1][1]                                     # ... and the right part repairs it
    circ.attach_code_object(img, name, 1, len(objs[1]))
    circ.eval_check(got, ev, [g_out, g_mix], poly_mix, seg.po2)
    assert np.array_equal(got.to_vec(), want)
