"""RECURSION on the device: witness generator, copy argument and seal against the CPU oracle, bit for
bit; then the real thing - segment seals lifted and joined into one receipt whose every node verified its children in-circuit."""
import json
import os
import time

import numpy as np
import pytest

import zko
from zeth_amd.circuits import rec_verify as V, recursion as R, syn_air
from zeth_amd.circuits.desc import P

pytestmark = pytest.mark.gpu
RM = (1 << 32) % P
RINV = pow(RM, -1, P)
ZK = 1994


def small_program():
    pr = R.Program()
    x, y = pr.input(0, 4), pr.input(4, 4)
    s, m = pr.add(x, y), pr.mul(x, y)
    iv = pr.inv(m)
    pr.eq(pr.mul(m, iv), pr.const(1))
    a, b, _, _ = pr.unpack(x)
    bits = pr.bits31(a, 12)
    sel = pr.mux(bits[0], s, m)
    h = pr.p2([x, y, s, m, pr.zero(), pr.zero()])
    h2 = pr.p2([h[0], h[1], sel, iv, h[4], h[5]])
    pr.eq(pr.is_zero(b), pr.zero())
    pk = pr.pack(2, x, y, s, m)
    pr.public(h2[0], h2[1], pk, bits[12])
    return pr


def _device_traces(hal, prog, inputs, noise=0x2E80):
    n = 1 << prog.po2
    code, data, accum = hal.alloc_elem("code", R.WC * n), hal.alloc_elem("data", R.WD * n), hal.alloc_elem("accum", R.WA * n)
    prog.code(code)
    out = prog.witgen(inputs, data, noise)
    return code, data, accum, out


def test_small_program_traces_accum_and_seal_match_the_oracle(hal, oracle):
    from zeth_amd.hal import HalError, HostCircuit, RecProgram
    pr = small_program()
    po2 = 12
    blob = pr.finish(po2, ZK)
    inputs = np.array([v * RM % P for v in (5, 6, 7, 8, 11, 12, 13, 14)], dtype=np.uint32)
    oc = zko.OracleCircuit(oracle, R.recursion_circuit())
    ocode, odata, oout = oc.rec_witgen(blob, inputs)
    circuit = hal.load_circuit(R.recursion_circuit())
    assert circuit.kernel_kind() == "builtin"
    prog = RecProgram(hal, circuit, blob)
    assert (prog.po2, prog.n_inputs, prog.n_p2) == (po2, 8, 2)
    assert prog.graph_steps == 0
    os.environ["ZKH_REC_GRAPH"] = "1"                               # opt-in: the same schedule captured as a hipGraph and replayed
    try:
        gprog = RecProgram(hal, circuit, blob)
    finally:
        del os.environ["ZKH_REC_GRAPH"]
    assert gprog.graph_steps > 0
    gdata = hal.alloc_elem("gdata", R.WD << po2)
    assert np.array_equal(gprog.witgen(inputs, gdata), oout) and np.array_equal(gdata.to_vec(), odata)
    code, data, accum, out = _device_traces(hal, prog, inputs)
    assert np.array_equal(code.to_vec(), ocode)
    assert np.array_equal(data.to_vec(), odata)
    assert np.array_equal(out, oout)
    assert np.array_equal(prog.root, oc.root_of_code(po2, ocode))
    mix = np.array([(i * 7919 + 13) * RM % P for i in range(20)], dtype=np.uint32)
    prog.accum(data, mix, accum)
    oaccum = oc.rec_accum(po2, ocode, odata, mix, ZK)
    assert np.array_equal(accum.to_vec(), oaccum)
    assert oc.check_rows(po2, oaccum, ocode, odata, oout, mix) == -1
    seal, out2 = prog.prove(inputs, 0x2E80)
    assert np.array_equal(out2, oout)
    assert np.array_equal(seal, oc.prove_traces(po2, ocode, odata, oout, ZK, 0x2E80))          # byte-identical seals
    HostCircuit(R.recursion_circuit()).verify_segment(seal, prog.root)
    assert oc.verify(seal, prog.root) is None
    # a witness that breaks an assertion of the program does not exist
    bad = inputs.copy()
    bad[1] = 0                                                      # b = 0: is_zero(b) == 0 fails
    with pytest.raises(HalError, match="assertion of the program fails"):
        prog.prove(bad)
    with pytest.raises(HalError, match="input words"):
        prog.prove(inputs[:7])


@pytest.mark.parametrize("seed", range(4))
def test_random_programs_device_witness_equals_the_oracles(hal, oracle, seed):
    """seeded random programs over every gate kind (tests/rec_programs.py): whatever the dependency levels, the persistent
    runs and the eight-lane permutations do, the device's trace and public output are the C interpreter's"""
    from rec_programs import random_program
    from zeth_amd.hal import RecProgram
    pr, words, _ = random_program(seed)
    blob = pr.finish(max(12, pr.min_po2(ZK)), ZK)
    inputs = np.array([w * RM % P for w in words], dtype=np.uint32)
    oc = zko.OracleCircuit(oracle, R.recursion_circuit())
    ocode, odata, oout = oc.rec_witgen(blob, inputs)
    prog = RecProgram(hal, hal.load_circuit(R.recursion_circuit()), blob)
    code, data, accum, out = _device_traces(hal, prog, inputs)
    assert np.array_equal(out, oout) and np.array_equal(data.to_vec(), odata) and np.array_equal(code.to_vec(), ocode)


def test_lift_of_a_small_segment_matches_the_oracle_and_rejects_forgeries(hal, oracle):
    from zeth_amd.hal import HalError, HostCircuit, RecProgram
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_tiny()
    sp = SegmentProver(hal, desc)
    cpo2 = 13                                                       # two FRI rounds
    rcpt = sp.prove_segment(Segment(0, cpo2, seed=77, noise_seed=5))
    croot = sp.control_root(cpo2)
    pr = V.build_lift(desc, cpo2, [int(w) * RINV % P for w in croot])
    po2 = pr.min_po2(ZK)
    blob = pr.finish(po2, ZK)
    A = np.arange(1, 9, dtype=np.uint32)
    inputs = np.concatenate([rcpt.seal, A])
    oc = zko.OracleCircuit(oracle, R.recursion_circuit())
    ocode, odata, oout = oc.rec_witgen(blob, inputs)
    circuit = hal.load_circuit(R.recursion_circuit())
    prog = RecProgram(hal, circuit, blob)
    code, data, accum, out = _device_traces(hal, prog, inputs)
    assert np.array_equal(data.to_vec(), odata) and np.array_equal(code.to_vec(), ocode) and np.array_equal(out, oout)
    # the lift's claim is the segment's claim digest (zkh_receipt_claim), its second half the allowed root it was handed
    from zeth_amd import recursion as host_rec
    assert np.array_equal(out[:8], host_rec.wrap_claim(HostCircuit(desc).receipt_claim(rcpt.seal, croot), 0, 0)) and np.array_equal(out[8:], A)
    seal, _ = prog.prove(inputs, 0x2E80)
    assert np.array_equal(seal, oc.prove_traces(po2, ocode, odata, oout, ZK, 0x2E80))
    HostCircuit(R.recursion_circuit()).verify_segment(seal, prog.root)
    # every part of the child seal is checked in-circuit: header, a Merkle top, coeff_u, an opened row, a path, FRI, the final poly
    n = rcpt.seal.size
    for k in (0, 4, 5 + 8 * 3, n // 5, n // 3, n // 2, 2 * n // 3, n - 300, n - 1):
        forged = inputs.copy()
        forged[k] = (int(forged[k]) + 1) % P
        with pytest.raises(HalError, match="assertion of the program fails"):
            prog.prove(forged)
    # ... and 300 more words drawn at random: no word of a seal is left unbound by the in-circuit verifier
    data_buf = hal.alloc_elem("data", R.WD << prog.po2)
    rng = np.random.default_rng(7)
    for k in rng.integers(0, n, 300):
        forged = inputs.copy()
        forged[k] = (int(forged[k]) + 1 + int(rng.integers(0, P - 1))) % P
        with pytest.raises(HalError, match="assertion of the program fails"):
            prog.witgen(forged, data_buf)
    # a seal under another control root (another circuit's code) is not lifted by this program
    other = V.build_lift(desc, cpo2, [(int(w) * RINV + 1) % P for w in croot])
    with pytest.raises(HalError, match="assertion of the program fails"):
        RecProgram(hal, circuit, other.finish(po2, ZK)).prove(inputs)


def test_block_of_segments_folds_to_one_receipt_that_verified_every_child_in_circuit(hal):
    """BASELINE.json config 5 at the BASELINE shape: po2-20 SYN-A segments + a po2-18 tail, lift each, join to one root."""
    from zeth_amd import recursion as rec
    from zeth_amd.hal import HalError, HostCircuit
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_a()
    sp = SegmentProver(hal, desc, resident_code_group=True)
    segs = [Segment(i, 20 if i < 4 else 18, seed=0x5EED0000 + i, noise_seed=100 + i) for i in range(5)]
    t0 = time.time()
    leaves = [sp.prove_segment(s) for s in segs]
    t_seg = time.time() - t0
    roots = {20: sp.control_root(20), 18: sp.control_root(18)}
    t0 = time.time()
    programs = rec.build_programs(desc, roots)
    rx = rec.Recursion(hal, programs)
    t_load = time.time() - t0
    assert [k[0] for k in rx.kinds] == ["lift", "lift", "lift2", "lift2", "lift2", "join", "join", "join", "join", "join3"]
    assert {p.po2 for p in rx.programs[:2]} == {17} and {p.po2 for p in rx.programs[2:]} == {18}
    hal.sync()
    t0 = time.time()
    lifted = [rx.lift(r, noise_seed=7) for r in leaves]
    hal.sync()
    t_lift = time.time() - t0
    t0 = time.time()
    root = rx.fold(lifted, noise_seed=9)
    hal.sync()
    t_join = time.time() - t0
    claims = [HostCircuit(desc).receipt_claim(r.seal, roots[r.po2]) for r in leaves]
    for l, c in zip(lifted, claims):        # a lift publishes claim' = hash_pair(receipt claim, (pre, post, 0..)); SYN-A has no state: (0, 0)
        assert np.array_equal(l.claim, rec.wrap_claim(c, 0, 0)) and np.array_equal(l.core, c) and np.array_equal(l.allowed, rx.allowed_root())
    assert root.n_leaves == 5 and root.po2 == 18
    root.verify(rx.allowed_roots(), claims)                         # ONE seal + the claim tree: nothing else is needed
    rec.succinct_verify(root.seal, rx.allowed_roots(), root.program, claims)       # the same as one host-only library call (zkh_succinct_verify)
    with pytest.raises(HalError, match="claim tree"):
        rec.succinct_verify(root.seal, rx.allowed_roots(), root.program, claims[::-1])
    # the same tree with the bottom level fused: lift2 = lift + lift + join as one proof per pair of segments (5 proofs, not 9)
    t0 = time.time()
    fused = rx.fold_segments(leaves, noise_seed=9)
    hal.sync()
    t_fused = time.time() - t0
    assert np.array_equal(fused.claim, root.claim) and fused.n_leaves == 5
    fused.verify(rx.allowed_roots(), claims)
    two = rx.lift2(leaves[0], leaves[1], noise_seed=9)
    assert np.array_equal(two.claim, rx.join(lifted[0], lifted[1], 9).claim)
    bad = SegmentReceipt_like(leaves[1])
    bad.seal[bad.seal.size // 3] ^= 1
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.lift2(leaves[0], bad)
    # above the bottom level a proof takes THREE nodes: join3(a, b, c) publishes exactly what join(join(a, b), c) does (the inner
    # claim' is computed in-circuit), verifies all three child seals, and refuses a forged one
    a3, b3, c3 = two, rx.lift2(leaves[2], leaves[3], 9), rx.lift2(leaves[0], leaves[1], 11)
    t0 = time.time()
    j3 = rx.join3(a3, b3, c3, 9)
    hal.sync()
    t_join3 = time.time() - t0
    nested = rx.join(rx.join(a3, b3, 9), c3, 9)
    assert j3.po2 == 18 and j3.n_leaves == 6 and np.array_equal(j3.claim, nested.claim) and np.array_equal(j3.core, nested.core)
    assert rx.kinds[j3.program] == ("join3", 18, 18, 18) and np.array_equal(rx.join_group([a3, b3, c3], 9).seal, j3.seal)
    j3.verify(rx.allowed_roots(), [claims[0], claims[1], claims[2], claims[3], claims[0], claims[1]])
    rec.succinct_verify(j3.seal, rx.allowed_roots(), j3.program, [claims[0], claims[1], claims[2], claims[3], claims[0], claims[1]])
    forged3 = rec.RecReceipt(c3.seal.copy(), c3.po2, c3.program, c3.control_root, 2, c3.core, 0, 0)
    forged3.seal[c3.seal.size // 2] ^= 1
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join3(a3, b3, forged3)
    with pytest.raises(HalError, match="assertion of the program fails"):          # the third child opened with another node's core
        rx.join3(a3, b3, rec.RecReceipt(c3.seal, c3.po2, c3.program, c3.control_root, 2, b3.core, 0, 0))
    with pytest.raises(HalError, match="claim tree"):
        root.verify(rx.allowed_roots(), claims[::-1])
    with pytest.raises(HalError, match="allowed set"):
        root.verify(rx.allowed_roots()[:2], claims)
    # a join refuses a child that is not a valid recursion seal, one that carries another allowed root, and one whose program
    # is not in the allowed set
    forged = rec.RecReceipt(lifted[0].seal.copy(), lifted[0].po2, lifted[0].program, lifted[0].control_root, 1, lifted[0].core, 0, 0)
    forged.seal[lifted[0].seal.size // 2] ^= 1
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join(forged, lifted[1])
    inputs = np.concatenate([leaves[0].seal, np.arange(8, dtype=np.uint32)])
    stranger, _ = rx.programs[0].prove(inputs, 3)                    # a valid lift, but under A' != A
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join(lifted[0], rec.RecReceipt(stranger, 17, 0, rx.programs[0].root, 1, lifted[0].core, 0, 0))
    wrong = rec.RecReceipt(lifted[1].seal, lifted[1].po2, 2, lifted[1].control_root, 1, lifted[1].core, 0, 0)     # membership path of another program
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join(lifted[0], wrong)
    # ... and a child whose claim' is opened with another core or another state range (the opening is checked in-circuit)
    lied = rec.RecReceipt(lifted[1].seal, lifted[1].po2, lifted[1].program, lifted[1].control_root, 1, lifted[0].core, 0, 0)
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join(lifted[0], lied)
    lied = rec.RecReceipt(lifted[1].seal, lifted[1].po2, lifted[1].program, lifted[1].control_root, 1, lifted[1].core, 0, 5)
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.join(lifted[0], lied)
    # the BASELINE shape against the oracle: the lift of a po2-20 SYN-A seal - the device's witness (72 x 2^17 words), its copy
    # argument and its seal are the CPU oracle's, word for word
    import zko
    oc = zko.OracleCircuit(zko.load(), R.recursion_circuit())
    blob20 = programs[0][1]
    inputs = np.concatenate([leaves[0].seal, rx.allowed_root()])
    ocode, odata, oout = oc.rec_witgen(blob20, inputs, noise_seed=7)
    p20 = rx.programs[0]
    n17 = 1 << p20.po2
    data_buf, accum_buf = hal.alloc_elem("data", R.WD * n17), hal.alloc_elem("accum", R.WA * n17)
    assert np.array_equal(p20.witgen(inputs, data_buf, 7), oout) and np.array_equal(data_buf.to_vec(), odata)
    mix = np.array([(i * 104729 + 7) * RM % P for i in range(20)], dtype=np.uint32)
    p20.accum(data_buf, mix, accum_buf, 7)
    assert np.array_equal(accum_buf.to_vec(), oc.rec_accum(p20.po2, ocode, odata, mix, ZK, 7))
    assert np.array_equal(lifted[0].seal, oc.prove_traces(p20.po2, ocode, odata, oout, ZK, 7))
    line = {"config": "succinct, recursive", "segments": 5, "segment_s": round(t_seg, 3), "program_load_s": round(t_load, 2),
            "lift_s_each": round(t_lift / 5, 4), "join_s_each": round(t_join / 4, 4), "fold_s_9_proofs": round(t_lift + t_join, 4),
            "fold_s_fused_5_proofs": round(t_fused, 4), "join3_s": round(t_join3, 4),
            "programs": [{"kind": list(k), "po2": p.po2, "permutations": p.n_p2, "gates": p.n_gates, "levels": p.n_levels,
                          "input_words": p.n_inputs} for k, p in zip(rx.kinds, rx.programs)]}
    print("RECURSION " + json.dumps(line))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/recursion_test.json", "w") as fh:
        json.dump(line, fh)


def test_native_session_executor_lifts_and_joins_like_the_python_driver(hal, tmp_path):
    """zkh_session_prove(join_tree = 2) (csrc/session.hip: C++ threads over lanes) against zeth_amd/recursion.py: the same root
    receipt word for word (fixed noise), zkh_session_verify accepts it, a swapped leaf is refused; and the compiled host
    examples/prove_session does the same from program files."""
    import subprocess
    from zeth_amd import build, recursion as rec
    from zeth_amd.hal import HalError
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_small()
    sp = SegmentProver(hal, desc)
    segs = [Segment(index=i, po2=13 if i < 4 else 12, seed=900 + i, noise_seed=0x51) for i in range(5)]
    roots = {13: sp.control_root(13), 12: sp.control_root(12)}
    programs = rec.build_programs(desc, roots)
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    with pytest.raises(HalError, match="set_recursion"):
        sess.prove(segs, join_tree=2)
    sess.set_recursion(programs)
    comp, root, stats = sess.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
    assert stats["n_lifts"] == 3 and stats["n_joins"] == 2 and stats["verified"] and root is not None      # 2 lift2 + 1 lift, 2 joins
    rx = rec.Recursion(hal, programs)
    leaves = [sp.prove_segment(s) for s in segs]
    for a, b in zip(comp.segments, leaves):
        assert np.array_equal(a.seal, b.seal)
    want = rx.fold_segments(leaves, 0x77)
    assert np.array_equal(root.seal, want.seal) and stats["root_program"] == want.program
    want.verify(rx.allowed_roots(), [HostCircuit_claim(desc, r, roots) for r in leaves])
    # the library's verification binds the root to THESE segments: another session's root is refused
    other = [Segment(index=i, po2=s.po2, seed=s.seed + 50, noise_seed=0x51) for i, s in enumerate(segs)]
    _, root2, _ = sess.prove(other, join_tree=2, join_noise_seed=0x77, verify=True)
    assert not np.array_equal(root2.seal[:8], root.seal[:8])
    sess.close()
    # the compiled host: programs from files
    d = tmp_path / "zkr"
    d.mkdir()
    np.asarray(R.recursion_circuit(), dtype="<u4").tofile(d / "recursion.desc")
    for kind, blob in programs:
        np.asarray(blob, dtype="<u4").tofile(d / ("-".join(str(x) for x in kind[:2 if kind[0] == "lift" else 4 if kind[0] == "join3" else 3]) + ".zkr1"))
    dpath = tmp_path / "syn_small.desc"
    np.asarray(desc, dtype="<u4").tofile(dpath)
    exe = os.path.join(os.path.dirname(build.build_examples()), "prove_session")
    r = subprocess.run([exe, "--desc", str(dpath), "--recursion-dir", str(d), "--po2", "13", "--tail-po2", "12", "--segments", "6", "--inflight", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    m = int(dict(programs)[("lift2", 13, 13)][2])                 # three lift2 nodes of size m: one join3 where the set has it, else two joins
    has3 = ("join3", m, m, m) in [k for k, _ in programs]
    assert out["verified"] is True and out["lifts"] == 3 and out["joins"] == (1 if has3 else 2) and out["in_circuit_verification"] is True   # 3 lift2


def test_cpp_host_builds_the_recursion_programs_itself_and_gets_the_python_root(hal, tmp_path):
    """examples/prove_session --circuit syn_small --build-recursion: NO files — the circuit description comes out of the library, the
    control roots from the GPU, the lift / lift2 / join / join3 programs from the library's C++ builder (csrc/rec_builder.hip), in the
    order zeth_amd/recursion.py build_programs uses: the root receipt's public output (claim tree root ‖ allowed-programs root) is
    the one the Python-built program set gives for the same segments and noise."""
    import subprocess
    from zeth_amd import build, recursion as rec
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_small()
    sp = SegmentProver(hal, desc)
    programs = rec.build_programs(desc, {13: sp.control_root(13), 12: sp.control_root(12)})
    segs = [Segment(index=i, po2=13 if i < 6 else 12, seed=0x5EED0000 + i, noise_seed=0x51) for i in range(7)]
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_recursion(programs)
    _, root, st = sess.prove(segs, join_tree=2, join_noise_seed=0x51, verify=True)
    sess.close()
    sess2 = Session(desc, devices=(0,), lanes_per_device=2)      # the same through the ctypes mirror: zkh_session_build_recursion
    sess2.build_recursion([s.po2 for s in segs])
    _, root2, st2 = sess2.prove(segs, join_tree=2, join_noise_seed=0x51, verify=True)
    sess2.close()
    assert np.array_equal(root2.seal, root.seal) and st2["root_program"] == st["root_program"]
    exe = os.path.join(os.path.dirname(build.build_examples()), "prove_session")
    csv = os.path.join(os.path.dirname(exe), "..", "gpurun_out", "prove_session_stats.csv")
    os.makedirs(os.path.dirname(csv), exist_ok=True)
    if os.path.exists(csv):
        os.remove(csv)
    r = subprocess.run([exe, "--circuit", "syn_small", "--build-recursion", "--po2", "13", "--tail-po2", "12", "--segments", "7", "--inflight", "2",
                        "--noise-seed", str(0x51), "--csv", csv, "--block-number", "19000000", "--gas-used", "15000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    # the reference's stats vocabulary (/root/reference/run-parallel.sh:15): header once, one row per session
    rows = open(csv).read().strip().splitlines()
    assert rows[0] == "block_number,execution_time,total_cycles,user_cycles,paging_cycles,keccak_calls,gas_used" and len(rows) == 2
    cols = rows[1].split(",")
    assert cols[0] == "19000000" and float(cols[1]) > 0 and int(cols[2]) == 6 * 8192 + 4096 and int(cols[3]) == int(cols[2]) - 7 * 1994 and cols[4:] == ["0", "0", "15000000"]
    assert out["verified"] is True and out["programs_built_by_library"] is True and out["in_circuit_verification"] is True
    assert out["lifts"] == st["n_lifts"] and out["joins"] == st["n_joins"]
    assert out["root_out"] == "".join(f"{int(w):08x}" for w in root.seal[:16])


def SegmentReceipt_like(r):
    from zeth_amd.prover import SegmentReceipt
    return SegmentReceipt(seal=r.seal.copy(), index=r.index, po2=r.po2)


def HostCircuit_claim(desc, receipt, roots):
    from zeth_amd.hal import HostCircuit
    return HostCircuit(desc).receipt_claim(receipt.seal, roots[receipt.po2])


def test_a_keccak_assumption_receipt_is_lifted_too(hal):
    """The lift program is built from a circuit DESCRIPTION: the same builder lifts a KECCAK-F receipt (3 840 columns, 43.8 k
    constraint steps evaluated gate by gate in-circuit) - how upstream resolves a keccak assumption on the way to the succinct
    receipt.  The lift's claim is the keccak receipt's claim digest, which binds the SHA-3 state it proves."""
    import hashlib
    from zeth_amd import recursion as rec
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import HalError, HostCircuit, fp_decode
    from zeth_amd.prover import Segment, SegmentProver
    kdesc = keccak_f.keccak_f_circuit()
    kp = SegmentProver(hal, kdesc)
    msg = b"assumption: one accelerator batch"
    pub = tuple(w for lane in keccak_f.sha3_256_block(msg) for w in (lane & 0xFFFFFFFF, lane >> 32))
    krec = kp.prove_segment(Segment(index=0, po2=13, seed=0xCECC, noise_seed=3, pub=pub))
    limbs = [fp_decode(int(w)) for w in krec.seal[:100]]
    assert keccak_f.digest_of_state([sum(limbs[4 * l + j] << (16 * j) for j in range(4)) for l in range(25)]) == hashlib.sha3_256(msg).digest()
    kroot = kp.control_root(13)
    programs = [p for p in rec.build_programs(kdesc, {13: kroot}) if p[0][0] == "lift"]
    rx = rec.Recursion(hal, programs, families=[(kdesc, {13: kroot})])
    lifted = rx.lift(krec, noise_seed=5)
    assert np.array_equal(lifted.claim, rec.wrap_claim(HostCircuit(kdesc).receipt_claim(krec.seal, kroot), 0, 0))
    lifted.verify(rx.allowed_roots())
    forged = krec.seal.copy()
    forged[7] = (int(forged[7]) + 1) % P                            # one limb of the proven state
    from zeth_amd.prover import SegmentReceipt
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.lift(SegmentReceipt(seal=forged, index=0, po2=13))
    print("KECCAK lift:", {"po2": lifted.po2, "permutations": rx.programs[0].n_p2, "gates": rx.programs[0].n_gates})


def test_segments_and_a_keccak_assumption_fold_into_one_receipt(hal):
    """upstream's flow for a block whose guest used the keccak accelerator: the segment receipts AND the keccak batch receipt
    are lifted, then joined into one receipt - every node verified in-circuit, three circuits (SYN, KECCAK-F, RECURSION) and
    three recursion sizes (17, 18, 19) in one allowed set"""
    from zeth_amd import recursion as rec
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import HostCircuit
    from zeth_amd.prover import Segment, SegmentProver
    sdesc, kdesc = syn_air.syn_small(), keccak_f.keccak_f_circuit()
    sp, kp = SegmentProver(hal, sdesc), SegmentProver(hal, kdesc)
    segs = [sp.prove_segment(Segment(index=i, po2=13, seed=60 + i, noise_seed=9)) for i in range(3)]
    krec = kp.prove_segment(Segment(index=0, po2=13, seed=0xCECC, noise_seed=3))
    sroot, kroot = sp.control_root(13), kp.control_root(13)
    programs = rec.build_programs(sdesc, {13: sroot}, assumptions=[(kdesc, {13: kroot})], fused_pairs=False)
    rx = rec.Recursion(hal, programs)
    sizes = sorted({p.po2 for p in rx.programs})
    assert len(programs) <= 16 and len(sizes) >= 3
    leaves = [rx.lift(r, 5) for r in segs] + [rx.lift(krec, 5, family=1)]
    root = rx.fold(leaves, 7)
    claims = [HostCircuit(sdesc).receipt_claim(r.seal, sroot) for r in segs] + [HostCircuit(kdesc).receipt_claim(krec.seal, kroot)]
    root.verify(rx.allowed_roots(), claims)
    assert root.n_leaves == 4
    print("RESOLVE", {"programs": len(programs), "sizes": sizes})


def test_keccak_assumptions_are_united_and_the_session_resolved_against_them(hal):
    """upstream's full shape (ProverServer::{lift, join, union, resolve}): the segment receipts fold into the session's root through the
    join tree; the keccak batch receipts are lifted and UNITED (each node the digest of the sorted pair: no order among assumptions);
    `resolve` binds the two.  The verifier recomputes the resolved claim from the segment leaves and the assumption leaves — in
    Python and as one host-only library call — and refuses another assumption set, a missing assumption, a plain join-tree reading."""
    import hashlib
    from zeth_amd import recursion as rec
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import HalError, HostCircuit
    from zeth_amd.prover import Segment, SegmentProver
    sdesc, kdesc = syn_air.syn_small(), keccak_f.keccak_f_circuit()
    sp, kp = SegmentProver(hal, sdesc), SegmentProver(hal, kdesc)
    segs = [sp.prove_segment(Segment(index=i, po2=13, seed=80 + i, noise_seed=9)) for i in range(3)]
    msgs = [b"assumption %d: one accelerator batch" % i for i in range(3)]
    kpubs = [tuple(w for lane in keccak_f.sha3_256_block(m) for w in (lane & 0xFFFFFFFF, lane >> 32)) for m in msgs]
    krecs = [kp.prove_segment(Segment(index=i, po2=13, seed=0xCECC + i, noise_seed=3, pub=kpubs[i])) for i in range(3)]
    sroot, kroot = sp.control_root(13), kp.control_root(13)
    t0 = time.time()
    programs = rec.build_programs(sdesc, {13: sroot}, assumptions=[(kdesc, {13: kroot})], fused_pairs=False, resolve=True)
    kinds = [k for k, _ in programs]
    assert len(programs) <= 16 and {k[0] for k in kinds} == {"lift", "join", "join3", "union", "resolve"}
    rx = rec.Recursion(hal, programs)
    t_build = time.time() - t0
    session_root = rx.fold([rx.lift(r, 5) for r in segs], 7)
    lifted = [rx.lift(k, 5, family=1) for k in krecs]
    # union(a, b) == union(b, a): the same claim' whichever way round (two proofs, one statement)
    ab, ba = rx.union(lifted[0], lifted[1], 11), rx.union(lifted[1], lifted[0], 12)
    assert np.array_equal(ab.claim, ba.claim) and np.array_equal(ab.claim, rec.union_node(lifted[0].claim, lifted[1].claim)[0])
    t0 = time.time()
    assumed = rx.union_fold(lifted, 13)
    resolved = rx.resolve(session_root, assumed, 17)
    t_fold = time.time() - t0
    sclaims = [HostCircuit(sdesc).receipt_claim(r.seal, sroot) for r in segs]
    kclaims = [HostCircuit(kdesc).receipt_claim(r.seal, kroot) for r in krecs]
    roots = rx.allowed_roots()
    assumed.verify(roots, assumption_claims=kclaims)
    resolved.verify(roots, sclaims, assumption_claims=kclaims)
    resolved.verify(roots, sclaims, assumption_claims=[kclaims[1], kclaims[0], kclaims[2]])        # within a pair the order is the union's
    rec.succinct_verify(resolved.seal, roots, resolved.program, sclaims, assumption_claims=kclaims)
    rec.succinct_verify(assumed.seal, roots, assumed.program, [], assumption_claims=kclaims)
    assert resolved.n_leaves == 6 and (resolved.pre, resolved.post) == (session_root.pre, session_root.post)
    for leaves, assumptions, what in ((sclaims, kclaims[:2], "resolved root"), (sclaims, [kclaims[0], kclaims[2], kclaims[1]], "resolved root"),
                                      (sclaims[:2], kclaims, "resolved root"), (sclaims, None, "claim tree")):
        with pytest.raises(HalError, match=what):
            resolved.verify(roots, leaves, assumption_claims=assumptions)
        with pytest.raises(HalError, match=what):
            rec.succinct_verify(resolved.seal, roots, resolved.program, leaves, assumption_claims=assumptions)
    with pytest.raises(HalError, match="root receipt"):                                 # the union root is not the resolved receipt
        rec.succinct_verify(assumed.seal, roots, resolved.program, sclaims, assumption_claims=kclaims)
    # a keccak receipt whose proven state was altered has no lift, so nothing to unite
    from zeth_amd.prover import SegmentReceipt
    forged = krecs[0].seal.copy()
    forged[7] = (int(forged[7]) + 1) % P
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.lift(SegmentReceipt(seal=forged, index=0, po2=13), family=1)
    # a resolve against a receipt that is NOT what the verifier is told was assumed: proves, but verifies only as what it is
    other = rx.resolve(session_root, lifted[2], 19)
    other.verify(roots, sclaims, assumption_claims=[kclaims[2]])
    with pytest.raises(HalError, match="resolved root"):
        other.verify(roots, sclaims, assumption_claims=kclaims)
    assert hashlib.sha3_256(msgs[0]).digest()                                          # (the batches are real SHA-3 blocks: test above)
    print("UNION/RESOLVE", {"programs": len(programs), "kinds": sorted({k[0] for k in kinds}), "sizes": sorted({p.po2 for p in rx.programs}),
                             "build_s": round(t_build, 2), "union_fold_and_resolve_s": round(t_fold, 3)})


def test_native_session_unites_its_assumption_receipts_and_resolves_the_root(hal):
    """zkh_session_set_assumptions + zkh_session_prove(join_tree = 2): the native executor's plan (csrc/scheduler.h) gets a lift per
    keccak receipt, the union tree and ONE resolve; programs built by the library (zkh_session_build_recursion) in build_programs'
    order.  The root receipt equals the Python driver's resolve(fold_segments(..), union_fold(..)) word for word (fixed noise),
    zkh_session_verify recomputes the resolved claim, a forged assumption receipt is refused when it is handed over."""
    from zeth_amd import recursion as rec
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import HalError, HostCircuit
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver, SegmentReceipt
    sdesc, kdesc = syn_air.syn_a(), keccak_f.keccak_f_circuit()      # (po2-13 SYN-A: program sizes {17, 18} - the set with unions and resolves fits 16)
    sp, kp = SegmentProver(hal, sdesc), SegmentProver(hal, kdesc)
    segs = [Segment(index=i, po2=13, seed=700 + i, noise_seed=0x51) for i in range(6)]
    krecs = [kp.prove_segment(Segment(index=i, po2=13, seed=0xCECC + i, noise_seed=3)) for i in range(3)]
    sroot, kroot = sp.control_root(13), kp.control_root(13)
    programs = rec.build_programs(sdesc, {13: sroot}, assumptions=[(kdesc, {13: kroot})], resolve=True)
    sess = Session(sdesc, devices=(0,), lanes_per_device=2)
    forged = krecs[1].seal.copy()
    forged[7] = (int(forged[7]) + 1) % P
    with pytest.raises(HalError, match="assumption receipt 1"):
        sess.set_assumptions(kdesc, [krecs[0], SegmentReceipt(seal=forged, index=1, po2=13), krecs[2]], {13: kroot})
    sess.set_assumptions(kdesc, krecs, {13: kroot})
    assert len(programs) <= 16 and {"union", "resolve"} <= {k[0] for k, _ in programs}
    with pytest.raises(ValueError, match="no room for resolve"):                    # SYN-small's three program sizes (16, 17, 18) need 9 joins
        rec.build_programs(syn_air.syn_small(), {13: sroot}, assumptions=[(kdesc, {13: kroot})], resolve=True)
    t0 = time.time()
    sess.build_recursion([13])
    t_build = time.time() - t0
    comp, root, stats = sess.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
    sess.close()
    # plan: 3 lift2 at the bottom; above: 1 join3 or 2 joins; 3 assumption lifts + 2 unions + 1 resolve
    m = int(dict(programs)[("lift2", 13, 13)][2])
    has3 = ("join3", m, m, m) in [k for k, _ in programs]
    assert stats["n_lifts"] == 3 + 3 and stats["n_joins"] == (1 if has3 else 2) + 2 + 1 and stats["verified"]
    rx = rec.Recursion(hal, programs)
    assert np.array_equal(root.seal[8:16], rx.allowed_root())                        # the library built build_programs' set, in its order
    leaves = [sp.prove_segment(s) for s in segs]
    want = rx.resolve(rx.fold_segments(leaves, 0x77), rx.union_fold([rx.lift(k, 0x77, family=1) for k in krecs], 0x77), 0x77)
    assert np.array_equal(root.seal, want.seal) and stats["root_program"] == want.program
    assert np.array_equal(stats["root_core"], want.core) and (stats["root_pre"], stats["root_post"]) == (want.pre, want.post)
    sclaims = [HostCircuit(sdesc).receipt_claim(r.seal, sroot) for r in comp.segments]
    kclaims = [HostCircuit(kdesc).receipt_claim(r.seal, kroot) for r in krecs]
    rec.succinct_verify(root.seal, rx.allowed_roots(), stats["root_program"], sclaims, assumption_claims=kclaims)
    with pytest.raises(HalError, match="resolved root"):
        rec.succinct_verify(root.seal, rx.allowed_roots(), stats["root_program"], sclaims, assumption_claims=kclaims[:2])
    # the same session WITHOUT assumptions is another statement: the plain join-tree root
    sess2 = Session(sdesc, devices=(0,), lanes_per_device=2)
    sess2.set_recursion(programs)
    _, plain, st2 = sess2.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
    sess2.close()
    assert not np.array_equal(plain.seal[:8], root.seal[:8]) and st2["n_joins"] == (1 if has3 else 2) and st2["n_lifts"] == 3
    # the g++ host: examples/prove_session --keccak-batches (no Python, no files): keccak batches sealed, handed over, united, resolved
    import subprocess
    from zeth_amd import build
    exe = os.path.join(os.path.dirname(build.build_examples()), "prove_session")
    csv = os.path.join(os.path.dirname(exe), "..", "gpurun_out", "prove_session_keccak.csv")
    os.makedirs(os.path.dirname(csv), exist_ok=True)
    if os.path.exists(csv):
        os.remove(csv)
    r = subprocess.run([exe, "--circuit", "syn_a", "--build-recursion", "--po2", "13", "--segments", "6", "--inflight", "2", "--keccak-batches", "3",
                        "--noise-seed", str(0x51), "--csv", csv, "--block-number", "19000001"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] is True and out["resolved"] is True and out["assumption_receipts"] == 3 and out["lifts"] == stats["n_lifts"] and out["joins"] == stats["n_joins"]
    assert out["root_out"][64:] == "".join(f"{int(w):08x}" for w in rx.allowed_root())        # the same program set: the same allowed-programs root
    cols = open(csv).read().strip().splitlines()[1].split(",")
    assert cols[0] == "19000001" and int(cols[5]) == 3 * ((8192 - 1994) // 25)                # keccak_calls: the permutations the batches prove
    print("NATIVE UNION/RESOLVE", {"programs": len(programs), "library_build_s": round(t_build, 2), "wall_s": round(stats["wall_s"], 3), "nodes": stats["n_lifts"] + stats["n_joins"]})


def test_a_chained_session_that_assumes_receipts_folds_resolves_and_names_them(hal):
    """Everything at once, natively: a CHAINED SYN-S session (exit codes, journal, continuity) that ASSUMES three keccak receipts is
    sealed, folded (lift2 / joins that assert continuity), its assumption receipts lifted and united, the root resolved — and the last
    segment's seal binds Output{journal, assumptions} over exactly those receipts: the Python verifier accepts the composite with that
    list and refuses it reordered (round-5 verdict, missing #5)."""
    from zeth_amd.circuits import keccak_f
    from zeth_amd.hal import HalError
    from zeth_amd.host import Receipt, Session, assumption_of, image_id, output_digest, segment_claim
    from zeth_amd.prover import Segment, SegmentProver
    sdesc, kdesc = syn_air.syn_session(), keccak_f.keccak_f_circuit()
    sp, kp = SegmentProver(hal, sdesc), SegmentProver(hal, kdesc)
    krecs = [kp.prove_segment(Segment(index=i, po2=13, seed=0xCECC + i, noise_seed=3)) for i in range(3)]
    kroot, sroot = kp.control_root(13), sp.control_root(13)
    assumed = [assumption_of(r, kdesc, kroot) for r in krecs]
    segs = [Segment(index=i, po2=13, seed=4700 + i, noise_seed=0x51) for i in range(6)]
    sess = Session(sdesc, devices=(0,), lanes_per_device=2)
    sess.set_assumptions(kdesc, krecs, {13: kroot})
    sess.set_chained(True, 4)
    sess.build_recursion([13])
    comp, root, st = sess.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
    sess.close()
    assert st["n_lifts"] == 3 + 3 and st["verified"] and root is not None
    claims = [segment_claim(r) for r in comp.segments]
    journal = int(claims[-1].post).to_bytes(4, "little")
    assert claims[-1].output == output_digest(journal, assumed) and all(c.output is None for c in claims[:-1])
    rec_ = Receipt(comp, journal, tuple(assumed))
    rec_.verify(image_id(sdesc, 4), sdesc, initial_state=4, control_root={13: sroot})
    rec_.verify_assumptions(kdesc, krecs, {13: kroot})
    with pytest.raises(HalError, match="do not hash"):
        Receipt(comp, journal, tuple(assumed[::-1])).verify(image_id(sdesc, 4), sdesc, initial_state=4, control_root={13: sroot})
