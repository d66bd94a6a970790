"""Round-2 parity tests on the GPU: byte-exactness at the BASELINE shape (po2-20 SYN-A) op by op and for the whole
seal, the control-root binding of the verifier, the host-witness ingress path (zkh_prove_begin / zkh_prove_finish +
pinned uploads), joins that commit to their children, the keccak-like third circuit, and several contexts per process."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import zko
from zeth_amd.circuits import syn_air
from zeth_amd.circuits.desc import Circuit
from zeth_amd.hal import HalError, HipHal
from zeth_amd.prover import Segment, SegmentProver, shipped_control_root

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


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


def test_contexts_on_every_visible_device_from_threads(oracle):
    """One context + prover per visible device, each driven by its own host thread, all sealing the same segments:
    identical seals everywhere (per-device kernel attributes, per-thread device binding, atomic handle refcounts)."""
    import ctypes as C
    from zeth_amd import hal as zhal
    zhal.load_library()
    n_dev = 0
    while True:
        try:
            h = HipHal(n_dev)
        except HalError:
            break
        h.close()
        n_dev += 1
        if n_dev >= 8:
            break
    assert n_dev >= 1
    desc = syn_air.syn_small()
    lanes = [(d, k) for d in range(n_dev) for k in range(2)]
    out, errs = {}, []

    def work(dev, k):
        try:
            h = HipHal(dev)
            prover = SegmentProver(h, desc)
            out[(dev, k)] = [prover.prove_segment(Segment(index=i, po2=12, seed=70 + i, noise_seed=5)).seal for i in range(3)]
            del prover
            h.close()
        except Exception as e:          # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=l) for l in lanes]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    oc = zko.OracleCircuit(oracle, desc)
    want = [oc.prove(12, 1994, 70 + i, 5) for i in range(3)]
    for l in lanes:
        for i in range(3):
            assert np.array_equal(out[l][i], want[i]), f"device {l[0]} lane {l[1]} segment {i}"


def test_version_names_the_provenance_of_the_poseidon2_tables():
    from zeth_amd import hal as zhal
    v = zhal.load_library().zkh_version().decode()
    assert "gfx950" in v and "poseidon2_consts=derived" in v


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


@pytest.mark.parametrize("bitrev", [False, True])
def test_batch_evaluate_any_runs_of_equal_columns(hal, oracle, bitrev):
    """Taps of one register = consecutive entries with the same `which`: one block streams the column once for up to 8
    points.  Runs of every length around that limit, interleaved with singletons, against the oracle."""
    rng = np.random.default_rng(77)
    po2, count = 14, 12
    n = 1 << po2
    coeffs = rng.integers(0, 2013265921, size=count * n, dtype=np.uint64).astype(np.uint32)
    which = []
    for col, run in zip([3, 0, 7, 7, 1, 11, 5, 2, 9, 4, 4, 6], [1, 5, 8, 9, 2, 17, 1, 3, 16, 7, 1, 24]):
        which += [col] * run
    which = np.array(which, dtype=np.uint32)
    xs = rng.integers(0, 2013265921, size=4 * which.size, dtype=np.uint64).astype(np.uint32)
    want = np.zeros(4 * which.size, np.uint32)
    oracle.zko_batch_evaluate_any(coeffs, coeffs.size, count, which, xs, which.size, want)
    dev = coeffs.copy()
    if bitrev:
        oracle.zko_batch_bit_reverse(dev, dev.size, count)       # the layout batch_interpolate_ntt leaves behind
    out = hal.alloc_elem("out", 4 * which.size)
    fn = hal.batch_evaluate_any_bitrev if bitrev else hal.batch_evaluate_any
    fn(hal.copy_from("c", dev), count, hal.copy_from("w", which), hal.copy_from("x", xs), out)
    assert np.array_equal(out.to_vec(), want)


def test_combos_divide_all_matches_sequential_division(hal, oracle):
    """Seven combo polynomials with 1..5 division points each (SYN-HEAVY's tap combos): the batched rounds give the same
    quotients and remainders as dividing every polynomial by its points one after the other on the CPU."""
    rng = np.random.default_rng(91)
    cycles, counts = 1 << 13, [1, 2, 3, 4, 5, 3, 2, 1]
    P = 2013265921
    combos = rng.integers(0, P, size=4 * cycles * len(counts), dtype=np.uint64).astype(np.uint32)
    pts = rng.integers(0, P, size=4 * sum(counts), dtype=np.uint64).astype(np.uint32)
    begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    want, want_rem = combos.copy(), np.zeros(4 * sum(counts), np.uint32)
    for i, cnt in enumerate(counts):
        poly = want[4 * cycles * i: 4 * cycles * (i + 1)]
        for k in range(cnt):
            j = int(begin[i]) + k
            rem = np.zeros(4, np.uint32)
            oracle.zko_poly_divide(poly, cycles, pts[4 * j: 4 * j + 4].copy(), rem)
            want_rem[4 * j: 4 * j + 4] = rem
    dev = hal.copy_from("combos", combos)
    rem_out = hal.alloc("rem", 4 * sum(counts), zero=True)
    hal.combos_divide_all(dev, cycles, pts, begin, rem_out)
    assert np.array_equal(dev.to_vec(), want)
    assert np.array_equal(rem_out.to_vec(), want_rem)


def test_cpp_host_attaches_code_objects_and_writes_receipts(tmp_path, monkeypatch):
    """A non-Python host with a circuit that is NOT built into the library: the generated eval_check kernels arrive as code
    objects (`python -m zeth_amd.circuits.jit` wrote them + a manifest ahead of time), examples/seal_segments attaches them
    through zkh_circuit_attach_code_object_part, seals, verifies against the control root and writes receipt containers,
    which the Python side parses back and verifies again."""
    import subprocess
    from zeth_amd import build
    from zeth_amd.circuits import jit, syn_heavy
    from zeth_amd.prover import SegmentReceipt
    monkeypatch.setenv("ZKH_CODEGEN_PART", "1600")
    desc = syn_heavy.syn_heavy_small()                      # two kernels, no built-in match
    desc_path = tmp_path / "c.desc"
    np.asarray(desc, dtype="<u4").tofile(desc_path)
    objs = tmp_path / "objs"
    assert jit.main(["jit", str(desc_path), str(objs)]) == 0
    assert (objs / "manifest.txt").read_text().count("\n") >= 2
    rdir = tmp_path / "receipts"
    rdir.mkdir()
    exe = build.build_examples()
    r = subprocess.run([exe, "--desc", str(desc_path), "--po2", "13", "--segments", "3", "--inflight", "2", "--noise-seed", "7",
                        "--code-objects", str(objs), "--receipts-dir", str(rdir)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] == 3
    blobs = []
    for i in range(3):
        blobs.append(np.fromfile(rdir / f"segment_{i}.zkr", dtype="<u4"))
        rec = SegmentReceipt.from_words(desc, blobs[-1])
        assert rec.index == i and rec.po2 == 13
        rec.verify(desc, rec.control_root)                  # the driver computed the root on the GPU; same circuit, same po2
    # without the code objects the same host still works (step interpreter), and gives the same seal for the same noise
    r2 = subprocess.run([exe, "--desc", str(desc_path), "--po2", "13", "--segments", "1", "--inflight", "1", "--noise-seed", "7",
                         "--receipts-dir", str(rdir)], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    assert np.array_equal(np.fromfile(rdir / "segment_0.zkr", dtype="<u4"), blobs[0])


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
