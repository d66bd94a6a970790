"""Round 4 on the GPU: the fused Merkle build (leaves + their parents in one pass), the session executor as one pipeline
(streamed fold, resident code groups, per-segment retry), host placement, and the compact-trace witness ingress."""
import os

import numpy as np
import pytest

from conftest import rand_fp
from zeth_amd.circuits import syn_air

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_rows,cols", [(17, 0), (17, 5), (17, 16), (17, 17), (18, 33), (17, 208), (12, 40)])
@pytest.mark.parametrize("fused", [False, True])
def test_merkle_build_equals_hash_rows_plus_fold_all(oracle, log_rows, cols, fused, monkeypatch):
    """zkh_merkle_build — the default path and the opt-in fused first pass (ZKH_MERKLE_FUSED=1: k_hash_rows_pair = two adjacent rows
    per lane + their parent) — gives the nodes zkh_hash_rows + zkh_merkle_fold_all give, and the oracle's whole tree."""
    import subprocess, sys
    if fused:
        # the switch is read once per process: run this case in a child interpreter
        env = dict(os.environ, ZKH_MERKLE_FUSED="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                            f"{__file__}::test_merkle_build_equals_hash_rows_plus_fold_all[False-{log_rows}-{cols}]"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        return
    from zeth_amd.hal import HipHal
    hal = HipHal(0)
    rng = np.random.default_rng(400 + cols + log_rows)
    rows = 1 << log_rows
    mat = rand_fp(rng, cols * rows) if cols else np.zeros(0, np.uint32)
    m = hal.copy_from("m", mat) if cols else hal.alloc("m", 0)
    fused = hal.alloc_digest("nodes", 2 * rows)
    hal.merkle_build(fused, m, rows)
    plain = hal.alloc_digest("nodes2", 2 * rows)
    hal.hash_rows(plain.slice(rows * 8, rows * 8), m)
    hal.merkle_fold_all(plain, rows)
    a, b = fused.to_vec(), plain.to_vec()
    assert np.array_equal(a[8:], b[8:])                      # nodes[1 .. 2 rows): node 0 is unused
    # oracle: a few leaves, their parent, and the whole tree's root
    want = np.zeros(rows * 8, dtype=np.uint32)
    oracle.zko_hash_rows(want, rows, np.ascontiguousarray(mat) if cols else np.zeros(1, np.uint32), rows * cols)
    assert np.array_equal(a[rows * 8:], want)
    nodes = np.zeros(2 * rows * 8, dtype=np.uint32)
    nodes[rows * 8:] = want
    size = rows
    while size > 1:
        oracle.zko_hash_fold(nodes, size, size // 2)
        size //= 2
    assert np.array_equal(a[8:], nodes[8:])


def test_session_streamed_fold_equals_two_phases_and_keeps_code_resident(hal):
    """zkh_session_prove(join_tree = 2) as one pipeline (fold nodes proven as their children appear, concurrently with the sealing
    lanes) gives the root receipt of the two-phase run WORD FOR WORD (fixed noise), with the code group resident or recomputed."""
    from zeth_amd import recursion as rec
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_small()
    sp = SegmentProver(hal, desc)
    segs = [Segment(index=i, po2=13 if i < 6 else 12, seed=1200 + i, noise_seed=0x51) for i in range(7)]
    roots = {13: sp.control_root(13), 12: sp.control_root(12)}
    programs = rec.build_programs(desc, roots)
    m = int(dict(programs)[("lift2", 13, 13)][2])             # size of a lift2 node: three of them are ONE proof if the set has that join3
    has3 = ("join3", m, m, m) in [k for k, _ in programs]
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_recursion(programs)
    results = {}
    for streamed in (True, False):
        for resident in (True, False):
            sess.set_streamed_fold(streamed)
            sess.set_resident_code(resident)
            comp, root, st = sess.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
            # 3 lift2 + 1 lift; above: (lift2 lift2 lift2) is one join3 where the set has it, else two joins; then one join with the lift
            assert st["streamed_fold"] == streamed and st["n_lifts"] == 4 and st["n_joins"] == (2 if has3 else 3) and st["n_retries"] == 0
            results[(streamed, resident)] = (root.seal.copy(), [r.seal.copy() for r in comp.segments])
    ref_root, ref_leaves = results[(False, False)]
    leaves = [sp.prove_segment(s) for s in segs]             # the Python mirror's seals (code group recomputed)
    for k, (root_seal, leaf_seals) in results.items():
        assert np.array_equal(root_seal, ref_root), k
        for a, b, c in zip(leaf_seals, ref_leaves, leaves):
            assert np.array_equal(a, b) and np.array_equal(a, c.seal), k
    want = rec.Recursion(hal, programs).fold_segments(leaves, 0x77)
    assert np.array_equal(ref_root, want.seal)
    sess.close()


def test_session_retries_a_failed_segment_on_another_lane(hal, monkeypatch):
    """ZKH_FAULT_SEGMENT=k makes the first attempt at segment k fail: the session hands it to another lane and finishes with the
    same receipts; a segment that fails every time (ZKH_FAULT_SEGMENT_ALWAYS) fails the session with its index in the error."""
    from zeth_amd.hal import HalError
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment
    desc = syn_air.syn_small()
    segs = [Segment(index=i, po2=12, seed=1300 + i, noise_seed=0x52) for i in range(6)]
    sess = Session(desc, devices=(0,), lanes_per_device=3)
    comp0, _, st0 = sess.prove(segs, verify=True)
    assert st0["n_retries"] == 0
    monkeypatch.setenv("ZKH_FAULT_SEGMENT", "2")
    comp1, _, st1 = sess.prove(segs, verify=True)
    assert st1["n_retries"] == 1
    for a, b in zip(comp0.segments, comp1.segments):
        assert np.array_equal(a.seal, b.seal)
    monkeypatch.delenv("ZKH_FAULT_SEGMENT")
    monkeypatch.setenv("ZKH_FAULT_SEGMENT_ALWAYS", "4")
    with pytest.raises(HalError, match=r"segment 4 \(after 2 attempt"):
        sess.prove(segs)
    monkeypatch.setenv("ZKH_SEGMENT_RETRIES", "0")
    with pytest.raises(HalError, match=r"segment 4 \(after 1 attempt"):
        sess.prove(segs)
    sess.close()


def test_host_placement_of_this_device():
    """the NUMA node of device 0 (or -1 where the host reports none) and binding the calling thread next to it never fail; when a
    node is reported the thread's CPU mask becomes a subset of that node's CPU list"""
    import threading
    from zeth_amd import hal as H
    node, bdf = H.device_numa_node(0)
    assert node >= -1 and len(bdf.split(":")) == 3
    out = {}

    def run():                       # in a thread of its own: the binding must not leak into the test process
        out["r"] = H.bind_to_device(0)
        out["mask"] = os.sched_getaffinity(0)
    th = threading.Thread(target=run)
    th.start(); th.join()
    assert out["r"]["numa_node"] in (-1, node)
    if out["r"]["numa_node"] >= 0:
        _, cpus = H.pci_numa_cpus(bdf)
        assert out["mask"] <= set(cpus) and out["r"]["cpus"] == len(out["mask"])
    assert H.placement_slot(0, [0]) == (0, 1)


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


def test_session_with_host_preflight_pipeline(hal, oracle, monkeypatch):
    """zkh_session_set_witness_source(1): producer threads replay every segment's cycles on the host ahead of the seals; receipts
    equal the op-by-op path's and the oracle's, the host CPU time and the PCIe bytes are reported, a faulted segment is retried."""
    import zko
    from zeth_amd import hal as H
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_a()
    oc = zko.OracleCircuit(oracle, desc)
    segs = [Segment(index=i, po2=13 if i < 5 else 12, seed=1400 + i, noise_seed=0x53) for i in range(6)]
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_witness_source(1, 2)
    comp, _, st = sess.prove(segs, verify=True)
    words = sum(16 * ((1 << s.po2) - 1994) + 4 * 1024 + 16 for s in segs)
    assert st["preflight_cpu_s_sum"] > 0 and st["trace_bytes"] == words and st["n_retries"] == 0
    sp = SegmentProver(hal, desc)
    for s, r in zip(segs, comp.segments):
        rec, ram, _ = H.syn_preflight(s.seed, s.po2)
        ocode, odata, oout = oc.witgen_trace(s.po2, rec, ram, s.noise_seed)
        assert np.array_equal(r.seal, oc.prove_traces(s.po2, ocode, odata, oout, noise_seed=s.noise_seed))
    monkeypatch.setenv("ZKH_FAULT_SEGMENT", "3")
    comp2, _, st2 = sess.prove(segs, verify=True)
    assert st2["n_retries"] == 1 and all(np.array_equal(a.seal, b.seal) for a, b in zip(comp.segments, comp2.segments))
    monkeypatch.delenv("ZKH_FAULT_SEGMENT")
    # back to the closed-form generator: different witnesses, hence different seals, same session
    sess.set_witness_source(0)
    comp3, _, st3 = sess.prove(segs, verify=True)
    assert st3["trace_bytes"] == 0 and not np.array_equal(comp3.segments[0].seal, comp.segments[0].seal)
    sess.close()


def test_combos_prepare_regs_takes_any_register_size_and_refuses_inconsistent_lists(hal):
    """Round-3 advisor finding: registers larger than 32 silently lost their terms and the sizes were never checked against coeff_u.
    Now: any size up to `cycles` (against the host aggregation), and an inconsistent register list is an ERROR before the launch."""
    import hal_only_prover as hop
    from zeth_amd.hal import HalError
    P = 2013265921
    rng = np.random.default_rng(11)
    cycles, combo_count = 128, 3
    sizes = np.array([1, 40, 3, 97, 32, 33, 128, 2], dtype=np.uint32)
    ids = rng.integers(0, combo_count, size=sizes.size).astype(np.uint32)
    n_u = int(sizes.sum()) + hop.CHECK_SIZE
    coeff_u = rng.integers(0, P, size=4 * n_u, dtype=np.uint64).astype(np.uint32)
    start = rng.integers(0, P, size=4 * cycles * (combo_count + 1), dtype=np.uint64).astype(np.uint32)
    mix = tuple(int(x) for x in rng.integers(1, P, size=4))
    a = hal.copy_from("combos", start)
    hal.combos_prepare_regs(a, hal.copy_from("cu", coeff_u), combo_count, cycles, hal.copy_from("s", sizes), hal.copy_from("i", ids), hop.e_words(mix))
    sub, cur, pos = {}, (1, 0, 0, 0), 0
    cu = [tuple(hop.dec(coeff_u[4 * k + i]) for i in range(4)) for k in range(n_u)]
    for sz, cid in zip(sizes, ids):
        for i in range(int(sz)):
            key = cycles * int(cid) + i
            sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos + i]))
        cur = hop.e_mul(cur, mix)
        pos += int(sz)
    for _ in range(hop.CHECK_SIZE):
        key = cycles * combo_count
        sub[key] = hop.e_add(sub.get(key, (0, 0, 0, 0)), hop.e_mul(cur, cu[pos]))
        pos += 1
        cur = hop.e_mul(cur, mix)
    b = hal.copy_from("combos", start)
    hal.combos_prepare(b, np.asarray(list(sub), dtype=np.uint32), np.asarray([w for v in sub.values() for w in hop.e_words(v)], dtype=np.uint32))
    assert np.array_equal(a.to_vec(), b.to_vec())

    def call(sz, idv, cu_words):
        hal.combos_prepare_regs(hal.copy_from("combos", start), hal.copy_from("cu", coeff_u[:cu_words]), combo_count, cycles,
                                hal.copy_from("s", np.asarray(sz, dtype=np.uint32)), hal.copy_from("i", np.asarray(idv, dtype=np.uint32)), hop.e_words(mix))
    with pytest.raises(HalError, match="add up to"):                  # sizes claim more U coefficients than coeff_u holds
        call(sizes, ids, 4 * (n_u - 5))
    with pytest.raises(HalError, match="has size"):                   # a register larger than the polynomial
        call([cycles + 1] + list(sizes[1:]), ids, coeff_u.size)
    with pytest.raises(HalError, match="has size"):
        call([0] + list(sizes[1:]), ids, coeff_u.size)
    with pytest.raises(HalError, match="names combo"):
        call(sizes, [combo_count] + list(ids[1:]), coeff_u.size)


def test_chained_session_on_the_gpu(hal, oracle):
    """Claim continuity through the native session executor: zkh_session_set_chained runs the executor's pass on the GPU (one launch:
    every segment's contribution to the running state), proves each segment with its pre-state as public input, and
    zkh_session_verify checks pre == prev.post on the seals.  Seals equal the oracle's for the same pre-states; the Python
    CompositeReceipt check agrees; a session started from another state is a different, equally continuous session."""
    import zko
    from zeth_amd.circuits.syn_air import syn_chain_small
    from zeth_amd.hal import P, fp_encode
    from zeth_amd.host import Session, chain_segments
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_chain_small()
    oc = zko.OracleCircuit(oracle, desc)
    sp = SegmentProver(hal, desc)
    segs = [Segment(index=i, po2=13 if i != 2 else 12, seed=1500 + i, noise_seed=0x54) for i in range(5)]
    # the executor's pass: GPU contributions equal the oracle's
    contrib = [sp.chain_contribution(s) for s in segs]
    for s, c in zip(segs, contrib):
        assert c == int(oc.witgen(s.po2, 1994, s.seed, s.noise_seed, pub=np.zeros(1, np.uint32))[2][0])
    want = chain_segments(segs, lambda s: contrib[s.index], initial_state=7)
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_chained(True, 7)
    comp, _, st = sess.prove(segs, verify=True)
    roots = {p: sp.control_root(p) for p in (12, 13)}
    comp.verify(desc, roots, chained=True, initial_state=7)
    for s, r in zip(want, comp.segments):
        assert int(r.seal[4]) == s.pub[0]
        assert np.array_equal(r.seal, oc.prove(s.po2, 1994, s.seed, s.noise_seed, pub=np.asarray(s.pub, dtype=np.uint32)))
    assert comp.final_state() == (fp_encode(7) + sum(contrib)) % P
    with pytest.raises(ValueError, match="not continuous"):
        comp.verify(desc, roots, chained=True, initial_state=0)
    sess.set_chained(True, 0)
    comp0, _, _ = sess.prove(segs, verify=True)
    comp0.verify(desc, roots, chained=True, initial_state=0)
    assert comp0.final_state() == sum(contrib) % P and not np.array_equal(comp0.segments[0].seal, comp.segments[0].seal)
    sess.close()


def test_chained_session_folds_to_one_receipt_whose_joins_asserted_continuity(hal):
    """Continuity IN-CIRCUIT: every recursion receipt publishes claim' = hash_pair(core, (pre, post, 0..)); lift2 and join open their
    children's claim' and assert post(left) = pre(right).  A chained SYN-C session folds to one receipt natively and in the Python
    driver (same root, word for word); the root follows from the leaves' (claim, pre, post); two segments that do NOT chain have no
    lift2 witness, and a claim tree over a broken chain is refused on the host as well."""
    from zeth_amd import recursion as rec
    from zeth_amd.circuits.syn_air import syn_chain_small
    from zeth_amd.hal import HalError, HostCircuit
    from zeth_amd.host import Session, chain_segments
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_chain_small()
    sp = SegmentProver(hal, desc)
    base = [Segment(index=i, po2=13, seed=1600 + i, noise_seed=0x55) for i in range(4)]
    roots = {13: sp.control_root(13)}
    programs = rec.build_programs(desc, roots)
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_recursion(programs)
    sess.set_chained(True, 9)
    comp, root, st = sess.prove(base, join_tree=2, join_noise_seed=0x78, verify=True)          # zkh_session_verify: seals, chain, claim tree
    assert st["n_lifts"] == 2 and st["n_joins"] == 1
    comp.verify(desc, roots, chained=True, initial_state=9)
    rx = rec.Recursion(hal, programs)
    want = rx.fold_segments(comp.segments, 0x78)
    assert np.array_equal(root.seal, want.seal)
    leaves = [(HostCircuit(desc).receipt_claim(r.seal, roots[13]), int(r.seal[4]), int(r.seal[0])) for r in comp.segments]
    want.verify(rx.allowed_roots(), leaves)
    assert (want.pre, want.post) == (int(comp.segments[0].seal[4]), comp.final_state())
    assert np.array_equal(want.claim, rec.fold_leaf_claims(leaves))
    # a pair that does not chain: segment 2 after segment 0 — the fused lift has no witness; the host-side tree refuses too
    with pytest.raises(HalError, match="assertion of the program fails"):
        rx.lift2(comp.segments[0], comp.segments[2])
    with pytest.raises(HalError, match="do not chain"):
        rec.fold_leaf_claims([leaves[0], leaves[2], leaves[1], leaves[3]])
    with pytest.raises(HalError):                                                              # plain claims (state 0, 0) are another tree
        want.verify(rx.allowed_roots(), [l[0] for l in leaves])
    # without the executor's pass (arbitrary public inputs) the session cannot be folded: its joins would not chain
    sess.set_chained(False)
    loose = [Segment(index=i, po2=13, seed=1600 + i, noise_seed=0x55, pub=(i + 1,)) for i in range(4)]
    with pytest.raises(HalError, match="do not chain|assertion of the program fails"):
        sess.prove(loose, join_tree=2, join_noise_seed=0x78)
    sess.close()


def test_session_over_a_device_list_with_streamed_fold(hal):
    """The G devices x K lanes shape of the executor (here the device list names GPU 0 twice: two "devices" x 2 lanes + their
    fold-only lanes): segments and fold nodes are pulled by lanes of both, receipts come back in index order and the root equals
    the single-device session's word for word."""
    from zeth_amd import recursion as rec
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment, SegmentProver
    desc = syn_air.syn_small()
    sp = SegmentProver(hal, desc)
    segs = [Segment(index=i, po2=13 if i != 5 else 12, seed=1700 + i, noise_seed=0x56) for i in range(9)]
    programs = rec.build_programs(desc, {13: sp.control_root(13), 12: sp.control_root(12)})
    roots = {}
    for devices in ((0,), (0, 0)):
        sess = Session(desc, devices=devices, lanes_per_device=2)
        sess.set_recursion(programs)
        comp, root, st = sess.prove(segs, join_tree=2, join_noise_seed=0x79, verify=True)
        assert [r.index for r in comp.segments] == list(range(9)) and st["n_retries"] == 0
        roots[devices] = (root.seal.copy(), [r.seal.copy() for r in comp.segments])
        sess.close()
    assert np.array_equal(roots[(0,)][0], roots[(0, 0)][0])
    assert all(np.array_equal(a, b) for a, b in zip(roots[(0,)][1], roots[(0, 0)][1]))


def test_gathered_power_tables_travel_inside_the_code_objects(hal, oracle, tmp_path, monkeypatch):
    """Round 4's eval_check generator gives every kernel its own mix-power table in emission order and exports the exponent list
    as `<kernel>_exps` inside the code object: a host that attaches the parts needs to know nothing about it.  Kernels generated
    WITH the table and WITHOUT it give the interpreter's words; a set that mixes the two is refused, not launched."""
    from zeth_amd.circuits import codegen, jit, syn_heavy
    from zeth_amd.hal import HalError
    from zeth_amd.prover import Segment, SegmentProver
    from test_round2_gpu import _evaluated_groups
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
