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
    sess = Session(desc, devices=(0,), lanes_per_device=2)
    sess.set_recursion(programs)
    results = {}
    for streamed in (True, False):
        for resident in (True, False):
            sess.set_streamed_fold(streamed)
            sess.set_resident_code(resident)
            comp, root, st = sess.prove(segs, join_tree=2, join_noise_seed=0x77, verify=True)
            assert st["streamed_fold"] == streamed and st["n_lifts"] == 4 and st["n_joins"] == 3 and st["n_retries"] == 0
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
