"""Runtime behaviour of the C-ABI library on the GPU: constant-table swap, several contexts driven from host threads,
caller-owned device memory (torch), profiling records, size limits, full-size properties."""
import ctypes as C
import threading

import numpy as np
import pytest

import zko
from conftest import P, rand_fp
from zeth_amd.circuits import syn_air
from zeth_amd.hal import HalError, HipHal
from zeth_amd.prover import Segment, SegmentProver

pytestmark = pytest.mark.gpu


def test_poseidon2_constants_are_data(oracle):
    """consts.rs is data: swapping the tables on both sides keeps HIP == oracle (and changes the digests)."""
    rng = np.random.default_rng(42)
    rc = rng.integers(0, P, size=24 * 29, dtype=np.uint64).astype(np.uint32)
    diag = rng.integers(1, P, size=24, dtype=np.uint64).astype(np.uint32)
    m = rand_fp(rng, 64 * 37)
    base = np.zeros(64 * 8, np.uint32)
    oracle.zko_hash_rows(base, 64, m, m.size)
    hal = HipHal(0)
    try:
        hal.poseidon2_set_constants(rc, diag)
        oracle.zko_poseidon2_set_constants(rc, diag)
        want = np.zeros(64 * 8, np.uint32)
        oracle.zko_hash_rows(want, 64, m, m.size)
        out = hal.alloc_digest("o", 64)
        hal.hash_rows(out, hal.copy_from("m", m))
        assert np.array_equal(out.to_vec(), want)
        assert not np.array_equal(want, base)
        # 2->1 compression through both fold kernels (one lane per parent, and the 8-lane cooperative one)
        nodes = np.zeros(128 * 8, np.uint32)
        nodes[64 * 8:] = want
        wn = nodes.copy()
        layer = 64
        while layer > 1:
            oracle.zko_hash_fold(wn, layer, layer // 2)
            layer //= 2
        nb = hal.copy_from("n", nodes)
        hal.merkle_fold_all(nb, 64)
        assert np.array_equal(nb.to_vec()[8:], wn[8:])
    finally:
        # restore the shipped tables in the oracle (process-global there)
        import re
        txt = open(zko._ORACLE_DIR + "/../include/zkh_poseidon2_consts.h").read()
        nums = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)]
        oracle.zko_poseidon2_set_constants(np.array(nums[24:], np.uint32), np.array(nums[:24], np.uint32))
        hal.close()


def _restore_shipped_constants(oracle):
    import re
    txt = open(zko._ORACLE_DIR + "/../include/zkh_poseidon2_consts.h").read()
    nums = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)]
    oracle.zko_poseidon2_set_constants(np.array(nums[24:], np.uint32), np.array(nums[:24], np.uint32))


@pytest.mark.parametrize("rc_fill,diag_fill", [(0, 1), (P - 1, P - 1), (P - 1, (P - 1) // 2), (1, (P + 1) // 2), ((P - 1) // 2, P - 2)])
def test_poseidon2_edge_constants_and_edge_inputs(oracle, rc_fill, diag_fill):
    """The kernels carry the sponge state in scaled, lazily reduced forms (poseidon2.h): tables and inputs at the edges
    of those forms' operand bounds (0, P-1, the centring boundary) must still give the literal oracle's digests."""
    rng = np.random.default_rng(rc_fill % 1000 + diag_fill % 977)
    rc = np.full(24 * 29, rc_fill, np.uint32)
    diag = np.full(24, diag_fill, np.uint32)
    rows, cols = 256, 37
    edge = np.array([0, 1, 268435454, P - 1, (P - 1) // 2, (P + 1) // 2, P - 2], np.uint32)   # every u32 < P is a Montgomery word
    mats = [np.zeros(rows * cols, np.uint32), np.full(rows * cols, P - 1, np.uint32),
            edge[rng.integers(0, edge.size, rows * cols)], rand_fp(rng, rows * cols)]
    hal = HipHal(0)
    try:
        hal.poseidon2_set_constants(rc, diag)
        oracle.zko_poseidon2_set_constants(rc, diag)
        for m in mats:
            want = np.zeros(rows * 8, np.uint32)
            oracle.zko_hash_rows(want, rows, m, m.size)
            out = hal.alloc_digest("o", rows)
            hal.hash_rows(out, hal.copy_from("m", m))
            assert np.array_equal(out.to_vec(), want)
            nodes = np.zeros(2 * rows * 8, np.uint32)
            nodes[rows * 8:] = want
            wn = nodes.copy()
            layer = rows
            while layer > 1:
                oracle.zko_hash_fold(wn, layer, layer // 2)
                layer //= 2
            nb = hal.copy_from("n", nodes)
            hal.hash_fold(nb, rows, rows // 2)                   # the lane-per-parent kernel on the widest layer
            assert np.array_equal(nb.to_vec()[rows // 2 * 8:rows * 8], wn[rows // 2 * 8:rows * 8])
            hal.merkle_fold_all(nb, rows)
            assert np.array_equal(nb.to_vec()[8:], wn[8:])
    finally:
        _restore_shipped_constants(oracle)
        hal.close()


def test_two_contexts_from_two_threads_give_identical_seals(oracle):
    desc = syn_air.syn_small()
    out = [None, None]

    def work(k):
        hal = HipHal(0)
        prover = SegmentProver(hal, desc)
        out[k] = [prover.prove_segment(Segment(index=i, po2=13, seed=900 + i, noise_seed=0x2E80)).seal for i in range(3)]
        del prover
        hal.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    oc = zko.OracleCircuit(oracle, desc)
    for i in range(3):
        assert np.array_equal(out[0][i], out[1][i])
        assert np.array_equal(out[0][i], oc.prove(13, 1994, 900 + i, 0x2E80))


def test_wrap_torch_memory_and_stream(hal, oracle):
    """The boundary takes plain device pointers: a torch tensor's storage can be used in place (no torch types in the ABI)."""
    import torch
    rng = np.random.default_rng(4)
    x = rand_fp(rng, 2 << 12)
    try:
        t32 = torch.from_numpy(x.view(np.int32).copy()).to("cuda:0")
    except (RuntimeError, AssertionError) as e:      # torch is plumbing here, not the product: its own HIP init may fail
        pytest.skip(f"torch cannot use the GPU in this process: {e}")
    torch.cuda.synchronize()
    buf = hal.wrap(t32.data_ptr(), x.size)
    assert buf.size() == x.size and buf.device_ptr() == t32.data_ptr()
    hal.batch_interpolate_ntt(buf, 2)
    hal.sync()
    want = x.copy()
    oracle.zko_batch_interpolate_ntt(want, want.size, 2)
    assert np.array_equal(t32.cpu().numpy().view(np.uint32), want)
    del buf            # never frees caller-owned memory
    assert int(t32[0].item()) == int(want.view(np.int32)[0])


def test_profiling_records(hal):
    hal.prof_reset()
    hal.prof_enable(True)
    a = hal.copy_from("a", rand_fp(np.random.default_rng(0), 1 << 16))
    hal.batch_interpolate_ntt(a, 1)
    hal.batch_bit_reverse(a, 1)
    recs = {r["name"]: r for r in hal.prof_get()}
    hal.prof_enable(False)
    # NTT records are per pass, "<Hal op>:<kernel>"; 2^16 = two generic LDS passes.  The §8d algorithmic bytes of the op
    # (operands once in, once out: 8 n) are charged once, to the first pass
    ntt = [r for n, r in recs.items() if n.startswith("batch_interpolate_ntt:")]
    assert sum(r["calls"] for r in ntt) == 2
    assert recs["batch_bit_reverse"]["calls"] == 1 and recs["batch_bit_reverse"]["total_ms"] > 0
    assert sum(r["alg_bytes"] for r in ntt) == 8.0 * (1 << 16)


def test_size_limits(hal):
    with pytest.raises(HalError, match="exceeds"):
        hal.batch_interpolate_ntt(hal.alloc_elem("big", 1 << 27), 1)        # the largest domain is 2^26 (a po2-24 segment x INV_RATE 4)
    a = hal.alloc_elem("z", 0)
    assert a.size() == 0


@pytest.mark.parametrize("log_n", [22, 24])
def test_ntt_roundtrip_max_sizes(hal, log_n):
    """interpolate then evaluate (expand 0) is the identity at the largest domains (2^22 = po2-20 domain, 2^24 = limit)."""
    rng = np.random.default_rng(log_n)
    x = rand_fp(rng, 1 << log_n)
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt(buf, 1)
    out = hal.alloc_elem("out", 1 << log_n)
    hal.batch_expand_into_evaluate_ntt(out, buf, 1, 0)
    assert np.array_equal(out.to_vec(), x)
    # and the coset shift composes: shift then bit-reverse twice is the shift
    hal.zk_shift(buf, 1)
    y = buf.to_vec()
    hal.batch_bit_reverse(buf, 1)
    hal.batch_bit_reverse(buf, 1)
    assert np.array_equal(buf.to_vec(), y)


def test_merkle_root_matches_independent_path_check(hal, oracle):
    """Full-size property: for a 2^22-leaf tree, an opened path recomputed with the oracle's hash_pair reaches the root."""
    rng = np.random.default_rng(77)
    rows, cols = 1 << 22, 3
    mat = hal.copy_from("m", rand_fp(rng, rows * cols))
    nodes = hal.alloc_digest("n", 2 * rows)
    hal.hash_rows(nodes.slice(rows * 8, rows * 8), mat)
    hal.merkle_fold_all(nodes, rows)
    root = nodes.slice(8, 8).to_vec()
    top = nodes.slice(32 * 8, 32 * 8).to_vec().reshape(32, 8)
    idx = np.array([0, 1, rows - 1, 123456, 4000000], np.uint32)
    wpq = cols + 8 * (22 - 5)
    out = hal.alloc("o", wpq * idx.size)
    hal.merkle_open(mat, nodes, rows, cols, idx, out)
    got = out.to_vec().reshape(idx.size, wpq)
    for q, i in enumerate(idx):
        cur = np.zeros(8, np.uint32)
        oracle.zko_hash_elem_slice(np.ascontiguousarray(got[q, :cols]), cols, 1, cur)
        j = int(i) + rows
        for lvl in range(17):
            sib = np.ascontiguousarray(got[q, cols + 8 * lvl: cols + 8 * lvl + 8])
            nxt = np.zeros(8, np.uint32)
            if j & 1:
                oracle.zko_hash_pair(sib, cur, nxt)
            else:
                oracle.zko_hash_pair(cur, sib, nxt)
            cur, j = nxt, j // 2
        assert np.array_equal(cur, top[j - 32])
    # the top layer folds to the root
    layer = [top[k] for k in range(32)]
    while len(layer) > 1:
        nxt = []
        for k in range(0, len(layer), 2):
            o = np.zeros(8, np.uint32)
            oracle.zko_hash_pair(np.ascontiguousarray(layer[k]), np.ascontiguousarray(layer[k + 1]), o)
            nxt.append(o)
        layer = nxt
    assert np.array_equal(layer[0], root)


def test_cpp_host_driver_seals_and_verifies(tmp_path):
    """examples/seal_segments: a compiled host (no Python, no torch, no HIP headers) drives the library through the C ABI:
    4 segments over 2 lanes with a shared work index, every seal accepted by the host verifier."""
    import json
    import os
    import subprocess
    from zeth_amd import build
    from zeth_amd.circuits import syn_air
    desc = tmp_path / "syn_small.desc"
    np.asarray(syn_air.syn_small(), dtype="<u4").tofile(desc)
    exe = build.build_examples()
    r = subprocess.run([exe, "--desc", str(desc), "--po2", "12", "--segments", "4", "--inflight", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] == 4 and out["segments"] == 4 and out["seal_words_total"] > 4 * 1000
    # the same session with the code group committed once per worker and kept resident: the receipts are the same bytes
    # (fixed noise seed), and the host verifier accepts every one against the control root
    da, db = tmp_path / "a", tmp_path / "b"
    for d, extra in ((da, []), (db, ["--resident-code-group"])):
        os.makedirs(d)
        r = subprocess.run([exe, "--desc", str(desc), "--po2", "11", "--segments", "5", "--inflight", "2", "--noise-seed", "99",
                            "--receipts-dir", str(d), *extra], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert json.loads(r.stdout.strip().splitlines()[-1])["verified"] == 5
    names = sorted(os.listdir(da))
    assert len(names) == 5 and names == sorted(os.listdir(db))
    # BASELINE config 5 in the compiled host: the session's receipts folded through the P2-JOIN tree to one root receipt,
    # then the compact receipt verified (root seal + the claim tree recomputed with zkh_poseidon2_mix_host)
    from zeth_amd.circuits import p2_join
    jdesc = tmp_path / "p2_join.desc"
    np.asarray(p2_join.p2_join_circuit(), dtype="<u4").tofile(jdesc)
    r = subprocess.run([exe, "--desc", str(desc), "--po2", "12", "--segments", "7", "--inflight", "2", "--join-desc", str(jdesc), "--join-po2", "13"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] == 7 and out["joins"] == 6 and out["compact_receipt_verified"] is True and out["root_receipt_words"] > 1000
    # the same session as ONE library call (zkh_session_create / _prove / _verify): examples/prove_session
    exe2 = os.path.join(os.path.dirname(exe), "prove_session")
    r = subprocess.run([exe2, "--desc", str(desc), "--join-desc", str(jdesc), "--po2", "13", "--tail-po2", "12", "--segments", "6", "--inflight", "2",
                        "--join-po2", "13"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified"] is True and out["segments"] == 6 and out["joins"] == 5 and out["tail_po2"] == 12 and out["lanes"] == 2
    for nm in names:
        assert open(da / nm, "rb").read() == open(db / nm, "rb").read(), nm


def test_pool_accounting_and_trim(oracle):
    """The free list recycles blocks by size: repeated seals of one shape do not grow memory, trim() returns the cache to
    the driver, and the next seal is still byte-identical."""
    from zeth_amd.hal import HipHal
    from zeth_amd.circuits import syn_air
    from zeth_amd.prover import Segment, SegmentProver
    h = HipHal(0)
    pr = SegmentProver(h, syn_air.syn_small())
    seg = Segment(index=0, po2=13, seed=11)
    code, data, out = pr.witgen(seg)
    first = pr.seal(seg, code, data, out).seal
    m1 = h.memory()
    for _ in range(5):
        assert np.array_equal(pr.seal(seg, code, data, out).seal, first)
    m2 = h.memory()
    assert m2["live"] == m1["live"] and m2["cached"] == m1["cached"] and m2["peak"] == m1["peak"]
    assert m1["cached"] > 0 and m1["peak"] >= m1["live"] > 0
    h.trim()
    m3 = h.memory()
    assert m3["cached"] == 0 and m3["live"] == m1["live"]
    assert np.array_equal(pr.seal(seg, code, data, out).seal, first)
    del pr, code, data
    h.close()
