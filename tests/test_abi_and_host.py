"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/zkhal.h declares (no
compute calls without a GPU), the circuit builder / code generator are consistent, and the host-side segment
partitioning logic (the mirror of BlockProcessor::prove, /root/reference/crates/host/src/lib.rs:123-143)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from zeth_amd import hal as zhal
from zeth_amd.circuits import codegen, syn_air
from zeth_amd.circuits.desc import OP_AND_COND, OP_AND_EQZ, OP_GET, Circuit
from zeth_amd.host import BlockProcessor, CompositeReceipt, partition_round_robin, session_segments
from zeth_amd.prover import Segment, SegmentReceipt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "zkhal.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = zhal.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/zkhal.h but not exported"
    # the python binding table covers the whole header too
    assert set(declared) == set(zhal.ABI), set(declared) ^ set(zhal.ABI)
    assert b"gfx950" in lib.zkh_version()


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a HIP device context creation must fail loudly (never silently fall back)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zhal.HalError, match="no CPU fallback|no HIP device|failed"):
        zhal.HipHal(0)


def test_product_never_imports_the_oracle():
    """The oracle is the checker only: nothing under zeth_amd/ may import, link, include or call it."""
    bad = re.compile(r"import\s+zko|libzkoracle|zkoracle\.h|#include\s+\"[^\"]*oracle|\bzko_[a-z]")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zeth_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(txt), f"{f} references the oracle"
    # bench.py and its parts: ONLY the cpu_baseline leg may touch it (as the thing measured beside the GPU, never on the GPU path)
    for f in ["bench.py"] + [os.path.join("benchlib", g) for g in sorted(os.listdir(os.path.join(ROOT, "benchlib"))) if g.endswith(".py")]:
        if f.endswith("cpu_baseline.py"):
            continue
        code = "\n".join(l for l in open(os.path.join(ROOT, f), errors="replace").read().split("\n") if not l.lstrip().startswith("#"))
        assert not re.search(r"import\s+zko|libzkoracle|\bzko\.", code), f"{f} references the oracle"


def test_syn_air_desc_roundtrip_and_shapes():
    for wc, wd, wa in [(6, 11, 4), (8, 20, 8), (16, 208, 32)]:
        desc = syn_air.build_syn_air(wc, wd, wa)
        c = Circuit.parse(desc)
        assert c.group_sizes == (wa, wc, wd) and c.global_sizes == (4, wa)
        # every column tapped at back 0; running sum and accum columns also at back 1
        assert len(c.taps) == wa + wc + wd + 1 + wa
        assert c.combos == [(0,), (0, 1)]
        assert c.taps == sorted(c.taps)
        regs = c.regs
        assert len(regs) == wa + wc + wd
        assert sum(1 for r in regs if r[3] == 1) == wa + 1
        gets = [s for s in c.steps if s[0] == OP_GET]
        assert all(s[1] < len(c.taps) for s in gets)
    assert codegen.desc_hash64(syn_air.syn_a()) != codegen.desc_hash64(syn_air.syn_small())


def test_codegen_static_mix_exponents():
    c = Circuit.parse(syn_air.syn_tiny())
    kinds, mix_exp, used_f, used_m, n_pows = codegen.analyse(c)
    n_eqz = sum(1 for s in c.steps if s[0] == OP_AND_EQZ)
    # the result's `mul` would be poly_mix^(number of AndEqz): every constraint gets its own power
    assert mix_exp[c.ret] == n_eqz
    assert n_pows <= n_eqz
    src, h, n2 = codegen.emit_kernel("t", syn_air.syn_tiny())
    assert "k_eval_check_t" in src and n2 == n_pows and h == codegen.desc_hash64(syn_air.syn_tiny())
    assert sum(1 for s in c.steps if s[0] == OP_AND_COND) == 4


def test_round_robin_partition():
    for n, g in [(0, 1), (1, 1), (7, 2), (256, 8), (5, 8)]:
        parts = [partition_round_robin(n, g, r) for r in range(g)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        for r, p in enumerate(parts):
            assert all(i % g == r for i in p)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        partition_round_robin(4, 2, 2)


def test_session_segments_tail():
    segs = session_segments(3 * (1 << 20) + 5000, 20)
    assert [s.po2 for s in segs] == [20, 20, 20, 13]
    assert [s.index for s in segs] == [0, 1, 2, 3]
    assert len({s.seed for s in segs}) == 4
    assert [s.po2 for s in session_segments(1 << 20, 20)] == [20]
    assert [s.po2 for s in session_segments((1 << 20) + (1 << 19) + 1, 20)] == [20, 20]


def test_block_processor_assembles_in_index_order():
    def fake(seg: Segment) -> SegmentReceipt:
        return SegmentReceipt(seal=np.array([seg.index, seg.po2], dtype=np.uint32), index=seg.index, po2=seg.po2)

    segs = session_segments(5 << 20, 20)
    rec = BlockProcessor(fake).prove(segs)
    assert [r.index for r in rec.segments] == [0, 1, 2, 3, 4]
    # two logical ranks, gathered by hand
    parts = [BlockProcessor(fake, rank=r, world_size=2).prove_local(segs) for r in range(2)]
    assert [r.index for r in parts[0]] == [0, 2, 4] and [r.index for r in parts[1]] == [1, 3]
    bp = BlockProcessor(fake, rank=0, world_size=2, gather=lambda local: parts)
    assert [r.index for r in bp.prove(segs).segments] == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        CompositeReceipt([parts[0][0], parts[0][1]]).verify_integrity()


def test_dev_mode_plumbing(monkeypatch):
    """BASELINE config 1: RISC0_DEV_MODE=1 'prove' = segment scheduling + fake receipts, no GPU."""
    from zeth_amd.hal import HalError
    from zeth_amd.host import DevModeProver, dev_mode_enabled
    monkeypatch.setenv("RISC0_DEV_MODE", "1")
    assert dev_mode_enabled()
    segs = session_segments(7 * (1 << 20) + 1, 20)
    rec = BlockProcessor(DevModeProver().prove_segment).prove(segs)
    assert [r.index for r in rec.segments] == list(range(8)) and rec.segments[-1].po2 == 13
    assert all(r.hashfn == "fake" and r.seal.size == 0 for r in rec.segments)
    with pytest.raises((HalError, IndexError)):
        rec.verify(syn_air.syn_tiny())                  # fake receipts never verify
    monkeypatch.delenv("RISC0_DEV_MODE")
    assert not dev_mode_enabled()


def test_eval_check_jit_cross_compiles_and_caches(tmp_path, monkeypatch):
    """circuits/jit.py: desc -> standalone HIP source -> gfx950 code object (hipcc --genco cross-compiles without a GPU)."""
    from zeth_amd.circuits import jit, syn_air
    if jit.hipcc_path() is None:
        pytest.skip("hipcc not installed")
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    desc = syn_air.build_syn_air(6, 14, 4)
    src, name = jit.eval_check_source(desc)
    assert f'extern "C" __global__ __launch_bounds__(256) void {name}(EvalCheckArgs a)' in src
    assert "static void launch_" not in src
    image, name2 = jit.compile_code_object(desc)
    assert name2 == name and len(image) > 1000
    assert image[:4] == b"\x7fELF" or image.startswith(b"__CLANG_OFFLOAD_BUNDLE__")
    cached = [f for f in tmp_path.iterdir() if f.name.endswith(".hsaco")]
    assert len(cached) == 1
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")        # second call must come from the cache, not the compiler
    image2, _ = jit.compile_code_object(desc)
    assert image2 == image
    with pytest.raises(jit.JitError):
        jit.compile_code_object(syn_air.build_syn_air(6, 15, 4))


def test_header_is_plain_c_and_example_driver_builds(tmp_path):
    """include/zkhal.h must be consumable from C (the boundary is a C ABI), and examples/seal_segments — a g++-only host
    driver over that header — must build and link against the library."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "zkhal.h")
    c_file = tmp_path / "t.c"
    c_file.write_text('#include "zkhal.h"\nint main(void) { return zkh_version() == 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr), str(c_file)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from zeth_amd import build
    exe = build.build_examples()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


def test_join_schedule_shape_and_placement():
    """ceil(log2 S) dependent levels, S/2^l-way parallel, a join runs where its left child was produced."""
    from zeth_amd.host import join_schedule
    assert join_schedule(1, 4) == []
    lv = join_schedule(1024, 8)
    assert [len(x) for x in lv] == [512, 256, 128, 64, 32, 16, 8, 4, 2, 1]
    assert sorted({t.device for t in lv[0]}) == [0, 2, 4, 6]          # leaf 2k lives on rank 2k mod 8
    assert [sorted({t.device for t in x}) for x in lv[-3:]] == [[0], [0], [0]]
    lv = join_schedule(5, 2)                                           # odd node carried up
    assert [len(x) for x in lv] == [2, 1, 1]
    assert [(t.left, t.right) for t in lv[0]] == [(0, 1), (2, 3)]
    assert (lv[2][0].left, lv[2][0].right) == (0, 1)
    for S in range(1, 40):
        for G in (1, 3, 8):
            lv = join_schedule(S, G)
            assert sum(len(x) for x in lv) == S - 1                     # a binary tree over S leaves has S-1 joins
            assert len(lv) == (S - 1).bit_length()
    with pytest.raises(ValueError):
        join_schedule(0, 1)


def test_codegen_value_numbering_windows_and_splitting(monkeypatch):
    """circuits/codegen.py on a constraint system of realistic shape: structurally identical sub-expressions collapse,
    the leaves are cut into parts that cover every constraint exactly once, every part is a self-contained kernel."""
    from zeth_amd.circuits import syn_heavy
    monkeypatch.setenv("ZKH_CODEGEN_PART", "1600")           # the small circuit in several parts (the shipped weight keeps it whole)
    desc = syn_heavy.syn_heavy_small()
    c = Circuit.parse(desc)
    plan = codegen.Plan.build(c)
    arith = sum(1 for s in c.steps if s[0] in (4, 5, 6))
    assert plan.n_unique < 0.85 * arith                      # Z_j and the shared terms are re-emitted by the circuit
    assert any(plan.ext)                                     # ConstExt operands make some values Fp4-typed
    weights = plan.leaf_weights()
    assert len(weights) == plan.n_leaves[c.ret] == sum(1 for s in c.steps if s[0] == OP_AND_EQZ)
    cuts = codegen.split_points(weights)
    assert len(cuts) >= 2 and cuts[0][0] == 0 and cuts[-1][1] == len(weights)
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    parts, h, n_pows = codegen.emit_parts("t", desc, standalone=True)
    assert [k for k, _ in parts] == codegen.part_kernel_names("t", len(cuts))
    assert n_pows == plan.n_pows and h == codegen.desc_hash64(desc)
    for k, src in parts:
        assert f'extern "C" __global__ __launch_bounds__(256) void {k}(EvalCheckArgs a)' in src
        assert "a.accumulate" in src and "tap_load(" in src and "volatile" not in src
        # the part's own power table: slots 0, 1, 2, ... in the order the code reads them, exponents exported beside the kernel
        import re
        slots = [int(x) for x in re.findall(r"pwp\[(\d+)\]", src)]
        assert sorted(set(slots)) == list(range(len(set(slots))))
        exps = [int(x) for x in re.search(rf"{k}_exps\[\] = \{{([^}}]*)\}}", src).group(1).split(",")]
        assert exps[0] == len(exps) - 1 == len(set(slots)) and all((e & 0x7FFFFFFF) < n_pows for e in exps[1:])      # (bit 31: the slot is read centred)
    # a small circuit stays one kernel with the historical name
    one, _, _ = codegen.emit_parts("syn_tiny", syn_air.syn_tiny())
    assert [k for k, _ in one] == ["k_eval_check_syn_tiny"]
    # the full SYN-HEAVY: ~54 k steps, ~1.1 k taps, 7 tap combos, > 8 kernels
    big = Circuit.parse(syn_heavy.syn_heavy())
    assert len(big.steps) > 50000 and len(big.taps) > 1000 and len(big.combos) == 7


def test_multi_part_jit_cross_compiles_with_verified_cache(tmp_path, monkeypatch):
    from zeth_amd.circuits import jit, syn_heavy
    if jit.hipcc_path() is None:
        pytest.skip("hipcc not installed")
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("ZKH_CODEGEN_PART", "1600")
    desc = syn_heavy.syn_heavy_small()
    objs = jit.compile_code_objects(desc)
    assert len(objs) >= 2 and all(img[:4] == b"\x7fELF" or img.startswith(b"__CLANG_OFFLOAD_BUNDLE__") for img, _ in objs)
    files = sorted(f.name for f in tmp_path.iterdir())
    assert sum(f.endswith(".hsaco") for f in files) == len(objs) == sum(f.endswith(".sha256") for f in files)
    assert (os.stat(tmp_path).st_mode & 0o022) == 0
    # a repeated load is served by the circuit-level manifest: the generator does not even run (round 6: SYN-HUGE's second load 3.0 -> 0.25 s)
    assert sum(f.endswith(".manifest.json") for f in files) == 1
    with monkeypatch.context() as m:
        m.setattr(jit, "eval_check_sources", lambda d: (_ for _ in ()).throw(AssertionError("the generator ran on a cache hit")))
        again = jit.compile_code_objects(desc)
        assert [n for _, n in again] == [n for _, n in objs] and all(a == b for (a, _), (b, _) in zip(again, objs))
        with pytest.raises(AssertionError, match="generator ran"):          # another knob = another set of kernels = another manifest
            m.setenv("ZKH_CODEGEN_PART", "1500")
            jit.compile_code_objects(desc)
    # a tampered cache entry is not used: it is recompiled (and with no compiler available, that fails loudly)
    victim = next(f for f in tmp_path.iterdir() if f.name.endswith(".hsaco"))
    victim.write_bytes(victim.read_bytes()[:-8] + b"tampered")
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    with pytest.raises(jit.JitError):
        jit.compile_code_objects(desc)
    # `python -m zeth_amd.circuits.jit` writes code objects + a manifest for hosts that attach them through the C ABI
    monkeypatch.delenv("HIPCC")
    out = tmp_path / "objs"
    assert jit.main(["jit", "syn_tiny", str(out)]) == 0
    import json
    man = json.load(open(out / "manifest.json"))
    assert man["arch"] == "gfx950" and len(man["parts"]) == 1 and (out / man["parts"][0]["file"]).exists()


def test_cached_input_reader_feeds_the_segment_list(tmp_path):
    """cli.rs:126-131 keeps `cache/input_<hash>.json`; the reader validates the shape and turns the block into a segment
    list: from a measured cycle count when a dev-mode run left one, from the declared gas heuristic otherwise."""
    import json
    from zeth_amd.host import CYCLES_PER_GAS_ESTIMATE, list_cached_inputs, read_cached_input
    h1, h2 = "0x" + "ab" * 32, "0x" + "cd" * 32
    (tmp_path / f"input_{h1}.json").write_text(json.dumps({"block": {"header": {"gasUsed": hex(29_500_000)}, "body": {}}, "witness": {"state": []}}))
    (tmp_path / f"input_{h2}.json").write_text(json.dumps({"block": {"header": {"gasUsed": "0x1c9c380"}}, "witness": {}}))
    (tmp_path / f"input_{h2}.cycles.json").write_text(json.dumps({"total_cycles": 5 * (1 << 20) + 70000, "user_cycles": 5_000_000,
                                                                 "paging_cycles": 300_000, "keccak_calls": 1234}))
    (tmp_path / "input_0xbad.json").write_text(json.dumps([1, 2, 3]))
    assert list_cached_inputs(str(tmp_path)) == sorted([h1, h2, "0xbad"])
    with pytest.raises(ValueError, match="header is incomplete"):                     # cli.rs:141 re-derives the hash: a stub header cannot pass
        read_cached_input(str(tmp_path), h1)
    a = read_cached_input(str(tmp_path), h1, check_hash=False)
    assert not a.hash_checked
    assert a.cycles_source == "gas-estimate" and a.total_cycles == int(29_500_000 * CYCLES_PER_GAS_ESTIMATE) and a.gas_used == 29_500_000
    segs = a.segments(20)
    assert len(segs) == -(-a.total_cycles // (1 << 20)) and all(s.po2 == 20 for s in segs[:-1])
    b = read_cached_input(str(tmp_path), h2, check_hash=False)
    assert b.cycles_source == "sidecar" and b.keccak_calls == 1234
    assert [s.po2 for s in b.segments(20)] == [20] * 5 + [17]
    with pytest.raises(ValueError, match="StatelessInput"):
        read_cached_input(str(tmp_path), "0xbad")


@pytest.mark.parametrize("world,n_leaves", [(2, 5), (3, 7), (4, 16), (8, 21), (8, 64), (5, 1), (8, 3)])
def test_join_executor_any_world_size_matches_single_rank(world, n_leaves):
    """The distributed join schedule for G ranks (threads + queues standing in for the gloo point-to-point transport, a
    deterministic fake 'prover'): every join runs exactly once, on the rank that holds its left child; only right children
    travel; the root lands on rank 0 and equals the single-rank tree — for world sizes up to the 8 GPUs of the north star."""
    import hashlib
    import queue
    import threading
    from zeth_amd.host import JoinExecutor, join_schedule

    def fake_receipt(index, po2, payload: bytes):
        words = np.frombuffer(hashlib.sha256(payload).digest(), dtype=np.uint32).copy()
        return SegmentReceipt(seal=words, index=index, po2=po2)

    def claim_of(rec, is_leaf):
        return np.frombuffer(hashlib.sha256(rec.seal.tobytes() + bytes([is_leaf])).digest(), dtype=np.uint32)[:8].copy() % 2013265921

    def prove_join(seg):
        return fake_receipt(seg.index, seg.po2, np.asarray(seg.pub, np.uint32).tobytes() + seg.seed.to_bytes(8, "little"))

    leaves = [fake_receipt(i, 20, b"leaf%d" % i) for i in range(n_leaves)]
    single_done, single_root = JoinExecutor(prove_join, claim_of, 0, 1, noise_seed=1).run(n_leaves, dict(enumerate(leaves)))
    chan = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
    sent = []
    results = [None] * world

    def rank_main(r):
        def send(obj, dst):
            sent.append((r, dst))
            chan[(r, dst)].put(obj)

        def recv(src):
            return chan[(src, r)].get(timeout=30)
        ex = JoinExecutor(prove_join, claim_of, r, world, noise_seed=1, send=send, recv=recv)
        results[r] = ex.run(n_leaves, {i: leaves[i] for i in range(r, n_leaves, world)})

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
        assert not t.is_alive()
    sched = join_schedule(n_leaves, world)
    merged = {}
    for r, (done, root) in enumerate(results):
        assert set(done) == {(t.level, t.index) for lvl in sched for t in lvl if t.device == r}
        merged.update(done)
        assert (root is not None) == (r == 0)               # the root always lands where leaf 0 was produced
    assert len(merged) == n_leaves - 1 == len(single_done)
    for k, rec in single_done.items():
        assert np.array_equal(merged[k].seal, rec.seal)
    if n_leaves > 1:
        assert np.array_equal(results[0][1].seal, single_root.seal)
    # only right children whose owner differs from the join's rank were sent
    assert len(sent) == sum(1 for lvl in sched for t in lvl if t.right_owner != t.device)


def test_poseidon2_fast_form_equals_literal_permutation_at_the_bounds(tmp_path):
    """tests/cpp/poseidon2_bounds.cpp: the scaled / lazily reduced permutation the kernels run (same header, host build)
    against a literal 29-round permutation, on edge-valued and random states, round constants and diagonals."""
    import subprocess
    exe = tmp_path / "poseidon2_bounds"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "zeth_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "poseidon2_bounds.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "150000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout


def test_poseidon2_tables_are_the_output_of_the_published_procedure():
    """tools/gen_poseidon2_consts.py restates the public parameter generation (Grain LFSR; internal-matrix candidates drawn
    until every power up to 2t has an irreducible characteristic polynomial).  The shipped header must be exactly its
    output, the run must reproduce the published instance's values on record (asserted inside the tool), and the accepted
    internal matrix must be the fifth candidate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_p2", os.path.join(ROOT, "tools", "gen_poseidon2_consts.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rc, diag, attempt = gen.generate(verbose=False)
    assert attempt == 5 and diag == gen.ANCHOR_DIAG and rc[:8] == gen.ANCHOR_EXTERNAL
    assert [rc[(4 + r) * 24] for r in range(4)] == gen.ANCHOR_INTERNAL
    txt = open(os.path.join(ROOT, "include", "zkh_poseidon2_consts.h")).read()
    nums = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)]
    assert nums[:24] == diag and nums[24:] == rc
    from zeth_amd.circuits import poseidon2_consts as pc          # the package's own copy (no dependency on include/ at import time)
    assert pc.M_INT_DIAG == diag and pc.ROUND_CONSTANTS == rc
    assert "#define ZKH_P2_CONSTS_ARE_PLACEHOLDER 0" in txt and "#define ZKH_P2_CONSTS_ARE_DERIVED 1" in txt
    # the routines the acceptance test stands on
    assert gen.irreducible([11, 0, 0, 0, 1])                      # x^4 + 11: the prover's extension field
    assert not gen.irreducible([gen.P - 1, 0, 1])                 # x^2 - 1
    m = [[(3 * i + 7 * j + i * j) % gen.P for j in range(5)] for i in range(5)]
    cp = gen.charpoly(m)
    assert cp[-1] == 1 and (-cp[-2]) % gen.P == sum(m[i][i] for i in range(5)) % gen.P
