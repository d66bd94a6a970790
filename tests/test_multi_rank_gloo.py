"""N > 1 path on CPU: world_size-2 and world_size-4 gloo processes partition a segment list round-robin, each "seals" its share with
the CPU oracle standing in for the GPU (test-only), receipts are gathered on rank 0 with no data-path collective,
and the composite equals the single-rank result bit for bit.  The same two ranks then run the distributed join
executor (BASELINE config 5): right children cross the control plane, the root lands on rank 0, and the succinct
receipt verifies — including every join's commitment to the claims of its children."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF_PO2, JOIN_PO2, ZK, N_LEAVES = 9, 9, 100, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _descs():
    from zeth_amd.circuits import p2_join, syn_air
    return syn_air.syn_tiny(), p2_join.p2_join_circuit()        # joins hash their children's claims in-circuit


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zko
    from zeth_amd.host import BlockProcessor, JoinExecutor, node_claim, torch_gather
    from zeth_amd.prover import Segment, SegmentReceipt
    leaf_desc, join_desc = _descs()
    lib = zko.load()
    oc, ocj = zko.OracleCircuit(lib, leaf_desc), zko.OracleCircuit(lib, join_desc)

    def prove(seg):
        return SegmentReceipt(seal=oc.prove(seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed), index=seg.index, po2=seg.po2)

    segs = [Segment(index=i, po2=LEAF_PO2, seed=100 + i, noise_seed=0x2E80, zk_cycles=ZK) for i in range(N_LEAVES)]
    bp = BlockProcessor(prove, rank=rank, world_size=world, gather=torch_gather(rank, world))
    # timing protocol of bench.py: barrier, work, barrier, MAX over ranks
    dist.barrier()
    local = bp.prove_local(segs)
    rec = bp.prove(segs)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)

    # ---- distributed join tree over the same leaves ----
    leaf_root, join_root = oc.control_root(LEAF_PO2, ZK), ocj.control_root(JOIN_PO2, ZK)

    def claim_of(r, is_leaf):
        return node_claim(r, leaf_desc if is_leaf else join_desc, leaf_root if is_leaf else join_root, is_leaf)

    def prove_join(seg):
        seal = ocj.prove(seg.po2, ZK, seg.seed, 0x2E81, pub=np.array(seg.pub, np.uint32))
        return SegmentReceipt(seal=seal, index=seg.index, po2=seg.po2)

    ex = JoinExecutor(prove_join, claim_of, rank, world, join_po2=JOIN_PO2)
    done, root = ex.run(N_LEAVES, {r.index: r for r in local})
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({k: v.seal.tobytes() for k, v in done.items()}, gathered, dst=0)
    if rank == 0:
        q.put(([r.index for r in rec.segments], [r.seal.tobytes() for r in rec.segments], float(t.item()),
               gathered, root.seal.tobytes() if root is not None else None))
    else:
        assert rec is None and root is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_round_robin_and_distributed_joins_over_gloo(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zko
    from zeth_amd.hal import HalError
    from zeth_amd.host import SuccinctReceipt, fold_claims, join_schedule, node_claim, prove_succinct, receipt_claim
    from zeth_amd.prover import SegmentReceipt
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    idx, seals, tmax, gathered, root_bytes = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert idx == list(range(N_LEAVES)) and tmax == float(world)
    leaf_desc, join_desc = _descs()
    lib = zko.load()
    oc, ocj = zko.OracleCircuit(lib, leaf_desc), zko.OracleCircuit(lib, join_desc)
    leaves = []
    for i, s in enumerate(seals):
        want = oc.prove(LEAF_PO2, ZK, 100 + i, 0x2E80)
        assert s == want.tobytes()
        leaves.append(SegmentReceipt(seal=want, index=i, po2=LEAF_PO2))
    # which rank ran which join: the one holding the left child; together the two ranks ran every join exactly once
    sched = join_schedule(N_LEAVES, world)
    assert any(t.right_owner != t.device for lvl in sched for t in lvl)          # some right child really crosses ranks
    for r, part in enumerate(gathered):
        assert set(part) == {(t.level, t.index) for lvl in sched for t in lvl if t.device == r}
    assert sum(len(p) for p in gathered) == N_LEAVES - 1
    # the distributed tree equals the single-rank tree bit for bit, and verifies
    leaf_root, join_root = oc.control_root(LEAF_PO2, ZK), ocj.control_root(JOIN_PO2, ZK)

    def claim_of(r, is_leaf):
        return node_claim(r, leaf_desc if is_leaf else join_desc, leaf_root if is_leaf else join_root, is_leaf)

    def prove_join(seg):
        return SegmentReceipt(seal=ocj.prove(seg.po2, ZK, seg.seed, 0x2E81, pub=np.array(seg.pub, np.uint32)), index=seg.index, po2=seg.po2)

    single = prove_succinct(leaves, prove_join, claim_of, join_po2=JOIN_PO2)
    merged = {k: v for part in gathered for k, v in part.items()}
    for lvl, tasks in zip(single.joins, join_schedule(N_LEAVES, 1)):
        for j, t in zip(lvl, tasks):
            assert merged[(t.level, t.index)] == j.seal.tobytes()
    assert root_bytes == single.root.seal.tobytes()
    single.verify(leaf_desc, join_desc, leaf_root, join_root)
    # P2-JOIN joins constrain parent = hash_pair(left, right): the verifier needs ONLY the root receipt and the leaves —
    # the joins below the root are dropped, the claim tree is recomputed on the host and compared with the root's output
    compact = single.compact()
    assert compact.joins == []
    compact.verify(leaf_desc, join_desc, leaf_root, join_root)
    claims = [receipt_claim(r, leaf_desc, leaf_root) for r in leaves]
    assert np.array_equal(single.root.seal[:8], fold_claims(claims))
    import copy
    copy.deepcopy(compact).verify(leaf_desc, join_desc, leaf_root, join_root)      # content, not object identity
    # a root that is not a join proof / a root over OTHER leaves / a leaf swapped for another valid leaf: all rejected
    with pytest.raises((ValueError, HalError)):
        SuccinctReceipt(root=leaves[0], joins=[], leaves=leaves).verify(leaf_desc, join_desc, leaf_root, join_root)
    fewer = prove_succinct(leaves[:4], prove_join, claim_of, join_po2=JOIN_PO2)
    with pytest.raises(ValueError, match="claim tree"):
        SuccinctReceipt(root=fewer.root, joins=[], leaves=leaves).verify(leaf_desc, join_desc, leaf_root, join_root)
    swapped = [leaves[1], leaves[0]] + leaves[2:]
    swapped = [SegmentReceipt(seal=r.seal, index=i, po2=r.po2) for i, r in enumerate(swapped)]
    with pytest.raises(ValueError, match="claim tree"):
        SuccinctReceipt(root=single.root, joins=[], leaves=swapped).verify(leaf_desc, join_desc, leaf_root, join_root)
    # a one-leaf session: the root is that leaf, and an arbitrary root next to a valid leaf is rejected
    one = prove_succinct(leaves[:1], prove_join, claim_of, join_po2=JOIN_PO2)
    one.verify(leaf_desc, join_desc, leaf_root, join_root)
    with pytest.raises(ValueError, match="root is not the top"):
        SuccinctReceipt(root=leaves[1], joins=[], leaves=leaves[:1]).verify(leaf_desc, join_desc, leaf_root, join_root)


def _chain_worker(rank, world, port, q):
    """A CHAINED session over N ranks: the executor's pass runs once (rank 0) and fixes every segment's pre-state, the pre-states
    travel with the segment list over the control plane, segment i is proven on rank i mod N (independent provers), rank 0 gathers."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.host import BlockProcessor, chain_segments, torch_gather
    from zeth_amd.prover import Segment, SegmentReceipt
    desc = syn_air.syn_chain_small()
    oc = zko.OracleCircuit(zko.load(), desc)
    po2, zk = 11, 1994
    base = [Segment(index=i, po2=po2, seed=300 + i, noise_seed=0x62) for i in range(5)]
    box = [None]
    if rank == 0:            # the executor: one sequential pass that fixes pre / post of every segment
        box[0] = chain_segments(base, lambda s: int(oc.witgen(s.po2, zk, s.seed, s.noise_seed, pub=np.zeros(1, np.uint32))[2][0]), initial_state=3)
    dist.broadcast_object_list(box, src=0)
    segs = box[0]

    def prove(seg):
        seal = oc.prove(seg.po2, zk, seg.seed, seg.noise_seed, pub=np.asarray(seg.pub, dtype=np.uint32))
        return SegmentReceipt(seal=seal, index=seg.index, po2=seg.po2, output=seal[:5].copy())
    rec = BlockProcessor(prove, rank=rank, world_size=world, gather=torch_gather(rank, world)).prove(segs)
    if rank == 0:
        root = oc.control_root(po2, zk)
        rec.verify(desc, root, chained=True, initial_state=3)
        q.put([r.seal.tobytes() for r in rec.segments])
    dist.destroy_process_group()


def test_chained_session_across_two_ranks_is_continuous():
    """claim continuity does not serialise the provers: with the pre-states fixed by the executor's pass, two ranks prove their
    round-robin shares independently and the gathered composite passes the pre == prev.post check"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seals = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert len(seals) == 5 and len(set(seals)) == 5
    words = [np.frombuffer(s, dtype=np.uint32) for s in seals]
    assert all(int(words[i + 1][4]) == int(words[i][0]) for i in range(4))
