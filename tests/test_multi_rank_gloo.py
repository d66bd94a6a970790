"""N > 1 path on CPU: world_size-2 gloo processes partition a segment list round-robin, each "seals" its share with
the CPU oracle standing in for the GPU (test-only), receipts are gathered on rank 0 with no data-path collective,
and the composite equals the single-rank result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zko
    from zeth_amd.circuits import syn_air
    from zeth_amd.host import BlockProcessor, torch_gather
    from zeth_amd.prover import Segment, SegmentReceipt
    oc = zko.OracleCircuit(zko.load(), syn_air.syn_tiny())

    def prove(seg):
        return SegmentReceipt(seal=oc.prove(seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed), index=seg.index, po2=seg.po2)

    segs = [Segment(index=i, po2=9, seed=100 + i, zk_cycles=100) for i in range(5)]
    bp = BlockProcessor(prove, rank=rank, world_size=world, gather=torch_gather(rank, world))
    # timing protocol of bench.py: barrier, work, barrier, MAX over ranks
    dist.barrier()
    rec = bp.prove(segs)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put(([r.index for r in rec.segments], [r.seal.tobytes() for r in rec.segments], float(t.item())))
    else:
        assert rec is None
    dist.destroy_process_group()


def test_world_size_2_round_robin_matches_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zko
    from zeth_amd.circuits import syn_air
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    idx, seals, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert idx == [0, 1, 2, 3, 4] and tmax == 2.0
    oc = zko.OracleCircuit(zko.load(), syn_air.syn_tiny())
    for i, s in enumerate(seals):
        want = oc.prove(9, 100, 100 + i, 0x2E80)
        assert s == want.tobytes()
