"""--config dev (BASELINE config 1: `RISC0_DEV_MODE=1` prove — executor + mock receipt, plumbing, no GPU;
/root/reference/README.md:104-109, CI at /root/reference/.github/workflows/main.yml:51-54).

No proving: `DevModeProver` hands out a fake receipt per segment (it only carries the claim metadata and never verifies).  What
this config exercises is everything AROUND the seals — the round-robin partition, the control plane (barriers, MAX over ranks,
gathers), result assembly in index order, a failed rank — on any host, which is how the N > 1 failure paths are tested on CPU
(tests/test_bench_faults.py).  The line says `"data": "dev-mode"`: it is never a measurement of the hot path."""
from __future__ import annotations

import os
import time

from .common import BASE_SEED, BENCH_NOISE, Run, block_segments, maybe_fault
from .control import RankFailed
from .segment import consensus_failed, secondary


def run_dev(run: Run):
    from zeth_amd.host import CompositeReceipt, DevModeProver, partition_round_robin
    from zeth_amd.prover import Segment
    args, ctl, rank, world = run.args, run.ctl, run.rank, run.world
    run.failed_ranks = {}
    fake_ms = float(os.environ.get("ZKH_DEV_SEAL_MS", "1.0"))
    prover = DevModeProver()

    def prove(seg):
        time.sleep(fake_ms * 1e-3)
        return prover.prove_segment(seg)

    # ---- headline: K fake seals per rank between two barriers, MAX over ranks ----
    try:
        mine = partition_round_robin((args.warmup + args.steps) * world, world, rank)
        for i in mine[:args.warmup]:
            prove(Segment(index=i, po2=args.po2, seed=BASE_SEED + i, noise_seed=BENCH_NOISE))
        maybe_fault(run, "headline")
        ctl.barrier()
        t0 = time.perf_counter()
        recs = [prove(Segment(index=i, po2=args.po2, seed=BASE_SEED + i, noise_seed=BENCH_NOISE)) for i in mine[args.warmup:]]
        ctl.barrier()
        dt = time.perf_counter() - t0
    except RankFailed:
        raise
    except Exception as e:
        if run.distributed:
            raise RankFailed(ctl.fail("the headline leg", e))
        raise
    dt = ctl.max(dt)
    steps_done = int(ctl.sum([float(len(recs))])[0])
    run.failed_ranks = consensus_failed(run)
    if run.failed_ranks:
        run.failed_in = "headline"
    headline_ranks = world - len(run.failed_ranks)         # the ranks whose steps are in `value`

    # ---- a block: S fake segments round-robin, gathered on rank 0 in index order ----
    S = args.block_segments if args.block_segments is not None else 64

    def block_leg():
        segs = block_segments(run, S)
        bmine = partition_round_robin(S, world, rank)
        maybe_fault(run, "block")
        ctl.barrier()
        tb = time.perf_counter()
        local = [prove(segs[i]) for i in bmine]
        ctl.barrier()
        dtb = ctl.max(time.perf_counter() - tb)
        parts = ctl.gather(local, dst=0)
        if rank == 0:
            comp = CompositeReceipt(sorted((r for p in parts.values() for r in p), key=lambda r: r.index))
            comp.verify_integrity()                       # every index once, in order
        return {"segments": S, "wall_clock_s": dtb, "segments_per_s": S / dtb, "assembled_in_index_order": True}
    block = secondary(run, "block", block_leg) if not args.no_block and S > 0 else None
    if rank != 0:
        return None, []
    line = {
        "metric": "segments/sec", "value": steps_done / dt, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
        "data": "dev-mode",
        "config": {"workload": f"RISC0_DEV_MODE plumbing: fake receipts ({fake_ms} ms of sleep per segment), no GPU work, no proving — "
                               f"BASELINE config 1; exercises the partition, the control plane and result assembly only",
                   "po2": args.po2, "circuit": "none", "parallelism": f"segments round-robin over {world} rank(s), no collectives",
                   "launcher": "ranks", "devices": run.devices, "devices_distinct": run.devices_distinct, "rccl_probe": run.rccl},
    }
    if run.failed_ranks:
        line["failed_ranks"] = sorted(run.failed_ranks)
        line["failed_ranks_detail"] = [run.failed_ranks[r] for r in sorted(run.failed_ranks)]
        line["ranks_reporting"] = headline_ranks
    if block is not None:
        line["block"] = block
    return line, []
