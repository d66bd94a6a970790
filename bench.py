#!/usr/bin/env python3
"""bench.py — segments/sec for sealing 2^20-cycle zkVM segments on N MI355X (BASELINE.json metric).

One "step" = one segment seal (commit code/data, accum, eval_check, DEEP, FRI, queries).
Configs (SURVEY.md §8d restatements of BASELINE.json's configs; one module each under benchlib/):

  --config segment  (default; BASELINE config 2)  every step seals one 2^po2-cycle segment whose witness is already
                    resident in HBM when the clock starts; N GPUs = N ranks each doing K steps ("scaling": "weak").
                    The same run then measures, as guarded secondary legs: the SYN-HEAVY constraint system, the resident
                    code group, a short block (S distinct segments, witgen in the clock, all verified), the host-preflight
                    witness pipeline and the block's fold to one receipt; then roofline{} (live HBM traffic and VALU issue
                    from rocprofv3 --pmc child runs) and cpu_baseline{}.
  --config block    (configs 3/4)  one block = S DISTINCT segments (seeds base+i, the last one a po2-18 tail), handed out
                    round-robin over the ranks and through a shared work index inside a rank; witness generation runs
                    inside the clock, every seal is verified on the host after the clock stops; value = S / wall ("strong").
  --config succinct (config 5)  S leaf segments sealed and folded to ONE root receipt (lift2 / join3 / join programs of the
                    RECURSION circuit: every node verifies its child seals in-circuit), one native call per rank.
  --config dev      (config 1)  RISC0_DEV_MODE plumbing: fake receipts, no GPU — the partition, control plane and assembly only.

    python bench.py                         # N=1, K=60, W=2 (about a minute)
    python bench.py --gpus 8                # self-launching: spawns 8 ranks (one GPU each), kills siblings if one dies
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W          # the driver's launch shape

N > 1: one process per GPU, NO data-path collective (segments are independent); the control plane (barriers, MAX over ranks,
gathers) is a key/value store that survives a dead rank (benchlib/control.py): rank 0 still prints the line, with `failed_ranks`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from datetime import timedelta

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import PO2, Run                                          # noqa: E402
from benchlib.control import TIMEOUT_S, ControlPlane, RankFailed, launch_ranks  # noqa: E402
from benchlib.roofline import add_roofline, by_op                              # noqa: E402,F401  (re-exported: tests/test_bench_contract.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed seals per GPU (60 x ~23.5 ms: 1.4 s of GPU time; run-to-run spread +-0.2 %%)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed seals per lane before the clock starts (also ramps the clocks)")
    ap.add_argument("--po2", type=int, default=PO2)
    ap.add_argument("--config", choices=("segment", "block", "succinct", "dev"),
                    default="dev" if os.environ.get("RISC0_DEV_MODE", "").lower() in ("1", "true", "yes") else "segment")
    ap.add_argument("--circuit", choices=("syn_a", "syn_heavy", "syn_huge"), default="syn_a",
                    help="syn_huge: the constraint system of a real circuit's size (255 k steps; not built in: generated, verified and compiled "
                         "when it is loaded, ~10 s the first time on this machine)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ZKH_INFLIGHT", "3")),
                    help="segments sealed concurrently per GPU (one host thread + HIP stream each); 1 = strictly serial")
    g = ap.add_argument_group("segment config: what else the default run measures")
    g.add_argument("--no-cpu-baseline", action="store_true")
    g.add_argument("--cpu-full-host", action="store_true",
                   help="cpu_baseline: a larger sample for both legs (the po2-20 unit itself alone, 1/16-unit seals on every core): +~45 s")
    g.add_argument("--no-cpu-all-cores", action="store_true",
                   help="cpu_baseline: skip the all-cores leg (floor(cores / threads) seals at once on disjoint core blocks; default on, ~10-15 s)")
    g.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events (no roofline object)")
    g.add_argument("--no-live-traffic", action="store_true",
                   help="roofline.traffic from the committed PMC file, no rocprofv3 --pmc child runs (no live VALU figures)")
    g.add_argument("--no-certify", action="store_true", help="do not verify the timed seals / compare with the golden digest after the clock")
    g.add_argument("--no-heavy", action="store_true", help="skip the SYN-HEAVY measurement")
    g.add_argument("--heavy-steps", type=int, default=18,
                   help="timed steps of each repetition of the SYN-HEAVY and resident-code legs (two repetitions after three warm-up seals per lane)")
    g.add_argument("--no-resident", action="store_true", help="skip the measurement with the code group kept resident")
    g.add_argument("--no-block", action="store_true", help="skip the short block leg (S distinct segments, witgen in the clock, all verified)")
    g.add_argument("--block-segments", type=int, default=None,
                   help="segments of the block leg (the last one a po2-18 tail); default 64 on one GPU, 256 on N > 1 "
                        "(the strong-scaling figure next to the weak-scaling `value`: BASELINE's metric is a block's wall-clock)")
    g.add_argument("--no-preflight-leg", action="store_true", help="skip the block leg with the host-preflight witness pipeline")
    g.add_argument("--preflight-producers", type=int, default=2, help="host preflight threads per sealing lane")
    g.add_argument("--no-recursive", action="store_true", help="skip the lift2 / join fold of the block leg's receipts")
    g.add_argument("--with-p2-join", action="store_true", help="also fold the block leg's receipts through round 3's P2-JOIN tree")
    g.add_argument("--ingress", choices=("device", "host"), default="device", help="host: additionally time the PCIe-inclusive path")
    g = ap.add_argument_group("block / succinct configs")
    g.add_argument("--segments", type=int, default=None, help="number of segments S (default 256 / 1024)")
    g.add_argument("--no-verify", action="store_true", help="skip the host verification after the clock stops")
    g.add_argument("--chained", action="store_true",
                   help="block: SYN-C segments whose pre-state is their predecessor's post-state (claim continuity), native session executor")
    g.add_argument("--recompute-code", action="store_true",
                   help="re-commit the code group for every segment (upstream's SegmentProver) instead of keeping it resident")
    g.add_argument("--join-circuit", choices=("recursion", "p2_join"), default="recursion",
                   help="succinct: joins that verify both child seals in-circuit (RECURSION programs), or round 3's P2-JOIN joins")
    g.add_argument("--join-po2", type=int, default=18)
    g.add_argument("--fold-inflight", type=int, default=int(os.environ.get("ZKH_FOLD_INFLIGHT", "6")),
                   help="lifts / joins in flight per GPU (a lift's witness schedule is a latency chain: more lanes pack the GPU)")
    g.add_argument("--executor", choices=("native", "python"), default="native",
                   help="succinct + recursion: the native session executor (zkh_session_prove: one pipeline) or round 3's Python-orchestrated fold")
    g.add_argument("--fold", choices=("streamed", "phased"), default="streamed",
                   help="native executor: prove a lift2 / join the moment its children exist, or seal everything first and fold afterwards")
    g.add_argument("--witness", choices=("device", "preflight"), default="device",
                   help="succinct (native): the closed-form generator on the device, or a sequential host preflight per segment + row fill on the GPU")
    g.add_argument("--no-join3", action="store_true", help="recursion: leave the join3 program out (same tree, same root claim, more proofs)")
    g.add_argument("--no-fused-lift", action="store_true", help="recursion: lift every segment on its own and join instead of lift2")
    g = ap.add_argument_group("N > 1")
    g.add_argument("--allow-shared-gpu", action="store_true", default=bool(os.environ.get("ZKH_SHARE_GPUS")),
                   help="let several ranks (or session devices) drive the SAME GPU: a dry run of the N-rank shape on a box with fewer GPUs.  "
                        "Without it an N > 1 run refuses to start unless its ranks hold N distinct devices (config.devices lists them)")
    g.add_argument("--launcher", choices=("ranks", "session"), default="ranks",
                   help="ranks: one process per GPU (the driver's shape).  session: ONE process, zkh_session_create with N devices x "
                        "--inflight lanes (the library's own executor: C++ threads, one work index)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)       # benchlib/roofline.py's rocprofv3 child runs
    return ap.parse_args(argv)


def main() -> int:
    args = parse_args()
    t_start = time.perf_counter()
    dev = args.config == "dev"
    # a source-only snapshot (the GPU box) has no built library: build it before anything else — once, under a file lock, so the
    # ranks of a launcher-started run serialise on it and all but the first find it done (a current tree costs milliseconds)
    build_s = 0.0
    if not dev:
        from zeth_amd import build as _build
        t_build = time.perf_counter()
        _build.ensure_built(oracle=not args.no_cpu_baseline and not args.pmc_child)
        build_s = time.perf_counter() - t_build
    if args.pmc_child:
        from benchlib.segment import run_pmc_child
        run = Run(args, ControlPlane(0, 1), 0, 0, 1)
        run.load_circuit()
        run_pmc_child(run)
        return 0
    if args.allow_shared_gpu:
        os.environ["ZKH_SHARE_GPUS"] = "1"        # (the ranks and the pmc child runs read the environment)
    if args.launcher == "session" and not dev:
        from benchlib.session_launcher import run_session_launcher
        return run_session_launcher(args, t_start, build_s)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Rendezvous through torch.distributed (gloo on CPU tensors; a rank that never shows up fails the group within the timeout
        # instead of gloo's 30 minutes).  The path has no exchange step, so no collective is invented for it; the control plane
        # proper is the store (benchlib/control.py).  RCCL is brought up AFTER the ranks have exchanged their device identities
        # (below): a health probe on a group of its own, by default whenever the ranks hold N distinct GPUs.
        sys.stdout.flush()
        saved_stdout = os.dup(1)                  # gloo announces its connections on stdout; keep stdout for the ONE JSON line
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", timeout=timedelta(seconds=TIMEOUT_S))
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    ctl = ControlPlane.connect(rank, world)
    run = Run(args, ctl, rank, local_rank, world)
    run.failed_ranks, run.failed_in = {}, None
    line, after = None, []
    try:
        ctl.barrier()
        if not dev:
            run.bind_device()
        # who holds which GPU: every rank's {pci_bus_id, uuid, numa_node} goes into config.devices, and an N > 1 run refuses to
        # start unless they are N distinct devices (--allow-shared-gpu: a dry run).  Then RCCL comes up (probe only).
        run.exchange_devices()
        if world > 1 and not dev:
            from benchlib.control import rccl_probe
            run.rccl = rccl_probe(run)
        if dev:
            from benchlib.dev import run_dev
            line, after = run_dev(run)
        else:
            run.load_circuit()
            if args.config == "segment":
                from benchlib.segment import run_segment
                line, after = run_segment(run)
            elif args.config == "block":
                from benchlib.block import run_block
                line, after = run_block(run)
            else:
                from benchlib.succinct import run_succinct
                line, after = run_succinct(run)
    except RankFailed as e:
        # this rank is out (its own leg failed, or its peers gave up on it): say so, stay until rank 0 has printed, exit non-zero
        sys.stderr.write(f"bench: {e}\n")
        if ctl.i_failed is None:
            ctl.fail("the control plane", e)
        ctl.wait_done()
        return 3
    except BaseException as e:
        if world > 1 and ctl.i_failed is None:
            try:
                ctl.fail(args.config, e)
            except Exception:
                pass
        raise
    if world > 1 and not run.failed_ranks and not run.rccl_hung:
        try:                                      # (with a dead rank the group is left alone: tearing it down can wait for the dead)
            import torch.distributed as dist
            t_d = time.perf_counter()
            dist.destroy_process_group()
            if os.environ.get("ZKH_BENCH_TRACE_EXIT"):
                sys.stderr.write(f"rank {rank}: destroy_process_group {time.perf_counter() - t_d:.2f} s\n")
        except Exception:
            pass
    rc = 0
    if rank == 0 and line is not None:
        for fn in after:
            fn()
        line["build_s"] = round(build_s, 2)
        line["command_wall_s"] = round(time.perf_counter() - t_start, 1)
        if run.failed_ranks:
            line["failed_in"] = run.failed_in
            rc = 4 if run.failed_in == "headline" else 0      # a secondary leg's failure is in the line; `value` is whole
        print(json.dumps(line))
        sys.stdout.flush()
        ctl.finish()
    if run.rccl_hung:                             # an RCCL probe that never returned: its communicator's teardown could block the exit
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(rc)
    return rc


if __name__ == "__main__":
    _rc = main()
    if os.environ.get("ZKH_BENCH_TRACE_EXIT"):
        sys.stderr.write(f"rank {os.environ.get('RANK')}: main returned {_rc} at {time.time():.2f}\n")
    raise SystemExit(_rc)
