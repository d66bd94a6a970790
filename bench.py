#!/usr/bin/env python3
"""bench.py — segments/sec for sealing 2^20-cycle zkVM segments on N MI355X (BASELINE.json metric).

One "step" = one segment seal (commit code/data, accum, eval_check, DEEP, FRI,
queries) on a witness that is already resident in HBM when the timed region starts.  Workload at every N is
BASELINE.json configs[1] per GPU: one 2^20-cycle SYN-A segment (W_code 16, W_data 208, W_accum 32; SYN-AIR is
the declared-synthetic stand-in for the un-obtainable rv32im circuit — DESIGN.md).  With N GPUs the segment
list is partitioned round-robin (segment i -> rank i mod N, one process per GPU, no data-path collective), so
per-GPU work is fixed as N grows: "scaling": "weak", value = N*K segments / max-over-ranks time.

    python bench.py --gpus 1 --steps 30 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
PO2 = 20
CPU_SAMPLE_PO2 = 17             # bounded CPU-baseline sample: one SYN-A segment at 2^17 cycles (1/8 of the unit)


def seal_algorithmic_bytes(wa: int, wc: int, wd: int, n_taps: int, n_combos: int, n: int) -> float:
    """SURVEY.md §8d per-op read-once + write-once bytes for ONE seal, parametric in the column counts."""
    groups = [wc, wd, wa]
    commit = sum(60 * w + 512 for w in groups) * n                 # iNTT, shift, expand-NTT, bitrev, hash_rows, hash_fold
    sigma = wa + wc + wd
    eval_check = (16 * sigma + 64) * n
    check_group = (128 + 44 * 16 + 512) * n
    deep = 4 * (sigma + 16) * n * 1                                 # each column streamed once per evaluation pass
    mix = (4 * (sigma + 16) + 4 * 2 * 16 * (n_combos + 1)) * n
    combos = (2 * 32 * (n_combos + 1) + 16 * (n_combos + 1) + 16 + 32) * n
    fri = 208 * n
    return float(commit + eval_check + check_group + deep + mix + combos + fri)


def cpu_baseline(desc) -> dict:
    """The CPU oracle (a from-spec port of the reference CPU prover's algorithm) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zko                                     # test infrastructure; used here ONLY as the reported CPU baseline
    lib = zko.load()
    oc = zko.OracleCircuit(lib, desc)
    # the oracle's OpenMP loops stop scaling long before a two-socket host is full (fork/join + memory bound): scan a
    # few thread counts on a small segment and quote the baseline at the fastest one
    avail = int(lib.zko_num_threads())
    best, best_dt = avail, None
    for t in sorted({c for c in (8, 16, 32, 64, avail) if c <= avail}):
        lib.zko_set_num_threads(t)
        t0 = time.perf_counter()
        oc.prove(CPU_SAMPLE_PO2 - 2, 1994, 0x5EED0000, 0x2E80)
        d = time.perf_counter() - t0
        if best_dt is None or d < best_dt:
            best, best_dt = t, d
    lib.zko_set_num_threads(best)
    t0 = time.perf_counter()
    seal = oc.prove(CPU_SAMPLE_PO2, 1994, 0x5EED0000, 0x2E80)
    dt = time.perf_counter() - t0
    scale = 1 << (PO2 - CPU_SAMPLE_PO2)
    return {"value": 1.0 / (dt * scale), "unit": "segments/s", "cores": best, "cores_available": avail, "kind": "port",
            "sample": f"one SYN-A segment seal at po2={CPU_SAMPLE_PO2} ({dt:.2f} s wall, OpenMP oracle incl. witgen, at the fastest of "
                      f"8/16/32/64/{avail} threads = {best}), "
                      f"scaled x1/{scale} to the po2={PO2} unit (work is ~linear in n)",
            "seal_words": int(seal.size)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed seals per GPU (30 x ~32 ms: about a second of GPU time)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed seals per lane before the clock starts (also ramps the clocks)")
    ap.add_argument("--po2", type=int, default=PO2)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ZKH_INFLIGHT", "3")),
                    help="segments sealed concurrently per GPU (one host thread + HIP stream each); 1 = strictly serial")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events (no roofline object)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane only — the path has no exchange step (segments are independent), so there is no collective on
        # the data path and none is invented: barrier and MAX-of-elapsed go over gloo on CPU tensors, which keeps the
        # timing protocol independent of torch's own HIP state.  ZKH_DIST_BACKEND=cpu:gloo,cuda:nccl additionally
        # brings RCCL up next to it and runs one all_reduce over xGMI before the timed region (health probe only).
        backend = os.environ.get("ZKH_DIST_BACKEND", "gloo")
        if "nccl" in backend:
            try:
                torch.cuda.set_device(local_rank)
            except (RuntimeError, AssertionError):
                backend = "gloo"
        ctrl_dev = "cpu" if "gloo" in backend else f"cuda:{local_rank}"

        def barrier():
            dist.all_reduce(torch.zeros(1, device=ctrl_dev))

        # gloo announces its connections on stdout; keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend)
            barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from zeth_amd.circuits import syn_air
    from zeth_amd.circuits.desc import Circuit
    from zeth_amd.hal import HipHal
    from zeth_amd.host import partition_round_robin
    from zeth_amd.prover import Segment, SegmentProver

    # one GPU per rank; ZKH_SHARE_GPUS=1 lets ranks wrap around the visible devices (dry runs on a 1-GPU box)
    device = local_rank % torch.cuda.device_count() if os.environ.get("ZKH_SHARE_GPUS") else local_rank
    desc = syn_air.syn_a()
    circ = Circuit.parse(desc)
    wa, wc, wd = circ.group_sizes
    import threading
    n = 1 << args.po2
    inflight = max(1, min(args.inflight, args.steps))

    # segment list of the "block": (warmup + steps) * world segments, partitioned round-robin over ranks; inside a
    # rank, `inflight` host threads (one HipHal context = one HIP stream each) seal different segments concurrently so
    # that the latency-bound phases of one seal (Merkle tree tops, scans, Fiat-Shamir round trips) overlap another's
    # throughput-bound phases.  Segments stay independent: no data is shared between the threads.
    total = (args.warmup + args.steps) * world
    mine = partition_round_robin(total, world, rank)

    class Worker:
        def __init__(self, w):
            self.hal = HipHal(device)                # raises if the HIP library / GPU is missing: no fallback
            self.prover = SegmentProver(self.hal, desc)
            self.sealed = 0
            self.seal_s = []
            ring = max(1, min(-(-args.steps // inflight) + args.warmup, 2))
            self.wit = []
            self.witgen_s = []
            for j in range(ring):                    # witnesses resident in HBM before the clock starts
                idx = mine[(w + j * inflight) % len(mine)]
                seg = Segment(index=idx, po2=args.po2, seed=0x5EED0000 + idx)
                t_w = time.perf_counter()
                self.wit.append((seg, *self.prover.witgen(seg)))
                self.hal.sync()
                self.witgen_s.append(time.perf_counter() - t_w)
            self.last = None
            self.err = None

        def seal(self, i):
            seg, code, data, out = self.wit[i % len(self.wit)]
            t_s = time.perf_counter()
            self.last = self.prover.seal(seg, code, data, out)   # returns with the seal words on the host
            self.seal_s.append(time.perf_counter() - t_s)

        def run(self):
            # the K timed steps are handed out through a shared work index (work stealing), so K need
            # not be a multiple of the number of seals in flight
            try:
                while next_step() is not None:
                    self.seal(args.warmup + self.sealed)
                    self.sealed += 1
                self.hal.sync()
            except Exception as e:                   # surfaced after join
                self.err = e

    def device_sync():
        """Both sides of the timed region: every library stream, then torch's device-wide synchronize (torch is only
        plumbing here; if its own HIP initialisation is unavailable the library's syncs already cover all our work)."""
        for wk in workers:
            wk.hal.sync()
        try:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except (RuntimeError, AssertionError):
            pass

    work_lock, work_next = threading.Lock(), [0]

    def next_step():
        with work_lock:
            k = work_next[0]
            if k >= args.steps:
                return None
            work_next[0] = k + 1
            return k

    workers = [Worker(w) for w in range(inflight)]
    for wk in workers:
        for i in range(args.warmup):
            wk.seal(i)
        wk.hal.sync()
    if not args.no_prof:
        for wk in workers:
            wk.hal.prof_reset()
            wk.hal.prof_enable(True)
    rccl = None
    if distributed and "nccl" in backend:
        try:
            probe = torch.ones(1, device=f"cuda:{device}")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            rccl = "ok" if int(probe.item()) == world else "wrong sum"
        except Exception as e:                       # control plane stays on gloo; the seals never needed RCCL
            rccl = f"unavailable ({type(e).__name__})"
    device_sync()
    if distributed:
        barrier()
    for wk in workers:
        wk.seal_s.clear()
    t0 = time.perf_counter()
    threads = [threading.Thread(target=wk.run) for wk in workers]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    device_sync()
    if distributed:
        barrier()
    dt = time.perf_counter() - t0
    for wk in workers:
        if wk.err is not None:
            raise wk.err
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=ctrl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = []
    if not args.no_prof:
        merged = {}
        for wk in workers:
            for p in wk.hal.prof_get():
                m = merged.setdefault(p["name"], {"name": p["name"], "calls": 0, "total_ms": 0.0, "alg_bytes": 0.0})
                m["calls"] += p["calls"]; m["total_ms"] += p["total_ms"]; m["alg_bytes"] += p["alg_bytes"]
            wk.hal.prof_enable(False)
        prof = list(merged.values())
    # With several seals in flight the HIP-event brackets of one stream include time its kernels spent sharing the GPU
    # with the other streams.  One more seal, alone on the GPU and outside the timed region, gives the unshared
    # per-kernel durations next to them (and names the kernel that really dominates the work).
    seal_times = [t for wk in workers for t in wk.seal_s]
    unloaded_seal_s = None
    ref = []
    if prof and inflight > 1 and rank == 0:
        w0 = workers[0]
        w0.hal.prof_reset(); w0.hal.prof_enable(True)
        w0.seal(args.warmup)
        w0.hal.sync()
        ref = w0.hal.prof_get()
        w0.hal.prof_enable(False)
        unloaded_seal_s = w0.seal_s[-1]              # one seal alone on the GPU: the single-segment latency
    last = next((wk.last for wk in workers if wk.last is not None), None)

    if rank == 0:
        value = world * args.steps / dt
        line = {
            "metric": "segments/sec", "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"single 2^{args.po2}-cycle segment seal per step per GPU, SYN-A circuit "
                                   f"(W_code {wc}, W_data {wd}, W_accum {wa}, check 16; {len(circ.taps)} taps), poseidon2, "
                                   "witness resident in HBM", "po2": args.po2,
                       "parallelism": f"segments round-robin over {world} GPU(s), no collectives; {inflight} segment(s) in flight per GPU",
                       "rccl_probe": rccl,
                       "inflight_per_gpu": inflight,
                       "seal_words": int(last.seal.size) if last is not None else 0},
            # wall-clock of one seal call (enqueue .. seal words on the host), mean over the timed seals of this rank;
            # with several seals in flight each one shares the GPU, so this is latency under load, not 1/value
            "seal_wall_clock_s": sum(seal_times) / max(1, len(seal_times)),
            # ... and of one seal with the GPU to itself (inflight 1: the same thing as seal_wall_clock_s)
            "seal_wall_clock_unloaded_s": unloaded_seal_s if unloaded_seal_s is not None else sum(seal_times) / max(1, len(seal_times)),
            # reported separately: synthetic witness generation on the device, outside the timed region
            "witgen_ms_per_segment": 1e3 * min(t for wk in workers for t in wk.witgen_s[1:] or wk.witgen_s),
        }
        alg = seal_algorithmic_bytes(wa, wc, wd, len(circ.taps), len(circ.combos), n)
        line["seal_roofline"] = {"alg_bytes": alg, "achieved": alg / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS}
        if prof:
            tot_ms = sum(p["total_ms"] for p in prof)
            unshared = {p["name"]: p for p in (ref or prof)}
            dom_name = max(unshared.values(), key=lambda p: p["total_ms"])["name"]
            dom = next(p for p in prof if p["name"] == dom_name)
            per_launch_ms = dom["total_ms"] / dom["calls"]
            per_launch_ms_unshared = unshared[dom_name]["total_ms"] / unshared[dom_name]["calls"]
            per_launch_bytes = dom["alg_bytes"] / dom["calls"]
            ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
            # FETCH doubled per the gfx950 correction; tools/pmc_summary.py) — bench.py cannot run rocprof on itself
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                kname = {"hash_rows": "k_hash_rows", "hash_fold": "k_hash_fold", "eval_check": "k_eval_check_syn_a"}.get(dom["name"])
                if kname in tj and args.po2 == PO2:
                    traffic = (tj[kname]["fetch_x2_bytes"] + tj[kname]["write_bytes"]) / tj[kname]["launches"]
            except Exception:
                traffic = None
            line["roofline"] = {"bound": "hbm", "kernel": dom["name"], "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": per_launch_ms,
                                "avg_launch_ms_unshared": per_launch_ms_unshared,
                                "achieved_unshared": per_launch_bytes / (per_launch_ms_unshared * 1e-3) / 1e9,
                                "alg_bytes_per_launch": per_launch_bytes,
                                "share_of_kernel_time": unshared[dom_name]["total_ms"] / sum(p["total_ms"] for p in unshared.values()),
                                "launches_overlap": inflight > 1,
                                "note": "dominant kernel is integer-VALU-bound by construction (Poseidon2: ~21 Montgomery "
                                        "products per absorbed byte); HBM fraction is reported as the contract asks; with "
                                        "inflight_per_gpu > 1 kernels of different seals overlap, so avg_launch_ms (timed region) "
                                        "includes time shared with other streams; *_unshared comes from one extra seal run "
                                        "alone after the timed region"}
            if dom["name"] == "hash_rows":
                # VALU view of the same kernel: permutations per launch x modelled issue cycles per 64-lane permutation
                # (8 full rounds x 2368 + 7 partial groups x 1576 + 1024 + 138 cycles) against 1024 SIMDs at 2.4 GHz
                perms = sum(-(-w // 16) for w in (wc, wd, wa, 16)) * 4 * n          # leaves of the 3 trace trees + check tree
                deg = n
                while deg > 256:                                                   # FRI rounds: 4*deg/16 rows of 64 words
                    perms += 4 * (4 * deg // 16)
                    deg //= 16
                cyc = 8 * 2368 + 7 * 1576 + 1024 + 138
                per_seal_ms = unshared[dom_name]["total_ms"] / (1 if ref else args.steps)
                line["roofline"]["valu"] = {"permutations_per_seal": perms, "model_cycles_per_wave_permutation": cyc,
                                            "issue_utilisation_at_2p4GHz": (perms / 64.0) * cyc / (1024 * 2.4e9 * per_seal_ms * 1e-3)}
            div = 1 if ref else args.steps
            line["kernels"] = [{"name": p["name"], "calls_per_seal": p["calls"] / div,
                                "ms_per_seal": p["total_ms"] / div,          # unshared (one seal alone on the GPU)
                                "ms_per_seal_timed_region": next((q["total_ms"] / args.steps for q in prof if q["name"] == p["name"]), None),
                                "alg_GBps": (p["alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9) if p["total_ms"] > 0 else 0.0}
                               for p in sorted(unshared.values(), key=lambda p: -p["total_ms"])]
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(desc)
            except Exception as e:       # the baseline is a reported number, never a dependency of the product path
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
