"""--config segment (BASELINE config 2; the default line): every step seals one 2^po2-cycle segment whose witness is already
resident in HBM when the clock starts; N GPUs = N ranks each doing K steps ("scaling": "weak").

The headline is measured first and on its own.  Everything after it is a SECONDARY leg of the same run — the heavy constraint
system, the resident code group, a short block (configs 3 / 4 in small), the host-preflight witness pipeline, the block folded to
one receipt — each guarded: a leg that fails is reported as `{"error": ...}` in its place and never takes the headline with it."""
from __future__ import annotations

import hashlib
import json
import os
import time

from .common import (BASE_SEED, BENCH_NOISE, HBM_PEAK_GBPS, ROOT, Fault, Run, WorkIndex, block_segments, config_common, fold_lanes,
                     maybe_fault, merged_prof, recursive_fold, recursive_prepare, run_lanes, seal_algorithmic_bytes, seal_block)
from .control import LegAborted, RankFailed


def consensus_failed(run: Run) -> dict:
    """what every live rank agrees has failed so far (one exchange: the union of the ranks' views)"""
    if not run.distributed:
        return {}
    views = run.ctl.allgather(run.ctl.failed())
    out = {}
    for v in views.values():
        out.update(v)
    return out


def secondary(run: Run, name: str, fn):
    """Run a secondary leg: the live ranks enter and leave it together.  A rank whose leg raises on its own leaves the group (the
    others stop waiting for it and report the leg as incomplete); a rank whose leg breaks BECAUSE a peer is gone (its share of a
    gather is missing, ...) aborts the leg for everybody and stays; with one rank the error is simply recorded in the leg's place."""
    if run.failed_ranks:
        return {"skipped": f"rank(s) {sorted(run.failed_ranks)} failed earlier in this run"}
    ctl = run.ctl
    ctl.begin_leg(name)
    out, err = None, None
    try:
        maybe_fault(run, name)
        out = fn()
    except RankFailed:
        raise
    except LegAborted as e:
        err = f"aborted: {e}"
    except Exception as e:
        if not run.distributed:
            ctl.leg = None
            return {"error": repr(e)}
        if not ctl.failed():                       # nobody else is to blame: this rank is broken and leaves
            raise RankFailed(ctl.fail(f"the {name} leg", e))
        err = repr(e)
        ctl.abort_leg(err)
    views = ctl.end_leg((ctl.failed(), err)) if run.distributed else {0: ({}, err)}
    failed = {}
    for v, _ in views.values():
        failed.update(v)
    errs = [e for _, e in views.values() if e]
    run.failed_ranks = failed
    if failed or errs:
        run.failed_in = run.failed_in or name
        why = [failed[r] for r in sorted(failed)] + errs
        return {"error": f"incomplete: {'; '.join(why)}", "partial": out}
    return out


def run_segment(run: Run):
    """-> (line or None on ranks other than 0, [callables for rank 0 to run after the other ranks are gone])"""
    from zeth_amd.hal import HipHal
    from zeth_amd.host import partition_round_robin
    from zeth_amd.prover import Segment, SegmentProver
    args, ctl, rank, world = run.args, run.ctl, run.rank, run.world
    wa, wc, wd = run.widths
    n = run.n
    inflight = run.inflight = max(1, min(args.inflight, args.steps))
    run.failed_ranks = {}
    # segment list of the "block": (warmup + steps) * world segments, partitioned round-robin over ranks; inside a rank, `inflight`
    # host threads (one HipHal context = one HIP stream each) seal different segments concurrently so that the latency-bound phases
    # of one seal (Merkle tree tops, scans, Fiat-Shamir round trips) overlap another's throughput-bound phases.  Segments stay
    # independent: no data is shared between the threads.
    total = (args.warmup + args.steps) * world
    mine = partition_round_robin(total, world, rank)
    idx = WorkIndex()

    def seal_one(ln, i):
        seg, code, data, out = ln.wit[i % len(ln.wit)]
        t_s = time.perf_counter()
        ln.last = ln.prover.seal(seg, code, data, out)   # returns with the seal words on the host
        ln.seal_s.append(time.perf_counter() - t_s)
        ln.sealed.append((seg, ln.last))                 # kept: every timed seal is verified after the clock

    def timed(ln):
        # the K timed steps are handed out through a shared work index (work stealing), so K need not be a
        # multiple of the number of seals in flight
        try:
            done = 0
            while idx.take(args.steps) is not None:
                seal_one(ln, args.warmup + done)
                done += 1
            ln.hal.sync()
        except Exception as e:                   # surfaced after join
            ln.err = e

    # ---------------------------------------------------------------- the headline ----
    try:
        lanes = [run.lane() for _ in range(inflight)]
        for w, ln in enumerate(lanes):
            ring = max(1, min(-(-args.steps // inflight) + args.warmup, 2))
            ln.wit = []
            for j in range(ring):                    # witnesses resident in HBM before the clock starts
                k = mine[(w + j * inflight) % len(mine)]
                seg = Segment(index=k, po2=args.po2, seed=BASE_SEED + k, noise_seed=BENCH_NOISE)
                t_w = time.perf_counter()
                ln.wit.append((seg, *ln.prover.witgen(seg)))
                ln.hal.sync()
                ln.witgen_s.append(time.perf_counter() - t_w)
        for ln in lanes:
            for i in range(args.warmup):
                seal_one(ln, i)
            ln.hal.sync()
        if not args.no_prof:
            for ln in lanes:
                ln.hal.prof_reset()
                ln.hal.prof_enable(True)
        maybe_fault(run, "headline")
        run.device_sync(lanes)
        ctl.barrier()
        for ln in lanes:
            ln.seal_s.clear()
            ln.sealed.clear()
        t0 = time.perf_counter()
        run_lanes(lanes, timed)
        run.device_sync(lanes)
        my_dt = time.perf_counter() - t0          # this rank's K steps, both device syncs inside
        ctl.barrier()
        dt = time.perf_counter() - t0
    except RankFailed:
        raise
    except Exception as e:
        if run.distributed:
            raise RankFailed(ctl.fail("the headline leg", e))
        raise
    # MAX over the ranks that finished (a rank that died inside the region is in `failed_ranks`, its steps are not counted)
    dt = ctl.max(dt)
    steps_done = int(ctl.sum([float(args.steps)])[0])
    run.failed_ranks = consensus_failed(run)
    if run.failed_ranks:
        run.failed_in = "headline"
    headline_ranks = world - len(run.failed_ranks)         # the ranks whose steps are in `value`
    prof = []
    if not args.no_prof:
        prof = merged_prof(lanes)
        for ln in lanes:
            ln.hal.prof_enable(False)

    # ---- after the clock: the timed work certifies itself.  EVERY seal produced inside the timed region goes through the host
    # verifier (`receipt.verify`, /root/reference/crates/host/src/bin/cli.rs:103) against the control root of its size, and the seal
    # of segment index 0 — whose seeds are exactly the CPU oracle's golden case (tests/golden/large_digests.json, made by
    # tests/golden/make_golden_large.py) — is compared with the oracle's seal by SHA-256.  No oracle code runs here: the digest
    # is a committed fixture.
    def certify_leg():
        t_v = time.perf_counter()
        sealed = [x for ln in lanes for x in ln.sealed]
        croot = lanes[0].prover.control_root(args.po2)
        for seg, rec in sealed:
            rec.verify(run.desc, croot)                      # raises HalError if a timed seal is rejected
        golden, matches = None, None
        try:
            cases = json.load(open(os.path.join(ROOT, "tests", "golden", "large_digests.json")))["cases"]
            golden = next((c for c in cases if c["shape"] == args.circuit and c["po2"] == args.po2 and c["seed"] == BASE_SEED
                           and c["noise_seed"] == BENCH_NOISE and c["zk_cycles"] == 1994), None)
        except (OSError, ValueError, KeyError):
            pass
        zero = [rec for seg, rec in sealed if seg.index == 0]
        if golden is not None and zero:
            matches = all(hashlib.sha256(rec.seal_bytes()).hexdigest() == golden["seal_sha256"] for rec in zero)
            if not matches:
                raise SystemExit("bench: the timed seal of segment 0 differs from the CPU oracle's golden seal (tests/golden/large_digests.json)")
        cnt = ctl.sum([float(len(sealed))])[0]
        return {"timed_seals_verified": int(cnt), "seal_matches_golden": matches,
                "golden_is": "the SHA-256 of THIS repository's CPU oracle seal for the same seeds (tests/golden/large_digests.json): a regression pin "
                             "that ties the timed GPU seal to the oracle, not a vector held by the reference (it holds none for this path)",
                "golden_seals_compared": len(zero) if golden is not None else 0,
                "verify_ms_per_seal_host": 1e3 * (time.perf_counter() - t_v) / max(1, len(sealed))}

    certify = None if args.no_certify else secondary(run, "certify", certify_leg)

    # With several seals in flight the HIP-event brackets of one stream include time its kernels spent sharing the GPU with the
    # other streams.  One more seal, alone on the GPU and outside the timed region, gives the unshared per-kernel durations next
    # to them (and names the kernel that really dominates the work).
    seal_times = [t for ln in lanes for t in ln.seal_s]
    ref_box = {"ref": [], "unloaded_seal_s": None}

    def reference_seal():
        ctl.barrier()                                    # every rank is past its certification: nothing else runs while rank 0 takes its reference seal
        if rank == 0:
            w0 = lanes[0]
            w0.hal.prof_reset(); w0.hal.prof_enable(True)
            seal_one(w0, args.warmup)
            w0.hal.sync()
            ref_box["ref"] = w0.hal.prof_get()
            w0.hal.prof_enable(False)
            ref_box["unloaded_seal_s"] = w0.seal_s[-1]   # one seal alone on the GPU: the single-segment latency
        ctl.barrier()
        return True
    if prof:
        secondary(run, "reference_seal", reference_seal)
    ref, unloaded_seal_s = ref_box["ref"], ref_box["unloaded_seal_s"]

    # PCIe-inclusive variant: the same K steps, but every step uploads its code + data traces from pinned host memory
    def pcie_leg():
        for ln in lanes:
            seg, code, data, out = ln.wit[0]
            ln.host = (ln.hal.host_alloc(code.size()), ln.hal.host_alloc(data.size()))
            ln.host[0][:] = code.to_vec()
            ln.host[1][:] = data.to_vec()

        def host_step(ln):
            seg, _, _, out = ln.wit[0]
            ln.last = ln.prover.seal_host_witness(seg, ln.host[0], ln.host[1], out)

        def timed_host(ln):
            try:
                while idx.take(args.steps) is not None:
                    host_step(ln)
                ln.hal.sync()
            except Exception as e:
                ln.err = e

        for ln in lanes:
            host_step(ln)
        run.device_sync(lanes)
        ctl.barrier()
        idx.reset()
        t1 = time.perf_counter()
        run_lanes(lanes, timed_host)
        run.device_sync(lanes)
        ctl.barrier()
        dth = ctl.max(time.perf_counter() - t1)
        up_bytes = 4.0 * (lanes[0].host[0].size + lanes[0].host[1].size)
        out = {"segments_per_s": world * args.steps / dth, "ms_per_step": 1e3 * dth / args.steps,
               "upload_bytes_per_segment": up_bytes, "upload_GBps_sustained": up_bytes * args.steps / dth / 1e9,
               "note": "code + data traces uploaded from pinned host memory (zkh_write_async) inside every step, "
                       "sealed through zkh_prove_begin / zkh_prove_finish; uploads of one lane overlap the kernels of the others"}
        for ln in lanes:
            for h in ln.host:
                ln.hal.host_free(h)
        return out
    pcie = secondary(run, "pcie", pcie_leg) if args.ingress == "host" else None

    REPS, EXTRA_WARM = 2, 2

    def rep_stats(times, steps):
        """seconds of each repetition of `steps` steps -> the figures a secondary leg reports: the rate over ALL repetitions, the
        slowest and the fastest repetition, and whether they agree (a leg whose repetitions differ by more than 8 % says so)"""
        rates = [world * steps / t for t in times]
        v = world * steps * len(times) / sum(times)
        spread = (max(rates) - min(rates)) / v
        out = {"segments_per_s": v, "min": min(rates), "max": max(rates), "reps": len(times), "spread_pct": 100.0 * spread,
               "ms_per_step": 1e3 * sum(times) / (steps * len(times)), "steps": steps}
        if spread > 0.08:
            out["unstable"] = f"the {len(times)} repetitions differ by {100.0 * spread:.1f} % (> 8 %): quote min .. max, not the mean"
        return out

    def timed_extra(make_prover, steps, with_prof):
        """`steps` more timed steps of the same resident witnesses under another prover per lane, REPS times over -> (seconds of
        each repetition, per-kernel times of one seal alone).  Before the first clock every lane seals once on its own (that seal
        loads the prover's code objects and sizes its pool blocks) and then EXTRA_WARM more times with the other lanes running —
        round 5 timed 9 steps after one seal per lane and the driver's box read 26 % low."""
        for ln in lanes:
            ln.extra = make_prover(ln)

        def one(ln):
            seg, code, data, out = ln.wit[0]
            ln.last_extra = ln.extra.seal(seg, code, data, out)

        def warm_x(ln):
            try:
                for _ in range(EXTRA_WARM):
                    one(ln)
                ln.hal.sync()
            except Exception as e:
                ln.err = e

        def timed_x(ln):
            try:
                while idx.take(steps) is not None:
                    one(ln)
                ln.hal.sync()
            except Exception as e:
                ln.err = e

        for ln in lanes:
            one(ln)
            ln.hal.sync()
        run_lanes(lanes, warm_x)
        kprof = {}
        if with_prof and rank == 0 and not args.no_prof:
            lanes[0].hal.prof_reset(); lanes[0].hal.prof_enable(True)
            one(lanes[0]); lanes[0].hal.sync()
            kprof = {p["name"]: p for p in lanes[0].hal.prof_get()}
            lanes[0].hal.prof_enable(False)
        times = []
        for _ in range(REPS):
            run.device_sync(lanes)
            ctl.barrier()
            idx.reset()
            t2 = time.perf_counter()
            run_lanes(lanes, timed_x)
            run.device_sync(lanes)
            ctl.barrier()
            times.append(ctl.max(time.perf_counter() - t2))
        return times, kprof

    # The same step under the realistically heavy constraint system (SYN-HEAVY: same trace shape and witness, ~54 k constraint steps
    # instead of ~1 k): SYN-A's eval_check is 4 % of a seal, upstream's is the largest kernel, so the headline flatters the real
    # workload and this one is reported next to it (same lanes, same resident witnesses, a few steps).
    def heavy_leg():
        from zeth_amd.circuits import syn_heavy
        from zeth_amd.circuits.desc import Circuit
        hdesc = syn_heavy.syn_heavy()
        hsteps = max(inflight, min(args.heavy_steps, args.steps))
        times, hprof = timed_extra(lambda ln: SegmentProver(ln.hal, hdesc), hsteps, True)
        hc = Circuit.parse(hdesc)
        out = rep_stats(times, hsteps)
        out["value"], out["unit"] = out["segments_per_s"], "segments/s"
        return {**out,
                "workload": f"same step with the SYN-HEAVY constraint system ({len(hc.steps)} steps, {len(hc.taps)} taps, "
                            f"{len(hc.combos)} tap combos, degree 5, ConstExt, nested AndCond; {lanes[0].extra.circuit.compiled_parts()} generated kernels)",
                "kernels_ms_per_seal_unshared": {k: round(v["total_ms"], 3) for k, v in sorted(hprof.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}}
    heavy = secondary(run, "syn_heavy", heavy_leg) if args.circuit == "syn_a" and not args.no_heavy and args.po2 >= 13 else None

    # The same step with the committed code (control) group of this segment size kept resident in HBM instead of being re-committed
    # for every segment (zkh_prover_cache_code; the group is a function of (circuit, po2) alone, 0.6 GB at po2 20).  Upstream's
    # SegmentProver recomputes it and so does `value`; this is what a deployment that keeps it gets.
    def resident_leg():
        import numpy as np
        rsteps = max(inflight, min(args.heavy_steps, args.steps))
        times, _ = timed_extra(lambda ln: SegmentProver(ln.hal, run.desc, resident_code_group=True), rsteps, False)
        same = all(np.array_equal(ln.last_extra.seal, ln.prover.seal(*ln.wit[0]).seal) for ln in lanes)   # same witness, recomputing prover
        return {**rep_stats(times, rsteps),
                "seals_identical_to_recomputing_prover": bool(same),
                "note": "NOT the headline: the code group's iNTT / expand-NTT / leaf hashing / Merkle fold are skipped because "
                        "its committed form is resident (opt-in: SegmentProver(resident_code_group=True))"}
    resident = secondary(run, "code_group_resident", resident_leg) if not args.no_resident and args.po2 >= 13 else None

    # A short block in the same run (BASELINE's metric is "segments/sec + seal wall-clock" of a block: configs 3/4): S DISTINCT
    # segments, the last one a po2-18 tail, round-robin over the ranks, witness generation INSIDE the clock, every seal verified on
    # the host after the clock.  `--config block` is the full-size version (S = 256).
    if args.block_segments is None:
        args.block_segments = 64 if world == 1 else 256
    S = args.block_segments
    bstate = {}

    def block_leg():
        bsegs = block_segments(run, S)
        bmine = partition_round_robin(S, world, rank)
        # The block leg keeps the committed code (control) group of each segment size RESIDENT per lane (it is a function
        # of (circuit, po2) alone; seals are byte-identical) — what the session executor does by default.  Upstream's SegmentProver
        # re-commits it per segment: that figure is reported next to it (`recompute_code_group`), and `value` above is measured that way.
        for ln in lanes:                                  # every size once, outside the clock (pool blocks, code objects, the resident groups)
            ln.block_prover = ln.prover if args.recompute_code else SegmentProver(ln.hal, run.desc, resident_code_group=True)
            for p2 in sorted({sg.po2 for sg in bsegs}):
                ln.block_prover.prove_segment(Segment(index=0, po2=p2, seed=1, noise_seed=BENCH_NOISE))
                ln.prover.prove_segment(Segment(index=0, po2=p2, seed=1, noise_seed=BENCH_NOISE))
            ln.hal.sync()
        broots = {p: lanes[0].prover.control_root(p) for p in sorted({sg.po2 for sg in bsegs})}
        recompute = None
        if not args.recompute_code:
            _, tr0, _, _ = seal_block(run, lanes, bsegs, bmine)
            ctl.barrier()
            trc = ctl.max(time.perf_counter() - tr0)
            recompute = {"wall_clock_s": trc, "segments_per_s": S / trc,
                         "note": "the same block with the code group re-committed for every segment, as upstream's SegmentProver does"}
        maybe_fault(run, "block")
        brec, tb0, bwit, bseal = seal_block(run, lanes, bsegs, bmine, prover_of=lambda ln: ln.block_prover)
        ctl.barrier()
        dtb = time.perf_counter() - tb0
        t_v = time.perf_counter()
        for i in bmine:
            brec[i].verify(run.desc, broots[bsegs[i].po2])
        verify_s = time.perf_counter() - t_v
        dtb = ctl.max(dtb)
        tb = ctl.sum([float(len(bmine)), sum(bwit), float(len(bwit))])
        bstate.update(bsegs=bsegs, bmine=bmine, broots=broots, brec=brec)
        return {"segments": S, "wall_clock_s": dtb, "segments_per_s": S / dtb,
                "tail_po2": bsegs[-1].po2, "witgen_in_clock": True, "verified_after_clock": int(tb[0]),
                "code_group": "recomputed per segment" if args.recompute_code else "resident per lane and size (byte-identical seals)",
                "recompute_code_group": recompute,
                "witgen_ms_per_segment": 1e3 * tb[1] / max(1.0, tb[2]),
                "verify_s_rank0": verify_s,
                "workload": f"{S} distinct 2^{args.po2}-cycle segments (last one 2^{bsegs[-1].po2}), round-robin over {world} GPU(s), "
                            f"{inflight} in flight per GPU; `--config block` runs S = 256"}
    block = secondary(run, "block", block_leg) if not args.no_block and args.po2 >= 13 and S > 0 else None
    block_ok = isinstance(block, dict) and "error" not in block and "skipped" not in block

    # The same block with upstream's witness SHAPE: a sequential host preflight per segment replays the cycles
    # on host threads that run AHEAD of the seals (2 per sealing lane), 16 bytes per cycle cross PCIe from pinned memory, the GPU
    # row-fill kernel expands them (csrc/preflight.hip), and the preload is a zkh_scatter — through the native session executor
    # (zkh_session_set_witness_source(1)).  The host CPU seconds per segment are the Amdahl term of the pipeline: with T producer
    # threads it sustains min(GPU rate, T / preflight seconds).
    def preflight_leg():
        from zeth_amd.hal import HalError
        from zeth_amd.host import Session
        bsegs, bmine, broots = bstate["bsegs"], bstate["bmine"], bstate["broots"]
        for ln in lanes:                       # the session brings its own lanes: hand the cached pool blocks of this rank's back first
            ln.hal.trim()
        psess, perr = None, None
        try:
            psess = Session(run.desc, devices=(run.device,), lanes_per_device=inflight)
            psess.set_witness_source(1, args.preflight_producers)
            psess.set_resident_code(not args.recompute_code)
            psess.prove([bsegs[0]] * (3 * inflight) + [bsegs[-1]] * inflight)   # warm-up: three seals per lane, every size
        except HalError as e:                  # (ranks sharing ONE GPU in a dry run can run out of HBM here)
            perr = str(e)
        if ctl.min(0.0 if perr else 1.0) < 1.0:                       # every rank skips the leg together
            if psess is not None:
                psess.close()
            return {"error": perr or "another rank could not set the leg up"}
        ptimes = []
        for _ in range(REPS):
            run.device_sync(lanes)
            ctl.barrier()
            tp0 = time.perf_counter()
            pcomp, _, pst = psess.prove([bsegs[i] for i in bmine])
            ctl.barrier()
            ptimes.append(ctl.max(time.perf_counter() - tp0))
        for r in pcomp.segments:
            r.verify(run.desc, broots[r.po2])
        tpv = ctl.sum([pst["preflight_cpu_s_sum"], pst["trace_bytes"], float(len(bmine)), pst["witgen_s_sum"]])
        psess.close()
        dtp = sum(ptimes) / len(ptimes)
        prates = [S / t for t in ptimes]
        pspread = (max(prates) - min(prates)) / (S / dtp)
        return {"segments": S, "wall_clock_s": dtp, "segments_per_s": S / dtp, "min": min(prates), "max": max(prates), "reps": len(ptimes),
                "spread_pct": 100.0 * pspread, **({"unstable": f"the repetitions differ by {100.0 * pspread:.1f} % (> 8 %)"} if pspread > 0.08 else {}),
                "host_preflight_cpu_ms_per_segment": 1e3 * tpv[0] / max(1.0, tpv[2]),
                "pcie_bytes_per_segment": tpv[1] / max(1.0, tpv[2]),
                "full_trace_bytes_per_segment": 4.0 * (wc + wd) * n,
                "upload_and_row_fill_ms_per_segment": 1e3 * tpv[3] / max(1.0, tpv[2]),
                "producer_threads_per_gpu": inflight * (args.preflight_producers or 2), "sealing_lanes_per_gpu": inflight,
                # the Amdahl term of an N-GPU node: host cores the preflight producers keep busy = N x segments/s per GPU x CPU seconds per segment
                "host_cores_needed": (S / dtp) * (tpv[0] / max(1.0, tpv[2])),
                "host_cores_needed_8_gpus": 8.0 * (S / dtp / world) * (tpv[0] / max(1.0, tpv[2])),
                "verified_after_clock": int(tpv[2]),
                "note": "the preflight is a sequential per-cycle machine (SYN-VM: 8 registers, 64 instructions, 1 KiB words of RAM) on host "
                        "threads; its 16-byte-per-cycle records are the ONLY witness input that crosses PCIe; the GPU expands them (one lane per "
                        "cycle), scans the running sum and scatters the preloaded RAM image; a DIFFERENT witness than the closed-form "
                        "generator's, same circuit, seals byte-identical to the CPU oracle's (tests/test_witness_gpu.py)"}
    if block_ok and not args.no_preflight_leg and args.circuit == "syn_a":
        block["host_preflight_pipeline"] = secondary(run, "host_preflight_pipeline", preflight_leg)

    # ... and, on one GPU, that block's receipts folded to ONE root receipt (BASELINE config 5 in small): lift2 / join3 / join programs
    # of the RECURSION circuit, every node verifies its child seals in-circuit.  Afterwards the root is verified the way a holder
    # would: one seal + the claim tree recomputed on the host from the leaf claims.  (--with-p2-join adds round 3's cheap join tree.)
    def recursive_leg():
        from zeth_amd.host import receipt_claim
        bsegs, broots, brec = bstate["bsegs"], bstate["broots"], bstate["brec"]
        rlanes = fold_lanes(run, lanes)
        prep = recursive_prepare(run, rlanes, broots, brec[0])
        rroot, rstats = recursive_fold(run, rlanes, [brec[i] for i in range(S)])
        rstats.update(prep)
        rstats["in_flight"] = len(rlanes)
        t_v = time.perf_counter()
        rroot.verify(lanes[0].rec.allowed_roots(), [receipt_claim(brec[i], run.desc, broots[bsegs[i].po2]) for i in range(S)])
        rstats["root_verify_s"] = time.perf_counter() - t_v
        rstats["root_verified_against_leaf_claims"] = True
        rstats["block_plus_fold_s"] = block["wall_clock_s"] + rstats["fold_s"]
        return rstats

    def p2_join_leg():
        import threading
        from zeth_amd.circuits import p2_join
        from zeth_amd.host import SuccinctReceipt, join_schedule, join_segment, node_claim, receipt_claim
        bsegs, broots, brec = bstate["bsegs"], bstate["broots"], bstate["brec"]
        join_desc = p2_join.p2_join_circuit()
        for ln in lanes:
            ln.join_prover = SegmentProver(ln.hal, join_desc)
            ln.join_prover.prove_segment(Segment(index=0, po2=args.join_po2, seed=1, noise_seed=BENCH_NOISE, pub=tuple([1] * 16)))
            ln.hal.sync()
        jroot = lanes[0].join_prover.control_root(args.join_po2)
        nodes = [(brec[i], receipt_claim(brec[i], run.desc, broots[bsegs[i].po2])) for i in range(S)]
        n_joins = 0
        run.device_sync(lanes)
        t_j = time.perf_counter()
        for tasks in join_schedule(S, 1):
            jsegs = [join_segment(t, nodes[t.left][1], nodes[t.right][1], args.join_po2, BENCH_NOISE) for t in tasks]
            out_recs = [None] * len(jsegs)
            jidx = WorkIndex()

            def jwork(ln):
                try:
                    while True:
                        k = jidx.take(len(jsegs))
                        if k is None:
                            return
                        out_recs[k] = ln.join_prover.prove_segment(jsegs[k])
                except Exception as e:
                    ln.err = e
            run_lanes(lanes, jwork)
            nxt = [(r, node_claim(r, join_desc, jroot, False)) for r in out_recs]
            if len(nodes) % 2:
                nxt.append(nodes[-1])
            nodes, n_joins = nxt, n_joins + len(jsegs)
        run.device_sync(lanes)
        join_s = time.perf_counter() - t_j
        SuccinctReceipt(root=nodes[0][0], joins=[], leaves=[brec[i] for i in range(S)]).verify(run.desc, join_desc, broots, jroot)
        return {"leaves": S, "joins": n_joins, "join_po2": args.join_po2, "join_tree_s": join_s,
                "block_plus_joins_s": block["wall_clock_s"] + join_s,
                "root_receipt_words": int(nodes[0][0].seal.size), "compact_receipt_verified": True,
                "note": "P2-JOIN: parent claim = Poseidon2 hash_pair(children's claims) constrained in-circuit; the verifier "
                        "needs the root receipt + the leaves only (`--config succinct --join-circuit p2_join` runs S = 1024)"}
    if block_ok and world == 1 and S > 1:
        if args.with_p2_join:
            block["succinct"] = secondary(run, "p2_join", p2_join_leg)
        if not args.no_recursive:
            block["recursive"] = secondary(run, "recursive_fold", recursive_leg)

    if rank != 0:
        return None, []
    # ---------------------------------------------------------------- the line ----
    last = next((ln.last for ln in lanes if ln.last is not None), None)
    value = steps_done / dt
    cfg = config_common(run)
    cfg.update({"workload": (f"single 2^{args.po2}-cycle segment seal per step per GPU, {run.workload}, witness resident in HBM; every group incl. the "
                             f"code (control) group is re-committed per segment as upstream's SegmentProver does (`value` does NOT keep it resident)"),
                "parallelism": f"segments round-robin over {world} GPU(s), no collectives; {inflight} segment(s) in flight per GPU",
                "seal_words": int(last.seal.size) if last is not None else 0,
                "value_recomputes_code_group": True})
    line = {
        "metric": "segments/sec", "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": cfg,
        # wall-clock of one seal call (enqueue .. seal words on the host), mean over the timed seals of this rank; with several
        # seals in flight each one shares the GPU, so this is latency under load, not 1/value
        "seal_wall_clock_s": sum(seal_times) / max(1, len(seal_times)),
        # ... and of one seal with the GPU to itself (inflight 1: the same thing as seal_wall_clock_s)
        "seal_wall_clock_unloaded_s": unloaded_seal_s if unloaded_seal_s is not None else sum(seal_times) / max(1, len(seal_times)),
        "timed_region_s_rank0": my_dt,
    }
    if run.failed_ranks:
        # a rank died: the line stands for the survivors (their steps over the slowest survivor's time) and says so
        line["failed_ranks"] = sorted(run.failed_ranks)
        line["failed_ranks_detail"] = [run.failed_ranks[r] for r in sorted(run.failed_ranks)]
        line["ranks_reporting"] = headline_ranks
    # witness generation (synthetic, on the device) is reported separately.  ONE meaning in every config: the MEAN
    # per segment measured INSIDE a clock with the other lanes sealing (here: the block leg's).
    if block_ok:
        line["witgen_ms_per_segment"] = block["witgen_ms_per_segment"]
    else:
        line["witgen_ms_per_segment_idle_gpu"] = 1e3 * sum(t for ln in lanes for t in ln.witgen_s) / max(1, sum(len(ln.witgen_s) for ln in lanes))
    if isinstance(certify, dict):
        if "timed_seals_verified" in certify:
            line.update(timed_seals_verified=certify["timed_seals_verified"], seal_matches_golden=certify["seal_matches_golden"])
        line["certify"] = certify
    if block is not None:
        line["block"] = block
        if block_ok:     # the strong-scaling figure (BASELINE's metric is a block's seal wall-clock): total work fixed at S segments
            line["block_wall_clock_s"] = block["wall_clock_s"]
            line["block_segments_per_s"] = block["segments_per_s"]
    if pcie is not None:
        line["pcie_inclusive"] = pcie
    if heavy is not None:
        line["syn_heavy"] = heavy
    if resident is not None:
        line["code_group_resident"] = resident
    alg = seal_algorithmic_bytes(wa, wc, wd, len(run.circ.taps), len(run.circ.combos), n)
    line["seal_roofline"] = {"alg_bytes": alg, "achieved": alg / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS}
    flatten_secondary(line)
    # roofline{} (its HBM traffic and VALU issue are measured by child runs under rocprofv3) and the CPU baseline are taken by rank 0
    # AFTER the other ranks are gone: the host's cores and rank 0's GPU are idle then.  With a failed rank the line goes out at once
    # (the failed rank is waiting for it): HIP-event roofline only, no child runs.
    after = []
    if prof:
        from .roofline import add_roofline
        live = not run.failed_ranks
        for ln in lanes:                       # the child runs bring their own context: hand this rank's cached pool blocks back first
            after.append(ln.hal.trim)
        after.append(lambda: add_roofline(line, prof, ref, args, inflight, (wa, wc, wd), n, run.device, live=live))
        if live:
            from .roofline import add_heavy_valu
            after.append(lambda: add_heavy_valu(line, args, n, run.device))
    if not args.no_cpu_baseline and not run.failed_ranks:
        def _cpu():
            from .cpu_baseline import cpu_baseline
            try:
                line["cpu_baseline"] = cpu_baseline(run.desc, args.circuit, run.cpus_before, full_host=args.cpu_full_host, all_cores=not args.no_cpu_all_cores)
            except Exception as e:       # the baseline is a reported number, never a dependency of the product path
                line["cpu_baseline"] = {"error": repr(e)}
        after.append(_cpu)
    return line, after


MAX_CONFIG_KEY = 32             # round 5: the driver's record cut `block_recompute_code_group_segments_per_` at 40 characters


def flatten_secondary(line: dict) -> None:
    """The secondary measurements of this run once more as SCALAR keys of `config` (a record that keeps only the contract's keys
    and drops nested objects then still holds them).  Nothing here is `value`."""
    cfg = line["config"]

    def put(key, obj, field, digits=3):
        if isinstance(obj, dict) and isinstance(obj.get(field), (int, float)):
            cfg[key] = round(obj[field], digits)
    put("syn_heavy_segments_per_s", line.get("syn_heavy"), "segments_per_s")
    put("syn_heavy_min_segments_per_s", line.get("syn_heavy"), "min")
    put("syn_heavy_max_segments_per_s", line.get("syn_heavy"), "max")
    put("syn_heavy_ms_per_step", line.get("syn_heavy"), "ms_per_step")
    put("resident_code_segments_per_s", line.get("code_group_resident"), "segments_per_s")
    blk = line.get("block")
    if isinstance(blk, dict):
        put("block_segments", blk, "segments", 0)
        put("block_segments_per_s", blk, "segments_per_s")
        put("block_wall_clock_s", blk, "wall_clock_s", 4)
        put("block_recompute_segments_per_s", blk.get("recompute_code_group"), "segments_per_s")
        pre = blk.get("host_preflight_pipeline")
        put("preflight_segments_per_s", pre, "segments_per_s")
        put("preflight_host_cpu_ms_per_seg", pre, "host_preflight_cpu_ms_per_segment", 2)
        put("preflight_host_cores_needed", pre, "host_cores_needed", 2)
        if isinstance(pre, dict) and isinstance(pre.get("pcie_bytes_per_segment"), (int, float)):
            cfg["preflight_pcie_MB_per_segment"] = round(pre["pcie_bytes_per_segment"] / 1e6, 2)
        put("block_fold_to_one_receipt_s", blk.get("recursive"), "fold_s", 4)
    put("seal_wall_clock_unloaded_s", line, "seal_wall_clock_unloaded_s", 5)
    put("seal_hbm_frac", line.get("seal_roofline"), "frac", 4)
    too_long = [k for k in cfg if len(k) > MAX_CONFIG_KEY]
    assert not too_long, f"config keys longer than {MAX_CONFIG_KEY} characters (a record that truncates keys loses them): {too_long}"


def run_pmc_child(run: Run) -> None:
    """What roofline.py's rocprofv3 --pmc child runs execute: one lane, one resident witness, one warm seal, ONE more seal — no
    torch, no control plane, no line (the counters are read from rocprofv3's CSV)."""
    from zeth_amd.prover import Segment
    ln = run.lane()
    seg = Segment(index=0, po2=run.args.po2, seed=BASE_SEED, noise_seed=BENCH_NOISE)
    code, data, out = ln.prover.witgen(seg)
    for _ in range(2):
        ln.prover.seal(seg, code, data, out)
        ln.hal.sync()
