"""--config succinct (BASELINE config 5): S leaf segments sealed AND folded to one root receipt.

Default: ONE native call per rank — zkh_session_prove(join_tree = 2) seals this rank's segments and folds them, by default as one
pipeline (a lift2 / join is proven the moment its children exist, on the fold lanes while the sealing lanes are busy), with
--fold phased as two phases.  No Python in the loop; this is what a Rust shim's Prover::prove would call once per session
(/root/reference/crates/host/src/lib.rs:137).  --executor python / --join-circuit p2_join: round 3's Python-orchestrated trees
(block.py run_python_orchestrated)."""
from __future__ import annotations

import os
import time

from .block import run_python_orchestrated
from .common import BENCH_NOISE, Run, block_segments, config_common, top_proofs


def run_succinct(run: Run):
    import numpy as np
    from zeth_amd import recursion as zrec
    from zeth_amd.host import Session, partition_round_robin, receipt_claim
    from zeth_amd.prover import Segment
    args, ctl, rank, world, inflight = run.args, run.ctl, run.rank, run.world, run.inflight
    desc = run.desc
    S = args.segments or 1024
    recursive = args.join_circuit == "recursion"
    segs = block_segments(run, S)
    mine = partition_round_robin(S, world, rank)
    if recursive and world > 1:
        # every rank folds a contiguous, equal range of leaves (zeth_amd/recursion.py fold_plan), and rank 0 folds the `world`
        # local roots by the same rule: N range trees under one top tree (the verifier: fold_leaf_claims(leaves, ranks = N))
        try:
            mine = list(zrec.aligned_range(S, world, rank))
        except ValueError as e:
            raise SystemExit(f"bench: --join-circuit recursion: {e}")
    if not (recursive and args.executor == "native"):
        return run_python_orchestrated(run, segs, mine, S, succinct=True, recursive=recursive), []

    os.environ["ZKH_FOLD_LANES"] = str(max(args.fold_inflight, inflight))
    probe = run.lane()                                 # control roots + (rank 0, N > 1) the top joins
    probe.prover.prove_segment(Segment(index=0, po2=args.po2, seed=1, noise_seed=BENCH_NOISE))
    roots = {p: probe.prover.control_root(p) for p in sorted({s.po2 for s in segs})}
    t_b = time.perf_counter()
    programs = zrec.build_programs(desc, roots, fused_pairs=not args.no_fused_lift, ternary=not args.no_join3)
    build_s = time.perf_counter() - t_b
    t_b = time.perf_counter()
    sess = Session(desc, devices=(run.device,), lanes_per_device=inflight)
    sess.set_recursion(programs)
    sess.set_streamed_fold(args.fold == "streamed")
    sess.set_resident_code(not args.recompute_code)
    if args.witness == "preflight":                    # upstream's whole shape: host preflight -> row fill -> seal -> join-as-you-go
        sess.set_witness_source(1, args.preflight_producers)
    load_s = time.perf_counter() - t_b
    # warm-up: a short session of the same shape (every segment size, every program kind, pools, clocks)
    wsegs = [segs[0]] * (2 * max(args.fold_inflight, inflight)) + [segs[0], segs[-1]]
    for _ in range(max(1, args.warmup)):
        sess.prove(wsegs, join_tree=2, join_noise_seed=BENCH_NOISE)
    if run.distributed and rank == 0:
        probe.rec = zrec.Recursion(probe.hal, programs)
    run.device_sync([probe])
    ctl.barrier()
    t0 = time.perf_counter()
    comp, local_root, st = sess.prove([segs[i] for i in mine], join_tree=2, join_noise_seed=BENCH_NOISE)
    kinds = [k for k, _ in programs]
    rp = st["root_program"]
    local = zrec.RecReceipt(local_root.seal, local_root.po2, rp, None, len(mine), st["root_core"], st["root_pre"], st["root_post"])
    tops, root = [local], local
    top_s = 0.0
    if run.distributed:
        got = ctl.gather(local, dst=0)
        if rank == 0:
            tops = [got[r] for r in sorted(got)]
            t_top = time.perf_counter()
            for t in tops:
                t.control_root = probe.rec.programs[t.program].root
            root = probe.rec.fold(tops, BENCH_NOISE)
            probe.hal.sync()
            top_s = time.perf_counter() - t_top
    ctl.barrier()
    dt = ctl.max(time.perf_counter() - t0)
    t_leaves, fold_tail = ctl.max(st["leaves_s"]), ctl.max(st["fold_tail_s"])
    tt = ctl.sum([st["fold_busy_s_sum"], float(st["n_retries"]), st["witgen_s_sum"], float(len(mine)), st["preflight_cpu_s_sum"], st["trace_bytes"]])
    # ---- after the clock: every leaf seal through the host verifier, the root seal, and the claim tree ----
    verified, follows = 0, None
    t_v = time.perf_counter()
    if not args.no_verify:
        for r in comp.segments:
            r.verify(desc, roots[r.po2])
            verified += 1
    verify_s = time.perf_counter() - t_v
    mine_claims = {i: receipt_claim(r, desc, roots[r.po2]) for i, r in zip(mine, comp.segments)} if not args.no_verify else {}
    parts = ctl.gather(mine_claims, dst=0)
    rstats = None
    if rank == 0:
        root_verify_s = None
        if not args.no_verify:
            if not run.distributed:
                probe.rec = zrec.Recursion(probe.hal, programs)       # only for the allowed set (host data), after the clock
                root.control_root = probe.rec.programs[root.program].root
            t_rv = time.perf_counter()
            root.verify(probe.rec.allowed_roots())                    # ONE seal; the claim tree is checked against the leaves below
            root_verify_s = time.perf_counter() - t_rv
            verified += 1
            allc = {k: v for part in parts.values() for k, v in part.items()}
            follows = bool(np.array_equal(root.seal[:8], zrec.fold_leaf_claims([allc[i] for i in range(S)], ranks=world)))
            if not follows:
                raise SystemExit("bench: the root receipt's output is not the claim tree of the leaves")
        n_fused = sum(1 for k in range(len(mine) // 2) if ("lift2", segs[mine[2 * k]].po2, segs[mine[2 * k + 1]].po2) in kinds)
        all_fused = n_fused == len(mine) // 2 and len(mine) > 1
        rstats = {"executor": "native: one zkh_session_prove(join_tree = 2) call per rank (csrc/session.hip), no Python in the loop",
                  "witness": ("host preflight: a sequential per-cycle machine on producer threads ahead of the seals, 16 bytes per cycle over PCIe, row fill "
                              "on the GPU" if args.witness == "preflight" else "closed-form generator on the device"),
                  "host_preflight_cpu_ms_per_segment": 1e3 * tt[4] / max(1.0, tt[3]) if args.witness == "preflight" else None,
                  "pcie_bytes_per_segment": tt[5] / max(1.0, tt[3]) if args.witness == "preflight" else None,
                  "fold": args.fold, "streamed_fold": st["streamed_fold"], "code_group": "recomputed per segment" if args.recompute_code else "resident per lane and size",
                  "bottom_level_proofs": st["n_lifts"] * world, "fused_lift2": (len(mine) // 2) * world if all_fused else 0,
                  "joins": st["n_joins"] * world + top_proofs(tops, kinds), "proofs": (st["n_lifts"] + st["n_joins"]) * world + top_proofs(tops, kinds),
                  "leaves_s": t_leaves, "fold_tail_s": fold_tail, "fold_busy_lane_s": tt[0], "top_joins": top_proofs(tops, kinds), "top_joins_s": top_s,
                  "segment_retries": int(tt[1]), "program_build_s": build_s, "program_load_s_all_lanes": load_s,
                  "in_flight": {"sealing_lanes": inflight, "fold_lanes": max(args.fold_inflight, inflight)},
                  "root_verify_s": root_verify_s,
                  "note": "every lift2 runs the STARK verifier on two segment seals and every join on both child seals INSIDE the RECURSION "
                          "circuit; fold_tail_s = last segment sealed -> root receipt (the part of the fold the leaves did not hide)"}
    cnt = ctl.sum([float(verified)])[0]
    if rank != 0:
        return None, []
    cfg = config_common(run)
    cfg.update({"workload": (f"block + fold to ONE succinct receipt: {S} distinct 2^{args.po2}-cycle segments (last one 2^{segs[-1].po2}), {run.workload}; "
                             f"witness generation inside the clock; {rstats['proofs']} proofs of the RECURSION circuit, every node runs the STARK "
                             f"verifier on its child seal(s) in-circuit; fold {args.fold}"),
                "segments": S,
                "parallelism": (f"{world} GPU(s): every rank seals AND folds its own contiguous, equal range of segments "
                                f"(a deviation from round-robin: a rank folds what it sealed), rank 0 folds "
                                f"the {world} local roots gathered over the control plane by the same plan; no data-path collective; {inflight} sealing + "
                                f"{max(args.fold_inflight, inflight) - inflight} fold-only lanes per GPU"),
                "join_circuit": "recursion (lift2 + join programs, in-circuit verification of every child seal)",
                "fold_proofs": rstats["proofs"], "fold_tail_s": round(fold_tail, 4), "leaves_s": round(t_leaves, 4)})
    line = {
        "metric": "segments/sec", "value": S / dt, "unit": "segments/s", "n_gpus": world, "steps": S,
        "warmup": max(1, args.warmup), "ms_per_step": 1e3 * dt / S, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": cfg,
        "block_wall_clock_s": dt, "leaf_phase_s": t_leaves, "join_phase_s": dt - t_leaves,
        "witgen_ms_per_segment": 1e3 * tt[2] / max(1.0, tt[3]),
        "verified_after_clock": int(cnt), "verify_s_rank0": verify_s,
        "root_receipt_words": int(root.seal.size), "succinct_root_follows_from_leaf_claims": follows,
        "recursion": rstats,
    }
    return line, []
