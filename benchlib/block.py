"""--config block (BASELINE configs 3 / 4): one block = S DISTINCT segments (seeds base + i, the last one a po2-18 tail), handed out
round-robin over the ranks and through a shared work index inside a rank; witness generation runs inside the clock (reported
separately), every seal is verified on the host after the clock stops; value = S / wall ("scaling": "strong").

Also here: the chained block (--chained: SYN-C segments, claim continuity) and the Python-orchestrated join trees that
--config succinct falls back to (--executor python / --join-circuit p2_join); the native one-call executor is succinct.py."""
from __future__ import annotations

import time

from .common import (BENCH_NOISE, Run, WorkIndex, block_segments, config_common, fold_lanes, recursive_fold, recursive_prepare, run_lanes,
                     seal_block, top_proofs)


def run_chained(run: Run, segs, mine, S):
    """A CHAINED block (claim continuity): SYN-C segments — SYN-A with the pre-state as public input, out = (post, 0,
    0, 0, pre) — through the native session executor.  Rank 0 runs the executor's pass for the WHOLE block (one launch: every
    segment's contribution to the running state; before the clock, as upstream's executor runs before any proving), the pre-states
    travel with the segment list, every rank proves its round-robin share independently, and after the clock the gathered composite
    must pass pre == prev.post (`CompositeReceipt::verify_integrity`)."""
    from zeth_amd.circuits import syn_air as _sa
    from zeth_amd.hal import HipHal
    from zeth_amd.host import CompositeReceipt, Session, chain_segments
    from zeth_amd.prover import SegmentProver
    args, ctl, rank, world, inflight = run.args, run.ctl, run.rank, run.world, run.inflight
    cdesc = _sa.syn_chain()
    cprobe = SegmentProver(HipHal(run.device), cdesc)
    croots = {p: cprobe.control_root(p) for p in sorted({sg.po2 for sg in segs})}
    t_e = time.perf_counter()
    csegs = chain_segments(segs, cprobe.chain_contribution, initial_state=1) if rank == 0 else None
    executor_s = time.perf_counter() - t_e
    if run.distributed:
        csegs = ctl.broadcast(csegs, src=0)
    sess = Session(cdesc, devices=(run.device,), lanes_per_device=inflight)
    sess.set_resident_code(not args.recompute_code)
    sess.prove([csegs[0]] * inflight + [csegs[-1]])            # warm-up (the library treats `pub` as given: not chained mode)
    run.device_sync([cprobe])
    ctl.barrier()
    t0 = time.perf_counter()
    comp, _, st = sess.prove([csegs[i] for i in mine])
    ctl.barrier()
    dt = time.perf_counter() - t0
    t_v = time.perf_counter()
    for r in comp.segments:
        r.verify(cdesc, croots[r.po2])
    verify_s = time.perf_counter() - t_v
    for r, i in zip(comp.segments, mine):
        r.index = i
    parts = ctl.gather(comp.segments, dst=0)
    dt = ctl.max(dt)
    if rank != 0:
        return None
    whole = CompositeReceipt(sorted((r for part in parts.values() for r in part), key=lambda r: r.index))
    whole.verify_integrity(chained=True, initial_state=1)           # raises if the session is not continuous
    cfg = config_common(run)
    cfg.update({"workload": f"one CHAINED block: {S} distinct 2^{args.po2}-cycle SYN-C segments (last one 2^{segs[-1].po2}); every segment's "
                            f"pre-state is its predecessor's post-state (out = post, 0, 0, 0, pre), fixed by the executor's pass before the clock; "
                            f"witness generation inside the clock", "circuit": "syn_chain", "segments": S,
                "parallelism": f"segments round-robin over {world} GPU(s), native session executor per rank, no data-path collective; {inflight} seal(s) in flight per GPU"})
    return {
        "metric": "segments/sec", "value": S / dt, "unit": "segments/s", "n_gpus": world, "steps": S, "warmup": 1,
        "ms_per_step": 1e3 * dt / S, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": cfg,
        "block_wall_clock_s": dt, "verified_after_clock": len(whole.segments), "verify_s_rank0": verify_s,
        "continuity": {"checked": "pre == prev.post over all segments (CompositeReceipt.verify_integrity), first pre == the initial state",
                       "executor_pass_s": executor_s, "initial_state": 1, "final_state_word": whole.final_state()},
    }


def run_python_orchestrated(run: Run, segs, mine, S, succinct: bool, recursive: bool):
    """block, and the Python-orchestrated forms of succinct (round 3): the committed code group of each segment size stays resident
    per lane (what the session executor does by default; byte-identical seals); --recompute-code re-commits it per segment like
    upstream's SegmentProver"""
    import numpy as np
    from zeth_amd.circuits import p2_join
    from zeth_amd.host import JoinExecutor, fold_claims, node_claim, receipt_claim
    from zeth_amd.prover import Segment
    args, ctl, rank, world, inflight = run.args, run.ctl, run.rank, run.world, run.inflight
    desc = run.desc
    join_desc = p2_join.p2_join_circuit()     # joins hash their children's claims in-circuit (Poseidon2 unrolled over trace rows)
    lanes = [run.lane(with_join=succinct and not recursive, resident=not args.recompute_code, join_desc=join_desc) for _ in range(inflight)]
    # warm-up: one seal of every size per lane (clocks, pools, code objects, resident groups), plus the control roots the verifier needs
    for ln in lanes:
        for _ in range(max(1, args.warmup)):
            for p2 in sorted({sg.po2 for sg in segs}, reverse=True):
                ln.prover.prove_segment(Segment(index=0, po2=p2, seed=1, noise_seed=BENCH_NOISE))
        if succinct and not recursive:
            ln.join_prover.prove_segment(Segment(index=0, po2=args.join_po2, seed=1, noise_seed=BENCH_NOISE, pub=tuple([1] * 16)))
        ln.hal.sync()
    roots = {p: lanes[0].prover.control_root(p) for p in sorted({s.po2 for s in segs})}
    join_root = lanes[0].join_prover.control_root(args.join_po2) if succinct and not recursive else None
    rstats, rlanes = None, None
    if recursive:
        rlanes = fold_lanes(run, lanes)
        rstats = recursive_prepare(run, rlanes, roots, lanes[0].prover.prove_segment(Segment(index=0, po2=args.po2, seed=1, noise_seed=BENCH_NOISE)))
        rstats["in_flight"] = len(rlanes)
    receipts, t0, wit_s, seal_s = seal_block(run, lanes, segs, mine)
    t_leaves = time.perf_counter() - t0
    joins_done, root = {}, None
    if recursive:
        local_root, st = recursive_fold(run, rlanes, [receipts[i] for i in mine])
        rstats.update(st)
        tops = ctl.gather(local_root, dst=0)
        if rank == 0:
            tops = [tops[r] for r in sorted(tops)]
            t_top = time.perf_counter()
            root = lanes[0].rec.fold(tops, BENCH_NOISE)
            lanes[0].hal.sync()
            rstats["top_joins"] = top_proofs(tops, lanes[0].rec.kinds)
            rstats["top_joins_s"] = time.perf_counter() - t_top
    elif succinct:
        # join tree: tasks of one level are independent -> spread over the lanes of this rank
        def claim_of(r, is_leaf):
            return node_claim(r, desc if is_leaf else join_desc, roots[r.po2] if is_leaf else join_root, is_leaf)

        def prove_joins_parallel(tasks_segs):
            """prove a list of join Segments on this rank's lanes concurrently -> receipts in the same order"""
            out = [None] * len(tasks_segs)
            jidx = WorkIndex()

            def work(ln):
                try:
                    while True:
                        k = jidx.take(len(tasks_segs))
                        if k is None:
                            return
                        out[k] = ln.join_prover.prove_segment(tasks_segs[k])
                except Exception as e:
                    ln.err = e
            run_lanes(lanes, work)
            return out

        class BatchedExecutor(JoinExecutor):
            """JoinExecutor whose per-level local joins run concurrently on the lanes (same schedule, same results)."""
            def run(self, n_leaves, local_leaves):
                from zeth_amd.host import join_schedule, join_segment
                nodes = {i: (r, True) for i, r in local_leaves.items()}
                n_nodes, done = n_leaves, {}
                for tasks in join_schedule(n_leaves, self.world_size):
                    right = {}
                    for t in tasks:
                        if t.right_owner == t.device:
                            continue
                        if self.rank == t.right_owner:
                            self._send(nodes[t.right], t.device)
                        elif self.rank == t.device:
                            right[t.index] = self._recv(t.right_owner)
                    local = [t for t in tasks if t.device == self.rank]
                    jsegs = []
                    for t in local:
                        l_rec, l_leaf = nodes[t.left]
                        r_rec, r_leaf = right[t.index] if t.index in right else nodes[t.right]
                        jsegs.append(join_segment(t, self.claim_of(l_rec, l_leaf), self.claim_of(r_rec, r_leaf), self.join_po2, self.noise_seed))
                    recs = prove_joins_parallel(jsegs)
                    nxt2 = {}
                    for t, j in zip(local, recs):
                        done[(t.level, t.index)] = j
                        nxt2[t.index] = (j, False)
                    if n_nodes % 2 and (n_nodes - 1) in nodes:
                        nxt2[n_nodes // 2] = nodes[n_nodes - 1]
                    nodes, n_nodes = nxt2, (n_nodes + 1) // 2
                return done, (nodes.get(0, (None, False))[0] if n_nodes == 1 else None)

        ex = BatchedExecutor(None, claim_of, rank, world, join_po2=args.join_po2, noise_seed=BENCH_NOISE,
                             send=ctl.send if run.distributed else None, recv=ctl.recv if run.distributed else None)
        joins_done, root = ex.run(S, {i: receipts[i] for i in mine})
        run.device_sync(lanes)
    ctl.barrier()
    dt = ctl.max(time.perf_counter() - t0)
    t_leaves = ctl.max(t_leaves)
    # ---- after the clock: verify every seal this rank produced (cli.rs:103 analogue) ----
    verified = 0
    t_v = time.perf_counter()
    if not args.no_verify:
        for i in mine:
            receipts[i].verify(desc, roots[segs[i].po2])
            verified += 1
        for j in joins_done.values():
            j.verify(join_desc, join_root)
            verified += 1
    verify_s = time.perf_counter() - t_v
    # succinct: what a holder of the COMPACT receipt (root + leaves, joins dropped) checks — the claim tree over the leaf
    # claims, recomputed on the host with hash_pair, must end in the root receipt's public output
    follows = None
    if recursive and not args.no_verify and rank == 0:
        t_rv = time.perf_counter()
        root.verify(lanes[0].rec.allowed_roots())            # ONE seal; the claim tree is checked against the leaves below
        rstats["root_verify_s"] = time.perf_counter() - t_rv
        verified += 1
    if succinct and not args.no_verify:
        mine_claims = {i: receipt_claim(receipts[i], desc, roots[segs[i].po2]) for i in mine}
        parts = ctl.gather(mine_claims, dst=0)
        if rank == 0 and root is not None and (S > 1 or recursive):
            allc = {k: v for part in parts.values() for k, v in part.items()}
            if recursive:
                from zeth_amd.recursion import fold_leaf_claims
                follows = bool(np.array_equal(root.seal[:8], fold_leaf_claims([allc[i] for i in range(S)], ranks=world)))
            else:
                follows = bool(np.array_equal(root.seal[:8], fold_claims([allc[i] for i in range(S)])))
            if not follows:
                raise SystemExit("bench: the root receipt's output is not the claim tree of the leaves")
    counts = ctl.sum([float(verified), float(len(joins_done))])
    if rank != 0:
        return None
    n_joins = int(counts[1])
    cfg = config_common(run)
    cfg.update({"workload": (f"{'block + join tree to one succinct receipt' if succinct else 'one block'}: {S} distinct "
                             f"2^{args.po2}-cycle segments (last one 2^{segs[-1].po2}), {run.workload}; witness generation inside the clock"
                             + (f"; {rstats['proofs']} proofs of the RECURSION circuit ({rstats['fused_lift2']} lift2 = lift + lift + join fused, {rstats['lifts']} lifts, {rstats['joins']} joins): every node runs the STARK verifier on its child seal(s) in-circuit" if recursive else
                                f"; {n_joins} P2-JOIN joins at po2 {args.join_po2} (parent claim = Poseidon2 hash_pair of the children's, proven in-circuit)" if succinct else "")),
                "segments": S,
                "parallelism": f"segments round-robin over {world} GPU(s) + shared work index inside a rank, no data-path collective; "
                               f"{inflight} seal(s) in flight per GPU" + ("; every rank folds its own aligned range of leaves, rank 0 joins the local roots (gathered over the control plane)" if recursive else
                                                                          "; joins on the rank of their left child, right child over the control plane" if succinct else ""),
                "code_group": "recomputed per segment" if args.recompute_code else "resident per lane and size (byte-identical seals)"})
    line = {
        "metric": "segments/sec", "value": S / dt, "unit": "segments/s", "n_gpus": world, "steps": S,
        "warmup": max(1, args.warmup), "ms_per_step": 1e3 * dt / S, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": cfg,
        "block_wall_clock_s": dt, "leaf_phase_s": t_leaves, "join_phase_s": dt - t_leaves if succinct else None,
        "witgen_ms_per_segment": 1e3 * sum(wit_s) / max(1, len(wit_s)),      # mean, in-clock (rank 0's segments)
        "seal_call_ms_mean": 1e3 * sum(seal_s) / max(1, len(seal_s)),
        "verified_after_clock": int(counts[0]), "verify_s_rank0": verify_s,
        "root_receipt_words": int(root.seal.size) if root is not None else None,
        "succinct_root_follows_from_leaf_claims": follows,
    }
    if recursive:
        line["recursion"] = rstats
        line["config"]["join_circuit"] = "recursion (lift + join programs, in-circuit verification of every child seal)"
    elif succinct:
        line["config"]["join_circuit"] = "p2_join"
    return line


def run_block(run: Run):
    from zeth_amd.host import partition_round_robin
    args = run.args
    S = args.segments or 256
    segs = block_segments(run, S)
    mine = partition_round_robin(S, run.world, run.rank)
    if args.chained:
        return run_chained(run, segs, mine, S), []
    return run_python_orchestrated(run, segs, mine, S, succinct=False, recursive=False), []
