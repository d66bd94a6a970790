"""The run context of one bench.py rank and the helpers the legs share."""
from __future__ import annotations

import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
N_SIMDS = 1024                  # 256 CUs x 4 SIMDs
PO2 = 20
TAIL_PO2 = 18                   # the short last segment of a block
BASE_SEED = 0x5EED0000
BENCH_NOISE = 0x2E80            # fixed blinding seed: bench seals must be reproducible run to run (product default: OS RNG)


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


class Fault(RuntimeError):
    """an injected failure (tests): ZKH_BENCH_FAULT_RANK=r [ZKH_BENCH_FAULT_LEG=headline|block|hard|hang]"""


def maybe_fault(run, leg: str) -> None:
    """Test hook of the control plane (never set in a measurement): the chosen rank fails when it enters `leg` — by an exception
    (reported through the control plane) or, leg "hard", by dying on the spot at the headline (the launcher's case)."""
    want = os.environ.get("ZKH_BENCH_FAULT_RANK", "")
    if want == "" or int(want) != run.rank:
        return
    where = os.environ.get("ZKH_BENCH_FAULT_LEG", "headline")
    if where == "hard" and leg == "headline":
        os._exit(17)
    if where == "hang" and leg == "headline":
        time.sleep(3600)
    if where == leg:
        raise Fault(f"injected fault on rank {run.rank} in the {leg} leg")


class Run:
    """Everything a leg needs: the arguments, this rank's place (rank / world / device / host placement), the control plane, the
    circuit.  One process per GPU; `ctl` is the only thing the ranks share."""

    def __init__(self, args, ctl, rank: int, local_rank: int, world: int):
        self.args, self.ctl, self.rank, self.local_rank, self.world = args, ctl, rank, local_rank, world
        self.distributed = world > 1
        self.device = local_rank
        self.torch = None
        self.placement = {"numa_node": -1, "cpus": 0}
        self.cpus_before = None
        self.rccl, self.rccl_hung = None, False
        self.devices, self.devices_distinct = None, None
        self.identity = {"rank": rank, "pci_bus_id": None, "uuid": None, "numa_node": -1}
        self.desc = self.circ = self.workload = None
        self.widths = (0, 0, 0)
        self.n = 1 << args.po2
        self.inflight = max(1, args.inflight)
        self.ranks_per_gpu = 1

    # ---- device + host placement ----
    def bind_device(self) -> None:
        """one GPU per rank; ZKH_SHARE_GPUS=1 lets ranks wrap around the visible devices (dry runs on a 1-GPU box); then this
        rank's threads (and the pinned blocks they allocate) next to its GPU's root port (csrc/topology.hip)"""
        try:
            import torch
            self.torch = torch
            visible = int(torch.cuda.device_count())
        except Exception:
            visible = 0
        if os.environ.get("ZKH_SHARE_GPUS") or getattr(self.args, "allow_shared_gpu", False):
            self.device = self.local_rank % max(1, visible)
            self.ranks_per_gpu = -(-self.world // max(1, visible))
        elif 0 < visible <= self.local_rank:
            # a launcher that narrows HIP_VISIBLE_DEVICES per rank (every rank sees ITS GPU as device 0): follow it instead of failing
            self.device = self.local_rank % visible
        try:
            self.cpus_before = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            self.cpus_before = None
        self.placement = {"numa_node": -1, "cpus": 0, "cpus_before": len(self.cpus_before or [])}
        try:
            from zeth_amd import hal as _zhal
            slot, share = (0, 1) if os.environ.get("ZKH_SHARE_GPUS") or visible < self.world else _zhal.placement_slot(self.device, list(range(self.world)))
            self.placement = _zhal.bind_to_device(self.device, slot, share)
            self.placement.update(slot=slot, share=share, pci_bus_id=_zhal.device_numa_node(self.device)[1])
        except Exception as e:                               # placement is an optimisation, never a dependency
            self.placement["error"] = repr(e)
        from zeth_amd import hal as _zhal
        ident = _zhal.device_identity(self.device)           # (raises without the HIP library / a GPU: there is no fallback)
        self.identity.update(ident, hip_device=self.device, hip_visible_devices=os.environ.get("HIP_VISIBLE_DEVICES"))

    def exchange_devices(self) -> None:
        """every rank learns which physical GPU every rank drives (PCI bus id + device UUID + NUMA node -> `config.devices`), and
        an N > 1 run REFUSES to start unless these are N distinct devices — `--allow-shared-gpu` (ZKH_SHARE_GPUS=1) is the dry
        run of the N-rank shape on fewer GPUs.  (Tests: ZKH_BENCH_FAKE_DEVICES="a,b,c,d" stands in for the identities of a
        `--config dev` run, which touches no GPU.)"""
        fake = os.environ.get("ZKH_BENCH_FAKE_DEVICES")
        if fake:
            ids = fake.split(",")
            self.identity.update(pci_bus_id=ids[self.rank % len(ids)], uuid=ids[self.rank % len(ids)], fake=True)
        views = self.ctl.allgather(self.identity) if self.distributed else {0: self.identity}
        if self.distributed:
            # nobody acts on the view before everybody HAS it: a rank that refuses below declares itself failed on its way out, and a
            # peer still polling inside the allgather would then drop that rank's entry, count one device fewer and start the run
            self.ctl.barrier()
        self.devices = [views[r] for r in sorted(views)]
        keys = [(d.get("uuid") or d.get("pci_bus_id")) for d in self.devices]
        known = [k for k in keys if k]
        self.devices_distinct = len(known) == len(keys) and len(set(known)) == len(keys) if known else None
        if self.distributed and self.devices_distinct is False and not getattr(self.args, "allow_shared_gpu", False):
            shared = {}
            for d, k in zip(self.devices, keys):
                shared.setdefault(k, []).append(d["rank"])
            dup = "; ".join(f"ranks {r} all drive {k}" for k, r in shared.items() if len(r) > 1)
            if self.rank != 0:                 # every rank reaches the same verdict from the same exchanged view: one message is enough
                raise SystemExit(2)
            raise SystemExit(f"bench: --gpus {self.world} but the ranks do not hold {self.world} distinct GPUs ({dup}).  A dry run of the "
                             f"N-rank shape on fewer GPUs needs --allow-shared-gpu; its line then says `devices_distinct: false`.")

    def load_circuit(self) -> None:
        from zeth_amd.circuits import syn_air
        from zeth_amd.circuits.desc import Circuit
        if self.args.circuit == "syn_heavy":
            from zeth_amd.circuits import syn_heavy
            self.desc = syn_heavy.syn_heavy()
        elif self.args.circuit == "syn_huge":
            from zeth_amd.circuits import syn_heavy
            self.desc = syn_heavy.syn_huge()
        else:
            self.desc = syn_air.syn_a()
        self.circ = Circuit.parse(self.desc)
        wa, wc, wd = self.circ.group_sizes
        self.widths = (wa, wc, wd)
        self.workload = (f"{self.args.circuit.upper().replace('_', '-')} circuit (W_code {wc}, W_data {wd}, W_accum {wa}, check 16; "
                         f"{len(self.circ.taps)} taps, {len(self.circ.steps)} constraint steps), poseidon2")

    # ---- both sides of a timed region ----
    def device_sync(self, workers) -> None:
        """every library stream, then torch's device-wide synchronize (torch is only plumbing here; if its own HIP initialisation is
        unavailable the library's syncs already cover all our work)"""
        for wk in workers:
            wk.hal.sync()
        try:
            if self.torch is not None and self.torch.cuda.is_available():
                self.torch.cuda.synchronize(self.device)       # this rank's GPU only (never touch another rank's device)
        except (RuntimeError, AssertionError):
            pass

    def lane(self, with_join=False, resident=False, join_desc=None):
        return Lane(self, with_join, resident, join_desc)


class Lane:
    """One seal in flight: a context (HIP stream) + circuit + prover, driven by one host thread."""

    def __init__(self, run: Run, with_join=False, resident=False, join_desc=None):
        from zeth_amd.hal import HipHal
        from zeth_amd.prover import SegmentProver
        self.hal = HipHal(run.device)                # raises if the HIP library / GPU is missing: no fallback
        self.prover = SegmentProver(self.hal, run.desc, resident_code_group=resident)
        self.join_prover = SegmentProver(self.hal, join_desc) if with_join else None
        self.seal_s, self.witgen_s, self.err = [], [], None
        self.last, self.sealed = None, []


def run_lanes(lanes, fn) -> None:
    threads = [threading.Thread(target=fn, args=(ln,)) for ln in lanes]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for ln in lanes:
        if ln.err is not None:
            err, ln.err = ln.err, None
            raise err


def merged_prof(lanes):
    merged = {}
    for ln in lanes:
        for p in ln.hal.prof_get():
            m = merged.setdefault(p["name"], {"name": p["name"], "calls": 0, "total_ms": 0.0, "alg_bytes": 0.0})
            m["calls"] += p["calls"]; m["total_ms"] += p["total_ms"]; m["alg_bytes"] += p["alg_bytes"]
    return list(merged.values())


def block_segments(run: Run, S: int):
    """S distinct segments of one block: seeds base + i, the last one the short po2-18 tail."""
    from zeth_amd.prover import Segment
    po2 = run.args.po2
    return [Segment(index=i, po2=po2 if i + 1 < S or S == 1 else min(po2, TAIL_PO2), seed=BASE_SEED + i, noise_seed=BENCH_NOISE)
            for i in range(S)]


class WorkIndex:
    """the shared work index of a rank's lanes (work stealing)"""

    def __init__(self):
        self.lock, self.next = threading.Lock(), 0

    def reset(self):
        self.next = 0

    def take(self, limit: int):
        with self.lock:
            k = self.next
            if k >= limit:
                return None
            self.next = k + 1
            return k


def seal_block(run: Run, lanes, segs, mine, prover_of=lambda ln: ln.prover):
    """Seal this rank's share `mine` of the block `segs` on the lanes (shared work index), witness generation inside the
    clock -> ({index: receipt}, t0 of the clock, witgen seconds[], seal-call seconds[]); both device syncs are inside."""
    receipts, wit_s, seal_s = {}, [], []
    lock, idx = threading.Lock(), WorkIndex()

    def seal_leaves(ln):
        try:
            while True:
                k = idx.take(len(mine))
                if k is None:
                    break
                i = mine[k]
                t_w = time.perf_counter()
                pv = prover_of(ln)
                code, data, out = pv.witgen(segs[i])             # inside the clock, reported separately
                ln.hal.sync()                                    # so that t_s - t_w is the witness generator alone
                t_s = time.perf_counter()
                rec = pv.seal(segs[i], code, data, out)
                t_e = time.perf_counter()
                with lock:
                    receipts[i] = rec
                    wit_s.append(t_s - t_w); seal_s.append(t_e - t_s)
            ln.hal.sync()
        except Exception as e:
            ln.err = e

    run.device_sync(lanes)
    run.ctl.barrier()
    t0 = time.perf_counter()
    run_lanes(lanes, seal_leaves)
    run.device_sync(lanes)
    return receipts, t0, wit_s, seal_s


def parity_pins():
    """tests/upstream/parity_pins.json (written by tests/test_upstream_dropbox.py when upstream artefacts have been dropped into
    tests/upstream/): which recalled details have been pinned against the real reference.  None: nothing has been dropped yet —
    parity against the crates is unpinned."""
    import json
    try:
        pins = json.load(open(os.path.join(ROOT, "tests", "upstream", "parity_pins.json")))
        return {k: {"ok": v.get("ok"), "checked_at": v.get("checked_at")} for k, v in pins.items()}
    except (OSError, ValueError):
        return None


def config_common(run: Run) -> dict:
    from zeth_amd.hal import HipHal
    v = HipHal.version()
    return {"po2": run.args.po2, "circuit": run.args.circuit, "inflight_per_gpu": run.inflight, "ranks_per_gpu": run.ranks_per_gpu, "library": v,
            "poseidon2_consts": v.split("poseidon2_consts=")[-1].rstrip(")"), "host_placement_rank0": run.placement,
            "launcher": "ranks", "devices": run.devices, "devices_distinct": run.devices_distinct,
            "distinct_devices": len({(d.get("uuid") or d.get("pci_bus_id")) for d in (run.devices or [])}),
            "rccl_probe": run.rccl, "rccl_world": run.world if run.rccl == "ok" else None, "parity_pins": parity_pins()}


# ---- the recursive fold driven from Python (round 3's two-phase form; the native executor is the default: succinct.py) ----
def top_proofs(tops, kinds) -> int:
    """proofs rank 0 spends on folding the ranks' local roots (zeth_amd/recursion.py fold_plan: pairs, then three at a time)"""
    from zeth_amd.recursion import fold_plan
    if not tops or len(tops) < 2:
        return 0
    po2 = tops[0].po2
    return sum((1 if len(g) == 2 or ("join3", po2, po2, po2) in kinds else 2) for groups in fold_plan(len(tops)) for g in groups if len(g) > 1)


def fold_lanes(run: Run, lanes):
    """the lanes of the fold: the sealing lanes plus extra contexts up to --fold-inflight"""
    return list(lanes) + [run.lane() for _ in range(max(0, run.args.fold_inflight - len(lanes)))]


def recursive_prepare(run: Run, lanes, leaf_roots, warm):
    """build the lift / join programs (host) and load them on every lane (code groups committed, resident), one warm
    lift + join per lane: before any clock, as upstream ships lift / join as precompiled .zkr programs"""
    from zeth_amd import recursion as zrec
    t_b = time.perf_counter()
    programs = zrec.build_programs(run.desc, leaf_roots, ternary=not run.args.no_join3)
    build_s = time.perf_counter() - t_b
    t_b = time.perf_counter()
    for ln in lanes:
        ln.rec = zrec.Recursion(ln.hal, programs)
        w = ln.rec.lift(warm, BENCH_NOISE)
        ln.rec.join(w, w, BENCH_NOISE)
        ln.hal.sync()
    return {"program_build_s": build_s, "program_load_s_all_lanes": time.perf_counter() - t_b}


def recursive_fold(run: Run, lanes, leaves):
    """lift every segment receipt of `leaves` (in order), then join level by level down to ONE receipt - each join runs the
    STARK verifier on both children INSIDE its circuit (zeth_amd/recursion.py).  Lifts and the joins of a level are
    independent: a shared work index spreads them over the lanes.  -> (root receipt, stats)"""
    from zeth_amd.recursion import fold_plan
    args = run.args

    def spread(jobs):
        """jobs: callables taking a lane -> results in order"""
        out, idx = [None] * len(jobs), WorkIndex()

        def work(ln):
            try:
                while True:
                    k = idx.take(len(jobs))
                    if k is None:
                        return
                    out[k] = jobs[k](ln)
            except Exception as e:
                ln.err = e
        run_lanes(lanes, work)
        return out
    run.device_sync(lanes)
    t0 = time.perf_counter()
    # bottom level: lift + lift + join fused into one proof per pair of segments (lift2) where the program set has it
    rx0 = lanes[0].rec
    jobs, n_fused = [], 0
    for k in range(len(leaves) // 2):
        a, b = leaves[2 * k], leaves[2 * k + 1]
        if not args.no_fused_lift and rx0.has_lift2(a, b):
            jobs.append(lambda ln, a=a, b=b: ln.rec.lift2(a, b, BENCH_NOISE))
            n_fused += 1
        else:
            jobs.append(lambda ln, a=a, b=b: ln.rec.join(ln.rec.lift(a, BENCH_NOISE), ln.rec.lift(b, BENCH_NOISE), BENCH_NOISE))
    if len(leaves) % 2:
        jobs.append(lambda ln, r=leaves[-1]: ln.rec.lift(r, BENCH_NOISE))
    level = spread(jobs)
    run.device_sync(lanes)
    lift_s = time.perf_counter() - t0
    n_joins = 0
    for groups in fold_plan(len(leaves))[1:]:           # above the bottom level: three nodes per proof (join3)
        n_joins += sum((1 if len(g) == 2 or ("join3",) + tuple(level[k].po2 for k in g) in rx0.kinds else 2) for g in groups if len(g) > 1)
        level = spread([(lambda ln, nodes=[level[k] for k in g]: ln.rec.join_group(nodes, BENCH_NOISE)) for g in groups])
    run.device_sync(lanes)
    total_s = time.perf_counter() - t0
    rx = lanes[0].rec
    n_unfused = len(leaves) // 2 - n_fused
    stats = {"segments_lifted": len(leaves), "fused_lift2": n_fused, "lifts": 2 * n_unfused + len(leaves) % 2, "joins": n_joins + n_unfused,
             "proofs": n_fused + 3 * n_unfused + len(leaves) % 2 + n_joins,
             "bottom_level_s": lift_s, "join_phase_s": total_s - lift_s, "fold_s": total_s,
             "bottom_ms_per_segment": 1e3 * lift_s / max(1, len(leaves)), "join_ms_each": 1e3 * (total_s - lift_s) / max(1, n_joins),
             "programs": [{"kind": "-".join(str(x) for x in k), "po2": p.po2, "permutations": p.n_p2, "gates": p.n_gates,
                           "levels": p.n_levels, "witness_words": p.n_inputs} for k, p in zip(rx.kinds, rx.programs)],
             "root_receipt_words": int(level[0].seal.size),
             "note": "every lift runs the STARK verifier on its segment seal and every join on both child seals INSIDE the RECURSION "
                     "circuit (Fiat-Shamir sponge, all Merkle openings, constraint check at z, DEEP, FRI of 50 queries); the root "
                     "receipt is checked below with ONE seal verification + the claim tree of the leaves"}
    return level[0], stats
