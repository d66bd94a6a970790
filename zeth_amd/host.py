"""Block-level orchestration: the mirror of `BlockProcessor::prove` (/root/reference/crates/host/src/lib.rs:123-143).

Upstream proves the segments of one block in a plain loop (`ProverImpl::prove_session`, risc0-zkvm 3.0.3,
un-vendored: /root/reference/Cargo.lock:5418) and concatenates the `SegmentReceipt`s into a
`CompositeReceipt`; `receipt.verify(image_id)` then checks every segment
(/root/reference/crates/host/src/bin/cli.rs:103).  Segments are independent, so with G GPUs (one process per
GPU, `torch.distributed` ranks) segment i goes to rank i mod G and NO data-path collective is needed; the
receipts (~0.25 MB each) are gathered on rank 0 over the control plane (gloo / RCCL object gather).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

from .prover import Segment, SegmentReceipt


def partition_round_robin(n_segments: int, world_size: int, rank: int) -> List[int]:
    """Segment indices owned by `rank`: i with i % world_size == rank (BASELINE.json north_star)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    return list(range(rank, n_segments, world_size))


def session_segments(total_cycles: int, segment_po2: int = 20, base_seed: int = 0x5EED0000) -> List[Segment]:
    """Split a session of `total_cycles` into 2^po2-cycle segments; the tail segment gets the smallest po2 >= 13
    that holds the remainder (upstream pads the last segment to a power of two, MIN_CYCLES_PO2 = 13)."""
    if total_cycles <= 0:
        raise ValueError("total_cycles must be positive")
    seg_cycles = 1 << segment_po2
    segs: List[Segment] = []
    full, rem = divmod(total_cycles, seg_cycles)
    for i in range(full):
        segs.append(Segment(index=i, po2=segment_po2, seed=base_seed + i))
    if rem:
        po2 = 13
        while (1 << po2) < rem:
            po2 += 1
        segs.append(Segment(index=full, po2=min(po2, segment_po2), seed=base_seed + full))
    return segs


@dataclass
class CompositeReceipt:
    """`CompositeReceipt{segments[]}` analogue: ordered by segment index."""
    segments: List[SegmentReceipt]

    def verify_integrity(self) -> None:
        idx = [s.index for s in self.segments]
        if idx != list(range(len(idx))):
            raise ValueError(f"composite receipt has missing or unordered segments: {idx}")

    def verify(self, circuit_desc) -> None:
        """`receipt.verify(image_id)` analogue: structural integrity + every segment seal through the host verifier."""
        self.verify_integrity()
        for s in self.segments:
            s.verify(circuit_desc)


class BlockProcessor:
    """Proves the segment list of one block on this rank's GPU and (optionally) gathers across ranks."""

    def __init__(self, prove_segment: Callable[[Segment], SegmentReceipt], rank: int = 0, world_size: int = 1,
                 gather: Optional[Callable[[List[SegmentReceipt]], Optional[List[List[SegmentReceipt]]]]] = None):
        self.prove_segment = prove_segment
        self.rank, self.world_size = rank, world_size
        self.gather = gather

    def prove_local(self, segments: Sequence[Segment]) -> List[SegmentReceipt]:
        mine = partition_round_robin(len(segments), self.world_size, self.rank)
        return [self.prove_segment(segments[i]) for i in mine]

    def prove(self, segments: Sequence[Segment]) -> Optional[CompositeReceipt]:
        local = self.prove_local(segments)
        if self.world_size == 1 or self.gather is None:
            rec = CompositeReceipt(sorted(local, key=lambda r: r.index))
            rec.verify_integrity()
            return rec
        parts = self.gather(local)
        if parts is None:           # non-root rank
            return None
        rec = CompositeReceipt(sorted((r for p in parts for r in p), key=lambda r: r.index))
        rec.verify_integrity()
        return rec


class DevModeProver:
    """`RISC0_DEV_MODE=1` analogue (BASELINE config 1; /root/reference/README.md:104-109, CI at
    /root/reference/.github/workflows/main.yml:51-54): no proving, a fake receipt per segment that only carries the
    claim metadata.  Plumbing for schedulers and tests — it never touches a GPU and its receipts never verify."""

    def prove_segment(self, seg: Segment) -> SegmentReceipt:
        import numpy as np
        return SegmentReceipt(seal=np.zeros(0, dtype=np.uint32), index=seg.index, po2=seg.po2, hashfn="fake")


def dev_mode_enabled() -> bool:
    import os
    return os.environ.get("RISC0_DEV_MODE", "").lower() in ("1", "true", "yes")


def torch_gather(rank: int, world_size: int):
    """Control-plane gather of receipts to rank 0 with torch.distributed (gloo on CPU, RCCL on GPU)."""
    import torch.distributed as dist

    def _g(local: List[SegmentReceipt]):
        out = [None] * world_size if rank == 0 else None
        dist.gather_object(local, out, dst=0)
        return out

    return _g


# ---------------------------------------------------------------------------------------------------------------
# Config 5 of BASELINE.json (lift/join to one succinct receipt), restated synthetically: the
# recursion circuit (risc0-circuit-recursion 4.0.2, un-vendored: /root/reference/Cargo.lock:5305) is unobtainable, so
# a "join" here is one seal of the declared-synthetic SYN-J shape (W_code 16, W_data 128, W_accum 16, po2 18 by
# default) whose witness seed is bound to the two child seals.  Same HAL ops, different circuit, tree-shaped schedule.
# ---------------------------------------------------------------------------------------------------------------
JOIN_PO2 = 18


@dataclass(frozen=True)
class JoinTask:
    level: int          # 1 = joins of leaves
    index: int          # position within the level
    left: int           # node index in the level below
    right: int
    device: int         # rank that runs it: the one that produced `left`


def join_schedule(n_leaves: int, world_size: int) -> List[List[JoinTask]]:
    """Binary join tree over `n_leaves` segment receipts: ceil(log2 S) dependent levels; level l pairs nodes (2k, 2k+1)
    of level l-1, an unpaired last node is carried up unchanged.  Leaf i lives on rank i mod G; every join runs where
    its left child was produced, so only the right child's receipt (~0.25 MB) crosses the control plane."""
    if n_leaves <= 0 or world_size <= 0:
        raise ValueError("need at least one leaf and one rank")
    owners = [i % world_size for i in range(n_leaves)]
    levels: List[List[JoinTask]] = []
    level = 0
    while len(owners) > 1:
        level += 1
        tasks = [JoinTask(level, k, 2 * k, 2 * k + 1, owners[2 * k]) for k in range(len(owners) // 2)]
        nxt = [t.device for t in tasks]
        if len(owners) % 2:
            nxt.append(owners[-1])
        levels.append(tasks)
        owners = nxt
    return levels


def join_seed(left: SegmentReceipt, right: SegmentReceipt) -> int:
    """64-bit witness seed of a join, bound to both child seals (stand-in for the recursion circuit reading them)."""
    import hashlib
    h = hashlib.sha256(left.seal_bytes() + right.seal_bytes()).digest()
    return int.from_bytes(h[:8], "little")


def prove_succinct(leaves: Sequence[SegmentReceipt], prove_join: Callable[[Segment], SegmentReceipt],
                   join_po2: int = JOIN_PO2) -> "SuccinctReceipt":
    """Single-rank driver of the join tree (every rank of `join_schedule` collapses onto the caller)."""
    nodes = list(leaves)
    joins: List[List[SegmentReceipt]] = []
    for tasks in join_schedule(len(nodes), 1):
        done = [prove_join(Segment(index=t.index, po2=join_po2, seed=join_seed(nodes[t.left], nodes[t.right])))
                for t in tasks]
        nxt = list(done)
        if len(nodes) % 2:
            nxt.append(nodes[-1])
        joins.append(done)
        nodes = nxt
    return SuccinctReceipt(root=nodes[0], joins=joins, leaves=list(leaves))


@dataclass
class SuccinctReceipt:
    root: SegmentReceipt
    joins: List[List[SegmentReceipt]]
    leaves: List[SegmentReceipt]

    def verify(self, segment_desc, join_desc) -> None:
        """Every leaf seal and every join seal is accepted by the host verifier, and the tree has the scheduled shape.
        (The synthetic join does not constrain its children — that is what the real recursion circuit adds.)"""
        shape = [len(t) for t in join_schedule(len(self.leaves), 1)]
        if [len(lvl) for lvl in self.joins] != shape:
            raise ValueError(f"join tree has levels {[len(lvl) for lvl in self.joins]}, expected {shape}")
        for s in self.leaves:
            s.verify(segment_desc)
        for lvl in self.joins:
            for j in lvl:
                j.verify(join_desc)
