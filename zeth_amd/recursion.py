"""lift / join: a block's segment receipts folded into ONE succinct receipt whose every node verified its children IN-CIRCUIT.

Host-side mirror of what `default_prover().prove(env, elf)` (/root/reference/crates/host/src/lib.rs:137) does after the
segments are sealed when `ProverOpts::succinct()` is in force (BASELINE.json config 5): risc0-zkvm 3.0.3
`ProverServer::{lift, join}` -> risc0-circuit-recursion 4.0.2 `Prover::run` on the lift / join programs (both un-vendored:
/root/reference/Cargo.lock:5418, :5305).  Here: the programs are built by circuits/rec_verify.py (this library's STARK verifier
restated as a RECURSION program), loaded once per GPU lane (`hal.RecProgram`: code group resident), and run per receipt.

    programs = build_programs(segment_desc, {20: root20, 18: root18})      # host only; [(kind, blob)]
    rec = Recursion(hal, programs)                                          # per GPU lane
    root = rec.fold([rec.lift(r) for r in segment_receipts])
    root.verify(rec.allowed_roots(), claims)         # host: one seal, one membership, one claim tree

A `RecReceipt` is a seal of the RECURSION circuit: out = claim (8 words) ‖ allowed-programs root A (8 words).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hal as _hal
from .circuits import rec_verify, recursion as rc
from .circuits.desc import P
from .host import fold_claims, hash_pair  # noqa: F401  (fold_claims: the P2-JOIN tree, re-exported for callers)
from .prover import SegmentReceipt

R = (1 << 32) % P
RINV = pow(R, -1, P)
N_ALLOWED = 1 << rec_verify.ALLOWED_DEPTH


def allowed_tree(roots: Sequence[np.ndarray]) -> List[List[np.ndarray]]:
    """levels of the allowed-programs tree over `roots` (padded with zero digests to 16 leaves): [leaves, ..., [root]]"""
    assert len(roots) <= N_ALLOWED
    level = [np.asarray(r, dtype=np.uint32) for r in roots] + [np.zeros(8, np.uint32)] * (N_ALLOWED - len(roots))
    levels = [level]
    while len(level) > 1:
        level = [hash_pair(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        levels.append(level)
    return levels


def membership_words(levels, index: int) -> np.ndarray:
    """the witness words `Verifier.allowed_member` reads: per level the direction bit, then the sibling digest"""
    out = []
    for lvl in levels[:-1]:
        bit = index & 1
        out.append(np.array([bit * R % P], dtype=np.uint32))
        out.append(lvl[index ^ 1])
        index >>= 1
    return np.concatenate(out)


def state_digest(pre: int, post: int) -> np.ndarray:
    """(pre, post, 0, 0, 0, 0, 0, 0): the second operand of a wrapped claim"""
    return np.array([pre, post, 0, 0, 0, 0, 0, 0], dtype=np.uint32)


def wrap_claim(core, pre: int, post: int) -> np.ndarray:
    """claim' = hash_pair(core, (pre, post, 0..)) — what every recursion receipt publishes (circuits/rec_verify.py _wrap): its core
    claim bound to the state range [pre, post] it covers (Montgomery words; 0, 0 for circuits without a state)"""
    return hash_pair(core, state_digest(pre, post))


def segment_state(desc, seal) -> Tuple[int, int]:
    """(pre, post) state words of a segment seal: SYN-C circuits carry them in out[4] / out[0]; every other circuit has none"""
    d = np.asarray(desc, dtype=np.uint32)
    cw = rec_verify.chain_words(rec_verify.Circuit.parse(d))
    return (int(seal[cw[0]]), int(seal[cw[1]])) if cw else (0, 0)


def _parent_node(l, r):
    """the node a join of l and r publishes, as (core, pre, post): core = hash_pair(claim'_l, claim'_r), range = pre(l) .. post(r);
    raises if they do not chain — the join could not have been proven: it asserts post(l) = pre(r) in-circuit"""
    (cl, pl, ql), (cr, pr_, qr) = l, r
    if ql != pr_:
        raise _hal.HalError("claim tree: two neighbouring nodes do not chain (post(l) != pre(r)): no join has a witness for them")
    return hash_pair(wrap_claim(cl, pl, ql), wrap_claim(cr, pr_, qr)), pl, qr


def fold_plan(n: int) -> List[List[Tuple[int, ...]]]:
    """THE shape of a fold over n nodes, level by level: groups of node indices of the level below (a group of one moves up
    unchanged).  The first level PAIRS (a pair of segments is one lift2; a pair of anything else one join), every level above takes
    THREE at a time (one join3 where the program set has it for their sizes, else join(join(a, b), c) — the same node either way),
    a remainder of two is a join, of one moves up.  Everything that folds — Recursion.fold / fold_segments, csrc/session.hip, the
    claim tree the verifiers recompute — follows this one rule."""
    levels, width, first = [], n, True
    while width > 1:
        g = 2 if first else 3
        groups = [tuple(range(k, k + g)) for k in range(0, width - width % g, g)]
        rest = tuple(range(width - width % g, width))
        if rest:
            groups.append(rest)
        levels.append(groups)
        width, first = len(groups), False
    return levels


def _fold_nodes(level):
    for groups in fold_plan(len(level)):
        nxt = []
        for g in groups:
            node = level[g[0]]
            for k in g[1:]:
                node = _parent_node(node, level[k])
            nxt.append(node)
        level = nxt
    return level[0]


def fold_leaf_claims(leaves, ranks: int = 1) -> np.ndarray:
    """The claim' a lift / lift2 / join / join3 tree ends in, recomputed on the host from the LEAVES: [(receipt claim, pre, post)] (a
    bare 8-word claim counts as state (0, 0)).  Node = (core, pre, post): a leaf's core is its receipt claim; a parent's core is
    hash_pair(claim'_l, claim'_r) and its state range runs from the left child's pre to the right child's post; a group of three is
    join(join(a, b), c).  The shape is `fold_plan`'s; ranks > 1: the leaves were folded in `ranks` contiguous equal ranges (one per
    GPU, `aligned_range`) whose roots were then folded by the same rule.  Raises if two neighbours do not chain."""
    level = [(np.asarray(l[0], dtype=np.uint32), int(l[1]), int(l[2])) if isinstance(l, (tuple, list)) and len(l) == 3 and not np.isscalar(l[0])
             else (np.asarray(l, dtype=np.uint32), 0, 0) for l in leaves]
    if ranks > 1:
        per = len(level) // ranks
        if per * ranks != len(level):
            raise ValueError("fold_leaf_claims: the leaves do not split into `ranks` equal ranges")
        level = [_fold_nodes(level[r * per:(r + 1) * per]) for r in range(ranks)]
    return wrap_claim(*_fold_nodes(level))


def succinct_verify(root_seal, allowed_roots: Sequence[np.ndarray], root_program: int, leaves, ranks: int = 1) -> None:
    """zkh_succinct_verify: `RecReceipt.verify` as ONE host-only library call (no GPU, no session) — the seal under an allowed
    program's control root, the allowed-programs root, the claim tree of the leaves ([(receipt claim, pre, post)] or bare claims).
    Raises HalError."""
    import ctypes as C
    lib = _hal.load_library()
    u32p = C.POINTER(C.c_uint32)
    seal = np.ascontiguousarray(root_seal, dtype=np.uint32)
    roots = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint32) for r in allowed_roots]))
    rows = []
    for l in leaves:
        core, pre, post = (l if isinstance(l, (tuple, list)) and len(l) == 3 and not np.isscalar(l[0]) else (l, 0, 0))
        rows.append(np.concatenate([np.asarray(core, dtype=np.uint32), np.array([pre, post], dtype=np.uint32)]))
    lv = np.ascontiguousarray(np.concatenate(rows))
    _hal._check(lib.zkh_succinct_verify(seal.ctypes.data_as(u32p), seal.size, roots.ctypes.data_as(u32p), len(allowed_roots), int(root_program),
                                        lv.ctypes.data_as(u32p), len(rows), int(ranks)))


@dataclass
class RecReceipt:
    """`SuccinctReceipt` analogue: one seal of the RECURSION circuit under program `program` (index into the allowed set).
    core / pre / post open the claim' the seal publishes (claim' = wrap_claim(core, pre, post)): a join needs them as witness for
    its children and checks them in-circuit, so they are carried next to the seal, never trusted."""
    seal: np.ndarray
    po2: int
    program: int
    control_root: np.ndarray
    n_leaves: int = 1
    core: Optional[np.ndarray] = None
    pre: int = 0
    post: int = 0

    @property
    def claim(self) -> np.ndarray:
        return np.asarray(self.seal[:8], dtype=np.uint32)

    @property
    def allowed(self) -> np.ndarray:
        return np.asarray(self.seal[8:16], dtype=np.uint32)

    def to_upstream_bytes(self, allowed_levels, journal: bytes = b"") -> bytes:
        """This receipt in upstream's wire format: bincode `Receipt{inner: Succinct{seal, control_id, claim, hashfn,
        verifier_parameters, control_inclusion_proof}, ..}` (receipt_codec.py, RECALLED layouts): control_id = the program's control
        root, the inclusion proof = its membership path in the allowed-programs tree (`allowed_tree` levels), the claim digest in
        the placeholder claim's pruned `post` slot."""
        from . import receipt_codec as rc
        idx, digests = self.program, []
        for lvl in allowed_levels[:-1]:
            digests.append([int(w) for w in lvl[idx ^ 1]])
            idx >>= 1
        return rc.succinct_receipt_bytes(self.seal, self.control_root, self.claim, journal, self.program, digests, verifier_parameters=allowed_levels[-1][0])

    def verify(self, allowed_roots: Sequence[np.ndarray], leaf_claims: Optional[Sequence[np.ndarray]] = None, ranks: int = 1) -> None:
        """Host check of the whole tree below this receipt: ONE seal verification, the program's membership in the allowed
        set, the allowed root the receipt carries, and (given the leaves: receipt claims, or (claim, pre, post) for circuits with a
        state) the claim tree — whose joins each asserted post(l) = pre(r) in-circuit.  Raises HalError."""
        roots = [np.asarray(r, dtype=np.uint32) for r in allowed_roots]
        if not any(np.array_equal(self.control_root, r) for r in roots):
            raise _hal.HalError("recursion receipt: its program is not in the allowed set")
        _hal.HostCircuit(rc.recursion_circuit()).verify_segment(self.seal, self.control_root)
        if not np.array_equal(self.allowed, allowed_tree(roots)[-1][0]):
            raise _hal.HalError("recursion receipt: it was produced under another allowed-programs root")
        if leaf_claims is not None and not np.array_equal(self.claim, fold_leaf_claims(list(leaf_claims), ranks)):
            raise _hal.HalError("recursion receipt: its claim is not the root of the leaves' claim tree")


class ProgramSet(list):
    """[(kind, blob)] + the leaf families [(circuit description, {po2: control root})] the lifts were built for"""
    families: list = []


def build_programs(segment_desc, segment_roots: Optional[Dict[int, np.ndarray]] = None, zk_cycles: int = _hal.ZK_CYCLES,
                   assumptions: Sequence[Tuple[np.ndarray, Dict[int, np.ndarray]]] = (), fused_pairs: bool = True,
                   ternary: bool = True) -> List[Tuple[Tuple, np.ndarray]]:
    """The program set of a block: one lift per segment size (`segment_roots`: {po2: control root of the segment circuit}) and
    per size of every assumption circuit (`assumptions`: [(circuit description, {po2: control root})], e.g. KECCAK-F batches:
    upstream lifts those receipts too and resolves them on the way to the succinct receipt), then joins for every pair of
    child sizes that can meet, until the set of sizes closes (po2-20 / po2-18 SYN-A segments: two lifts at po2 17,
    join(17,17) -> 18, join(17,18), join(18,17), join(18,18) -> 18).  Pure host work, no GPU: [(kind, blob)], kind =
    ("lift", segment po2, family) with family 0 = the segment circuit, 1.. = the assumption circuits, ("lift2", po2_l, po2_r) =
    lift + lift + join fused for a pair of segment receipts (fused_pairs), ("join", po2_l, po2_r), or (ternary) ("join3", m, m, m)
    for the largest size m when three such children fit one proof of that size again — the levels above the bottom of a large
    block are all of that size, and take three nodes per proof instead of two."""
    rdesc = rc.recursion_circuit()
    out: List[Tuple[Tuple, np.ndarray]] = []
    sizes = set()

    def add(kind, pr):
        po2 = pr.min_po2(zk_cycles)
        out.append((kind, pr.finish(po2, zk_cycles)))
        sizes.add(po2)
    families = [(segment_desc, segment_roots or {})] + list(assumptions)
    canon = lambda root: [int(w) * RINV % P for w in np.asarray(root, dtype=np.uint32)]
    for fam, (desc, roots) in enumerate(families):
        desc = np.asarray(desc, dtype=np.uint32)
        for po2, root in sorted(roots.items(), reverse=True):
            add(("lift", po2, fam), rec_verify.build_lift(desc, po2, canon(root)))
    if fused_pairs:
        # lift + lift + join as one program for pairs of SEGMENT receipts (a block's segments come largest first, the short
        # tail last: pairs (a, b) with a >= b)
        po2s = sorted(segment_roots or {}, reverse=True)
        for i, a in enumerate(po2s):
            for b in po2s[i:]:
                add(("lift2", a, b), rec_verify.build_lift2(np.asarray(segment_desc, dtype=np.uint32), a, canon(segment_roots[a]), b, canon(segment_roots[b])))
    done = set()
    while True:
        todo = [(a, b) for a in sorted(sizes) for b in sorted(sizes) if (a, b) not in done]
        if not todo:
            break
        for a, b in todo:
            done.add((a, b))
            add(("join", a, b), rec_verify.build_join(rdesc, a, b))
    if ternary and sizes and len(out) < N_ALLOWED:
        m = max(sizes)
        pr3 = rec_verify.build_join(rdesc, m, m, m)
        if pr3.min_po2(zk_cycles) == m:
            out.append((("join3", m, m, m), pr3.finish(m, zk_cycles)))
    assert len(out) <= N_ALLOWED, f"{len(out)} programs do not fit the allowed set"
    ps = ProgramSet(out)
    ps.families = [(np.asarray(d, dtype=np.uint32), dict(r)) for d, r in families]
    return ps


def aligned_range(n_leaves: int, world_size: int, rank: int) -> range:
    """The leaves rank `rank` folds when a block is spread over `world_size` GPUs: contiguous, equal ranges; every rank folds its
    range to a local root (`fold_plan`), rank 0 folds the `world_size` local roots by the same rule - the one exchange of the
    path: world_size - 1 receipts (~210 KB each) gathered over the control plane - and the verifier recomputes that shape
    (`fold_leaf_claims(leaves, ranks=world_size)`).  world_size and n_leaves / world_size are kept powers of two (every range
    then has the same shape, and a block's short tail segment lands in the last range)."""
    per = n_leaves // world_size
    if n_leaves % world_size or per & (per - 1) or world_size & (world_size - 1) or per == 0:
        raise ValueError("the recursive fold on N ranks needs N and S / N to be powers of two")
    return range(rank * per, (rank + 1) * per)


class Recursion:
    """The lift / join programs of one GPU lane (one HipHal): loaded once, code groups resident."""

    def __init__(self, hal: "_hal.HipHal", programs: Sequence[Tuple[Tuple, np.ndarray]], families: Sequence = ()):
        """families: [(circuit description, {po2: control root})] in build_programs' order (family 0 = the segment circuit) — what
        lift / lift2 need to open a leaf's claim (receipt claim + state words); without it leaves are taken to have no state and
        their claims must be supplied (`lift(..., claim=)`)."""
        self.hal = hal
        self.families = list(families) or list(getattr(programs, "families", ()))
        self.circuit = hal.load_circuit(rc.recursion_circuit())
        self.kinds = [k for k, _ in programs]
        self.programs = [_hal.RecProgram(hal, self.circuit, blob) for _, blob in programs]
        self.levels = allowed_tree([p.root for p in self.programs])

    def allowed_roots(self) -> List[np.ndarray]:
        return [p.root for p in self.programs]

    def allowed_root(self) -> np.ndarray:
        return self.levels[-1][0]

    def _leaf(self, receipt: SegmentReceipt, family: int):
        """(receipt claim, pre, post) of a leaf: what the lift proves about it"""
        if family >= len(self.families):
            raise _hal.HalError("Recursion: no circuit description / control roots for this leaf family (pass families= to Recursion)")
        desc, roots = self.families[family]
        claim = _hal.HostCircuit(desc).receipt_claim(receipt.seal, roots[receipt.po2])
        pre, post = segment_state(desc, receipt.seal)
        return claim, pre, post

    def lift(self, receipt: SegmentReceipt, noise_seed: Optional[int] = None, family: int = 0) -> RecReceipt:
        """family 0: a receipt of the segment circuit; 1..: of the corresponding assumption circuit of build_programs"""
        i = self.kinds.index(("lift", receipt.po2, family))
        inputs = np.concatenate([np.asarray(receipt.seal, dtype=np.uint32), self.allowed_root()])
        seal, _ = self.programs[i].prove(inputs, _seed(noise_seed))
        core, pre, post = self._leaf(receipt, family)
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, 1, core, pre, post)

    def has_lift2(self, left: SegmentReceipt, right: SegmentReceipt) -> bool:
        return ("lift2", left.po2, right.po2) in self.kinds

    def lift2(self, left: SegmentReceipt, right: SegmentReceipt, noise_seed: Optional[int] = None) -> RecReceipt:
        """the node lift(left), lift(right), join would produce, as one proof (the program asserts post(left) = pre(right))"""
        i = self.kinds.index(("lift2", left.po2, right.po2))
        inputs = np.concatenate([np.asarray(left.seal, dtype=np.uint32), np.asarray(right.seal, dtype=np.uint32), self.allowed_root()])
        seal, _ = self.programs[i].prove(inputs, _seed(noise_seed))
        (cl, pl, ql), (cr, pr_, qr) = self._leaf(left, 0), self._leaf(right, 0)
        core = hash_pair(wrap_claim(cl, pl, ql), wrap_claim(cr, pr_, qr))
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, 2, core, pl, qr)

    def join(self, left: RecReceipt, right: RecReceipt, noise_seed: Optional[int] = None) -> RecReceipt:
        """inputs per child: seal, membership path, then the opening of its claim' (core, pre, post); the program checks the opening,
        both memberships, and post(left) = pre(right)"""
        i = self.kinds.index(("join", left.po2, right.po2))
        parts = []
        for ch in (left, right):
            if ch.core is None:
                raise _hal.HalError("join: a child receipt without the opening of its claim (core, pre, post)")
            parts += [ch.seal, membership_words(self.levels, ch.program), np.asarray(ch.core, dtype=np.uint32), np.array([ch.pre, ch.post], dtype=np.uint32)]
        seal, _ = self.programs[i].prove(np.concatenate(parts), _seed(noise_seed))
        core = hash_pair(left.claim, right.claim)
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, left.n_leaves + right.n_leaves, core, left.pre, right.post)

    def join3(self, a: RecReceipt, b: RecReceipt, c: RecReceipt, noise_seed: Optional[int] = None) -> RecReceipt:
        """the node join(join(a, b), c) would produce, as ONE proof (the inner node's claim' is computed in-circuit)"""
        i = self.kinds.index(("join3", a.po2, b.po2, c.po2))
        parts = []
        for ch in (a, b, c):
            if ch.core is None:
                raise _hal.HalError("join: a child receipt without the opening of its claim (core, pre, post)")
            parts += [ch.seal, membership_words(self.levels, ch.program), np.asarray(ch.core, dtype=np.uint32), np.array([ch.pre, ch.post], dtype=np.uint32)]
        seal, _ = self.programs[i].prove(np.concatenate(parts), _seed(noise_seed))
        inner = wrap_claim(hash_pair(a.claim, b.claim), a.pre, b.post)
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, a.n_leaves + b.n_leaves + c.n_leaves,
                          hash_pair(inner, c.claim), a.pre, c.post)

    def join_group(self, nodes: Sequence[RecReceipt], noise_seed: Optional[int] = None) -> RecReceipt:
        """one group of `fold_plan`: a single node moves up, two are a join, three a join3 — or join(join(a, b), c) where the
        program set has no join3 for their sizes (the same node: same claim, one proof more)"""
        if len(nodes) == 1:
            return nodes[0]
        if len(nodes) == 2:
            return self.join(nodes[0], nodes[1], noise_seed)
        a, b, c = nodes
        if ("join3", a.po2, b.po2, c.po2) in self.kinds:
            return self.join3(a, b, c, noise_seed)
        return self.join(self.join(a, b, noise_seed), c, noise_seed)

    def fold_segments(self, receipts: Sequence[SegmentReceipt], noise_seed: Optional[int] = None) -> RecReceipt:
        """segment receipts -> one receipt: the bottom level pairs them with lift2 where the program set has it (else lift, lift,
        join), an unpaired last receipt is lifted; the levels above are `fold_plan`'s (three nodes per proof).  Same tree, same
        claims as lifting everything and calling `fold`."""
        level: List[RecReceipt] = []
        for k in range(len(receipts) // 2):
            a, b = receipts[2 * k], receipts[2 * k + 1]
            level.append(self.lift2(a, b, noise_seed) if self.has_lift2(a, b) else self.join(self.lift(a, noise_seed), self.lift(b, noise_seed), noise_seed))
        if len(receipts) % 2:
            level.append(self.lift(receipts[-1], noise_seed))
        for groups in fold_plan(len(receipts))[1:]:
            level = [self.join_group([level[k] for k in g], noise_seed) for g in groups]
        return level[0]

    def fold(self, leaves: Sequence[RecReceipt], noise_seed: Optional[int] = None) -> RecReceipt:
        """the tree of `fold_plan` over recursion receipts: the first level pairs, every level above takes three at a time"""
        level = list(leaves)
        for groups in fold_plan(len(level)):
            level = [self.join_group([level[k] for k in g], noise_seed) for g in groups]
        return level[0]


def _seed(noise_seed: Optional[int]) -> int:
    from .prover import fresh_noise_seed
    return fresh_noise_seed() if noise_seed is None else noise_seed
