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
    return wrap_claim(*_fold_root(leaves, ranks))


def _fold_root(leaves, ranks: int = 1):
    """(core, pre, post) of the root node of `fold_leaf_claims`' tree"""
    level = [(np.asarray(l[0], dtype=np.uint32), int(l[1]), int(l[2])) if isinstance(l, (tuple, list)) and len(l) == 3 and not np.isscalar(l[0])
             else (np.asarray(l, dtype=np.uint32), 0, 0) for l in leaves]
    if ranks > 1:
        per = len(level) // ranks
        if per * ranks != len(level):
            raise ValueError("fold_leaf_claims: the leaves do not split into `ranks` equal ranges")
        level = [_fold_nodes(level[r * per:(r + 1) * per]) for r in range(ranks)]
    return _fold_nodes(level)


def _canonical(claim) -> Tuple[int, ...]:
    """a digest as canonical residues: what `union` orders by"""
    return tuple(int(w) * RINV % P for w in np.asarray(claim, dtype=np.uint32))


def union_node(left, right) -> Tuple[np.ndarray, bool]:
    """(claim' of union(left, right), swapped?): wrap(hash_pair(lo, hi), 0, 0) with (lo, hi) the two claim' SORTED by their canonical
    words, lexicographically (upstream's UnionClaim keeps left <= right the same way: risc0-zkvm 3.0.3, recalled)"""
    l, r = np.asarray(left, dtype=np.uint32), np.asarray(right, dtype=np.uint32)
    swap = _canonical(r) < _canonical(l)
    lo, hi = (r, l) if swap else (l, r)
    return wrap_claim(hash_pair(lo, hi), 0, 0), swap


def union_claims(claims: Sequence[np.ndarray]) -> np.ndarray:
    """The claim' a union tree over assumption receipts ends in, from their claim' (a lifted keccak receipt publishes
    wrap_claim(receipt claim, 0, 0)): neighbours are united pairwise, level by level (an odd one moves up).  The SET decides the
    pairs' order, not the caller: union(a, b) = union(b, a)."""
    level = [np.asarray(c, dtype=np.uint32) for c in claims]
    if not level:
        raise _hal.HalError("union: no assumption claims")
    while len(level) > 1:
        nxt = [union_node(level[k], level[k + 1])[0] for k in range(0, len(level) - 1, 2)]
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return level[0]


def resolved_claim(cond_claim, cond_pre: int, cond_post: int, assumption_claim) -> np.ndarray:
    """claim' of resolve(cond, assum): the conditional's state range over hash_pair(claim'_cond, claim'_assum)"""
    return wrap_claim(hash_pair(np.asarray(cond_claim, dtype=np.uint32), np.asarray(assumption_claim, dtype=np.uint32)), cond_pre, cond_post)


def succinct_verify(root_seal, allowed_roots: Sequence[np.ndarray], root_program: int, leaves, ranks: int = 1,
                    assumption_claims: Optional[Sequence[np.ndarray]] = None) -> None:
    """zkh_succinct_verify: `RecReceipt.verify` as ONE host-only library call (no GPU, no session) — the seal under an allowed
    program's control root, the allowed-programs root, the claim tree of the leaves ([(receipt claim, pre, post)] or bare claims).
    assumption_claims (zkh_succinct_verify_resolved): the receipt claims of the assumption receipts a RESOLVED receipt was bound to;
    leaves empty: a union-tree root alone.  Raises HalError."""
    import ctypes as C
    lib = _hal.load_library()
    u32p = C.POINTER(C.c_uint32)
    seal = np.ascontiguousarray(root_seal, dtype=np.uint32)
    roots = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint32) for r in allowed_roots]))
    rows = []
    for l in leaves or ():
        core, pre, post = (l if isinstance(l, (tuple, list)) and len(l) == 3 and not np.isscalar(l[0]) else (l, 0, 0))
        rows.append(np.concatenate([np.asarray(core, dtype=np.uint32), np.array([pre, post], dtype=np.uint32)]))
    lv = np.ascontiguousarray(np.concatenate(rows)) if rows else np.zeros(1, np.uint32)
    if assumption_claims is None:
        _hal._check(lib.zkh_succinct_verify(seal.ctypes.data_as(u32p), seal.size, roots.ctypes.data_as(u32p), len(allowed_roots), int(root_program),
                                            lv.ctypes.data_as(u32p), len(rows), int(ranks)))
        return
    ac = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in assumption_claims])) if len(assumption_claims) else np.zeros(1, np.uint32)
    _hal._check(lib.zkh_succinct_verify_resolved(seal.ctypes.data_as(u32p), seal.size, roots.ctypes.data_as(u32p), len(allowed_roots), int(root_program),
                                                 lv.ctypes.data_as(u32p), len(rows), int(ranks), ac.ctypes.data_as(u32p), len(assumption_claims)))


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

    def verify(self, allowed_roots: Sequence[np.ndarray], leaf_claims: Optional[Sequence[np.ndarray]] = None, ranks: int = 1,
               assumption_claims: Optional[Sequence[np.ndarray]] = None) -> None:
        """Host check of the whole tree below this receipt: ONE seal verification, the program's membership in the allowed
        set, the allowed root the receipt carries, and (given the leaves: receipt claims, or (claim, pre, post) for circuits with a
        state) the claim tree — whose joins each asserted post(l) = pre(r) in-circuit.  assumption_claims (receipt claims of the
        assumption receipts, e.g. keccak batches): the receipt is a RESOLVED one — resolve(join tree over the leaves, union tree
        over the lifted assumptions); without leaf_claims: a union-tree root over them alone.  Raises HalError."""
        roots = [np.asarray(r, dtype=np.uint32) for r in allowed_roots]
        if not any(np.array_equal(self.control_root, r) for r in roots):
            raise _hal.HalError("recursion receipt: its program is not in the allowed set")
        _hal.HostCircuit(rc.recursion_circuit()).verify_segment(self.seal, self.control_root)
        if not np.array_equal(self.allowed, allowed_tree(roots)[-1][0]):
            raise _hal.HalError("recursion receipt: it was produced under another allowed-programs root")
        if assumption_claims is not None:
            assumed = union_claims([wrap_claim(c, 0, 0) for c in assumption_claims])
            if leaf_claims is None:
                expect = assumed
            else:
                core, pre, post = _fold_root(list(leaf_claims), ranks)
                expect = resolved_claim(wrap_claim(core, pre, post), pre, post, assumed)
            if not np.array_equal(self.claim, expect):
                raise _hal.HalError("recursion receipt: its claim is not the resolved root of the leaves' claim tree and the assumptions' union tree")
        elif leaf_claims is not None and not np.array_equal(self.claim, fold_leaf_claims(list(leaf_claims), ranks)):
            raise _hal.HalError("recursion receipt: its claim is not the root of the leaves' claim tree")


class ProgramSet(list):
    """[(kind, blob)] + the leaf families [(circuit description, {po2: control root})] the lifts were built for"""
    families: list = []


def build_programs(segment_desc, segment_roots: Optional[Dict[int, np.ndarray]] = None, zk_cycles: int = _hal.ZK_CYCLES,
                   assumptions: Sequence[Tuple[np.ndarray, Dict[int, np.ndarray]]] = (), fused_pairs: bool = True,
                   ternary: bool = True, resolve: bool = False) -> List[Tuple[Tuple, np.ndarray]]:
    """The program set of a block: one lift per segment size (`segment_roots`: {po2: control root of the segment circuit}) and
    per size of every assumption circuit (`assumptions`: [(circuit description, {po2: control root})], e.g. KECCAK-F batches:
    upstream lifts those receipts too and resolves them on the way to the succinct receipt), then joins for every pair of
    child sizes that can meet, until the set of sizes closes (po2-20 / po2-18 SYN-A segments: two lifts at po2 17,
    join(17,17) -> 18, join(17,18), join(18,17), join(18,18) -> 18).  Pure host work, no GPU: [(kind, blob)], kind =
    ("lift", segment po2, family) with family 0 = the segment circuit, 1.. = the assumption circuits, ("lift2", po2_l, po2_r) =
    lift + lift + join fused for a pair of segment receipts (fused_pairs), ("join", po2_l, po2_r), or (ternary) ("join3", m, m, m)
    for the largest size m when three such children fit one proof of that size again — the levels above the bottom of a large
    block are all of that size, and take three nodes per proof instead of two.
    resolve: upstream's shape for assumptions — their lifts do NOT join the segments' tree: ("union", a, b) programs close over THEIR
    sizes (the union tree of `Recursion.union_fold`), and ("resolve", s, u) binds a session root of size s to a union root of size u
    (`Recursion.resolve`).  Without it (the default) assumption lifts are leaves of the one join tree."""
    rdesc = rc.recursion_circuit()
    out: List[Tuple[Tuple, np.ndarray]] = []
    sizes = set()

    asizes = set()                        # resolve: the sizes on the assumptions' side (their lifts, the unions)

    def add(kind, pr, into=None):
        po2 = pr.min_po2(zk_cycles)
        out.append((kind, pr.finish(po2, zk_cycles)))
        (sizes if into is None else into).add(po2)
    families = [(segment_desc, segment_roots or {})] + list(assumptions)
    canon = lambda root: [int(w) * RINV % P for w in np.asarray(root, dtype=np.uint32)]
    for fam, (desc, roots) in enumerate(families):
        desc = np.asarray(desc, dtype=np.uint32)
        for po2, root in sorted(roots.items(), reverse=True):
            add(("lift", po2, fam), rec_verify.build_lift(desc, po2, canon(root)), asizes if resolve and fam else None)
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
    if resolve and asizes:
        # the pairs `union_fold` can meet: two leaves, or a union result on the LEFT of anything (an odd node moves up at the END of its
        # level, so the right operand is never newer than the left one)
        leaf_sizes, usizes, done = set(asizes), set(), set()
        while True:
            todo = [(a, b) for a in sorted(leaf_sizes | usizes) for b in sorted(leaf_sizes | usizes)
                    if (a, b) not in done and (a in usizes or (a in leaf_sizes and b in leaf_sizes))]
            if not todo:
                break
            for a, b in todo:
                done.add((a, b))
                add(("union", a, b), rec_verify.build_union(rdesc, a, b), usizes)
        # a session root of size s against a union root of size u: the largest session sizes first, as far as the allowed set has room
        for a in sorted(sizes, reverse=True):
            for b in sorted(leaf_sizes | usizes):
                if len(out) < N_ALLOWED:
                    add(("resolve", a, b), rec_verify.build_resolve(rdesc, a, b), set())
                elif a == max(sizes):
                    raise ValueError(f"build_programs: the allowed set ({N_ALLOWED} programs) has no room for resolve({a}, {b}): this block's segment sizes give "
                                     f"{len(sizes)} program sizes (their join closure alone is {len(sizes) ** 2} programs); drop join3 / the fused pairs")
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

    def union(self, a: RecReceipt, b: RecReceipt, noise_seed: Optional[int] = None) -> RecReceipt:
        """`ProverServer::union`: two receipts of ANY claims -> one whose claim' is wrap(hash_pair of the SORTED pair, 0, 0); the swap
        bit the program reads is decided here (union_node); union(a, b) and union(b, a) publish the same claim'"""
        i = self.kinds.index(("union", a.po2, b.po2))
        claim, swap = union_node(a.claim, b.claim)
        parts = [a.seal, membership_words(self.levels, a.program), b.seal, membership_words(self.levels, b.program),
                 np.array([R if swap else 0], dtype=np.uint32)]
        seal, _ = self.programs[i].prove(np.concatenate(parts), _seed(noise_seed))
        lo, hi = (b, a) if swap else (a, b)
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, a.n_leaves + b.n_leaves, hash_pair(lo.claim, hi.claim), 0, 0)

    def union_fold(self, leaves: Sequence[RecReceipt], noise_seed: Optional[int] = None) -> RecReceipt:
        """the union tree over assumption receipts (lifted keccak batches): neighbours pairwise, level by level (`union_claims`)"""
        level = list(leaves)
        if not level:
            raise _hal.HalError("union: no assumption receipts")
        while len(level) > 1:
            nxt = [self.union(level[k], level[k + 1], noise_seed) for k in range(0, len(level) - 1, 2)]
            if len(level) % 2:
                nxt.append(level[-1])
            level = nxt
        return level[0]

    def resolve(self, cond: RecReceipt, assum: RecReceipt, noise_seed: Optional[int] = None) -> RecReceipt:
        """`ProverServer::resolve`: the conditional receipt (a session's join-tree root; its claim' is opened in-circuit) bound to the
        receipt of what it assumed (a union-tree root): claim' = wrap(hash_pair(claim'_cond, claim'_assum), pre, post) of the session"""
        i = self.kinds.index(("resolve", cond.po2, assum.po2))
        if cond.core is None:
            raise _hal.HalError("resolve: a conditional receipt without the opening of its claim (core, pre, post)")
        parts = [cond.seal, membership_words(self.levels, cond.program), np.asarray(cond.core, dtype=np.uint32), np.array([cond.pre, cond.post], dtype=np.uint32),
                 assum.seal, membership_words(self.levels, assum.program)]
        seal, _ = self.programs[i].prove(np.concatenate(parts), _seed(noise_seed))
        return RecReceipt(seal, self.programs[i].po2, i, self.programs[i].root, cond.n_leaves + assum.n_leaves,
                          hash_pair(cond.claim, assum.claim), cond.pre, cond.post)

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
