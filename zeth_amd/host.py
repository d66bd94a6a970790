"""Block-level orchestration: the mirror of `BlockProcessor::prove` (/root/reference/crates/host/src/lib.rs:123-143).

Upstream proves the segments of one block in a plain loop (`ProverImpl::prove_session`, risc0-zkvm 3.0.3,
un-vendored: /root/reference/Cargo.lock:5418) and concatenates the `SegmentReceipt`s into a
`CompositeReceipt`; `receipt.verify(image_id)` then checks every segment
(/root/reference/crates/host/src/bin/cli.rs:103).  Segments are independent, so with G GPUs (one process per
GPU, `torch.distributed` ranks) segment i goes to rank i mod G and NO data-path collective is needed; the
receipts (~0.25 MB each) are gathered on rank 0 over the control plane (gloo / RCCL object gather).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

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


# ---------------------------------------------------------------------------------------------------------------
# Cached block inputs: /root/reference/crates/host/src/bin/cli.rs:113-145 (`get_cached_input`) keeps one
# `cache/input_<blockhash>.json` per block = serde_json of `StatelessInput{block, witness}`; run-parallel.sh:93 iterates
# over exactly those files.  The segment count of a block comes from EXECUTING the guest on that input (the "N total
# cycles" line run-parallel.sh:67 scrapes), which needs the rv32im executor + guest ELF (not available offline), so this
# reader validates the file shape and, as cli.rs:141 does (`ensure!(input.block.hash_slow() == header.hash)`), that the block
# header hashes to the hash in the file name (zeth_amd/eth_header.py); the CYCLES stay a stub: taken from a sidecar written by a previous dev-mode run
# (`input_<hash>.cycles.json`: the run-parallel.sh columns) when there is one, and otherwise falls back to a declared
# estimate from the block's gas — enough to drive `session_segments` / the block bench with realistic segment counts.
# ---------------------------------------------------------------------------------------------------------------
CYCLES_PER_GAS_ESTIMATE = 9.0        # declared heuristic (order of magnitude of RISC Zero's published zeth runs); replace with measured data


@dataclass
class CachedInput:
    block_hash: str
    path: str
    input_bytes: int
    gas_used: Optional[int]
    total_cycles: int
    cycles_source: str              # "sidecar" (measured by a dev-mode run) | "gas-estimate" | "size-estimate"
    keccak_calls: Optional[int] = None
    hash_checked: bool = False      # the header was complete and keccak256(rlp(header)) == block_hash (cli.rs:141)
    block_number: Optional[int] = None

    def segments(self, segment_po2: int = 20, base_seed: int = 0x5EED0000) -> List[Segment]:
        return session_segments(self.total_cycles, segment_po2, base_seed)


def read_cached_input(cache_dir: str, block_hash: str, check_hash: bool = True) -> CachedInput:
    """`cache/input_<hash>.json` -> CachedInput (see the banner above for what is real and what is estimated).  check_hash (default): the
    header must be complete and hash to `block_hash`, as cli.rs:141 insists; False accepts a file whose header cannot be hashed (a stub
    that only carries `gasUsed`) and says so in `hash_checked`."""
    import json
    import os
    path = os.path.join(cache_dir, f"input_{block_hash}.json")
    with open(path) as fh:
        doc = json.load(fh)
    if not isinstance(doc, dict) or "block" not in doc or "witness" not in doc:
        raise ValueError(f"{path}: not a StatelessInput (expected an object with `block` and `witness`)")
    header = doc["block"].get("header", {}) if isinstance(doc["block"], dict) else {}
    gas = header.get("gasUsed", header.get("gas_used"))
    if isinstance(gas, str):
        gas = int(gas, 16) if gas.startswith("0x") else int(gas)
    from .eth_header import header_hash, header_is_complete
    checked, number = False, None
    if header_is_complete(header):
        got = header_hash(header)
        if got != block_hash.lower():
            raise ValueError(f"{path}: the block header hashes to {got}, not to the hash the file is named after")
        checked = True
        number = header["number"]
        number = int(number, 16) if isinstance(number, str) and number.startswith("0x") else int(number)
    elif check_hash:
        raise ValueError(f"{path}: the block header is incomplete: its hash cannot be re-derived (pass check_hash=False for a stub input)")
    size = os.path.getsize(path)
    side = os.path.join(cache_dir, f"input_{block_hash}.cycles.json")
    keccak = None
    if os.path.exists(side):
        with open(side) as fh:
            s = json.load(fh)
        cycles, source, keccak = int(s["total_cycles"]), "sidecar", s.get("keccak_calls")
    elif gas:
        cycles, source = int(gas * CYCLES_PER_GAS_ESTIMATE), "gas-estimate"
    else:
        cycles, source = max(1 << 20, size * 64), "size-estimate"
    return CachedInput(block_hash, path, size, gas, cycles, source, keccak, checked, number)


def list_cached_inputs(cache_dir: str) -> List[str]:
    """Block hashes with a cached input, the `cache/input_0x*.json` glob of run-parallel.sh:93."""
    import glob
    import os
    return sorted(os.path.basename(p)[len("input_"):-len(".json")] for p in glob.glob(os.path.join(cache_dir, "input_0x*.json"))
                  if not p.endswith(".cycles.json"))


def _root_for(control_root, po2: int):
    """control_root argument of the verify methods: None (shipped table), one root, {po2: root} or callable(po2)."""
    if control_root is None or hasattr(control_root, "shape") or isinstance(control_root, (list, tuple)):
        return control_root
    if isinstance(control_root, dict):
        return control_root.get(po2)
    return control_root(po2)


@dataclass
class CompositeReceipt:
    """`CompositeReceipt{segments[], assumption_receipts[]}` analogue: segments ordered by index; `assumptions` holds
    the coprocessor proofs the guest requested (keccak batches, `prove_keccak` upstream) — each verified with its own circuit."""
    segments: List[SegmentReceipt]
    assumptions: List[SegmentReceipt] = field(default_factory=list)

    def verify_integrity(self, chained: bool = False, initial_state: int = 0) -> None:
        """`CompositeReceipt::verify_integrity`: the segments are all there, in order — and, for a session of SYN-C segments
        (`chained`, circuits/syn_air.py syn_chain: out = (post, 0, 0, 0, pre)), CONTINUOUS: the first segment starts from
        `initial_state` and every segment's pre-state is its predecessor's post-state (upstream: `pre == prev.post` over
        `ReceiptClaim`s).  The state words are read from the seals' `out` headers, which the seal verification binds."""
        idx = [s.index for s in self.segments]
        if idx != list(range(len(idx))):
            raise ValueError(f"composite receipt has missing or unordered segments: {idx}")
        if chained:
            from .circuits.syn_air import CHAIN_POST, CHAIN_PRE
            from .hal import fp_encode
            prev = fp_encode(initial_state)
            for s in self.segments:
                if int(s.seal[CHAIN_PRE]) != prev:
                    raise ValueError(f"composite receipt is not continuous: segment {s.index} starts from state word {int(s.seal[CHAIN_PRE])}, "
                                     f"its predecessor ended in {prev}")
                prev = int(s.seal[CHAIN_POST])

    def to_upstream_bytes(self, circuit_desc, control_root=None, journal: bytes = b"") -> bytes:
        """This composite in upstream's wire format: bincode `Receipt{inner: Composite{segments, ..}, journal, metadata}`
        (receipt_codec.py: layouts RECALLED from risc0-zkvm 3.0.3).  Every segment carries its seal, index, hashfn and — in the
        pruned `post` slot of a placeholder `ReceiptClaim` — its claim digest (these circuits have no rv32im SystemState)."""
        from . import receipt_codec as rc
        from .prover import shipped_control_root
        root_of = lambda s: _root_for(control_root, s.po2) if control_root is not None else shipped_control_root(circuit_desc, s.po2)
        if _is_session_circuit(circuit_desc):       # SYN-S: the REAL claim values (states, exit code, output digest) in upstream's layout
            return rc.composite_receipt_bytes([(s.seal, s.index, segment_claim(s).to_codec_value()) for s in self.segments], journal=journal)
        segs = [(s.seal, s.index, receipt_claim(s, circuit_desc, root_of(s))) for s in self.segments]
        return rc.composite_receipt_bytes(segs, journal=journal)

    @staticmethod
    def from_upstream_bytes(data: bytes, out_size: int) -> "CompositeReceipt":
        """decode the bincode container back into segment receipts (seal, index; po2 from the seal header)"""
        import numpy as np
        from . import receipt_codec as rc
        from .hal import fp_decode
        val = rc.decode(rc.Receipt, data)
        if val["inner"][0] != "Composite":
            raise ValueError(f"not a composite receipt: {val['inner'][0]}")
        segs = []
        for s in val["inner"][1]["segments"]:
            seal = np.asarray(s["seal"], dtype=np.uint32)
            segs.append(SegmentReceipt(seal=seal, index=s["index"], po2=fp_decode(int(seal[out_size])), hashfn=s["hashfn"], output=seal[:out_size].copy()))
        return CompositeReceipt(segs)

    def final_state(self) -> int:
        """post-state word of the last segment of a chained session (Montgomery form)"""
        from .circuits.syn_air import CHAIN_POST
        return int(self.segments[-1].seal[CHAIN_POST])

    def verify(self, circuit_desc, control_root=None, assumption_desc=None, assumption_control_root=None, chained: bool = False,
               initial_state: int = 0) -> None:
        """`receipt.verify(image_id)` analogue: structural integrity (+ continuity of a chained session) + every segment seal through
        the host verifier against the expected control root (None: the shipped table; or {po2: root} / callable for other sizes)."""
        self.verify_integrity(chained, initial_state)
        for s in self.segments:
            s.verify(circuit_desc, _root_for(control_root, s.po2))
        if self.assumptions and assumption_desc is None:
            raise ValueError("composite receipt carries assumption receipts but no circuit was given for them")
        for a in self.assumptions:
            a.verify(assumption_desc, _root_for(assumption_control_root, a.po2))


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


def chain_segments(segments: Sequence[Segment], contribution: Callable[[Segment], int], initial_state: int = 0) -> List[Segment]:
    """The executor's part of a chained session (SYN-C): give every segment its pre-state as its public input.
    `contribution(seg)` = the Montgomery word the segment adds to the running state (its `post` when started from state 0:
    `SegmentProver.chain_contribution` on the GPU, the oracle in tests); upstream's executor likewise fixes every segment's
    pre / post `SystemState` before any segment is proven, which is what keeps the segments independent for the provers."""
    from dataclasses import replace
    from .hal import P, fp_encode
    out, state = [], fp_encode(initial_state)
    for seg in segments:
        out.append(replace(seg, pub=(state,)))
        state = (state + contribution(seg)) % P           # Montgomery words add like the elements they stand for
    return out


# ---------------------------------------------------------------------------------------------------------------
# ReceiptClaim: what `receipt.verify(image_id)` + the journal comparison (/root/reference/crates/host/src/bin/cli.rs:103-107) are
# about.  Upstream (risc0-zkvm 3.0.3 receipt_claim.rs, risc0-binfmt {exit_code.rs, tagged_struct}; un-vendored, RECALLED):
#     ReceiptClaim { pre: SystemState, post: SystemState, exit_code: ExitCode, input: Option<Input>, output: Option<Output> }
#     digest = tagged_struct("risc0.ReceiptClaim", [input, pre, post, output], [sys_exit, user_exit])
#     tagged_struct(tag, down, data) = SHA-256( SHA-256(tag) ‖ down digests ‖ data as u32 LE ‖ len(down) as u16 LE )
#     SystemState digest = tagged_struct("risc0.SystemState", [merkle_root], [pc]);  Output = ("risc0.Output", [journal, assumptions], [])
#     ExitCode::into_pair: Halted(u) -> (0, u), Paused(u) -> (1, u), SystemSplit -> (2, 0), SessionLimit -> (2, 2)
# A SYN-S segment (circuits/syn_air.py syn_session) binds the pieces in its seal: pre / post state words, the exit pair, the 16 limbs
# of the OUTPUT digest tagged_struct("risc0.Output", [SHA-256(journal), Assumptions digest]) (round 6; until then SHA-256(journal) alone:
# which receipts a session had assumed was then whatever list the verifier was handed).  The SHA-256 claim digests below are computed on the host BESIDE the Poseidon2 claim the recursion circuit
# uses (zkh_receipt_claim); tools/check_upstream_receipt.py is the one-command check of the recalled layouts against a real receipt.
# ---------------------------------------------------------------------------------------------------------------
EXIT_HALTED, EXIT_PAUSED, EXIT_SYSTEM_SPLIT, EXIT_SESSION_LIMIT = "Halted", "Paused", "SystemSplit", "SessionLimit"


def exit_code_pair(exit_code) -> tuple:
    """ExitCode -> (system, user) as upstream's `into_pair`; exit_code = ("Halted", u) / ("Paused", u) / ("SystemSplit", None) / .."""
    name, user = exit_code
    if name == EXIT_HALTED:
        return (0, int(user or 0))
    if name == EXIT_PAUSED:
        return (1, int(user or 0))
    if name == EXIT_SYSTEM_SPLIT:
        return (2, 0)
    if name == EXIT_SESSION_LIMIT:
        return (2, 2)
    raise ValueError(f"unknown exit code {exit_code!r}")


def exit_code_from_pair(sys: int, user: int):
    table = {(2, 0): (EXIT_SYSTEM_SPLIT, None), (2, 2): (EXIT_SESSION_LIMIT, None)}
    if (sys, user) in table:
        return table[(sys, user)]
    if sys == 0:
        return (EXIT_HALTED, user)
    if sys == 1:
        return (EXIT_PAUSED, user)
    raise ValueError(f"({sys}, {user}) is not an exit code pair")


def sha256_words(data: bytes) -> List[int]:
    """SHA-256 as a risc0 `Digest`: eight u32 words, each the little-endian read of four digest bytes"""
    import hashlib
    d = hashlib.sha256(data).digest()
    return [int.from_bytes(d[4 * i:4 * i + 4], "little") for i in range(8)]


def tagged_struct(tag: str, down: Sequence[Sequence[int]], data: Sequence[int] = ()) -> List[int]:
    import hashlib
    import struct
    body = hashlib.sha256(tag.encode()).digest()
    for d in down:
        body += struct.pack("<8I", *[int(w) for w in d])
    body += struct.pack(f"<{len(data)}I", *[int(w) for w in data]) + struct.pack("<H", len(down))
    return sha256_words(body)


ZERO_DIGEST = [0] * 8


def union_claim_digest(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """`UnionClaim{left, right}.digest()` (risc0-zkvm 3.0.3, RECALLED): the two claim digests SORTED (left <= right, compared word by
    word), then tagged_struct("risc0.UnionClaim", [left, right], []) — the SHA-256 statement beside the Poseidon2 claim' a `union`
    program publishes (zeth_amd/recursion.py union_node): union(a, b) and union(b, a) are the same claim."""
    a, b = [int(w) for w in a], [int(w) for w in b]
    left, right = (a, b) if a <= b else (b, a)
    return tagged_struct("risc0.UnionClaim", [left, right])


def assumptions_digest(assumptions: Sequence[Tuple[Sequence[int], Sequence[int]]]) -> List[int]:
    """`Assumptions(Vec<Assumption{claim, control_root}>).digest()` (RECALLED): a cons list folded from the END over the zero digest,
    every element tagged_struct("risc0.Assumption", [claim, control_root], []) and every cons cell tagged_struct("risc0.Assumptions",
    [head, tail], []) — what `Output{journal, assumptions}` commits to for a session that assumed receipts (keccak batches)."""
    acc = list(ZERO_DIGEST)
    for claim, control_root in reversed(list(assumptions)):
        head = tagged_struct("risc0.Assumption", [list(claim), list(control_root)])
        acc = tagged_struct("risc0.Assumptions", [head, acc])
    return acc


def output_digest(journal: bytes, assumptions: Sequence[Tuple[Sequence[int], Sequence[int]]] = ()) -> List[int]:
    """`Output{journal, assumptions}.digest()` (RECALLED) = tagged_struct("risc0.Output", [SHA-256(journal), Assumptions digest], []) —
    what the LAST segment of a session binds: the journal AND which receipts the session assumed ([(claim digest, control root)] of
    its keccak batches, in order; none: the zero digest).  Twin of csrc/verifier.hip session_output_limbs."""
    return tagged_struct("risc0.Output", [sha256_words(bytes(journal)), assumptions_digest(assumptions)], [])


def output_limbs(journal: bytes, assumptions=()) -> List[int]:
    """the output digest as the sixteen 16-bit limbs a SYN-S seal binds (Montgomery words; limb k = digest bytes 2k, 2k + 1 LE)"""
    from .hal import fp_encode
    out = []
    for w in output_digest(journal, assumptions):
        out += [fp_encode(w & 0xFFFF), fp_encode(w >> 16)]
    return out


def assumption_of(receipt: "SegmentReceipt", circuit_desc, control_root) -> Tuple[List[int], List[int]]:
    """`Assumption{claim, control_root}` of a receipt a session assumes: its claim digest (the 8 words zkh_receipt_claim gives: Poseidon2
    over its public output, size and control root) and the control root of ITS circuit"""
    return [int(w) for w in receipt_claim(receipt, circuit_desc, control_root)], [int(w) for w in control_root]


@dataclass
class ReceiptClaim:
    """`risc0_zkvm::ReceiptClaim` for a segment (or a whole session) of a SYN-S circuit: the state words stand where upstream has
    `SystemState{pc, merkle_root}` (pc 0, merkle_root = (state word, 0, ..)), `input` is None as in every zkVM-2 receipt, and the
    output is the digest of `Output{journal, assumptions}` for a halted segment (what the seal binds: `output_digest`), None otherwise."""
    pre: int                                   # canonical residues of the state words
    post: int
    exit_code: tuple = (EXIT_HALTED, 0)
    output: Optional[List[int]] = None          # 8 words: Output{journal, assumptions}.digest(); None = no output (the segment did not halt)

    def state_digest(self, word: int) -> List[int]:
        return tagged_struct("risc0.SystemState", [[int(word)] + [0] * 7], [0])

    def output_digest(self) -> List[int]:
        return ZERO_DIGEST if self.output is None else [int(w) for w in self.output]

    def digest(self) -> List[int]:
        sys, user = exit_code_pair(self.exit_code)
        return tagged_struct("risc0.ReceiptClaim", [ZERO_DIGEST, self.state_digest(self.pre), self.state_digest(self.post), self.output_digest()], [sys, user])

    def to_codec_value(self) -> dict:
        """this claim as a value of receipt_codec.ReceiptClaim (upstream's bincode layout, recalled)"""
        st = lambda w: ("Value", {"pc": 0, "merkle_root": [int(w)] + [0] * 7})
        # the output travels PRUNED (MaybePruned::Pruned(digest)): the seal binds the digest of Output{journal, assumptions}, not its parts
        out = ("Value", None) if self.output is None else ("Pruned", [int(w) for w in self.output])
        return {"pre": st(self.pre), "post": st(self.post), "exit_code": self.exit_code, "input": ("Value", None), "output": out}

    @staticmethod
    def from_codec_value(v: dict) -> "ReceiptClaim":
        word = lambda s: int(s[1]["merkle_root"][0])
        if v["output"][0] == "Pruned":
            od = [int(w) for w in v["output"][1]]
        elif v["output"][1] is None:
            od = None
        else:                                  # Value{journal: Pruned(d), assumptions: Pruned(d)}: the digest of the pair
            od = tagged_struct("risc0.Output", [[int(w) for w in v["output"][1]["journal"][1]], [int(w) for w in v["output"][1]["assumptions"][1]]], [])
        return ReceiptClaim(pre=word(v["pre"]), post=word(v["post"]), exit_code=tuple(v["exit_code"]), output=od)


def segment_claim(receipt: SegmentReceipt) -> ReceiptClaim:
    """The claim a SYN-S segment's seal binds, read from its `out` words (trust it only after the seal has been verified)."""
    from .circuits.syn_air import CHAIN_POST, CHAIN_PRE, SESSION_EXIT_SYS, SESSION_EXIT_USER, SESSION_JOURNAL, SESSION_JOURNAL_LIMBS, SESSION_OUT_WORDS
    from .hal import HalError, fp_decode
    seal = receipt.seal
    if len(seal) <= SESSION_OUT_WORDS:
        raise HalError("segment_claim: not a SYN-S seal")
    sys, user = fp_decode(int(seal[SESSION_EXIT_SYS])), fp_decode(int(seal[SESSION_EXIT_USER]))
    try:
        code = exit_code_from_pair(sys, user)
    except ValueError as e:
        raise HalError(f"segment_claim: {e}")
    limbs = [fp_decode(int(w)) for w in seal[SESSION_JOURNAL:SESSION_JOURNAL + SESSION_JOURNAL_LIMBS]]
    if any(l >> 16 for l in limbs):
        raise HalError("segment_claim: an output-digest limb does not fit 16 bits")
    od = None
    if code[0] == EXIT_HALTED or any(limbs):
        raw = b"".join(int(l).to_bytes(2, "little") for l in limbs)
        od = [int.from_bytes(raw[4 * i:4 * i + 4], "little") for i in range(8)]
    return ReceiptClaim(pre=fp_decode(int(seal[CHAIN_PRE])), post=fp_decode(int(seal[CHAIN_POST])), exit_code=code, output=od)


def session_pub_words(pre_state_mont: int, is_last: bool, journal: bytes, assumptions=()) -> tuple:
    """the 19 public words of a SYN-S segment: pre-state, exit pair, output-digest limbs (Montgomery words)"""
    from .hal import fp_encode
    sys, user = exit_code_pair((EXIT_HALTED, 0) if is_last else (EXIT_SYSTEM_SPLIT, None))
    return (int(pre_state_mont), fp_encode(sys), fp_encode(user)) + tuple(output_limbs(journal, assumptions) if is_last else [0] * 16)


def chain_session(segments: Sequence[Segment], contribution: Callable[[Segment], int], initial_state: int = 0, assumptions=(),
                  journal: Optional[bytes] = None):
    """The executor's part of a SYN-S session: pre-states (as `chain_segments`), exit codes (SystemSplit .. SystemSplit, Halted(0)) and
    the journal — the bytes the guest commits (zeth: the 32-byte block hash, guests/stateless-client/src/lib.rs:33); None: the session's
    final state word, canonical, 4 bytes little-endian — whose OUTPUT digest Output{journal, assumptions}
    the LAST segment binds; `assumptions` = [(claim digest, control root)] of the receipts the session assumes (`assumption_of`).
    -> (segments with their 19 public words, journal bytes)"""
    from dataclasses import replace
    from .hal import P, fp_decode, fp_encode
    pres, state = [], fp_encode(initial_state)
    for seg in segments:
        pres.append(state)
        state = (state + contribution(seg)) % P
    journal = int(fp_decode(state)).to_bytes(4, "little") if journal is None else bytes(journal)
    out = [replace(seg, pub=session_pub_words(pre, i + 1 == len(segments), journal, assumptions)) for i, (seg, pre) in enumerate(zip(segments, pres))]
    return out, journal


def verify_session_integrity(receipts: Sequence[SegmentReceipt], initial_state: int, journal: Optional[bytes], assumptions=()) -> ReceiptClaim:
    """`CompositeReceipt::verify_integrity` + the exit-code and journal checks on VERIFIED SYN-S seals: indices 0 .. n-1 in order, the
    first segment starts from `initial_state`, every pre-state is the predecessor's post-state, every segment but the last ends in
    SystemSplit with no output, the last in Halted(0) with the digest of `journal` (None: the final state word).  A session with
    trailing segments cut off ends in a SystemSplit: refused.  -> the session's claim (pre of the first, post / exit / output of the last)."""
    from .hal import HalError
    if [r.index for r in receipts] != list(range(len(receipts))) or not receipts:
        raise HalError(f"session: missing or unordered segments: {[r.index for r in receipts]}")
    claims = [segment_claim(r) for r in receipts]
    prev = int(initial_state)
    for i, c in enumerate(claims):
        if c.pre != prev:
            raise HalError(f"session: segment {i} starts from state {c.pre}, its predecessor ended in {prev}")
        prev = c.post
        last = i + 1 == len(claims)
        if not last and (c.exit_code != (EXIT_SYSTEM_SPLIT, None) or c.output is not None):
            raise HalError(f"session: segment {i} of {len(claims)} does not end in SystemSplit ({c.exit_code}): the segments are not those of one session")
        if last and c.exit_code != (EXIT_HALTED, 0):
            raise HalError(f"session: the last segment says {c.exit_code}, not Halted(0): the session was cut short or did not succeed")
    j = journal if journal is not None else int(claims[-1].post).to_bytes(4, "little")
    if claims[-1].output != output_digest(bytes(j), assumptions):
        raise HalError("session: the journal and these assumption receipts do not hash to the output digest the last segment's seal binds "
                       "(another journal, or the session assumed other receipts / in another order)" if assumptions else
                       "session: the journal does not hash to the output digest the last segment's seal binds")
    return ReceiptClaim(pre=claims[0].pre, post=claims[-1].post, exit_code=claims[-1].exit_code, output=claims[-1].output)



IMAGE_ID_SCHEME = "zeth_amd.ImageId.v2"


def image_id(circuit_desc, initial_state: int = 0) -> "np.ndarray":
    """The 8-word identifier a verifier is handed for a session — the analogue of zeth's Image ID (`compute_image_id(elf)`,
    /root/reference/crates/host/src/lib.rs:74-84: a SHA-256 Merkle digest of the guest program's initial memory image).  Here the
    "program" is the circuit and the state every session starts from:
        tagged_struct("zeth_amd.ImageId.v2", [SHA-256(description words)], [initial state])
    Collision resistant over the WHOLE description.  Scheme v2 (round 6): the id binds the program ONLY.  v1 also hashed every
    control root zeth_amd/circuits/control_roots.json happened to hold for po2 1 .. 24, so regenerating or widening that file (round 5
    did: 13 .. 22 -> 13 .. 24) silently changed the id of every circuit and broke every issued `Receipt.verify(expected_image_id)`
    (round-5 advisor finding).  A control root is a FUNCTION of (description, po2, zk cycles) — the Merkle root of the code group the
    description generates — so binding the description binds them all; which root a seal is checked against is the verifier's own
    table (`shipped_control_root`), not something the id has to carry.  tests/test_session_claims.py pins the value."""
    import numpy as np
    d = np.ascontiguousarray(circuit_desc, dtype="<u4")
    return np.array(tagged_struct(IMAGE_ID_SCHEME, [sha256_words(d.tobytes())], [int(initial_state) & 0xFFFFFFFF]), dtype=np.uint32)


def _is_session_circuit(circuit_desc) -> bool:
    from .circuits.syn_air import SESSION_OUT_WORDS
    return int(circuit_desc[13]) == 1 and int(circuit_desc[7]) == SESSION_OUT_WORDS


@dataclass
class Receipt:
    """`risc0_zkvm::Receipt{inner, journal}` analogue for a session: what `BlockProcessor::prove` returns next to the image id
    (/root/reference/crates/host/src/lib.rs:123-143) and what the CLI then checks (/root/reference/crates/host/src/bin/cli.rs:103-107):
    `receipt.verify(image_id)`, then the journal against the expected value.  The journal is the session's final state word
    (canonical, 4 bytes little-endian); the LAST segment's seal binds Output{journal, assumptions} (SYN-S), `assumptions` being the
    [(claim digest, control root)] of the receipts the session assumed (its keccak batches) — a conditional receipt names them, it does
    not carry them: `verify_assumptions` checks the receipts a holder presents against that list."""
    inner: CompositeReceipt
    journal: bytes
    assumptions: tuple = ()

    def claim(self) -> ReceiptClaim:
        """the session's claim (SYN-S): pre of the first segment, post / exit code / output of the last (upstream: `Receipt::claim`)"""
        first, last = segment_claim(self.inner.segments[0]), segment_claim(self.inner.segments[-1])
        return ReceiptClaim(pre=first.pre, post=last.post, exit_code=last.exit_code, output=last.output)

    def verify_assumptions(self, circuit_desc, receipts: Sequence[SegmentReceipt], control_roots) -> None:
        """The receipts a holder presents for the session's assumptions (seals of `circuit_desc`, e.g. KECCAK-F batches): every one is
        verified against its control root, and their (claim digest, control root) list must be EXACTLY the one the session names — which
        `verify` has tied to the output digest the last seal binds.  A session cannot be resolved against other receipts.  Raises."""
        from .hal import HalError
        got = []
        for r in receipts:
            root = control_roots[r.po2]
            r.verify(circuit_desc, root)
            got.append(assumption_of(r, circuit_desc, root))
        want = [([int(w) for w in c], [int(w) for w in k]) for c, k in self.assumptions]
        if got != want:
            raise HalError(f"receipt.verify_assumptions: the session names {len(want)} assumption(s); the {len(got)} receipt(s) presented are not those (or not in that order)")

    def verify(self, expected_image_id, circuit_desc, initial_state: int = 0, control_root=None, n_segments: Optional[int] = None) -> None:
        """Every segment seal against its control root; the image id the caller expected = the one (circuit, control roots, initial
        state) hash to; and the session is WHOLE:
          SYN-S circuits — continuity from `initial_state`, SystemSplit .. SystemSplit, Halted(0), Output{SHA-256(journal), assumptions} = the
          output digest the last seal binds (`verify_session_integrity`): trailing segments cannot be cut off, the journal cannot be
          rewritten, the assumption list cannot be swapped;
          SYN-C circuits bind no exit code: continuity + journal == the last seal's post-state, and the caller MUST say how many segments
          the session has (`n_segments`) — without it a holder could drop trailing segments and rewrite the 4-byte journal.  Raises."""
        import numpy as np
        from .hal import HalError, fp_decode
        d = np.asarray(circuit_desc, dtype=np.uint32)
        if not np.array_equal(np.asarray(expected_image_id, dtype=np.uint32), image_id(d, initial_state)):
            raise HalError("receipt.verify: the image id does not match this circuit and initial state")
        if n_segments is not None and len(self.inner.segments) != n_segments:
            raise HalError(f"receipt.verify: the receipt holds {len(self.inner.segments)} segments, the session has {n_segments}")
        if _is_session_circuit(d):
            self.inner.verify(d, control_root)                       # order + every seal
            verify_session_integrity(self.inner.segments, initial_state, self.journal, self.assumptions)
            return
        if n_segments is None:
            raise HalError("receipt.verify: a SYN-C session does not bind its termination (no exit code in its seals): pass the expected "
                           "segment count, or prove the session with a SYN-S circuit (circuits/syn_air.py syn_session)")
        self.inner.verify(d, control_root, chained=True, initial_state=initial_state)
        if self.journal != int(fp_decode(self.inner.final_state())).to_bytes(4, "little"):
            raise HalError("receipt.verify: the journal is not the final state the last segment's seal binds")

    def check_block_hash(self, block_hash) -> None:
        """The CLI's last step after `receipt.verify(image_id)` (/root/reference/crates/host/src/bin/cli.rs:103-107): the journal decodes
        as a 32-byte hash (`B256::try_from(journal)`) and is the hash of the block that was to be proven.  `block_hash`: bytes or 0x-hex.  Raises."""
        from .hal import HalError
        want = bytes.fromhex(block_hash[2:] if block_hash.startswith("0x") else block_hash) if isinstance(block_hash, str) else bytes(block_hash)
        if len(self.journal) != 32:
            raise HalError(f"failed to decode journal: {len(self.journal)} bytes, a block hash has 32")
        if self.journal != want:
            raise HalError("journal output mismatch")

    def to_upstream_bytes(self, circuit_desc, control_root=None) -> bytes:
        """upstream's bincode `Receipt{inner: Composite{..}, journal, metadata}`; SYN-S segments carry their REAL claim values"""
        return self.inner.to_upstream_bytes(circuit_desc, control_root, journal=self.journal)

    @staticmethod
    def from_upstream_bytes(data: bytes, circuit_desc) -> "Receipt":
        """Decode the bincode container.  For a SYN-S circuit the `ReceiptClaim` each segment CARRIES must be the one its seal BINDS
        (states, exit code, output digest read from the seal's `out` words): a container whose claim fields were edited is refused here,
        before any seal is verified (the seal itself is what `verify` trusts)."""
        from . import receipt_codec as rc
        from .hal import HalError
        val = rc.decode(rc.Receipt, data)
        comp = CompositeReceipt.from_upstream_bytes(data, int(circuit_desc[7]))
        if _is_session_circuit(circuit_desc):
            for seg, v in zip(comp.segments, val["inner"][1]["segments"]):
                carried, bound = ReceiptClaim.from_codec_value(v["claim"]), segment_claim(seg)
                if carried != bound:
                    raise HalError(f"receipt: segment {seg.index} carries the claim {carried}, its seal binds {bound}")
        return Receipt(comp, bytes(val["journal"]["bytes"]))


def prove_chained_block(prove_segment: Callable[[Segment], SegmentReceipt], contribution: Callable[[Segment], int], circuit_desc,
                        segments: Sequence[Segment], initial_state: int = 0, assumptions=(), journal: Optional[bytes] = None):
    """`BlockProcessor::prove(input, po2) -> (Receipt, image id)` (/root/reference/crates/host/src/lib.rs:123-143) for a chained
    session on this rank: the executor's pass (pre-states; SYN-S: exit codes and the journal digest too), the segment seals, the
    composite with its journal."""
    from .hal import fp_decode
    if _is_session_circuit(circuit_desc):
        chained, journal = chain_session(segments, contribution, initial_state, assumptions, journal)
        comp = CompositeReceipt([prove_segment(s) for s in chained])
        verify_session_integrity(comp.segments, initial_state, journal, assumptions)
        return Receipt(comp, journal, tuple(assumptions)), image_id(circuit_desc, initial_state)
    chained = chain_segments(segments, contribution, initial_state)
    comp = CompositeReceipt([prove_segment(s) for s in chained])
    comp.verify_integrity(chained=True, initial_state=initial_state)
    return Receipt(comp, int(fp_decode(comp.final_state())).to_bytes(4, "little")), image_id(circuit_desc, initial_state)


class Session:
    """`ProverServer::prove_session` as ONE native call (zkh_session_*, csrc/session.hip): the segments of a session sealed on
    `devices` x `lanes_per_device` lanes through one shared work index inside the library (C++ threads, no Python in the loop),
    receipts in index order, optionally folded through the P2-JOIN tree to one root receipt.  What a Rust shim's
    `Prover::prove` would call once per session (/root/reference/crates/host/src/lib.rs:137)."""

    def __init__(self, circuit_desc, devices: Sequence[int] = (0,), lanes_per_device: int = 3, join_desc=None):
        import ctypes as C
        import numpy as np
        from . import hal as _hal
        _hal.load_library()
        self._hal, self._C, self._np = _hal, C, np
        self.desc = np.ascontiguousarray(circuit_desc, dtype=np.uint32)
        self.join_desc = None if join_desc is None else np.ascontiguousarray(join_desc, dtype=np.uint32)
        dev = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _hal._check(_hal._lib.zkh_session_create(dev, len(devices), lanes_per_device, _hal._ptr(self.desc), self.desc.size,
                                                 _hal._ptr(self.join_desc) if self.join_desc is not None else None,
                                                 0 if self.join_desc is None else self.join_desc.size, C.byref(h)))
        self.h = h

    def close(self) -> None:
        h, self.h = getattr(self, "h", None), None
        if h:
            self._hal._lib.zkh_session_destroy(h)

    __del__ = close

    def set_recursion(self, programs) -> None:
        """programs: [(kind, blob)] from zeth_amd.recursion.build_programs - loaded on every lane (code groups resident); then
        `prove(..., join_tree=2)` lifts every receipt and joins them with in-circuit verification of every child seal"""
        C, np = self._C, self._np
        from .circuits import recursion as rc
        rdesc = np.ascontiguousarray(rc.recursion_circuit(), dtype=np.uint32)
        blobs = [np.ascontiguousarray(b, dtype=np.uint32) for _, b in programs]
        u32p = C.POINTER(C.c_uint32)
        ptrs = (u32p * len(blobs))(*[b.ctypes.data_as(u32p) for b in blobs])
        words = (C.c_size_t * len(blobs))(*[b.size for b in blobs])
        code = {"lift": 0, "join": 1, "lift2": 2, "join3": 3, "union": 4, "resolve": 5}
        for k, _ in programs:
            if k[0] == "join3" and k[1] != k[2]:
                raise ValueError("set_recursion: a join3 whose first two children differ in size has no kind code")
        kinds = np.array([[code[k[0]], k[1], k[3] if k[0] == "join3" else k[2]] for k, _ in programs], dtype=np.uint32).reshape(-1)
        self._hal._check(self._hal._lib.zkh_session_set_recursion(self.h, self._hal._ptr(rdesc), rdesc.size, ptrs, words, self._hal._ptr(kinds), len(blobs)))

    def set_assumptions(self, circuit_desc, receipts: Sequence[SegmentReceipt], control_roots) -> None:
        """the session's assumption receipts (keccak batches: seals of `circuit_desc`, proven beforehand): verified on the host here;
        `prove(..., join_tree=2)` then lifts them, unites them pairwise (sorted pairs) and resolves the session's root against the
        union root — ProverServer::{lift, union, resolve}.  control_roots: {po2: control root} of that circuit.  Before
        build_recursion / set_recursion(build_programs(..., assumptions=[...], resolve=True)).  No receipts: cleared."""
        C, np = self._C, self._np
        u32p = C.POINTER(C.c_uint32)
        if not receipts:
            self._hal._check(self._hal._lib.zkh_session_set_assumptions(self.h, None, 0, None, None, None, None, 0))
            return
        desc = np.ascontiguousarray(circuit_desc, dtype=np.uint32)
        seals = [np.ascontiguousarray(r.seal, dtype=np.uint32) for r in receipts]
        ptrs = (u32p * len(seals))(*[x.ctypes.data_as(u32p) for x in seals])
        words = (C.c_size_t * len(seals))(*[x.size for x in seals])
        po2s = np.array([r.po2 for r in receipts], dtype=np.uint32)
        roots = np.ascontiguousarray(np.concatenate([np.asarray(control_roots[r.po2], dtype=np.uint32) for r in receipts]))
        self._hal._check(self._hal._lib.zkh_session_set_assumptions(self.h, self._hal._ptr(desc), desc.size, ptrs, words, self._hal._ptr(po2s),
                                                                    self._hal._ptr(roots), len(seals)))

    def build_recursion(self, po2s, join3: bool = True) -> None:
        """the program set of a block with segments of the sizes `po2s`, built by the LIBRARY (csrc/rec_builder.hip: the C++ twin of
        zeth_amd.recursion.build_programs — same programs, same order, same allowed-programs root) and loaded on every lane"""
        np = self._np
        sizes = np.array(sorted({int(p) for p in po2s}, reverse=True), dtype=np.uint32)
        self._hal._check(self._hal._lib.zkh_session_build_recursion(self.h, self._hal._ptr(sizes), sizes.size, int(join3)))

    def set_streamed_fold(self, on: bool) -> None:
        """join_tree=2: prove a lift2 / join the moment its children exist, concurrently with the sealing lanes (default), or hold
        the fold back until the last segment is sealed (two phases).  Same tree, same receipts."""
        self._hal._lib.zkh_session_set_streamed_fold(self.h, int(on))

    def set_witness_source(self, source: int, producers_per_lane: int = 0) -> None:
        """0: closed-form witness generators on the device (default).  1: a sequential host preflight per segment runs ahead of the
        seals on `producers_per_lane` host threads per lane (0 = 2); its compact per-cycle records (16 bytes per cycle) are uploaded
        from pinned memory and expanded by the GPU row-fill kernel (csrc/preflight.hip) — upstream's preflight -> witgen shape."""
        self._hal._check(self._hal._lib.zkh_session_set_witness_source(self.h, int(source), int(producers_per_lane)))

    def set_chained(self, on: bool, initial_state: int = 0) -> None:
        """SYN-C sessions: the library runs the executor's pass (every segment's pre-state = initial + the contributions of its
        predecessors, one launch), proves the segments with those pre-states as public inputs, and `verify=True` additionally checks
        continuity (pre == prev.post) on the seals."""
        self._hal._check(self._hal._lib.zkh_session_set_chained(self.h, int(on), int(initial_state)))

    def set_journal(self, journal: Optional[bytes]) -> None:
        """SYN-S sessions: the bytes the guest commits (zeth: the block hash); the last seal binds their digest and `verify=True` checks
        the seals against them.  None: the default, the session's final state word."""
        self._hal._check(self._hal._lib.zkh_session_set_journal(self.h, None if journal is None else bytes(journal), 0 if journal is None else len(journal)))

    def set_resident_code(self, on: bool) -> None:
        """built-in circuits: keep the committed code group of each segment size resident per lane (default) or re-commit it per
        segment like upstream's SegmentProver.  Seals are byte-identical."""
        self._hal._lib.zkh_session_set_resident_code(self.h, int(on))

    def _specs(self, segments: Sequence[Segment], host_traces=None):
        C, np = self._C, self._np
        arr = (self._hal.SegmentSpec * len(segments))()
        keep = []
        u32p = C.POINTER(C.c_uint32)
        for i, (s, seg) in enumerate(zip(arr, segments)):
            s.po2, s.seed = seg.po2, seg.seed & (2**64 - 1)
            k = self._hal.noise_key(seg.noise_seed)          # None / 0: all-zero = a fresh OS key inside the library
            if k is not None:
                s.noise_key[:] = [int(w) for w in k]
            if seg.pub:
                p = np.asarray(seg.pub, dtype=np.uint32)
                keep.append(p)
                s.pub, s.n_pub = p.ctypes.data_as(u32p), p.size
            if host_traces is not None and host_traces[i] is not None:
                code, data, out = (np.ascontiguousarray(a, dtype=np.uint32) for a in host_traces[i])
                keep.extend((code, data, out))
                s.host_code, s.host_data, s.out_global = code.ctypes.data_as(u32p), data.ctypes.data_as(u32p), out.ctypes.data_as(u32p)
        return arr, keep

    def prove(self, segments: Sequence[Segment], join_tree: int = 0, join_po2: int = 18, join_noise_seed: int = 0,
              verify: bool = False, host_traces=None):
        """-> (CompositeReceipt, root SegmentReceipt or None, stats dict).  Segments use the protocol's ZK_CYCLES; noise_seed 0
        = fresh OS randomness per segment.  verify=True additionally runs `receipt.verify` inside the library
        (zkh_session_verify: every leaf seal against its control root and, with a join tree, the root seal + the claim tree
        recomputed from the leaf claims) and raises HalError if anything is rejected.
        host_traces: per segment None or (code, data, out_global) host arrays produced by the caller (upstream's flow: CPU
        preflight + witgen) — uploaded and sealed through zkh_prove_begin / accumulate / zkh_prove_finish inside the library."""
        C, np = self._C, self._np
        specs, keep = self._specs(segments, host_traces)
        info = self._hal.ProveInfo()
        _jk, jkp = self._hal._key_ptr(join_noise_seed)
        self._hal._check(self._hal._lib.zkh_session_prove(self.h, specs, len(segments), int(join_tree), join_po2, jkp, C.byref(info)))
        try:
            if verify:
                self._hal._check(self._hal._lib.zkh_session_verify(self.h, specs, C.byref(info), join_po2))
            out_size = int(self.desc[7])
            recs = []
            for i, seg in enumerate(segments):
                seal = np.ctypeslib.as_array(info.seals[i], shape=(info.seal_words[i],)).copy()
                recs.append(SegmentReceipt(seal=seal, index=seg.index, po2=seg.po2, output=seal[:out_size].copy()))
            root = None
            if info.root_seal:
                rs = np.ctypeslib.as_array(info.root_seal, shape=(info.root_seal_words,)).copy()
                recursive = bool(info.n_lifts)
                root = SegmentReceipt(seal=rs, index=0, po2=self._hal.fp_decode(int(rs[16])) if recursive else join_po2, output=rs[:16 if recursive else 24].copy())
            stats = {"wall_s": info.wall_s, "leaves_s": info.leaves_s, "join_s": info.join_s, "n_joins": int(info.n_joins),
                     "n_lifts": int(info.n_lifts), "lift_s": info.lift_s, "root_program": int(info.root_program),
                     "witgen_s_sum": info.witgen_s_sum, "seal_s_sum": info.seal_s_sum, "verified": bool(verify),
                     "n_retries": int(info.n_retries), "fold_tail_s": info.fold_tail_s, "fold_busy_s_sum": info.fold_busy_s_sum,
                     "streamed_fold": bool(info.streamed), "preflight_cpu_s_sum": info.preflight_cpu_s_sum, "trace_bytes": info.trace_bytes,
                     # join_tree = 2: the opening of the root's claim' (what a further join needs as witness for this receipt)
                     "root_core": np.array(list(info.root_core), dtype=np.uint32), "root_pre": int(info.root_pre), "root_post": int(info.root_post)}
            return CompositeReceipt(recs), root, stats
        finally:
            self._hal._lib.zkh_prove_info_free(C.byref(info))


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
# default) that takes the CLAIM DIGESTS of its two children as public inputs: they sit in its witness, are bound to
# its `out` globals by constraints, and the verifier of a succinct receipt checks every join's `out` against the claims
# of the receipts below it.  (The real join additionally verifies the child seals in-circuit; that is what lets
# upstream drop the tree and keep only the root.)  Same HAL ops, different circuit, tree-shaped schedule.
# ---------------------------------------------------------------------------------------------------------------
JOIN_PO2 = 18


@dataclass(frozen=True)
class JoinTask:
    level: int          # 1 = joins of leaves
    index: int          # position within the level
    left: int           # node index in the level below
    right: int
    device: int         # rank that runs it: the one that produced `left`
    right_owner: int    # rank that holds the right child (its receipt crosses the control plane when != device)


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
        tasks = [JoinTask(level, k, 2 * k, 2 * k + 1, owners[2 * k], owners[2 * k + 1]) for k in range(len(owners) // 2)]
        nxt = [t.device for t in tasks]
        if len(owners) % 2:
            nxt.append(owners[-1])
        levels.append(tasks)
        owners = nxt
    return levels


def receipt_claim(receipt: SegmentReceipt, circuit_desc, control_root) -> "np.ndarray":
    """Claim digest of a receipt: Poseidon2 over (out globals, po2, control root) — what a parent join commits to."""
    from . import hal as _hal
    return _hal.HostCircuit(circuit_desc).receipt_claim(receipt.seal, control_root)


def join_segment(task: JoinTask, left_claim, right_claim, join_po2: int = JOIN_PO2, noise_seed: Optional[int] = None) -> Segment:
    """The SYN-J segment of one join: public inputs = the two child claims; witness seed derived from them."""
    import hashlib
    import numpy as np
    pub = np.concatenate([np.asarray(left_claim, dtype=np.uint32), np.asarray(right_claim, dtype=np.uint32)])
    seed = int.from_bytes(hashlib.sha256(pub.astype("<u4").tobytes()).digest()[:8], "little")
    kw = {} if noise_seed is None else {"noise_seed": noise_seed}
    return Segment(index=task.index, po2=join_po2, seed=seed, pub=tuple(int(x) for x in pub), **kw)


def hash_pair(left, right) -> "np.ndarray":
    """Poseidon2 `hash_pair` of two 8-word digests on the host (the library's zkh_poseidon2_mix_host): what a Merkle node is,
    and what a P2-JOIN join proves about its two children's claims."""
    import ctypes as C
    import numpy as np
    from . import hal as _hal
    _hal.load_library()
    st = np.zeros(24, dtype=np.uint32)
    st[:8], st[8:16] = np.asarray(left, dtype=np.uint32), np.asarray(right, dtype=np.uint32)
    _hal._check(_hal._lib.zkh_poseidon2_mix_host(None, None, st.ctypes.data_as(C.POINTER(C.c_uint32)), 1))
    return st[:8].copy()


def _is_p2_join(join_desc) -> bool:
    return int(join_desc[13]) == 3


def node_claim(receipt: SegmentReceipt, circuit_desc, control_root, is_leaf: bool) -> "np.ndarray":
    """Claim of a node of the join tree.  Leaves: `receipt_claim`.  P2-JOIN joins: the parent digest the circuit constrains to
    hash_pair(left, right) — `out[0..8)`; SYN-J joins (round 2): `receipt_claim` of the join receipt."""
    import numpy as np
    if not is_leaf and _is_p2_join(circuit_desc):
        return np.asarray(receipt.seal[:8], dtype=np.uint32).copy()
    return receipt_claim(receipt, circuit_desc, control_root)


def fold_claims(claims) -> "np.ndarray":
    """Root of the claim tree over `claims` along `join_schedule` (an unpaired last node is carried up unchanged)."""
    level = [c for c in claims]
    while len(level) > 1:
        nxt = [hash_pair(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return level[0]


@dataclass
class SuccinctReceipt:
    """Root join receipt + the leaves (+ optionally the joins below the root).

    With P2-JOIN joins (zeth_amd/circuits/p2_join.py) every join CONSTRAINS parent = hash_pair(claim_left, claim_right), so
    the join tree is a Merkle tree of claims: `verify` needs only the root receipt and the leaf receipts — it recomputes the
    claim tree on the host and compares its root with the root receipt's public output; the joins below the root can be
    dropped (`compact`).  (The synthetic join does not verify the child SEALS in-circuit — upstream's does, which is what lets
    it drop the leaves as well; declared.)  With the round-2 SYN-J joins the whole tree is kept and walked."""
    root: SegmentReceipt
    joins: List[List[SegmentReceipt]]
    leaves: List[SegmentReceipt]
    n_assumptions: int = 0          # the LAST n_assumptions leaves are coprocessor (keccak) receipts of another circuit

    def compact(self) -> "SuccinctReceipt":
        return SuccinctReceipt(root=self.root, joins=[], leaves=self.leaves, n_assumptions=self.n_assumptions)

    def verify(self, segment_desc, join_desc, leaf_root=None, join_root=None, assumption_desc=None, assumption_root=None) -> None:
        """leaf_root / join_root / assumption_root: the expected control roots (None = the shipped table, or {po2: root}).
        With assumption leaves (upstream: the keccak receipts a succinct receipt `resolve`s; here they are further leaves of
        the claim tree, verified with their own circuit) `assumption_desc` is required."""
        import numpy as np
        from .prover import shipped_control_root

        def root_of(desc, given, po2):
            r = _root_for(given, po2)
            return shipped_control_root(desc, po2) if r is None else r
        if self.n_assumptions and assumption_desc is None:
            raise ValueError("succinct receipt carries assumption leaves but no circuit was given for them")
        n_seg = len(self.leaves) - self.n_assumptions
        leaf_claims = []
        for k, s in enumerate(self.leaves):
            d, given = (segment_desc, leaf_root) if k < n_seg else (assumption_desc, assumption_root)
            r = root_of(d, given, s.po2)
            s.verify(d, r)
            leaf_claims.append(receipt_claim(s, d, r))
        if len(self.leaves) == 1:
            top = self.leaves[0]
            if (self.root.index != top.index or self.root.po2 != top.po2
                    or not np.array_equal(np.asarray(self.root.seal, dtype=np.uint32), np.asarray(top.seal, dtype=np.uint32))):
                raise ValueError("root is not the top of the verified tree")
            return
        if _is_p2_join(join_desc):
            # the root seal is a valid P2-JOIN proof, and what it hashed is the top of the claim tree over the verified leaves
            self.root.verify(join_desc, root_of(join_desc, join_root, self.root.po2))
            level = leaf_claims
            while len(level) > 2:
                nxt = [hash_pair(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
                if len(level) % 2:
                    nxt.append(level[-1])
                level = nxt
            out = np.asarray(self.root.seal[:24], dtype=np.uint32)
            if not np.array_equal(out[8:16], level[0]) or not np.array_equal(out[16:24], level[1]):
                raise ValueError("root join does not commit to the claim tree of these leaves")
            if not np.array_equal(out[:8], hash_pair(level[0], level[1])):
                raise ValueError("root join's parent claim is not hash_pair of its children")     # implied by the seal; cheap to restate
            # joins that were kept are checked too (they are not needed)
            for lvl in self.joins:
                for j in lvl:
                    j.verify(join_desc, root_of(join_desc, join_root, j.po2))
            return
        # ---- SYN-J joins (round 2): walk the whole tree ----
        shape = [len(t) for t in join_schedule(len(self.leaves), 1)]
        if [len(lvl) for lvl in self.joins] != shape:
            raise ValueError(f"join tree has levels {[len(lvl) for lvl in self.joins]}, expected {shape}")
        nodes = list(zip(self.leaves, leaf_claims))
        for lvl, tasks in zip(self.joins, join_schedule(len(self.leaves), 1)):
            nxt = []
            for j, t in zip(lvl, tasks):
                jr = root_of(join_desc, join_root, j.po2)
                j.verify(join_desc, jr)
                want = np.concatenate([nodes[t.left][1], nodes[t.right][1]])
                if not np.array_equal(np.asarray(j.seal[4:4 + want.size], dtype=np.uint32), want):
                    raise ValueError(f"join (level {t.level}, index {t.index}) does not commit to the claims of its children")
                nxt.append((j, receipt_claim(j, join_desc, jr)))
            if len(nodes) % 2:
                nxt.append(nodes[-1])
            nodes = nxt
        # the stored root must BE the top of the tree that was just verified — compared by content (a receipt that went
        # through a container round trip is a different object)
        top = nodes[0][0]
        if (self.root.index != top.index or self.root.po2 != top.po2
                or not np.array_equal(np.asarray(self.root.seal, dtype=np.uint32), np.asarray(top.seal, dtype=np.uint32))):
            raise ValueError("root is not the top of the verified tree")


class JoinExecutor:
    """Runs `join_schedule` on this rank (BASELINE config 5, `ProverImpl::{lift, join}` upstream).  World size 1: every
    task is local.  World size G: a task runs on `task.device`; when the right child lives elsewhere its receipt is
    sent over the control plane (torch.distributed point-to-point objects on gloo — ~0.25 MB, no data-path collective).

    `claim_of(receipt, is_leaf) -> 8 words`, `prove_join(Segment) -> SegmentReceipt` are supplied by the caller (on a GPU
    rank: SegmentProver bound to the SYN-J circuit)."""

    def __init__(self, prove_join: Callable[[Segment], SegmentReceipt], claim_of: Callable[[SegmentReceipt, bool], "np.ndarray"],
                 rank: int = 0, world_size: int = 1, join_po2: int = JOIN_PO2, noise_seed: Optional[int] = None,
                 send=None, recv=None):
        self.prove_join, self.claim_of = prove_join, claim_of
        self.rank, self.world_size, self.join_po2, self.noise_seed = rank, world_size, join_po2, noise_seed
        self._send, self._recv = send, recv
        if world_size > 1 and (send is None or recv is None):
            import torch.distributed as dist

            def _s(obj, dst):
                dist.send_object_list([obj], dst=dst)

            def _r(src):
                buf = [None]
                dist.recv_object_list(buf, src=src)
                return buf[0]
            self._send, self._recv = _s, _r

    def run(self, n_leaves: int, local_leaves: dict):
        """local_leaves: {leaf index: SegmentReceipt} for the leaves this rank produced (index mod G == rank).
        -> (joins_done_here: {(level, index): SegmentReceipt}, root receipt or None when it lives on another rank)."""
        nodes = {i: (r, True) for i, r in local_leaves.items()}          # node index in the current level -> (receipt, is_leaf)
        n_nodes = n_leaves
        done = {}
        for tasks in join_schedule(n_leaves, self.world_size):
            # exchange step of the level, in task order on every rank (pairs up sends and receives without deadlock)
            right = {}
            for t in tasks:
                if t.right_owner == t.device:
                    continue
                if self.rank == t.right_owner:
                    self._send(nodes[t.right], t.device)
                elif self.rank == t.device:
                    right[t.index] = self._recv(t.right_owner)
            nxt = {}
            for t in tasks:
                if t.device != self.rank:
                    continue
                l_rec, l_leaf = nodes[t.left]
                r_rec, r_leaf = right[t.index] if t.index in right else nodes[t.right]
                seg = join_segment(t, self.claim_of(l_rec, l_leaf), self.claim_of(r_rec, r_leaf), self.join_po2, self.noise_seed)
                j = self.prove_join(seg)
                done[(t.level, t.index)] = j
                nxt[t.index] = (j, False)
            if n_nodes % 2 and (n_nodes - 1) in nodes:                    # the unpaired last node is carried up where it lives
                nxt[n_nodes // 2] = nodes[n_nodes - 1]
            nodes, n_nodes = nxt, (n_nodes + 1) // 2
        root = nodes.get(0, (None, False))[0] if n_nodes == 1 else None
        return done, root


def prove_succinct(leaves: Sequence[SegmentReceipt], prove_join: Callable[[Segment], SegmentReceipt],
                   claim_of: Callable[[SegmentReceipt, bool], "np.ndarray"], join_po2: int = JOIN_PO2,
                   noise_seed: Optional[int] = None) -> "SuccinctReceipt":
    """Single-rank driver of the join tree (every rank of `join_schedule` collapses onto the caller)."""
    ex = JoinExecutor(prove_join, claim_of, 0, 1, join_po2, noise_seed)
    done, root = ex.run(len(leaves), dict(enumerate(leaves)))
    sched = join_schedule(len(leaves), 1)
    joins = [[done[(t.level, t.index)] for t in tasks] for tasks in sched]
    return SuccinctReceipt(root=root if root is not None else leaves[0], joins=joins, leaves=list(leaves))
