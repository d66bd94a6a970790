"""bincode codec for the receipts zeth gets back from `default_prover().prove(env, elf)` (row f3 of SURVEY.md §8).

`/root/reference/crates/host/src/lib.rs:137-141` returns `ProveInfo.receipt`; `/root/reference/crates/host/src/bin/cli.rs:103-106`
calls `receipt.verify(image_id)` and reads `receipt.journal.bytes`.  On the wire and on disk that `Receipt` is bincode 1.x in
its default configuration — little-endian fixed-width integers, `u64` lengths for `Vec` / `String`, `u32` variant indices for
enums, one tag byte for `Option`, no framing for structs, tuples and fixed arrays (risc0-zkvm 3.0.3 `src/receipt.rs` and
`src/receipt/{composite,segment,succinct}.rs`, risc0-binfmt `SystemState` / `ExitCode`, un-vendored:
/root/reference/Cargo.lock:5418).  The TYPE LAYOUTS below are RECALLED from those sources, not read from them (no crate is
available offline): field order and enum variant order are the parts a maintainer must re-check, and
`tools/check_upstream_receipt.py <receipt.bin>` is the one-command check — it decodes a real receipt with this table,
re-encodes it and compares byte for byte; the first offset that disagrees names the field.

A schema is data (`Struct`, `Enum`, `Vec`, ...); `encode(schema, value)` / `decode(schema, bytes)` are generic.  Values are plain
Python: dicts for structs, `(variant_name, payload)` for enums, lists, `bytes`, `str`, ints, `None` for an absent Option.
Pure host code; no GPU, no library call.
"""
from __future__ import annotations

import struct
from typing import Any, List, Optional, Tuple

DIGEST_WORDS = 8


# ---------------------------------------------------------------------------------------------------------------
# schema nodes
# ---------------------------------------------------------------------------------------------------------------
class Int:
    def __init__(self, fmt: str, name: str):
        self.fmt, self.name, self.size = fmt, name, struct.calcsize(fmt)


U8, U32, U64 = Int("<B", "u8"), Int("<I", "u32"), Int("<Q", "u64")


class Bytes:            # Vec<u8>: u64 length + raw bytes
    name = "Vec<u8>"


class String:           # String: u64 length + utf-8
    name = "String"


class Vec:
    def __init__(self, item):
        self.item = item


class Opt:
    def __init__(self, item):
        self.item = item


class Array:            # [T; n]: no length prefix
    def __init__(self, item, n: int):
        self.item, self.n = item, n


class Struct:
    def __init__(self, name: str, fields: List[Tuple[str, Any]]):
        self.name, self.fields = name, fields


class Enum:
    """variants: [(name, payload schema or None)] in declaration order (bincode writes the index as u32)"""

    def __init__(self, name: str, variants: List[Tuple[str, Any]]):
        self.name, self.variants = name, variants


class Lazy:             # a recursive type: resolved on first use
    def __init__(self, thunk):
        self.thunk, self._v = thunk, None

    def get(self):
        if self._v is None:
            self._v = self.thunk()
        return self._v


class Never:            # an uninhabited type (`Input`): can be neither encoded nor decoded
    name = "!"


class CodecError(ValueError):
    pass


def encode(schema, value) -> bytes:
    out = bytearray()
    _enc(schema, value, out, "")
    return bytes(out)


def _enc(s, v, out: bytearray, path: str) -> None:
    if isinstance(s, Lazy):
        return _enc(s.get(), v, out, path)
    if isinstance(s, Int):
        if not isinstance(v, int) or not 0 <= v < 1 << (8 * s.size):
            raise CodecError(f"{path}: {v!r} is not a {s.name}")
        out += struct.pack(s.fmt, v)
    elif isinstance(s, Bytes) or s is Bytes:
        b = bytes(v)
        out += struct.pack("<Q", len(b)) + b
    elif isinstance(s, String) or s is String:
        b = v.encode("utf-8")
        out += struct.pack("<Q", len(b)) + b
    elif isinstance(s, Vec):
        out += struct.pack("<Q", len(v))
        if isinstance(s.item, Int) and s.item is U32:
            try:
                out += struct.pack(f"<{len(v)}I", *[int(x) for x in v])       # seals: hundreds of thousands of words
            except struct.error as e:
                raise CodecError(f"{path}: not u32 words ({e})")
        else:
            for i, x in enumerate(v):
                _enc(s.item, x, out, f"{path}[{i}]")
    elif isinstance(s, Opt):
        if v is None:
            out += b"\x00"
        else:
            out += b"\x01"
            _enc(s.item, v, out, path + "?")
    elif isinstance(s, Array):
        if len(v) != s.n:
            raise CodecError(f"{path}: expected {s.n} items, got {len(v)}")
        for i, x in enumerate(v):
            _enc(s.item, x, out, f"{path}[{i}]")
    elif isinstance(s, Struct):
        missing = [f for f, _ in s.fields if f not in v]
        if missing:
            raise CodecError(f"{path or s.name}: missing field(s) {missing}")
        for f, fs in s.fields:
            _enc(fs, v[f], out, f"{path}.{f}" if path else f"{s.name}.{f}")
    elif isinstance(s, Enum):
        name, payload = v
        idx = next((i for i, (n, _) in enumerate(s.variants) if n == name), None)
        if idx is None:
            raise CodecError(f"{path}: {s.name} has no variant '{name}'")
        out += struct.pack("<I", idx)
        if s.variants[idx][1] is not None:
            _enc(s.variants[idx][1], payload, out, f"{path}::{name}")
    elif isinstance(s, Never) or s is Never:
        raise CodecError(f"{path}: a value of an uninhabited type")
    else:
        raise CodecError(f"{path}: unknown schema node {s!r}")


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p = memoryview(data), 0

    def take(self, n: int, path: str) -> memoryview:
        if n < 0 or self.p + n > len(self.d):
            raise CodecError(f"{path}: needs {n} bytes at offset {self.p}, {len(self.d) - self.p} left")
        v = self.d[self.p:self.p + n]
        self.p += n
        return v


def decode(schema, data: bytes, allow_trailing: bool = False):
    r = _Reader(data)
    v = _dec(schema, r, "")
    if r.p != len(data) and not allow_trailing:
        raise CodecError(f"{len(data) - r.p} trailing bytes after offset {r.p}")
    return v


def _dec(s, r: _Reader, path: str):
    if isinstance(s, Lazy):
        return _dec(s.get(), r, path)
    if isinstance(s, Int):
        return struct.unpack(s.fmt, r.take(s.size, path))[0]
    if isinstance(s, Bytes) or s is Bytes:
        n = struct.unpack("<Q", r.take(8, path))[0]
        return bytes(r.take(n, path))
    if isinstance(s, String) or s is String:
        n = struct.unpack("<Q", r.take(8, path))[0]
        try:
            return bytes(r.take(n, path)).decode("utf-8")
        except UnicodeDecodeError:
            raise CodecError(f"{path}: not utf-8 (offset {r.p - n})")
    if isinstance(s, Vec):
        n = struct.unpack("<Q", r.take(8, path))[0]
        if isinstance(s.item, Int) and s.item is U32:
            return list(struct.unpack(f"<{n}I", r.take(4 * n, path)))
        if n > len(r.d):                     # every item is at least one byte: a corrupt length cannot allocate the world
            raise CodecError(f"{path}: length {n} at offset {r.p - 8} exceeds the input")
        return [_dec(s.item, r, f"{path}[{i}]") for i in range(n)]
    if isinstance(s, Opt):
        tag = r.take(1, path)[0]
        if tag > 1:
            raise CodecError(f"{path}: Option tag {tag} at offset {r.p - 1}")
        return _dec(s.item, r, path + "?") if tag else None
    if isinstance(s, Array):
        return [_dec(s.item, r, f"{path}[{i}]") for i in range(s.n)]
    if isinstance(s, Struct):
        return {f: _dec(fs, r, f"{path}.{f}" if path else f"{s.name}.{f}") for f, fs in s.fields}
    if isinstance(s, Enum):
        idx = struct.unpack("<I", r.take(4, path))[0]
        if idx >= len(s.variants):
            raise CodecError(f"{path}: {s.name} variant index {idx} at offset {r.p - 4} (has {len(s.variants)})")
        name, ps = s.variants[idx]
        return (name, None if ps is None else _dec(ps, r, f"{path}::{name}"))
    if isinstance(s, Never) or s is Never:
        raise CodecError(f"{path}: a value of an uninhabited type at offset {r.p}")
    raise CodecError(f"{path}: unknown schema node {s!r}")


# ---------------------------------------------------------------------------------------------------------------
# the types (RECALLED: see the module docstring)
# ---------------------------------------------------------------------------------------------------------------
Digest = Array(U32, DIGEST_WORDS)                      # risc0_zkp::core::digest::Digest([u32; 8]), non-human-readable form


def MaybePruned(t):                                     # risc0_binfmt / risc0_zkvm::MaybePruned<T>
    return Enum("MaybePruned", [("Value", t), ("Pruned", Digest)])


SystemState = Struct("SystemState", [("pc", U32), ("merkle_root", Digest)])
ExitCode = Enum("ExitCode", [("Halted", U32), ("Paused", U32), ("SystemSplit", None), ("SessionLimit", None)])
Input = Never()                                         # `Input` has no inhabitant: the Option around it is always None
Assumption = Struct("Assumption", [("claim", Digest), ("control_root", Digest)])
Assumptions = Vec(MaybePruned(Assumption))              # struct Assumptions(pub Vec<MaybePruned<Assumption>>): a newtype adds nothing
Output = Struct("Output", [("journal", MaybePruned(Bytes())), ("assumptions", MaybePruned(Assumptions))])
ReceiptClaim = Struct("ReceiptClaim", [("pre", MaybePruned(SystemState)), ("post", MaybePruned(SystemState)), ("exit_code", ExitCode),
                                       ("input", MaybePruned(Opt(Input))), ("output", MaybePruned(Opt(Output)))])
SegmentReceipt = Struct("SegmentReceipt", [("seal", Vec(U32)), ("index", U32), ("hashfn", String()), ("verifier_parameters", Digest),
                                           ("claim", ReceiptClaim)])
MerkleProof = Struct("MerkleProof", [("index", U32), ("digests", Vec(Digest))])


def SuccinctReceipt(claim):
    return Struct("SuccinctReceipt", [("seal", Vec(U32)), ("control_id", Digest), ("claim", MaybePruned(claim)), ("hashfn", String()),
                                      ("verifier_parameters", Digest), ("control_inclusion_proof", MerkleProof)])


def Groth16Receipt(claim):
    return Struct("Groth16Receipt", [("seal", Bytes()), ("claim", MaybePruned(claim)), ("verifier_parameters", Digest)])


def FakeReceipt(claim):
    return Struct("FakeReceipt", [("claim", MaybePruned(claim))])


Unknown = Never()                                       # claims of unresolved assumptions are always Pruned
InnerAssumptionReceipt = Enum("InnerAssumptionReceipt", [("Composite", Lazy(lambda: CompositeReceipt)), ("Succinct", SuccinctReceipt(Unknown)),
                                                         ("Groth16", Groth16Receipt(Unknown)), ("Fake", FakeReceipt(Unknown))])
CompositeReceipt = Struct("CompositeReceipt", [("segments", Vec(SegmentReceipt)), ("assumption_receipts", Vec(InnerAssumptionReceipt)),
                                               ("verifier_parameters", Digest)])
InnerReceipt = Enum("InnerReceipt", [("Composite", CompositeReceipt), ("Succinct", SuccinctReceipt(ReceiptClaim)),
                                     ("Groth16", Groth16Receipt(ReceiptClaim)), ("Fake", FakeReceipt(ReceiptClaim))])
Journal = Struct("Journal", [("bytes", Bytes())])
ReceiptMetadata = Struct("ReceiptMetadata", [("verifier_parameters", Digest)])
Receipt = Struct("Receipt", [("inner", InnerReceipt), ("journal", Journal), ("metadata", ReceiptMetadata)])

SCHEMAS = {"Receipt": Receipt, "SegmentReceipt": SegmentReceipt, "CompositeReceipt": CompositeReceipt, "SuccinctReceipt": SuccinctReceipt(ReceiptClaim),
           "ReceiptClaim": ReceiptClaim}

ZERO_DIGEST = [0] * DIGEST_WORDS


# ---------------------------------------------------------------------------------------------------------------
# this repository's receipts in upstream's containers
# ---------------------------------------------------------------------------------------------------------------
def claim_placeholder(claim_digest, exit_code=("Halted", 0)) -> dict:
    """A `ReceiptClaim` for a receipt of this repository's circuits.  They have no rv32im `SystemState`: what binds a seal is its
    claim digest (zkh_receipt_claim = Poseidon2(out ‖ po2 ‖ control root)), carried here as the PRUNED `post` state so that the
    container round-trips it; `pre`, `input` and `output` are pruned to zero.  Declared: upstream's verifier would not accept
    this claim — the container layout is what is exercised, not the claim semantics."""
    return {"pre": ("Pruned", ZERO_DIGEST), "post": ("Pruned", [int(w) for w in claim_digest]), "exit_code": exit_code,
            "input": ("Pruned", ZERO_DIGEST), "output": ("Pruned", ZERO_DIGEST)}


def segment_receipt_value(seal, index: int, claim_digest, verifier_parameters=None, hashfn: str = "poseidon2") -> dict:
    """claim_digest: 8 words (a circuit without ReceiptClaim semantics: the placeholder above), or a dict — a full `ReceiptClaim` value
    (zeth_amd.host.ReceiptClaim.to_codec_value: SYN-S segments carry real states, exit code and output digest)"""
    return {"seal": [int(w) for w in seal], "index": int(index), "hashfn": hashfn,
            "verifier_parameters": [int(w) for w in (verifier_parameters if verifier_parameters is not None else ZERO_DIGEST)],
            "claim": claim_digest if isinstance(claim_digest, dict) else claim_placeholder(claim_digest)}


def composite_receipt_bytes(segments, journal: bytes = b"", verifier_parameters=None) -> bytes:
    """[(seal words, index, claim digest)] -> the bincode of `Receipt{inner: Composite{..}, journal, metadata}`"""
    vp = [int(w) for w in (verifier_parameters if verifier_parameters is not None else ZERO_DIGEST)]
    inner = ("Composite", {"segments": [segment_receipt_value(s, i, c, vp) for s, i, c in segments], "assumption_receipts": [],
                           "verifier_parameters": vp})
    return encode(Receipt, {"inner": inner, "journal": {"bytes": bytes(journal)}, "metadata": {"verifier_parameters": vp}})


def succinct_receipt_bytes(seal, control_id, claim_digest, journal: bytes = b"", proof_index: int = 0, proof_digests=(), verifier_parameters=None) -> bytes:
    """a recursion root (`RecReceipt`) in `Receipt{inner: Succinct{..}}`: control_id = the program's control root, the inclusion proof =
    its membership path in the allowed-programs tree"""
    vp = [int(w) for w in (verifier_parameters if verifier_parameters is not None else ZERO_DIGEST)]
    inner = ("Succinct", {"seal": [int(w) for w in seal], "control_id": [int(w) for w in control_id], "claim": ("Value", claim_placeholder(claim_digest)),
                          "hashfn": "poseidon2", "verifier_parameters": vp,
                          "control_inclusion_proof": {"index": int(proof_index), "digests": [[int(w) for w in d] for d in proof_digests]}})
    return encode(Receipt, {"inner": inner, "journal": {"bytes": bytes(journal)}, "metadata": {"verifier_parameters": vp}})


def describe(value, depth: int = 0, max_depth: int = 4) -> str:
    """a short structural summary of a decoded value (what tools/check_upstream_receipt.py prints)"""
    pad = "  " * depth
    if isinstance(value, dict):
        if depth >= max_depth:
            return pad + "{...}"
        return "\n".join(f"{pad}{k}: " + (describe(v, depth + 1, max_depth).lstrip() if not isinstance(v, (dict, list, tuple)) or _short(v) else "\n" + describe(v, depth + 1, max_depth))
                         for k, v in value.items())
    if isinstance(value, tuple) and len(value) == 2 and isinstance(value[0], str):
        inner = value[1]
        if inner is None:
            return pad + value[0]
        return pad + value[0] + ("(" + describe(inner, 0, max_depth) + ")" if _short(inner) else ":\n" + describe(inner, depth + 1, max_depth))
    if isinstance(value, list):
        if len(value) == DIGEST_WORDS and all(isinstance(x, int) for x in value):
            return pad + "digest " + "".join(f"{x:08x}" for x in value)
        if value and all(isinstance(x, int) for x in value):
            return pad + f"[{len(value)} words]"
        if depth >= max_depth:
            return pad + f"[{len(value)} items]"
        return "\n".join(f"{pad}[{i}]\n" + describe(v, depth + 1, max_depth) for i, v in enumerate(value[:4])) + (f"\n{pad}... {len(value) - 4} more" if len(value) > 4 else "") if value else pad + "[]"
    if isinstance(value, bytes):
        return pad + f"{len(value)} bytes" + (f" ({value[:32].hex()})" if len(value) <= 32 else "")
    return pad + repr(value)


def _short(v) -> bool:
    if isinstance(v, list):
        return (len(v) == DIGEST_WORDS and all(isinstance(x, int) for x in v)) or not v or all(isinstance(x, int) for x in v)
    if isinstance(v, tuple):
        return v[1] is None or _short(v[1])
    return not isinstance(v, dict)
