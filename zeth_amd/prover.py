"""Segment prover — the host-side mirror of what `default_prover().prove(env, elf)` runs per segment.

Reference call site: /root/reference/crates/host/src/lib.rs:137 (`BlockProcessor::prove`), which (through the
un-vendored risc0-zkvm 3.0.3 `ProverServer::prove_session` -> `prove_segment`, /root/reference/Cargo.lock:5418)
seals every 2^po2-cycle segment independently and returns `SegmentReceipt{seal, index, hashfn, claim}`.
Here `SegmentProver.prove_segment` does the same for one segment on one MI355X via `zkh_prove_segment`
(zeth_amd/csrc/prover.hip).  Because the rv32im circuit and the guest execution trace cannot be obtained
offline, a segment is described by a `Segment` (po2 + witness seeds for the SYN-AIR stand-in circuit).

Two ways a witness reaches the seal:
  * `prove_segment(seg)`            — SYN witness generated on the device (zkh_syn_witgen), then sealed;
  * `seal_host_witness(seg, ...)`   — code/data traces produced on the HOST (upstream's flow: CPU preflight + witgen),
                                      uploaded from pinned memory (zkh_host_alloc + zkh_write_async), sealed through
                                      zkh_prove_begin -> accum -> zkh_prove_finish, the two halves upstream's
                                      SegmentProver drives `Prover` in.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from . import hal as _hal
from .circuits import syn_air

DEFAULT_SEGMENT_PO2 = 20      # upstream default segment_limit_po2 (lib.rs:132-135 passes None -> 20)
_CONTROL_ROOTS_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "circuits", "control_roots.json")


def fresh_noise_seed() -> int:
    """Seed of the zero-knowledge blinding rows: OS randomness, like upstream (which fills the last ZK_CYCLES rows of
    every trace column from an OS RNG).  Tests, benchmarks and golden fixtures pass an explicit seed instead."""
    return int.from_bytes(os.urandom(32), "little") | 1          # 256 bits: the ChaCha12 key of the blinding stream (csrc/noise.h); never 0


@dataclass(frozen=True)
class Segment:
    """One continuation segment: index within the session, size, and the synthetic witness seeds."""
    index: int
    po2: int = DEFAULT_SEGMENT_PO2
    seed: int = 0x5EED0000
    noise_seed: int = field(default_factory=fresh_noise_seed)
    zk_cycles: int = _hal.ZK_CYCLES
    pub: Tuple[int, ...] = ()        # public input words (Montgomery) for circuits with OUTPUT_SIZE > 4 (SYN-J joins)


def desc_key(desc) -> str:
    from .circuits.codegen import desc_hash64
    return f"{desc_hash64(np.asarray(desc, dtype=np.uint32)):016x}"


_shipped_roots: Optional[Dict[str, Dict[str, list]]] = None


def shipped_control_root(desc, po2: int) -> Optional[np.ndarray]:
    """Control root registered for (circuit, po2) at the protocol's ZK_CYCLES — the control-ID table upstream compiles
    into the verifier.  Generated on a GPU box by `python -m zeth_amd.prover` (this module's main)."""
    global _shipped_roots
    if _shipped_roots is None:
        try:
            with open(_CONTROL_ROOTS_JSON) as fh:
                _shipped_roots = json.load(fh)["roots"]
        except (OSError, ValueError, KeyError):
            _shipped_roots = {}
    words = _shipped_roots.get(desc_key(desc), {}).get(str(po2))
    return None if words is None else np.asarray(words, dtype=np.uint32)


@dataclass
class SegmentReceipt:
    """`risc0_zkvm::SegmentReceipt` analogue: seal words + the metadata upstream carries."""
    seal: np.ndarray
    index: int
    po2: int
    hashfn: str = "poseidon2"
    output: Optional[np.ndarray] = None      # `out` globals (the claim-bearing public outputs)
    control_root: Optional[np.ndarray] = None  # what the prover committed the code group to (informational; the
    #                                            verifier is given the EXPECTED root, it never trusts this field)

    def seal_bytes(self) -> bytes:
        return np.asarray(self.seal, dtype="<u4").tobytes()

    def to_words(self, circuit_desc, control_root) -> np.ndarray:
        """Receipt container (zkh_receipt_encode): what gets stored / shipped instead of upstream's bincode SegmentReceipt."""
        return _hal.HostCircuit(circuit_desc).receipt_encode(self.seal, self.index, control_root)

    @staticmethod
    def from_words(circuit_desc, blob) -> "SegmentReceipt":
        """Parse + integrity-check a container.  The control root inside is informational: `verify` takes the expected one."""
        hdr, seal = _hal.HostCircuit(circuit_desc).receipt_decode(blob)
        return SegmentReceipt(seal=seal, index=hdr["index"], po2=hdr["po2"], output=seal[:hdr["out_size"]].copy(),
                              control_root=hdr["control_root"])

    def verify(self, circuit_desc, control_root=None) -> None:
        """`Receipt::verify` for this segment (/root/reference/crates/host/src/bin/cli.rs:103): host-side, no GPU needed.
        `control_root`: the expected code commitment; None looks it up in the shipped table (zk_cycles = ZK_CYCLES).
        Raises `zeth_amd.hal.HalError` (the VerificationError analogue) if the seal is rejected."""
        desc = np.asarray(circuit_desc, dtype=np.uint32)
        if control_root is None:
            control_root = shipped_control_root(desc, self.po2)
            if control_root is None:
                raise _hal.HalError(f"verify_segment: no control root registered for circuit {desc_key(desc)} at po2 {self.po2}")
        hc = _hal.HostCircuit(desc)
        hc.verify_segment(self.seal, control_root)
        out_size = int(desc[7])
        if _hal.fp_decode(int(self.seal[out_size])) != self.po2:
            raise _hal.HalError("verify_segment: receipt metadata does not match the seal (po2)")
        if self.output is not None and not np.array_equal(np.asarray(self.output, dtype=np.uint32), self.seal[:out_size]):
            raise _hal.HalError("verify_segment: receipt output does not match the seal's output globals")


class SegmentProver:
    """`SegmentProverImpl<H, C>` analogue bound to one HipHal (one GPU)."""

    def __init__(self, hal: "_hal.HipHal", circuit_desc=None, resident_code_group: bool = False):
        """resident_code_group: keep the committed code (control) group of each segment size in HBM instead of re-committing
        it for every segment (zkh_prover_cache_code): the group is a function of (circuit, po2, zk_cycles) alone.  Off by
        default — upstream's SegmentProver recomputes it, and so does the benchmark's headline number."""
        self.hal = hal
        self.circuit = hal.load_circuit(syn_air.syn_a() if circuit_desc is None else circuit_desc)
        h = C.c_void_p()
        _hal._check(_hal._lib.zkh_prover_create(hal.ctx, self.circuit.h, C.byref(h)))
        self.h = h
        self.resident_code_group = resident_code_group
        self._resident: Dict[int, int] = {}                 # po2 -> zk_cycles of the resident code group
        self._roots: Dict[Tuple[int, int], np.ndarray] = {}
        self._group_sizes = tuple(int(x) for x in self.circuit.desc[3:6])       # (accum, code, data): desc header words

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        # a prover may hold device buffers (the resident code group): like Buffer, it is only released into a live context
        if h and _hal._lib is not None and getattr(self.hal, "ctx", None):
            _hal._lib.zkh_prover_destroy(h)

    def group_sizes(self):
        return self._group_sizes

    def out_size(self) -> int:
        return int(self.circuit.desc[7])

    def control_root(self, po2: int, zk_cycles: int = _hal.ZK_CYCLES) -> np.ndarray:
        """Merkle root of the committed SYN code group for (po2, zk_cycles): computed on the GPU once and cached."""
        key = (po2, zk_cycles)
        if key not in self._roots:
            root = np.zeros(8, dtype=np.uint32)
            _hal._check(_hal._lib.zkh_syn_control_root(self.h, po2, zk_cycles, root.ctypes.data_as(C.POINTER(C.c_uint32))))
            self._roots[key] = root
        return self._roots[key]

    def code_root(self, code, po2: int) -> np.ndarray:
        root = np.zeros(8, dtype=np.uint32)
        _hal._check(_hal._lib.zkh_code_root(self.h, code.h, po2, root.ctypes.data_as(C.POINTER(C.c_uint32))))
        return root

    def witgen(self, seg: Segment):
        """Witness generation (code + data traces resident in HBM) — reported separately from the seal."""
        wa, wc, wd = self.group_sizes()
        n = 1 << seg.po2
        # with this size's committed code group resident the trace is not needed again (kinds 1 / 2: the data generator does
        # not read it) -> code is None, and `seal` passes NULL
        held = self.resident_code_group and self._resident.get(seg.po2) == seg.zk_cycles and int(self.circuit.desc[13]) in (1, 2)
        code = None if held else self.hal.alloc_elem("code", wc * n)
        data = self.hal.alloc_elem("data", wd * n)
        out = self.hal.syn_witgen(self.circuit, seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed, code, data,
                                  pub=np.asarray(seg.pub, dtype=np.uint32) if seg.pub else None)
        return code, data, out

    def chain_contribution(self, seg: Segment) -> int:
        """What a SYN-C segment adds to the running state: its post-state when started from state 0 (the witness generator run
        with pre = 0; the executor's pass of a chained session, outside any proving clock)."""
        seeds = (C.c_uint64 * 1)(seg.seed & (2**64 - 1))
        po2s, out = np.array([seg.po2], dtype=np.uint32), np.zeros(1, dtype=np.uint32)
        _hal._check(_hal._lib.zkh_syn_chain_contributions(self.hal.ctx, self.circuit.h, seeds, po2s.ctypes.data_as(C.POINTER(C.c_uint32)), 1, seg.zk_cycles,
                                                          out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return int(out[0])

    def _take_seal(self, seal_p, n) -> np.ndarray:
        seal = np.ctypeslib.as_array(seal_p, shape=(n.value,)).copy()
        _hal._lib.zkh_free_seal(seal_p)
        return seal

    def _code_handle(self, seg: Segment, code):
        """The code operand of a seal: the trace, or NULL once this size's committed group is resident."""
        if not self.resident_code_group:
            return code.h
        if self._resident.get(seg.po2) != seg.zk_cycles:
            if code is None:
                raise _hal.HalError("seal: no code trace and no resident code group of this size")
            _hal._check(_hal._lib.zkh_prover_cache_code(self.h, seg.po2, code.h))
            self._resident[seg.po2] = seg.zk_cycles
        return None

    def seal(self, seg: Segment, code, data, out_global) -> SegmentReceipt:
        """Steps 3-7 of SURVEY.md §3.2 on traces already resident in HBM: the timed unit of work."""
        out = np.ascontiguousarray(out_global, dtype=np.uint32)
        seal_p = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        _k, kp = _hal._key_ptr(seg.noise_seed)
        _hal._check(_hal._lib.zkh_prove_segment(self.h, seg.po2, seg.zk_cycles, kp, self._code_handle(seg, code), data.h,
                                                out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(seal_p), C.byref(n)))
        return SegmentReceipt(seal=self._take_seal(seal_p, n), index=seg.index, po2=seg.po2, output=out.copy())

    def seal_with_accum(self, seg: Segment, code, data, out_global,
                        accumulate: Callable[[np.ndarray], "_hal.Buffer"]) -> SegmentReceipt:
        """The same seal through the circuit-agnostic halves: zkh_prove_begin (header, code, data -> mix challenges),
        `accumulate(mix_global) -> accum buffer` supplied by the caller (CircuitHal::accumulate), zkh_prove_finish."""
        out = np.ascontiguousarray(out_global, dtype=np.uint32)
        wa = self.group_sizes()[0]
        job = C.c_void_p()
        mix = np.zeros(max(1, int(self.circuit.desc[8])), dtype=np.uint32)
        _hal._check(_hal._lib.zkh_prove_begin(self.h, seg.po2, self._code_handle(seg, code), data.h, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              C.byref(job), mix.ctypes.data_as(C.POINTER(C.c_uint32))))
        try:
            accum = accumulate(mix[: int(self.circuit.desc[8])])
            if accum.size() != wa << seg.po2:
                raise _hal.HalError("accumulate returned a buffer of the wrong shape")
        except BaseException:
            _hal._lib.zkh_prove_abort(job)
            raise
        seal_p = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        _hal._check(_hal._lib.zkh_prove_finish(job, accum.h, C.byref(seal_p), C.byref(n)))
        return SegmentReceipt(seal=self._take_seal(seal_p, n), index=seg.index, po2=seg.po2, output=out.copy())

    def syn_accumulate(self, seg: Segment, data):
        """`accumulate` callback for the SYN-AIR family (zkh_syn_accum)."""
        wa = self.group_sizes()[0]

        def acc(mix_global):
            accum = self.hal.alloc_elem("accum", wa << seg.po2)
            self.hal.syn_accum(self.circuit, seg.po2, seg.zk_cycles, seg.noise_seed, data, mix_global, accum)
            return accum
        return acc

    def seal_host_witness(self, seg: Segment, host_code: np.ndarray, host_data: np.ndarray, out_global) -> SegmentReceipt:
        """Seal a segment whose code/data traces live in pinned HOST memory (hal.host_alloc views): enqueue both uploads
        on the context's stream (no host sync), then run the two-halves seal.  This is the PCIe-inclusive path."""
        code = self.hal.alloc_elem("code", host_code.size)
        data = self.hal.alloc_elem("data", host_data.size)
        self.hal.write_async(code, host_code)
        self.hal.write_async(data, host_data)
        return self.seal_with_accum(seg, code, data, out_global, self.syn_accumulate(seg, data))

    def prove_segment(self, seg: Segment) -> SegmentReceipt:
        code, data, out = self.witgen(seg)
        return self.seal(seg, code, data, out)


def _generate_control_roots(po2s=range(13, 25)) -> None:
    """`python -m zeth_amd.prover` on a GPU box: (re)generate circuits/control_roots.json for the shipped circuits."""
    from .circuits import codegen
    hal = _hal.HipHal(0)
    roots: Dict[str, Dict[str, list]] = {}
    names = {}
    for name, desc in codegen.shipped().items():
        if int(np.asarray(desc)[13]) not in (1, 2, 3):     # only circuits with a built-in code generator have a per-size control root
            continue                                      # (a RECURSION program's control root belongs to the program: zkh_rec_program_info)
        prover = SegmentProver(hal, desc)
        key = desc_key(desc)
        names[key] = name
        roots[key] = {str(p): [int(x) for x in prover.control_root(p)] for p in po2s}
    with open(_CONTROL_ROOTS_JSON, "w") as fh:
        json.dump({"generator": "python -m zeth_amd.prover (zkh_syn_control_root on the GPU), zk_cycles = 1994",
                   "library": _hal.load_library().zkh_version().decode(), "names": names, "roots": roots}, fh, indent=1)
    print(f"wrote {_CONTROL_ROOTS_JSON}: {len(roots)} circuits x {len(list(po2s))} sizes")


if __name__ == "__main__":
    _generate_control_roots()
