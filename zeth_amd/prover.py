"""Segment prover — the host-side mirror of what `default_prover().prove(env, elf)` runs per segment.

Reference call site: /root/reference/crates/host/src/lib.rs:137 (`BlockProcessor::prove`), which (through the
un-vendored risc0-zkvm 3.0.3 `ProverServer::prove_session` -> `prove_segment`, /root/reference/Cargo.lock:5418)
seals every 2^po2-cycle segment independently and returns `SegmentReceipt{seal, index, hashfn, claim}`.
Here `SegmentProver.prove_segment` does the same for one segment on one MI355X via `zkh_prove_segment`
(zeth_amd/csrc/prover.hip).  Because the rv32im circuit and the guest execution trace cannot be obtained
offline, a segment is described by a `Segment` (po2 + witness seeds for the SYN-AIR stand-in circuit).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import hal as _hal
from .circuits import syn_air

DEFAULT_SEGMENT_PO2 = 20      # upstream default segment_limit_po2 (lib.rs:132-135 passes None -> 20)


@dataclass(frozen=True)
class Segment:
    """One continuation segment: index within the session, size, and the synthetic witness seeds."""
    index: int
    po2: int = DEFAULT_SEGMENT_PO2
    seed: int = 0x5EED0000
    noise_seed: int = 0x2E80
    zk_cycles: int = _hal.ZK_CYCLES


@dataclass
class SegmentReceipt:
    """`risc0_zkvm::SegmentReceipt` analogue: seal words + the metadata upstream carries."""
    seal: np.ndarray
    index: int
    po2: int
    hashfn: str = "poseidon2"
    output: Optional[np.ndarray] = None      # `out` globals (the claim-bearing public outputs)

    def seal_bytes(self) -> bytes:
        return np.asarray(self.seal, dtype="<u4").tobytes()

    def verify(self, circuit_desc) -> None:
        """`Receipt::verify` for this segment (/root/reference/crates/host/src/bin/cli.rs:103): host-side, no GPU needed.
        Raises `zeth_amd.hal.HalError` (the VerificationError analogue) if the seal is rejected."""
        hc = _hal.HostCircuit(circuit_desc)
        hc.verify_segment(self.seal)
        if int(self.seal[4]) != self.po2:
            raise _hal.HalError("receipt metadata does not match the seal (po2)")


class SegmentProver:
    """`SegmentProverImpl<H, C>` analogue bound to one HipHal (one GPU)."""

    def __init__(self, hal: "_hal.HipHal", circuit_desc=None):
        self.hal = hal
        self.circuit = hal.load_circuit(syn_air.syn_a() if circuit_desc is None else circuit_desc)
        h = C.c_void_p()
        _hal._check(_hal._lib.zkh_prover_create(hal.ctx, self.circuit.h, C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _hal._lib is not None:
            _hal._lib.zkh_prover_destroy(h)

    def group_sizes(self):
        from .circuits.desc import Circuit
        return Circuit.parse(self.circuit.desc).group_sizes

    def witgen(self, seg: Segment):
        """Witness generation (code + data traces resident in HBM) — reported separately from the seal."""
        wa, wc, wd = self.group_sizes()
        n = 1 << seg.po2
        code = self.hal.alloc_elem("code", wc * n)
        data = self.hal.alloc_elem("data", wd * n)
        out = self.hal.syn_witgen(self.circuit, seg.po2, seg.zk_cycles, seg.seed, seg.noise_seed, code, data)
        return code, data, out

    def seal(self, seg: Segment, code, data, out_global) -> SegmentReceipt:
        """Steps 3-7 of SURVEY.md §3.2 on traces already resident in HBM: the timed unit of work."""
        out = np.ascontiguousarray(out_global, dtype=np.uint32)
        seal_p = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        _hal._check(_hal._lib.zkh_prove_segment(self.h, seg.po2, seg.zk_cycles, seg.noise_seed, code.h, data.h,
                                                out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(seal_p), C.byref(n)))
        seal = np.ctypeslib.as_array(seal_p, shape=(n.value,)).copy()
        _hal._lib.zkh_free_seal(seal_p)
        return SegmentReceipt(seal=seal, index=seg.index, po2=seg.po2, output=out.copy())

    def prove_segment(self, seg: Segment) -> SegmentReceipt:
        code, data, out = self.witgen(seg)
        return self.seal(seg, code, data, out)
