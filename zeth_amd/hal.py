"""Python mirror of `risc0_zkp::hal::{Hal, CircuitHal, Buffer}` over libzkhal_mi355x.so (ctypes, no torch types).

Method names and argument meaning follow the upstream trait (risc0-zkp 3.0.2 src/hal/mod.rs, un-vendored:
/root/reference/Cargo.lock:5393) so that tests read like upstream's cpu-vs-gpu HAL parity tests.  The product
path is the HIP library ONLY: importing this module without the built .so, or creating a HipHal without a
GPU, raises — there is no CPU fallback here (the CPU oracle lives in oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKH_LIBRARY") or os.path.join(_HERE, "libzkhal_mi355x.so")   # override: another build of the same ABI

INV_RATE, QUERIES, FRI_FOLD, FRI_MIN_DEGREE, ZK_CYCLES, CHECK_SIZE, EXT_SIZE, DIGEST_WORDS = 4, 50, 16, 256, 1994, 16, 4, 8
P = 2013265921                      # BabyBear
_RINV = pow(1 << 32, -1, P)


def fp_decode(word: int) -> int:
    """Montgomery word -> canonical residue (seal words are raw Montgomery `Elem`s, like upstream's)."""
    return (int(word) * _RINV) % P


def noise_key(noise_seed) -> Optional[np.ndarray]:
    """The 8-word blinding key of include/zkhal.h (BLINDING ROWS) from the `noise_seed` this package's API takes: an integer below
    2^256 (little-endian words; 0x2E80 -> (0x2E80, 0, ..)), 32 bytes, or 8 words; None / 0 -> None = NULL at the ABI = a fresh 256-bit
    key from the OS for that call (the product default)."""
    if noise_seed is None:
        return None
    if isinstance(noise_seed, (bytes, bytearray)):
        noise_seed = int.from_bytes(noise_seed, "little")
    if isinstance(noise_seed, (int, np.integer)):
        v = int(noise_seed)
        if v == 0:
            return None
        if not 0 < v < (1 << 256):
            raise ValueError("noise seed out of range (256 bits)")
        return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
    k = np.ascontiguousarray(noise_seed, dtype=np.uint32)
    if k.shape != (8,):
        raise ValueError("a noise key is 8 words")
    return k if k.any() else None


def _key_ptr(noise_seed):
    """-> (keep-alive array or None, ctypes pointer or None)"""
    k = noise_key(noise_seed)
    return k, (None if k is None else k.ctypes.data_as(C.POINTER(C.c_uint32)))


def fp_encode(x: int) -> int:
    return ((int(x) % P) << 32) % P

class SegmentSpec(C.Structure):
    """`zkh_segment` (include/zkhal.h): one segment of a session."""
    _fields_ = [("po2", C.c_uint32), ("seed", C.c_uint64), ("noise_key", C.c_uint32 * 8), ("pub", C.POINTER(C.c_uint32)), ("n_pub", C.c_size_t),
                ("host_code", C.POINTER(C.c_uint32)), ("host_data", C.POINTER(C.c_uint32)), ("out_global", C.POINTER(C.c_uint32))]


class ProveInfo(C.Structure):
    """`zkh_prove_info`: what zkh_session_prove returns."""
    _fields_ = [("n_segments", C.c_size_t), ("seals", C.POINTER(C.POINTER(C.c_uint32))), ("seal_words", C.POINTER(C.c_size_t)),
                ("root_seal", C.POINTER(C.c_uint32)), ("root_seal_words", C.c_size_t), ("n_joins", C.c_size_t),
                ("wall_s", C.c_double), ("leaves_s", C.c_double), ("join_s", C.c_double), ("witgen_s_sum", C.c_double), ("seal_s_sum", C.c_double),
                ("n_lifts", C.c_size_t), ("root_program", C.c_size_t), ("lift_s", C.c_double),
                ("n_retries", C.c_size_t), ("fold_tail_s", C.c_double), ("fold_busy_s_sum", C.c_double), ("streamed", C.c_int),
                ("preflight_cpu_s_sum", C.c_double), ("trace_bytes", C.c_double),
                ("root_core", C.c_uint32 * 8), ("root_pre", C.c_uint32), ("root_post", C.c_uint32)]


# every symbol include/zkhal.h declares: (restype, argtypes)
_sz, _u32, _u64, _vp, _i = C.c_size_t, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int
_u32p = C.POINTER(C.c_uint32)
_err = C.c_void_p    # heap error string (freed with zkh_free_error)


class ProfRec(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("calls", C.c_uint64), ("total_ms", C.c_double), ("alg_bytes", C.c_double)]


ABI = {
    "zkh_free_error": (None, [_vp]),
    "zkh_version": (C.c_char_p, []),
    "zkh_ctx_create": (_err, [_i, C.c_char_p, C.POINTER(_vp)]),
    "zkh_ctx_destroy": (None, [_vp]),
    "zkh_sync": (_err, [_vp]),
    "zkh_ctx_memory": (None, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "zkh_ctx_trim": (_err, [_vp]),
    "zkh_ctx_stream": (_vp, [_vp]),
    "zkh_poseidon2_set_constants": (_err, [_vp, _u32p, _u32p]),
    "zkh_alloc": (_err, [_vp, C.c_char_p, _sz, _i, C.POINTER(_vp)]),
    "zkh_copy_from": (_err, [_vp, C.c_char_p, _u32p, _sz, C.POINTER(_vp)]),
    "zkh_wrap": (_err, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "zkh_slice": (_err, [_vp, _sz, _sz, C.POINTER(_vp)]),
    "zkh_retain": (None, [_vp]),
    "zkh_release": (None, [_vp]),
    "zkh_size": (_sz, [_vp]),
    "zkh_device_ptr": (_vp, [_vp]),
    "zkh_read": (_err, [_vp, _vp, _u32p, _sz, _sz]),
    "zkh_write": (_err, [_vp, _vp, _u32p, _sz, _sz]),
    "zkh_host_alloc": (_err, [_vp, _sz, C.POINTER(_u32p)]),
    "zkh_host_free": (None, [_vp, _u32p]),
    "zkh_write_async": (_err, [_vp, _vp, _u32p, _sz, _sz]),
    "zkh_batch_interpolate_ntt": (_err, [_vp, _vp, _sz]),
    "zkh_batch_expand_into_evaluate_ntt": (_err, [_vp, _vp, _vp, _sz, _sz]),
    "zkh_batch_bit_reverse": (_err, [_vp, _vp, _sz]),
    "zkh_zk_shift": (_err, [_vp, _vp, _sz]),
    "zkh_batch_interpolate_ntt_zk_shift": (_err, [_vp, _vp, _sz]),
    "zkh_batch_interpolate_ntt_from": (_err, [_vp, _vp, _vp, _sz, _i]),
    "zkh_hash_rows": (_err, [_vp, _vp, _vp]),
    "zkh_hash_fold": (_err, [_vp, _vp, _sz, _sz]),
    "zkh_merkle_fold_all": (_err, [_vp, _vp, _sz]),
    "zkh_merkle_build": (_err, [_vp, _vp, _vp, _sz]),
    "zkh_batch_evaluate_any": (_err, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "zkh_batch_evaluate_any_bitrev": (_err, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "zkh_batch_bit_reverse_extelem": (_err, [_vp, _vp, _sz]),
    "zkh_mix_poly_coeffs": (_err, [_vp, _vp, _u32p, _u32p, _vp, _vp, _sz, _sz]),
    "zkh_combos_prepare": (_err, [_vp, _vp, _u32p, _u32p, _sz]),
    "zkh_combos_prepare_regs": (_err, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp, _u32p]),
    "zkh_combos_divide": (_err, [_vp, _vp, _sz, _sz, _u32p, _sz, _vp]),
    "zkh_combos_divide_all": (_err, [_vp, _vp, _sz, _sz, _u32p, _u32p, _vp]),
    "zkh_eltwise_add_elem": (_err, [_vp, _vp, _vp, _vp]),
    "zkh_eltwise_copy_elem": (_err, [_vp, _vp, _vp]),
    "zkh_eltwise_zeroize_elem": (_err, [_vp, _vp]),
    "zkh_eltwise_sum_extelem": (_err, [_vp, _vp, _vp]),
    "zkh_fri_fold": (_err, [_vp, _vp, _vp, _u32p]),
    "zkh_gather_sample": (_err, [_vp, _vp, _vp, _sz, _sz, _sz]),
    "zkh_scatter": (_err, [_vp, _vp, _u32p, _u32p, _u32p, _sz, _sz]),
    "zkh_prefix_products": (_err, [_vp, _vp]),
    "zkh_merkle_open": (_err, [_vp, _vp, _vp, _sz, _sz, _u32p, _sz, _vp]),
    "zkh_circuit_load": (_err, [_vp, _u32p, _sz, C.POINTER(_vp)]),
    "zkh_circuit_destroy": (None, [_vp]),
    "zkh_circuit_has_compiled_kernel": (_i, [_vp]),
    "zkh_circuit_attach_code_object": (_err, [_vp, C.c_char_p, _sz, C.c_char_p]),
    "zkh_circuit_attach_code_object_part": (_err, [_vp, C.c_char_p, _sz, C.c_char_p, _sz, _sz]),
    "zkh_circuit_compiled_parts": (_sz, [_vp]),
    "zkh_eval_check": (_err, [_vp, _vp, _vp, C.POINTER(_vp), _sz, C.POINTER(_vp), _sz, _u32p, _sz, _sz, _i]),
    "zkh_syn_code": (_err, [_vp, _vp, _sz, _sz, _vp]),
    "zkh_sha256": (None, [C.c_char_p, _sz, C.POINTER(C.c_uint8)]),
    "zkh_session_check_termination": (_err, [_vp, C.POINTER(_u32p), C.POINTER(_sz), _sz, C.c_char_p, _sz]),
    "zkh_session_check_output": (_err, [_vp, C.POINTER(_u32p), C.POINTER(_sz), _sz, C.c_char_p, _sz, _u32p]),
    "zkh_assumptions_digest": (None, [_u32p, _u32p, _sz, _u32p]),
    "zkh_noise_cell_host": (_u32, [_u32p, _u32, _u32, _u32]),
    "zkh_chacha_block_host": (None, [_u32p, _u32p, _i, _u32p]),
    "zkh_syn_witgen": (_err, [_vp, _vp, _sz, _sz, _u64, _u32p, _u32p, _vp, _vp, _u32p]),
    "zkh_syn_accum": (_err, [_vp, _vp, _sz, _sz, _u32p, _vp, _u32p, _vp]),
    "zkh_syn_chain_contributions": (_err, [_vp, _vp, C.POINTER(_u64), _u32p, _sz, _sz, _u32p]),
    "zkh_syn_preflight_ram_words": (_sz, []),
    "zkh_syn_preflight": (_err, [_u64, _sz, _sz, _u32p, _u32p, C.POINTER(C.c_double)]),
    "zkh_syn_witgen_trace": (_err, [_vp, _vp, _sz, _sz, _u32p, _vp, _u32p, _vp, _vp, _u32p]),
    "zkh_poseidon2_mix": (_err, [_vp, _vp, _sz]),
    "zkh_poseidon2_mix_host": (_err, [_u32p, _u32p, _u32p, _sz]),
    "zkh_prover_create": (_err, [_vp, _vp, C.POINTER(_vp)]),
    "zkh_prover_destroy": (None, [_vp]),
    "zkh_prover_cache_code": (_err, [_vp, _sz, _vp]),
    "zkh_prover_drop_code_cache": (None, [_vp]),
    "zkh_prover_cached_code_root": (_err, [_vp, _sz, _u32p]),
    "zkh_prove_segment": (_err, [_vp, _sz, _sz, _u32p, _vp, _vp, _u32p, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_free_seal": (None, [_u32p]),
    "zkh_prove_begin": (_err, [_vp, _sz, _vp, _vp, _u32p, C.POINTER(_vp), _u32p]),
    "zkh_prove_finish": (_err, [_vp, _vp, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_prove_abort": (None, [_vp]),
    "zkh_code_root": (_err, [_vp, _vp, _sz, _u32p]),
    "zkh_syn_control_root": (_err, [_vp, _sz, _sz, _u32p]),
    "zkh_verify_segment": (_err, [_vp, _u32p, _sz, _u32p, _u32p, _u32p]),
    "zkh_receipt_claim": (_err, [_vp, _u32p, _sz, _u32p, _u32p, _u32p, _u32p]),
    "zkh_receipt_encode": (_err, [_vp, _u32p, _sz, _u32, _u32p, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_receipt_decode": (_err, [_vp, _u32p, _sz, _u32p, C.POINTER(_sz)]),
    "zkh_shipped_circuit_count": (_sz, []),
    "zkh_shipped_circuit_name": (C.c_char_p, [_sz]),
    "zkh_shipped_circuit_desc": (_err, [C.c_char_p, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_rec_build_program": (_err, [_u32, _u32p, _sz, _u32p, _u32p, _u32, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_rec_program_load": (_err, [_vp, _vp, _u32p, _sz, C.POINTER(_vp)]),
    "zkh_rec_program_destroy": (None, [_vp]),
    "zkh_rec_program_info": (_err, [_vp, _u32p, _u32p]),
    "zkh_rec_program_has_graph": (_i, [_vp]),
    "zkh_rec_code": (_err, [_vp, _vp]),
    "zkh_rec_witgen": (_err, [_vp, _u32p, _sz, _u32p, _vp, _u32p]),
    "zkh_rec_accum": (_err, [_vp, _u32p, _vp, _u32p, _vp]),
    "zkh_rec_prove": (_err, [_vp, _u32p, _sz, _u32p, _u32p, C.POINTER(_u32p), C.POINTER(_sz)]),
    "zkh_session_create": (_err, [C.POINTER(_i), _sz, _sz, _u32p, _sz, _u32p, _sz, C.POINTER(_vp)]),
    "zkh_session_destroy": (None, [_vp]),
    "zkh_session_lanes": (_sz, [_vp]),
    "zkh_session_circuit": (_vp, [_vp, _sz, _i]),
    "zkh_session_set_accumulate": (None, [_vp, _vp, _vp]),
    "zkh_session_set_resident_code": (None, [_vp, _i]),
    "zkh_session_set_streamed_fold": (None, [_vp, _i]),
    "zkh_session_set_witness_source": (_err, [_vp, _i, _sz]),
    "zkh_session_set_chained": (_err, [_vp, _i, _u32]),
    "zkh_session_set_journal": (_err, [_vp, C.c_char_p, _sz]),
    "zkh_session_set_recursion": (_err, [_vp, _u32p, _sz, C.POINTER(_u32p), C.POINTER(_sz), _u32p, _sz]),
    "zkh_succinct_verify": (_err, [_u32p, _sz, _u32p, _sz, _sz, _u32p, _sz, _sz]),
    "zkh_succinct_verify_resolved": (_err, [_u32p, _sz, _u32p, _sz, _sz, _u32p, _sz, _sz, _u32p, _sz]),
    "zkh_session_build_recursion": (_err, [_vp, _u32p, _sz, _i]),
    "zkh_session_set_assumptions": (_err, [_vp, _u32p, _sz, C.POINTER(_u32p), C.POINTER(_sz), _u32p, _u32p, _sz]),
    "zkh_session_prove": (_err, [_vp, C.POINTER(SegmentSpec), _sz, _i, _sz, _u32p, C.POINTER(ProveInfo)]),
    "zkh_prove_info_free": (None, [C.POINTER(ProveInfo)]),
    "zkh_session_verify": (_err, [_vp, C.POINTER(SegmentSpec), C.POINTER(ProveInfo), _sz]),
    "zkh_parse_cpulist": (_err, [C.c_char_p, C.POINTER(_i), _sz, C.POINTER(_sz)]),
    "zkh_pci_numa_cpus": (_err, [C.c_char_p, C.c_char_p, C.POINTER(_i), C.POINTER(_i), _sz, C.POINTER(_sz)]),
    "zkh_device_numa_node": (_err, [_i, C.POINTER(_i), C.c_char_p]),
    "zkh_device_identity": (_err, [_i, C.c_char_p, C.c_char_p, C.POINTER(_i), C.c_char_p, C.POINTER(_i)]),
    "zkh_bind_thread_to_device": (_err, [_i, _sz, _sz, C.POINTER(_i), C.POINTER(_sz)]),
    "zkh_prof_enable": (_err, [_vp, _i]),
    "zkh_prof_get": (_err, [_vp, C.POINTER(ProfRec), _sz, C.POINTER(_sz)]),
    "zkh_prof_reset": (_err, [_vp]),
}

_lib = None


class HalError(RuntimeError):
    pass


def load_library():
    """dlopen libzkhal_mi355x.so and bind every symbol of include/zkhal.h (fails loudly if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HalError(f"{LIB_PATH} is missing: run `python -m zeth_amd.build` (no CPU fallback exists)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(err):
    if err:
        lib = load_library()
        msg = C.cast(err, C.c_char_p).value.decode(errors="replace")
        lib.zkh_free_error(err)
        raise HalError(msg)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_u32p)


class Buffer:
    """`hal::Buffer<T>`: size / slice / to_vec (view) / get_at."""

    def __init__(self, hal: "HipHal", handle):
        self.hal, self.h = hal, handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None and getattr(self.hal, "ctx", None):
            _lib.zkh_release(h)      # (a context that is already closed has freed its pool)

    def size(self) -> int:
        return _lib.zkh_size(self.h)

    def slice(self, offset: int, size: int) -> "Buffer":
        out = _vp()
        _check(_lib.zkh_slice(self.h, offset, size, C.byref(out)))
        return Buffer(self.hal, out)

    def to_vec(self) -> np.ndarray:
        out = np.empty(self.size(), dtype=np.uint32)
        _check(_lib.zkh_read(self.hal.ctx, self.h, _ptr(out), 0, out.size))
        return out

    def get_at(self, idx: int) -> int:
        out = np.empty(1, dtype=np.uint32)
        _check(_lib.zkh_read(self.hal.ctx, self.h, _ptr(out), idx, 1))
        return int(out[0])

    def write(self, host, offset: int = 0) -> None:
        a = _u32(host)
        _check(_lib.zkh_write(self.hal.ctx, self.h, _ptr(a), offset, a.size))

    def device_ptr(self) -> int:
        return _lib.zkh_device_ptr(self.h)


class Circuit:
    """A loaded circuit description (TapSet + PolyExtStep list) — the `CircuitHal` side."""

    def __init__(self, hal: "HipHal", desc):
        self.hal = hal
        self.desc = _u32(desc)
        h = _vp()
        _check(_lib.zkh_circuit_load(hal.ctx, _ptr(self.desc), self.desc.size, C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.zkh_circuit_destroy(h)

    def has_compiled_kernel(self) -> bool:
        return bool(_lib.zkh_circuit_has_compiled_kernel(self.h))

    def kernel_kind(self) -> str:
        """'attached' (code object compiled at load time), 'builtin' (generated at build time) or 'interpreter'."""
        return ("interpreter", "builtin", "attached")[_lib.zkh_circuit_has_compiled_kernel(self.h)]

    def attach_code_object(self, image: bytes, kernel_name: str, part: int = 0, n_parts: int = 1) -> None:
        _check(_lib.zkh_circuit_attach_code_object_part(self.h, image, len(image), kernel_name.encode(), part, n_parts))

    def compiled_parts(self) -> int:
        return int(_lib.zkh_circuit_compiled_parts(self.h))

    def jit(self, use_cache: bool = True) -> None:
        """Generate + compile (hipcc --genco per part, in parallel, disk-cached) + attach the straight-line eval_check
        kernels for this desc."""
        from .circuits import jit as _jit
        objs = _jit.compile_code_objects(self.desc, use_cache=use_cache)
        for i, (image, name) in enumerate(objs):
            self.attach_code_object(image, name, i, len(objs))

    def eval_check(self, check: Buffer, groups: Sequence[Buffer], globals_: Sequence[Buffer], poly_mix, po2: int,
                   use_interpreter: bool = False) -> None:
        g = (_vp * len(groups))(*[b.h for b in groups])
        gl = (_vp * len(globals_))(*[b.h for b in globals_])
        pm = _u32(poly_mix)
        _check(_lib.zkh_eval_check(self.hal.ctx, self.h, check.h, g, len(groups), gl, len(globals_), _ptr(pm), po2,
                                   1 << po2, int(use_interpreter)))


class HostCircuit:
    """A circuit description loaded WITHOUT a GPU context: enough for `verify_segment` (upstream verifies on the CPU)."""

    def __init__(self, desc):
        load_library()
        self.desc = _u32(desc)
        h = _vp()
        _check(_lib.zkh_circuit_load(None, _ptr(self.desc), self.desc.size, C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.zkh_circuit_destroy(h)

    def verify_segment(self, seal, control_root, rc=None, diag=None) -> None:
        """`Receipt::verify` for one segment seal: raises HalError (VerificationError analogue) if rejected.
        `control_root` (8 words) is the expected code commitment for (circuit, po2): required, like upstream's control ID."""
        s = _u32(seal)
        cr = _u32(control_root) if control_root is not None else None
        if cr is not None and cr.size != 8:
            raise HalError("verify_segment: control root must be 8 words")
        r = _ptr(_u32(rc)) if rc is not None else None
        d = _ptr(_u32(diag)) if diag is not None else None
        _check(_lib.zkh_verify_segment(self.h, _ptr(s), s.size, _ptr(cr) if cr is not None else None, r, d))


    def receipt_encode(self, seal, segment_index: int, control_root) -> np.ndarray:
        """Seal -> receipt container words (zkh_receipt_encode)."""
        s, cr = _u32(seal), _u32(control_root)
        if cr.size != 8:
            raise HalError("receipt_encode: control root must be 8 words")
        blob, n = _u32p(), _sz()
        _check(_lib.zkh_receipt_encode(self.h, _ptr(s), s.size, segment_index, _ptr(cr), C.byref(blob), C.byref(n)))
        out = np.ctypeslib.as_array(blob, shape=(n.value,)).copy()
        _lib.zkh_free_seal(blob)
        return out

    def receipt_decode(self, blob):
        """Container words -> (header dict, seal words); raises HalError on any integrity failure."""
        b = _u32(blob)
        info = np.zeros(26, dtype=np.uint32)
        off = _sz()
        _check(_lib.zkh_receipt_decode(self.h, _ptr(b), b.size, _ptr(info), C.byref(off)))
        hdr = {"version": int(info[1]), "circuit_hash": int(info[2]) | (int(info[3]) << 32), "po2": int(info[4]),
               "hashfn": "poseidon2", "placeholder_tables": bool(info[6] & 1),
               "tables": "placeholder" if info[6] & 1 else "derived" if info[6] & 2 else "upstream", "index": int(info[7]), "out_size": int(info[8]),
               "control_root": info[10:18].copy(), "claim": info[18:26].copy()}
        return hdr, b[off.value: off.value + int(info[9])].copy()

    def receipt_claim(self, seal, control_root, rc=None, diag=None) -> np.ndarray:
        """Claim digest (8 words) of a sealed segment: Poseidon2(out globals, po2, control root)."""
        s, cr = _u32(seal), _u32(control_root)
        if cr.size != 8:
            raise HalError("receipt_claim: control root must be 8 words")
        out = np.zeros(8, dtype=np.uint32)
        r = _ptr(_u32(rc)) if rc is not None else None
        d = _ptr(_u32(diag)) if diag is not None else None
        _check(_lib.zkh_receipt_claim(self.h, _ptr(s), s.size, _ptr(cr), r, d, _ptr(out)))
        return out


class RecProgram:
    """A loaded RECURSION program (lift / join): zkh_rec_program_*.  `circuit` = the RECURSION description on the same HAL."""

    def __init__(self, hal: "HipHal", circuit: Circuit, blob):
        self.hal, self.circuit = hal, circuit
        b = _u32(blob)
        h = _vp()
        _check(_lib.zkh_rec_program_load(hal.ctx, circuit.h, _ptr(b), b.size, C.byref(h)))
        self.h = h
        root, info = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
        _check(_lib.zkh_rec_program_info(h, _ptr(root), _ptr(info)))
        self.root = root
        self.po2, self.zk_cycles, self.n_inputs, self.n_p2, self.n_gates, self.n_ops, self.n_levels, self.n_vars = (int(x) for x in info)
        self.graph_steps = int(_lib.zkh_rec_program_has_graph(h))       # > 0: the witness schedule is a hipGraph of that many plan steps

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        # a program holds device buffers and a prover (the resident code group): only released into a live context
        if h and _lib is not None and getattr(self.hal, "ctx", None):
            _lib.zkh_rec_program_destroy(h)

    def code(self, code: Buffer) -> None:
        _check(_lib.zkh_rec_code(self.h, code.h))

    def witgen(self, inputs, data: Buffer, noise_seed: int = 0x2E80) -> np.ndarray:
        i = _u32(inputs)
        out = np.zeros(16, np.uint32)
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_rec_witgen(self.h, _ptr(i), i.size, kp, data.h, _ptr(out)))
        return out

    def accum(self, data: Buffer, mix_global, accum: Buffer, noise_seed: int = 0x2E80) -> None:
        m = _u32(mix_global)
        assert m.size == 20
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_rec_accum(self.h, kp, data.h, _ptr(m), accum.h))

    def prove(self, inputs, noise_seed: int = 0x2E80):
        """-> (seal words, out globals): the witness exists only if the program's in-circuit verifier accepts `inputs`"""
        i = _u32(inputs)
        out = np.zeros(16, np.uint32)
        seal, n = _u32p(), _sz()
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_rec_prove(self.h, _ptr(i), i.size, kp, _ptr(out), C.byref(seal), C.byref(n)))
        words = np.ctypeslib.as_array(seal, shape=(n.value,)).copy()
        _lib.zkh_free_seal(seal)
        return words, out


def syn_preflight(seed: int, po2: int, zk_cycles: int = ZK_CYCLES, records: Optional[np.ndarray] = None):
    """The sequential host machine (csrc/preflight.hip): -> (records: 4 words per active cycle, RAM image, CPU seconds).
    `records` may be a pinned view from HipHal.host_alloc (filled in place)."""
    load_library()
    A = (1 << po2) - zk_cycles
    rec = records if records is not None else np.empty(4 * A, dtype=np.uint32)
    assert rec.size == 4 * A and rec.dtype == np.uint32
    ram = np.empty(int(_lib.zkh_syn_preflight_ram_words()), dtype=np.uint32)
    secs = C.c_double(0.0)
    _check(_lib.zkh_syn_preflight(seed & (2**64 - 1), po2, zk_cycles, _ptr(rec), _ptr(ram), C.byref(secs)))
    return rec, ram, secs.value


# ---- host placement (csrc/topology.hip) ----
def parse_cpulist(text: str) -> List[int]:
    load_library()
    n = _sz()
    _check(_lib.zkh_parse_cpulist(text.encode(), None, 0, C.byref(n)))
    cpus = (_i * max(1, n.value))()
    _check(_lib.zkh_parse_cpulist(text.encode(), cpus, n.value, C.byref(n)))
    return list(cpus[: n.value])


def pci_numa_cpus(bdf: str, sysfs_root: str = "/sys"):
    """-> (NUMA node of the PCI function or -1, the node's CPU ids)"""
    load_library()
    node, n = _i(-1), _sz()
    cpus = (_i * 4096)()
    _check(_lib.zkh_pci_numa_cpus(sysfs_root.encode(), bdf.encode(), C.byref(node), cpus, 4096, C.byref(n)))
    return node.value, list(cpus[: min(n.value, 4096)])


def device_numa_node(device: int):
    """-> (NUMA node of the HIP device or -1, its PCI bus id)"""
    load_library()
    node = _i(-1)
    bdf = C.create_string_buffer(32)
    _check(_lib.zkh_device_numa_node(device, C.byref(node), bdf))
    return node.value, bdf.value.decode()


def device_identity(device: int) -> dict:
    """-> {"pci_bus_id", "uuid", "numa_node", "name", "visible_devices"} of a HIP device: what tells one GPU from another across
    processes (bench.py gathers it from every rank)"""
    load_library()
    bdf, uuid, name = C.create_string_buffer(32), C.create_string_buffer(40), C.create_string_buffer(64)
    node, count = _i(-1), _i(0)
    _check(_lib.zkh_device_identity(device, bdf, uuid, C.byref(node), name, C.byref(count)))
    return {"pci_bus_id": bdf.value.decode(), "uuid": uuid.value.decode(), "numa_node": node.value, "name": name.value.decode(),
            "visible_devices": count.value}


def bind_to_device(device: int, slot: int = 0, share: int = 1) -> dict:
    """Bind the calling thread (threads created afterwards inherit it) to the cores next to `device`: slice `slot` of `share`
    slices of its NUMA node when several ranks share the node.  -> {"numa_node": n or -1, "cpus": count}"""
    load_library()
    node, n = _i(-1), _sz()
    _check(_lib.zkh_bind_thread_to_device(device, slot, share, C.byref(node), C.byref(n)))
    return {"numa_node": node.value, "cpus": n.value}


def placement_slot(device: int, devices: Sequence[int]):
    """Which slice of its NUMA node the rank driving `device` takes when the ranks of `devices` share the host:
    -> (slot, share) = (index of `device` among the devices on the same node, their number); (0, 1) if the node is unknown."""
    mine = device_numa_node(device)[0]
    if mine < 0:
        return 0, 1
    same = [d for d in devices if device_numa_node(d)[0] == mine]
    return same.index(device), len(same)


class HipHal:
    """`impl Hal for HipHal` — one MI355X, one HIP stream, one driving thread."""

    def __init__(self, device: int = 0, hash_suite: str = "poseidon2"):
        load_library()
        ctx = _vp()
        _check(_lib.zkh_ctx_create(device, hash_suite.encode(), C.byref(ctx)))
        self.ctx = ctx
        self.device = device

    @staticmethod
    def version() -> str:
        return load_library().zkh_version().decode()

    def close(self):
        ctx, self.ctx = getattr(self, "ctx", None), None
        if ctx and _lib is not None:
            _lib.zkh_ctx_destroy(ctx)

    def __del__(self):
        # buffers hold a reference to the hal, so the context outlives them
        self.close()

    def memory(self) -> dict:
        """Bytes of device memory: held by live buffers, cached by the free list, live high-water mark."""
        live, cached, peak = _sz(), _sz(), _sz()
        _lib.zkh_ctx_memory(self.ctx, C.byref(live), C.byref(cached), C.byref(peak))
        return {"live": live.value, "cached": cached.value, "peak": peak.value}

    def trim(self) -> None:
        """Drain the stream and hand the cached blocks back to the driver."""
        _check(_lib.zkh_ctx_trim(self.ctx))

    def sync(self) -> None:
        _check(_lib.zkh_sync(self.ctx))

    # ---- allocation (alloc_elem / alloc_extelem / alloc_digest / alloc_u32 / copy_from_*) ----
    def alloc(self, name: str, n_words: int, zero: bool = False) -> Buffer:
        out = _vp()
        _check(_lib.zkh_alloc(self.ctx, name.encode(), n_words, int(zero), C.byref(out)))
        return Buffer(self, out)

    def alloc_elem(self, name: str, size: int) -> Buffer:
        return self.alloc(name, size)

    def alloc_extelem(self, name: str, size: int) -> Buffer:
        return self.alloc(name, size * EXT_SIZE)

    def alloc_digest(self, name: str, size: int) -> Buffer:
        return self.alloc(name, size * DIGEST_WORDS)

    def copy_from(self, name: str, host) -> Buffer:
        a = _u32(host).reshape(-1)
        out = _vp()
        _check(_lib.zkh_copy_from(self.ctx, name.encode(), _ptr(a), a.size, C.byref(out)))
        return Buffer(self, out)

    copy_from_elem = copy_from_extelem = copy_from_u32 = copy_from_digest = copy_from

    def host_alloc(self, n_words: int) -> np.ndarray:
        """Pinned host memory (a numpy view) for witnesses produced on the CPU; release with host_free."""
        p = _u32p()
        _check(_lib.zkh_host_alloc(self.ctx, n_words, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(n_words,))

    def host_free(self, arr: np.ndarray) -> None:
        _lib.zkh_host_free(self.ctx, arr.ctypes.data_as(_u32p))

    def write_async(self, buf: Buffer, pinned: np.ndarray, offset: int = 0) -> None:
        """Enqueue H2D from a host_alloc block on the context's stream (no host sync)."""
        _check(_lib.zkh_write_async(self.ctx, buf.h, pinned.ctypes.data_as(_u32p), offset, pinned.size))

    def wrap(self, device_ptr: int, n_words: int) -> Buffer:
        out = _vp()
        _check(_lib.zkh_wrap(self.ctx, device_ptr, n_words, C.byref(out)))
        return Buffer(self, out)

    # ---- trait Hal ----
    def batch_interpolate_ntt(self, io: Buffer, count: int) -> None:
        _check(_lib.zkh_batch_interpolate_ntt(self.ctx, io.h, count))

    def batch_expand_into_evaluate_ntt(self, out: Buffer, inp: Buffer, count: int, expand_bits: int) -> None:
        _check(_lib.zkh_batch_expand_into_evaluate_ntt(self.ctx, out.h, inp.h, count, expand_bits))

    def batch_bit_reverse(self, io: Buffer, count: int) -> None:
        _check(_lib.zkh_batch_bit_reverse(self.ctx, io.h, count))

    def zk_shift(self, io: Buffer, count: int) -> None:
        _check(_lib.zkh_zk_shift(self.ctx, io.h, count))

    def batch_interpolate_ntt_zk_shift(self, io: Buffer, count: int) -> None:
        _check(_lib.zkh_batch_interpolate_ntt_zk_shift(self.ctx, io.h, count))

    def batch_interpolate_ntt_from(self, out: Buffer, inp: Buffer, count: int, zk_shift: bool) -> None:
        _check(_lib.zkh_batch_interpolate_ntt_from(self.ctx, out.h, inp.h, count, int(zk_shift)))

    def hash_rows(self, output: Buffer, matrix: Buffer) -> None:
        _check(_lib.zkh_hash_rows(self.ctx, output.h, matrix.h))

    def hash_fold(self, io: Buffer, input_size: int, output_size: int) -> None:
        _check(_lib.zkh_hash_fold(self.ctx, io.h, input_size, output_size))

    def merkle_fold_all(self, nodes: Buffer, rows: int) -> None:
        _check(_lib.zkh_merkle_fold_all(self.ctx, nodes.h, rows))

    def merkle_build(self, nodes: Buffer, matrix: Buffer, rows: int) -> None:
        """`MerkleTreeProver::new`: leaves = hash_rows(matrix) at nodes[rows..2 rows), then every layer above."""
        _check(_lib.zkh_merkle_build(self.ctx, nodes.h, matrix.h, rows))

    def batch_evaluate_any(self, coeffs: Buffer, poly_count: int, which: Buffer, xs: Buffer, out: Buffer) -> None:
        _check(_lib.zkh_batch_evaluate_any(self.ctx, coeffs.h, poly_count, which.h, xs.h, out.h))

    def batch_evaluate_any_bitrev(self, coeffs: Buffer, poly_count: int, which: Buffer, xs: Buffer, out: Buffer) -> None:
        _check(_lib.zkh_batch_evaluate_any_bitrev(self.ctx, coeffs.h, poly_count, which.h, xs.h, out.h))

    def batch_bit_reverse_extelem(self, io: Buffer, count: int) -> None:
        _check(_lib.zkh_batch_bit_reverse_extelem(self.ctx, io.h, count))

    def mix_poly_coeffs(self, output: Buffer, mix_start, mix, inp: Buffer, combos: Buffer, input_size: int, count: int) -> None:
        ms, m = _u32(mix_start), _u32(mix)
        _check(_lib.zkh_mix_poly_coeffs(self.ctx, output.h, _ptr(ms), _ptr(m), inp.h, combos.h, input_size, count))

    def combos_prepare(self, combos: Buffer, pos, vals_ext) -> None:
        p, v = _u32(pos), _u32(vals_ext).reshape(-1)
        _check(_lib.zkh_combos_prepare(self.ctx, combos.h, _ptr(p), _ptr(v), p.size))

    def combos_prepare_regs(self, combos: Buffer, coeff_u: Buffer, combo_count: int, cycles: int, reg_sizes: Buffer,
                            reg_combo_ids: Buffer, mix) -> None:
        """`Hal::combos_prepare` with upstream's literal argument list (all operands on the device)."""
        m = _u32(mix)
        _check(_lib.zkh_combos_prepare_regs(self.ctx, combos.h, coeff_u.h, combo_count, cycles, reg_sizes.size(), reg_sizes.h,
                                            reg_combo_ids.h, _ptr(m)))

    def combos_divide(self, combos: Buffer, combo: int, cycles: int, pts_ext, rem_out: Buffer) -> None:
        p = _u32(pts_ext).reshape(-1)
        _check(_lib.zkh_combos_divide(self.ctx, combos.h, combo, cycles, _ptr(p), p.size // 4, rem_out.h))

    def combos_divide_all(self, combos: Buffer, cycles: int, pts_ext, pts_begin, rem_out: Buffer) -> None:
        p, b = _u32(pts_ext).reshape(-1), _u32(pts_begin)
        _check(_lib.zkh_combos_divide_all(self.ctx, combos.h, cycles, b.size - 1, _ptr(p), _ptr(b), rem_out.h))

    def eltwise_add_elem(self, out: Buffer, a: Buffer, b: Buffer) -> None:
        _check(_lib.zkh_eltwise_add_elem(self.ctx, out.h, a.h, b.h))

    def eltwise_copy_elem(self, out: Buffer, inp: Buffer) -> None:
        _check(_lib.zkh_eltwise_copy_elem(self.ctx, out.h, inp.h))

    def eltwise_zeroize_elem(self, io: Buffer) -> None:
        _check(_lib.zkh_eltwise_zeroize_elem(self.ctx, io.h))

    def eltwise_sum_extelem(self, out: Buffer, inp: Buffer) -> None:
        _check(_lib.zkh_eltwise_sum_extelem(self.ctx, out.h, inp.h))

    def fri_fold(self, out: Buffer, inp: Buffer, mix) -> None:
        m = _u32(mix)
        _check(_lib.zkh_fri_fold(self.ctx, out.h, inp.h, _ptr(m)))

    def gather_sample(self, dst: Buffer, src: Buffer, idx: int, size: int, stride: int) -> None:
        _check(_lib.zkh_gather_sample(self.ctx, dst.h, src.h, idx, size, stride))

    def scatter(self, into: Buffer, index, offsets, values) -> None:
        i, o, v = _u32(index), _u32(offsets), _u32(values)
        _check(_lib.zkh_scatter(self.ctx, into.h, _ptr(i), _ptr(o), _ptr(v), o.size - 1, v.size))

    def prefix_products(self, io: Buffer) -> None:
        _check(_lib.zkh_prefix_products(self.ctx, io.h))

    def merkle_open(self, matrix: Buffer, nodes: Buffer, rows: int, cols: int, idx, out: Buffer) -> None:
        i = _u32(idx)
        _check(_lib.zkh_merkle_open(self.ctx, matrix.h, nodes.h, rows, cols, _ptr(i), i.size, out.h))

    def poseidon2_mix(self, states: "Buffer") -> None:
        """The bare permutation on `states.size() // 24` states (24 Montgomery words each), in place."""
        _check(_lib.zkh_poseidon2_mix(self.ctx, states.h, states.size() // 24))

    def poseidon2_set_constants(self, rc, diag) -> None:
        r, d = _u32(rc), _u32(diag)
        assert r.size == 24 * 29 and d.size == 24
        _check(_lib.zkh_poseidon2_set_constants(self.ctx, _ptr(r), _ptr(d)))

    # ---- circuit + SYN witness ----
    def load_circuit(self, desc, jit: Optional[bool] = None) -> Circuit:
        """CircuitHal for a description blob.  jit=None (default): a desc without a built-in eval_check kernel gets one
        compiled at load time when hipcc is on the machine, and runs on the step interpreter otherwise; jit=True:
        always compile (raises if that is impossible); jit=False: never."""
        c = Circuit(self, desc)
        if jit or (jit is None and not c.has_compiled_kernel()):
            from .circuits import jit as _jit
            if jit or _jit.hipcc_path() is not None:
                c.jit()
        return c

    def syn_code(self, circuit: Circuit, po2: int, zk_cycles: int, code: Buffer) -> None:
        _check(_lib.zkh_syn_code(self.ctx, circuit.h, po2, zk_cycles, code.h))

    def syn_witgen(self, circuit: Circuit, po2: int, zk_cycles: int, seed: int, noise_seed: int, code: Buffer, data: Buffer,
                   pub=None) -> np.ndarray:
        """-> out globals.  `code` may be None for kinds 1 / 2 (the code group of this size is already held, e.g. resident).
        SYN-AIR (kind 1): OUTPUT_SIZE words s, 0, 0, 0, then the public input words `pub`.
        KECCAK-F (kind 2): `pub` = optional input state of the LAST permutation (25 lanes = 50 words, low word first);
        out = that permutation's output state as 100 16-bit limbs."""
        out_size = int(circuit.desc[7])
        out = np.zeros(out_size, dtype=np.uint32)
        p = _u32(pub) if pub is not None else np.zeros(0, np.uint32)
        if int(circuit.desc[13]) == 2:
            if p.size not in (0, 50):
                raise HalError(f"syn_witgen: KECCAK-F takes an optional 50-word input state, got {p.size} words")
        elif int(circuit.desc[13]) == 3:
            if p.size != 16:
                raise HalError(f"syn_witgen: P2-JOIN takes the two child claims (16 words), got {p.size}")
        elif p.size != out_size - 4:
            raise HalError(f"syn_witgen: circuit takes {out_size - 4} public input words, got {p.size}")
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_syn_witgen(self.ctx, circuit.h, po2, zk_cycles, seed & (2**64 - 1), kp,
                                   _ptr(p) if p.size else None, code.h if code is not None else None, data.h, _ptr(out)))
        return out

    def syn_witgen_trace(self, circuit: Circuit, po2: int, zk_cycles: int, noise_seed: int, records: Buffer, ram_image, code: Optional[Buffer],
                         data: Buffer) -> np.ndarray:
        """records (device, 4 words per active cycle) + the RAM image -> code (None: already held), data; returns the out globals"""
        out = np.zeros(int(circuit.desc[7]), dtype=np.uint32)
        ram = _u32(ram_image) if ram_image is not None else None
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_syn_witgen_trace(self.ctx, circuit.h, po2, zk_cycles, kp, records.h, _ptr(ram) if ram is not None else None,
                                         code.h if code is not None else None, data.h, _ptr(out)))
        return out

    def syn_accum(self, circuit: Circuit, po2: int, zk_cycles: int, noise_seed: int, data: Buffer, mix_global, accum: Buffer) -> None:
        m = _u32(mix_global)
        _k, kp = _key_ptr(noise_seed)
        _check(_lib.zkh_syn_accum(self.ctx, circuit.h, po2, zk_cycles, kp, data.h, _ptr(m), accum.h))

    # ---- profiling ----
    def prof_enable(self, on: bool = True) -> None:
        _check(_lib.zkh_prof_enable(self.ctx, int(on)))

    def prof_reset(self) -> None:
        _check(_lib.zkh_prof_reset(self.ctx))

    def prof_get(self) -> List[dict]:
        recs = (ProfRec * 128)()
        n = _sz()
        _check(_lib.zkh_prof_get(self.ctx, recs, 128, C.byref(n)))
        return [{"name": recs[i].name.decode(), "calls": int(recs[i].calls), "total_ms": float(recs[i].total_ms),
                 "alg_bytes": float(recs[i].alg_bytes)} for i in range(n.value)]
