"""ctypes binding of the CPU oracle (oracle/libzkoracle.so).  TEST-ONLY: product code never imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
_LIB = os.path.join(_ORACLE_DIR, "libzkoracle.so")

u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    sz, u32, u64, vp = C.c_size_t, C.c_uint32, C.c_uint64, C.c_void_p
    sig = {
        "zko_fp_mul": (u32, [u32, u32]), "zko_fp_encode": (u32, [u32]), "zko_fp_decode": (u32, [u32]),
        "zko_fp_inv": (u32, [u32]), "zko_rou_fwd": (u32, [C.c_uint]), "zko_rou_rev": (u32, [C.c_uint]),
        "zko_fp4_mul": (None, [u32p, u32p, u32p]), "zko_fp4_inv": (None, [u32p, u32p]),
        "zko_poseidon2_mix": (None, [u32p]),
        "zko_poseidon2_set_constants": (None, [u32p, u32p]),
        "zko_hash_elem_slice": (None, [u32p, sz, sz, u32p]), "zko_hash_pair": (None, [u32p, u32p, u32p]),
        "zko_batch_interpolate_ntt": (None, [u32p, sz, sz]),
        "zko_batch_expand_into_evaluate_ntt": (None, [u32p, sz, u32p, sz, sz, sz]),
        "zko_batch_bit_reverse": (None, [u32p, sz, sz]), "zko_zk_shift": (None, [u32p, sz, sz]),
        "zko_hash_rows": (None, [u32p, sz, u32p, sz]), "zko_hash_fold": (None, [u32p, sz, sz]),
        "zko_batch_evaluate_any": (None, [u32p, sz, sz, u32p, u32p, sz, u32p]),
        "zko_mix_poly_coeffs": (None, [u32p, u32p, u32p, u32p, u32p, sz, sz]),
        "zko_eltwise_add_elem": (None, [u32p, u32p, u32p, sz]),
        "zko_eltwise_sum_extelem": (None, [u32p, sz, u32p, sz]),
        "zko_fri_fold": (None, [u32p, sz, u32p, u32p]),
        "zko_gather_sample": (None, [u32p, u32p, sz, sz, sz]),
        "zko_prefix_products": (None, [u32p, sz]),
        "zko_poly_interpolate": (None, [u32p, u32p, u32p, sz]),
        "zko_poly_eval": (None, [u32p, sz, u32p, u32p]),
        "zko_poly_divide": (None, [u32p, sz, u32p, u32p]),
        "zko_circuit_load": (C.c_char_p, [u32p, sz, C.POINTER(vp)]), "zko_circuit_free": (None, [vp]),
        "zko_poly_ext": (None, [vp, u32p, u32p, C.POINTER(vp), u32p]),
        "zko_eval_check": (None, [vp, u32p, C.POINTER(vp), C.POINTER(vp), u32p, C.c_uint]),
        "zko_syn_cell": (u32, [u64, u32, u32, u32]),
        "zko_syn_code": (None, [vp, C.c_uint, C.c_uint, u32p]),
        "zko_syn_witgen": (None, [vp, C.c_uint, C.c_uint, u64, u32p, C.c_void_p, u32p, u32p, u32p]),
        "zko_syn_accum": (None, [vp, C.c_uint, C.c_uint, u32p, u32p, u32p, u32p]),
        "zko_noise_cell": (u32, [u32p, u32, u32, u32]), "zko_chacha_block": (None, [u32p, u32p, C.c_int, u32p]),
        "zko_prove_segment": (C.POINTER(u32), [vp, C.c_uint, C.c_uint, u64, u32p, C.c_void_p, C.POINTER(sz), C.POINTER(C.c_char_p)]),
        "zko_control_root": (None, [vp, C.c_uint, C.c_uint, u32p]),
        "zko_verify_segment": (C.c_char_p, [vp, u32p, sz, C.c_void_p]),
        "zko_free": (None, [vp]),
        "zko_prove_traces": (C.POINTER(u32), [vp, C.c_uint, C.c_uint, u32p, u32p, u32p, u32p, C.POINTER(sz), C.POINTER(C.c_char_p)]),
        "zko_root_of_code": (None, [vp, C.c_uint, u32p, u32p]),
        "zko_rec_code": (C.c_char_p, [u32p, sz, u32p]),
        "zko_rec_witgen": (C.c_char_p, [u32p, sz, u32p, sz, u32p, u32p, u32p, u32p]),
        "zko_rec_accum": (None, [vp, C.c_uint, C.c_uint, u32p, u32p, u32p, u32p, u32p]),
        "zko_check_rows": (C.c_long, [vp, C.c_uint, C.POINTER(vp), C.POINTER(vp), sz, sz]),
        "zko_syn_preflight_ram_words": (sz, []),
        "zko_syn_preflight": (None, [u64, C.c_uint, C.c_uint, u32p, u32p]),
        "zko_syn_witgen_trace": (None, [vp, C.c_uint, C.c_uint, u32p, u32p, u32p, C.c_void_p, u32p, u32p]),
        "zko_num_threads": (C.c_int, []),
        "zko_set_num_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


def key_words(noise_seed) -> np.ndarray:
    """the 8-word blinding key of an integer noise seed (little-endian words: the convention of zeth_amd.hal.noise_key); the oracle
    is deterministic, so 0 / None is not "draw from the OS" here but an error"""
    if isinstance(noise_seed, np.ndarray):
        k = np.ascontiguousarray(noise_seed, dtype=np.uint32)
        assert k.shape == (8,)
        return k
    v = int(noise_seed)
    assert 0 < v < (1 << 256), "the oracle needs an explicit noise key"
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)


class OracleCircuit:
    def __init__(self, lib, desc):
        self.lib = lib
        self.desc = np.ascontiguousarray(desc, dtype=np.uint32)
        h = C.c_void_p()
        err = lib.zko_circuit_load(self.desc, self.desc.size, C.byref(h))
        if err:
            raise RuntimeError(err.decode())
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.zko_circuit_free(self.h)
            self.h = None

    @property
    def out_size(self):
        return int(self.desc[7])

    def witgen(self, po2, zk_cycles=1994, seed=0x5EED0000, noise_seed=0x2E80, pub=None):
        """-> (code, data, out_global) host arrays, the oracle's SYN witness."""
        wa, wc, wd = (int(x) for x in self.desc[3:6])
        n = 1 << po2
        code, data, out = np.zeros(wc * n, np.uint32), np.zeros(wd * n, np.uint32), np.zeros(self.out_size, np.uint32)
        self.lib.zko_syn_witgen(self.h, po2, zk_cycles, seed, key_words(noise_seed), self._pub(pub), code, data, out)
        return code, data, out

    def preflight(self, seed, po2, zk_cycles=1994):
        """-> (records: 4 words per active cycle, RAM image): the oracle's sequential machine (oracle/preflight.c)"""
        A = (1 << po2) - zk_cycles
        rec, ram = np.zeros(4 * A, np.uint32), np.zeros(int(self.lib.zko_syn_preflight_ram_words()), np.uint32)
        self.lib.zko_syn_preflight(seed, po2, zk_cycles, rec, ram)
        return rec, ram

    def witgen_trace(self, po2, records, ram, noise_seed=0x2E80, zk_cycles=1994):
        """records + RAM image -> (code, data, out_global): the oracle's row fill"""
        wa, wc, wd = (int(x) for x in self.desc[3:6])
        n = 1 << po2
        code, data, out = np.zeros(wc * n, np.uint32), np.zeros(wd * n, np.uint32), np.zeros(self.out_size, np.uint32)
        self.lib.zko_syn_witgen_trace(self.h, po2, zk_cycles, key_words(noise_seed), np.ascontiguousarray(records, dtype=np.uint32),
                                      np.ascontiguousarray(ram, dtype=np.uint32), code.ctypes.data_as(C.c_void_p), data, out)
        return code, data, out

    def _pub(self, pub):
        if int(self.desc[13]) == 2:                       # KECCAK-F: optional input state of the last permutation (50 words)
            if pub is None:
                return None
            self._pub_arr = np.ascontiguousarray(pub, dtype=np.uint32)
            assert self._pub_arr.size == 50, "KECCAK-F takes 25 lanes = 50 words"
            return self._pub_arr.ctypes.data
        n_pub = 16 if int(self.desc[13]) == 3 else self.out_size - 4          # P2-JOIN: the two child claims
        if n_pub == 0:
            return None
        self._pub_arr = np.ascontiguousarray(pub, dtype=np.uint32)
        assert self._pub_arr.size == n_pub, f"circuit takes {n_pub} public input words"
        return self._pub_arr.ctypes.data

    def control_root(self, po2, zk_cycles=1994):
        key = (po2, zk_cycles)
        cache = self.__dict__.setdefault("_roots", {})
        if key not in cache:
            root = np.zeros(8, np.uint32)
            self.lib.zko_control_root(self.h, po2, zk_cycles, root)
            cache[key] = root
        return cache[key]

    def prove(self, po2, zk_cycles=1994, seed=0x5EED0000, noise_seed=0x2E80, pub=None):
        n = C.c_size_t()
        err = C.c_char_p()
        p = self.lib.zko_prove_segment(self.h, po2, zk_cycles, seed, key_words(noise_seed), self._pub(pub), C.byref(n), C.byref(err))
        if not p:
            raise RuntimeError((err.value or b"?").decode())
        seal = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        self.lib.zko_free(p)
        return seal

    # ---- RECURSION (kind 4): the code group is a program blob (zeth_amd/circuits/recursion.py Program.finish) ----
    def rec_witgen(self, prog, inputs, noise_seed=0x2E80):
        """-> (code, data, out_global); raises if the program's assertions fail on `inputs` (raw Montgomery words)"""
        prog = np.ascontiguousarray(prog, dtype=np.uint32)
        inputs = np.ascontiguousarray(inputs, dtype=np.uint32)
        if inputs.size == 0:
            inputs = np.zeros(1, np.uint32)[:0]
        wa, wc, wd = (int(x) for x in self.desc[3:6])
        n = 1 << int(prog[2])
        code, data, out = np.zeros(wc * n, np.uint32), np.zeros(wd * n, np.uint32), np.zeros(self.out_size, np.uint32)
        err = self.lib.zko_rec_witgen(prog, prog.size, inputs, inputs.size, key_words(noise_seed), code, data, out)
        if err:
            raise RuntimeError(err.decode())
        return code, data, out

    def rec_accum(self, po2, code, data, mix, zk_cycles=1994, noise_seed=0x2E80):
        accum = np.zeros(int(self.desc[3]) << po2, np.uint32)
        self.lib.zko_rec_accum(self.h, po2, zk_cycles, key_words(noise_seed), code, data, np.ascontiguousarray(mix, dtype=np.uint32), accum)
        return accum

    def check_rows(self, po2, accum, code, data, out, mix, lo=0, hi=None):
        """first row of the trace on which some constraint of the step list does not vanish, or -1"""
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (accum, code, data, out, mix)]
        groups = (C.c_void_p * 3)(*(a.ctypes.data for a in arrs[:3]))
        globals_ = (C.c_void_p * 2)(*(a.ctypes.data for a in arrs[3:]))
        return int(self.lib.zko_check_rows(self.h, po2, groups, globals_, lo, (1 << po2) if hi is None else hi))

    def root_of_code(self, po2, code):
        root = np.zeros(8, np.uint32)
        self.lib.zko_root_of_code(self.h, po2, np.ascontiguousarray(code, dtype=np.uint32), root)
        return root

    def prove_traces(self, po2, code, data, out, zk_cycles=1994, noise_seed=0x2E80):
        n = C.c_size_t()
        err = C.c_char_p()
        p = self.lib.zko_prove_traces(self.h, po2, zk_cycles, key_words(noise_seed), np.ascontiguousarray(code, dtype=np.uint32),
                                      np.ascontiguousarray(data, dtype=np.uint32), np.ascontiguousarray(out, dtype=np.uint32),
                                      C.byref(n), C.byref(err))
        if not p:
            raise RuntimeError((err.value or b"?").decode())
        seal = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        self.lib.zko_free(p)
        return seal

    def verify(self, seal, control_root=None, zk_cycles=1994):
        """None if accepted.  control_root defaults to the oracle's own root for the po2 in the seal header (the test
        passes an explicit one when it wants to check the binding itself)."""
        seal = np.ascontiguousarray(seal, dtype=np.uint32)
        if control_root is None:
            po2 = 0
            if seal.size > self.out_size:
                po2 = int(self.lib.zko_fp_decode(int(seal[self.out_size])))
            if not 1 <= po2 <= 22:
                return "bad po2"
            control_root = self.control_root(po2, zk_cycles)
        cr = np.ascontiguousarray(control_root, dtype=np.uint32)
        err = self.lib.zko_verify_segment(self.h, seal, seal.size, cr.ctypes.data)
        return None if not err else err.decode()
