"""csrc/rec_builder.hip — the recursion program builder in C++ (host only) — against its Python statement (circuits/rec_verify.py +
recursion.py Program.finish): IDENTICAL blobs, word for word, for every program of the committed manifest (SHA-256) and for other
circuit shapes (ConstExt operands and nested AndCond: SYN-HEAVY; a chained circuit: SYN-C); and the circuit descriptions compiled
into the library are the ones circuits/*.py build.  No GPU."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from zeth_amd import hal
from zeth_amd.circuits import codegen, rec_verify as V, recursion as R, syn_air
from zeth_amd.circuits.desc import P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U32P = C.POINTER(C.c_uint32)
RM = (1 << 32) % P


def cpp_build(kind, desc, po2s, roots_canonical=None, zk=R.ZK_CYCLES):
    """zkh_rec_build_program; roots are handed over the way the library hands control roots out: Montgomery words"""
    lib = hal.load_library()
    desc = np.ascontiguousarray(desc, dtype=np.uint32)
    p = (C.c_uint32 * 3)(*(list(po2s) + [0] * (3 - len(po2s))))
    r = None
    if roots_canonical is not None:
        r = np.ascontiguousarray([int(w) * RM % P for root in roots_canonical for w in root], dtype=np.uint32)
    blob, words = U32P(), C.c_size_t()
    err = lib.zkh_rec_build_program(kind, desc.ctypes.data_as(U32P), desc.size, p, None if r is None else r.ctypes.data_as(U32P), zk,
                                    C.byref(blob), C.byref(words))
    if err:
        msg = C.cast(err, C.c_char_p).value.decode()
        lib.zkh_free_error(err)
        raise hal.HalError(msg)
    out = np.ctypeslib.as_array(blob, shape=(words.value,)).copy()
    lib.zkh_free_seal(blob)
    return out


def test_shipped_circuit_descriptions_are_the_python_ones():
    lib = hal.load_library()
    names = [lib.zkh_shipped_circuit_name(i).decode() for i in range(lib.zkh_shipped_circuit_count())]
    assert names == list(codegen.shipped()) and lib.zkh_shipped_circuit_name(len(names)) is None
    for name, d in codegen.shipped().items():
        w, k = U32P(), C.c_size_t()
        assert not lib.zkh_shipped_circuit_desc(name.encode(), C.byref(w), C.byref(k))
        assert np.array_equal(np.ctypeslib.as_array(w, shape=(k.value,)), np.asarray(d, dtype=np.uint32)), name
    err = lib.zkh_shipped_circuit_desc(b"rv32im", C.byref(w), C.byref(k))
    assert err and b"rv32im" in C.cast(err, C.c_char_p).value
    lib.zkh_free_error(err)


def test_cpp_builder_emits_the_committed_program_set_word_for_word():
    """every program of examples/recursion_programs.manifest.json (2 lifts, 3 lift2, 4 joins, the join3 of a SYN-A block) from the C++
    builder: the manifest's sizes and SHA-256 - i.e. exactly the blobs `python -m zeth_amd.circuits.rec_verify` writes"""
    from zeth_amd.prover import shipped_control_root
    man = json.load(open(os.path.join(ROOT, "examples", "recursion_programs.manifest.json")))
    desc, rdesc = syn_air.syn_a(), R.recursion_circuit()
    rinv = pow(RM, -1, P)
    canon = {p: [int(w) * rinv % P for w in shipped_control_root(desc, p)] for p in (20, 18)}
    jobs = [("lift-20.zkr1", 0, desc, [20], [canon[20]]), ("lift-18.zkr1", 0, desc, [18], [canon[18]]),
            ("lift2-20-20.zkr1", 2, desc, [20, 20], [canon[20], canon[20]]), ("lift2-20-18.zkr1", 2, desc, [20, 18], [canon[20], canon[18]]),
            ("lift2-18-18.zkr1", 2, desc, [18, 18], [canon[18], canon[18]]),
            ("join-17-17.zkr1", 1, rdesc, [17, 17], None), ("join-17-18.zkr1", 1, rdesc, [17, 18], None), ("join-18-17.zkr1", 1, rdesc, [18, 17], None),
            ("join-18-18.zkr1", 1, rdesc, [18, 18], None), ("join3-18-18-18.zkr1", 3, rdesc, [18, 18, 18], None)]
    assert sorted(n for n, *_ in jobs) == sorted(k for k in man["files"] if k.endswith(".zkr1"))
    for name, kind, d, po2s, roots in jobs:
        blob = cpp_build(kind, d, po2s, roots)
        want = man["files"][name]
        assert blob.size == want["words"] and int(blob[2]) == want["po2"], name
        assert hashlib.sha256(blob.astype("<u4").tobytes()).hexdigest() == want["sha256"], name


def test_cpp_builder_equals_python_on_other_circuit_shapes():
    from zeth_amd.circuits import syn_heavy
    root, other = [5, 6, 7, 8, 9, 10, 11, 12], [12, 11, 10, 9, 8, 7, 6, 5]
    for name, desc, po2, zk in (("syn_tiny", syn_air.syn_tiny(), 8, 50), ("syn_heavy_small", syn_heavy.syn_heavy_small(), 10, R.ZK_CYCLES),
                                ("syn_chain_small", syn_air.syn_chain_small(), 9, R.ZK_CYCLES)):
        py = V.build_lift(desc, po2, root)
        assert np.array_equal(cpp_build(0, desc, [po2], [root], zk), py.finish(py.min_po2(zk), zk)), name
    from zeth_amd.circuits import keccak_f, p2_join
    for name, desc in (("keccak_f", keccak_f.keccak_f_circuit()), ("p2_join", p2_join.p2_join_circuit())):       # 3 840 columns, 43.8 k steps: a po2-19 lift
        py = V.build_lift(desc, 13, root)
        assert np.array_equal(cpp_build(0, desc, [13], [root]), py.finish(py.min_po2())), name
    chain = syn_air.syn_chain_small()                                # lift2 of a CHAINED circuit: the continuity assertion and the state words
    py = V.build_lift2(chain, 9, root, 8, other)
    assert np.array_equal(cpp_build(2, chain, [9, 8], [root, other]), py.finish(py.min_po2()))
    lift_po2 = V.build_lift(syn_air.syn_tiny(), 8, root).min_po2()
    for po2s in ([lift_po2, lift_po2], [lift_po2, lift_po2, lift_po2]):
        py = V.build_join(R.recursion_circuit(), *po2s)
        assert np.array_equal(cpp_build(1 if len(po2s) == 2 else 3, R.recursion_circuit(), po2s), py.finish(py.min_po2()))
    # union (kind 4: the sorted pair, a swap bit) and resolve (kind 5: the conditional receipt opened, bound to its assumption receipt)
    for kind, build in ((4, V.build_union), (5, V.build_resolve)):
        py = build(R.recursion_circuit(), lift_po2, lift_po2 + 1)
        assert np.array_equal(cpp_build(kind, R.recursion_circuit(), [lift_po2, lift_po2 + 1]), py.finish(py.min_po2())), kind


def test_cpp_builder_refuses_what_it_cannot_build():
    rdesc = R.recursion_circuit()
    with pytest.raises(hal.HalError, match="joins take none"):
        cpp_build(1, rdesc, [17, 17], [[1] * 8, [2] * 8])
    with pytest.raises(hal.HalError, match="RECURSION circuit"):
        cpp_build(1, syn_air.syn_a(), [17, 17])
    with pytest.raises(hal.HalError, match="control roots"):
        cpp_build(0, syn_air.syn_a(), [20])
    with pytest.raises(hal.HalError, match="bad circuit description"):
        cpp_build(0, np.arange(64, dtype=np.uint32), [12], [[1] * 8])
    bad = np.asarray(syn_air.syn_tiny(), dtype=np.uint32).copy()
    bad[-5 * 3 + 1] = 0xFFFF                                          # a step whose first operand names no value
    with pytest.raises(hal.HalError, match="operand out of range|result is not"):
        cpp_build(0, bad, [8], [[1] * 8], 50)
    with pytest.raises(hal.HalError, match="child po2"):
        cpp_build(0, syn_air.syn_a(), [40], [[1] * 8])
