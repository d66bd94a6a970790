"""All-maximal and alternating input vectors through every lazy-arithmetic path on the device (round-5 verdict, item 2 b / c).

Random inputs almost never meet the worst case of a lazy representative: a value in [0, 2P) is near 2P only for operands near P,
a 64-bit sum of products is near its bound only if EVERY term is.  Round 5 shipped an accumulator overflow that 197 random circuits
with edge words sprinkled on 1/8 of the cells did not hit.  Here every tap word, global, mix word and poly_mix component is THE SAME
extreme word — P-1, (P-1)/2, (P+1)/2, 1 — or alternates between P-1 and 0 by row / by column, so every operand of every product is at
its extreme at once; the generated kernels, the on-device interpreter and the oracle's literal interpreter must still agree word for
word.  The same columns go through hash_rows at the BASELINE width (208 columns: 13 absorb blocks, partial rounds three at a time on
unreduced weighted sums, csrc/poseidon2.h) and through the Merkle layers above.  tools/check_bounds.py proves the eval_check side
for ALL inputs; these vectors are the measured end of the same claim, and the only check the Poseidon2 fast path has beyond its
host-side bounds test (tests/cpp/poseidon2_bounds.cpp)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import zko  # noqa: E402

from zeth_amd.circuits import codegen  # noqa: E402

pytestmark = pytest.mark.gpu
P = 2013265921
PATTERNS = ("max", "half_lo", "half_hi", "one", "rows_alternate", "cols_alternate", "max_zero_mix")


def _fill(pattern: str, w: int, dom: int) -> np.ndarray:
    """a W x dom column-major matrix of raw Montgomery words"""
    g = np.empty((w, dom), np.uint32)
    if pattern in ("max", "max_zero_mix"):
        g[:] = P - 1
    elif pattern == "half_lo":
        g[:] = (P - 1) // 2
    elif pattern == "half_hi":
        g[:] = (P + 1) // 2
    elif pattern == "one":
        g[:] = 1
    elif pattern == "rows_alternate":
        g[:, 0::2] = P - 1
        g[:, 1::2] = 0
    else:
        g[0::2, :] = P - 1
        g[1::2, :] = 0
    return np.ascontiguousarray(g.reshape(-1))


def _word(pattern: str) -> int:
    return {"max": P - 1, "half_lo": (P - 1) // 2, "half_hi": (P + 1) // 2, "one": 1, "rows_alternate": P - 1, "cols_alternate": P - 1,
            "max_zero_mix": 0}[pattern]


def _three_evaluators(hal, oracle, circ, desc, po2, patterns=PATTERNS, interpreter=True):
    oc = zko.OracleCircuit(oracle, desc)
    dom = 4 << po2
    widths = [int(x) for x in desc[3:6]]
    n_out, n_mix = int(desc[7]), int(desc[8])
    for pat in patterns:
        gs = [_fill(pat, w, dom) for w in widths]
        out = np.full(max(1, n_out), P - 1 if pat != "one" else 1, np.uint32)
        mix = np.full(max(1, n_mix), P - 1 if pat != "one" else 1, np.uint32)
        poly_mix = np.full(4, _word(pat), np.uint32)
        want = np.zeros(4 * dom, np.uint32)
        gp = (C.c_void_p * 3)(*[a.ctypes.data for a in gs])
        glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
        oracle.zko_eval_check(oc.h, want, gp, glp, poly_mix, po2)
        dev = [hal.copy_from("g", g) if g.size else hal.alloc("g", 0) for g in gs]
        g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
        for interp in ((False, True) if interpreter else (False,)):
            check = hal.alloc_elem("check", 4 * dom)
            circ.eval_check(check, dev, [g_out, g_mix], poly_mix, po2, use_interpreter=interp)
            got = check.to_vec()
            assert np.array_equal(got, want), (f"eval_check differs from the oracle on the `{pat}` vector "
                                               f"({'interpreter' if interp else circ.kernel_kind() + ' kernels'}): first word {int(np.argmax(got != want))}")


@pytest.mark.parametrize("name", list(codegen.shipped().keys()))
def test_shipped_circuits_on_extreme_vectors(hal, oracle, name):
    """every circuit whose kernels are compiled into the library, on its built-in kernels and on the interpreter"""
    desc = codegen.shipped()[name]
    circ = hal.load_circuit(desc)
    assert circ.kernel_kind() == "builtin"
    # the interpreter keeps a circuit's live values in LDS: the two widest circuits do not fit (the library says so), two evaluators then
    interp = name not in ("keccak_f",)
    _three_evaluators(hal, oracle, circ, desc, 6, interpreter=interp)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_circuits_on_extreme_vectors(hal, oracle, seed, tmp_path, monkeypatch):
    """the random constraint systems of tests/test_fuzz_gpu.py (same generator arguments; tools/check_bounds.py checks the same
    kernels statically), compiled at load time"""
    from zeth_amd.circuits import syn_random
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    groups = [(4, 6, 12), (8, 5, 20), (4, 16, 33)][seed % 3]
    desc = syn_random.random_circuit(seed, groups=groups, n_values=160 + 40 * (seed % 4), n_constraints=30 + 15 * (seed % 3), max_back=1 + seed % 4)
    circ = hal.load_circuit(desc, jit=True)
    assert circ.kernel_kind() == "attached"
    _three_evaluators(hal, oracle, circ, desc, 5)


@pytest.mark.parametrize("cols", [208, 16, 32, 33, 1])
@pytest.mark.parametrize("pattern", ["max", "half_lo", "half_hi", "rows_alternate", "cols_alternate"])
def test_hash_rows_and_tree_on_extreme_columns(hal, oracle, cols, pattern):
    """k_hash_rows at the BASELINE width (208 columns) and at the block edges, then every Merkle layer above (lane-per-parent, the
    8-lane cooperative kernels of the narrow layers), on columns that hold one extreme word / alternate between P-1 and 0: the
    f64 external rounds and the grouped partial rounds (three at a time on unreduced weighted sums) at their worst-case operands"""
    rows = 1 << 13
    mat = _fill(pattern, cols, rows)
    m = hal.copy_from("m", mat)
    nodes = hal.alloc_digest("nodes", 2 * rows)
    hal.merkle_build(nodes, m, rows)
    got = nodes.to_vec()
    want = np.zeros(2 * rows * 8, np.uint32)
    leaves = np.zeros(rows * 8, np.uint32)
    oracle.zko_hash_rows(leaves, rows, mat, rows * cols)
    want[rows * 8:] = leaves
    size = rows
    while size > 1:
        oracle.zko_hash_fold(want, size, size // 2)
        size //= 2
    assert np.array_equal(got[8:], want[8:])
    # digests of extreme inputs as the NEXT layer's input are ordinary words; an all-(P-1) digest layer is not reachable from a
    # matrix, so it is fed directly: hash_fold of 2^12 pairs of all-(P-1) digests, through both fold kernels
    for parents in (1 << 12, 1 << 5):
        io = np.zeros(4 * parents * 8, np.uint32)
        io[2 * parents * 8:] = P - 1
        want2 = io.copy()
        oracle.zko_hash_fold(want2, 2 * parents, parents)
        d = hal.copy_from("io", io)
        hal.hash_fold(d, 2 * parents, parents)
        assert np.array_equal(d.to_vec()[parents * 8: 2 * parents * 8], want2[parents * 8: 2 * parents * 8])


def test_poseidon2_mix_on_extreme_states(hal, oracle):
    """the bare permutation (zkh_poseidon2_mix: the device kernel with the context's tables) on states whose 24 words are all
    P-1, all (P+-1)/2, one-hot P-1 and alternating — against the oracle's literal permutation"""
    states = [np.full(24, P - 1, np.uint32), np.full(24, (P - 1) // 2, np.uint32), np.full(24, (P + 1) // 2, np.uint32)]
    for k in range(24):
        s = np.zeros(24, np.uint32); s[k] = P - 1
        states.append(s)
    a = np.zeros(24, np.uint32); a[0::2] = P - 1
    b = np.zeros(24, np.uint32); b[1::2] = P - 1
    states += [a, b]
    flat = np.ascontiguousarray(np.concatenate(states))
    want = flat.copy()
    for k in range(len(states)):
        st = np.ascontiguousarray(want[24 * k: 24 * k + 24])
        oracle.zko_poseidon2_mix(st)
        want[24 * k: 24 * k + 24] = st
    d = hal.copy_from("states", flat)
    hal.poseidon2_mix(d)
    assert np.array_equal(d.to_vec(), want)


def test_syn_huge_loads_as_data_and_three_evaluators_agree(hal, oracle, tmp_path, monkeypatch):
    """SYN-HUGE (circuits/syn_heavy.py syn_huge: > 250 k steps, > 2 k taps, ~15 k constraints) is not compiled into the library: it
    arrives as data, its ~26 kernels are generated and compiled at load time (parts in parallel) and attached; generated == interpreter
    == oracle on random and extreme vectors, and a whole small seal is byte-identical to the oracle's."""
    from zeth_amd.circuits import syn_heavy
    from zeth_amd.prover import Segment, SegmentProver
    monkeypatch.setenv("ZKH_JIT_CACHE", str(tmp_path))
    desc = syn_heavy.syn_huge()
    prover = SegmentProver(hal, desc)
    circ = prover.circuit
    assert circ.kernel_kind() == "attached" and circ.compiled_parts() >= 16
    _three_evaluators(hal, oracle, circ, desc, 5, patterns=("max", "half_hi", "rows_alternate"))
    rec = prover.prove_segment(Segment(index=0, po2=9, seed=21, noise_seed=22, zk_cycles=200))
    want = zko.OracleCircuit(oracle, desc).prove(9, 200, 21, 22)
    assert np.array_equal(rec.seal, want)


EXTREME_WORDS = (P - 1, (P - 1) // 2, (P + 1) // 2, 1)


@pytest.mark.parametrize("word", EXTREME_WORDS)
def test_hand_written_lazy_sums_on_extreme_inputs(hal, oracle, word):
    """The hand-written kernels that sum unreduced products — batch_evaluate_any (four Fp x Fp4 products per 64-bit accumulator, also
    the bit-reversed form the prover uses from po2 14 on), mix_poly_coeffs (a running mix power times every coefficient), fri_fold
    (sixteen Fp4 x Fp4 products), eltwise_sum_extelem, the combos scans — with EVERY operand the same extreme word: coefficients,
    evaluation points, mix values.  Bit-exact against the oracle."""
    rng = np.random.default_rng(word % 1000)
    full = lambda k: np.full(k, word, np.uint32)                     # noqa: E731
    # batch_evaluate_any: natural and bit-reversed coefficient order
    for po, polys, evals in ((1 << 12, 3, 5), (70000, 2, 3), (1 << 16, 2, 4)):
        coeffs, which, xs = full(po * polys), rng.integers(0, polys, size=evals).astype(np.uint32), full(4 * evals)
        want = np.zeros(4 * evals, np.uint32)
        oracle.zko_batch_evaluate_any(coeffs, coeffs.size, polys, which, xs, evals, want)
        out = hal.alloc_extelem("out", evals)
        hal.batch_evaluate_any(hal.copy_from("c", coeffs), polys, hal.copy_from("w", which), hal.copy_from("x", xs), out)
        assert np.array_equal(out.to_vec(), want), ("batch_evaluate_any", po)
        if po == 1 << 16:
            br = coeffs.copy()
            oracle.zko_batch_bit_reverse(br, br.size, polys)
            out2 = hal.alloc_extelem("out", evals)
            hal.batch_evaluate_any_bitrev(hal.copy_from("c", br), polys, hal.copy_from("w", np.sort(which)), hal.copy_from("x", xs), out2)
            want2 = np.zeros(4 * evals, np.uint32)
            oracle.zko_batch_evaluate_any(coeffs, coeffs.size, polys, np.sort(which), xs, evals, want2)
            assert np.array_equal(out2.to_vec(), want2), "batch_evaluate_any_bitrev"
    # mix_poly_coeffs
    count, input_size, ncombo = 4096, 17, 3
    inp, combos = full(input_size * count), np.sort(rng.integers(0, ncombo, size=input_size)).astype(np.uint32)
    out0, mix_start, mix = full(4 * ncombo * count), full(4), full(4)
    want = out0.copy()
    oracle.zko_mix_poly_coeffs(want, mix_start, mix, inp, combos, input_size, count)
    out = hal.copy_from("out", out0)
    hal.mix_poly_coeffs(out, mix_start, mix, hal.copy_from("in", inp), hal.copy_from("cb", combos), input_size, count)
    assert np.array_equal(out.to_vec(), want), "mix_poly_coeffs"
    # fri_fold
    count = 4096
    inp, mix = full(4 * 16 * count), full(4)
    want = np.zeros(4 * count, np.uint32)
    oracle.zko_fri_fold(want, want.size, inp, mix)
    out = hal.alloc_elem("o", 4 * count)
    hal.fri_fold(out, hal.copy_from("i", inp), mix)
    assert np.array_equal(out.to_vec(), want), "fri_fold"
    # eltwise_sum_extelem
    count, k = 777, 5
    inp = full(4 * count * k)
    want = np.zeros(4 * count, np.uint32)
    oracle.zko_eltwise_sum_extelem(want, want.size, inp, count * k)
    out = hal.alloc_elem("s", 4 * count)
    hal.eltwise_sum_extelem(out, hal.copy_from("i", inp))
    assert np.array_equal(out.to_vec(), want), "eltwise_sum_extelem"


@pytest.mark.parametrize("log_n,bits", [(20, 2), (22, 2), (18, 2), (16, 0)])
def test_ntt_round_trip_on_extreme_columns(hal, oracle, log_n, bits):
    """expand-NTT (the lazy signed butterflies at 2^20 / 2^22) and the inverse transform with the fused zk shift on columns that hold
    one extreme word or alternate: the forward result equals the oracle's, and interpolating it back returns the input"""
    n_out, n_in = 1 << log_n, (1 << log_n) >> bits
    cols = [np.full(n_in, w, np.uint32) for w in EXTREME_WORDS] + [np.where(np.arange(n_in) % 2 == 0, P - 1, 0).astype(np.uint32)]
    x = np.concatenate(cols)
    want = np.zeros(len(cols) * n_out, np.uint32)
    oracle.zko_batch_expand_into_evaluate_ntt(want, want.size, x, x.size, len(cols), bits)
    out = hal.alloc_elem("out", len(cols) * n_out)
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from("in", x), len(cols), bits)
    assert np.array_equal(out.to_vec(), want)
    if bits == 0:
        hal.batch_interpolate_ntt(out, len(cols))
        back = x.copy()
        assert np.array_equal(out.to_vec(), back)
    inv = x.copy()
    oracle.zko_batch_interpolate_ntt(inv, inv.size, len(cols))
    oracle.zko_zk_shift(inv, inv.size, len(cols))
    buf = hal.copy_from("io", x)
    hal.batch_interpolate_ntt_zk_shift(buf, len(cols))
    assert np.array_equal(buf.to_vec(), inv)
