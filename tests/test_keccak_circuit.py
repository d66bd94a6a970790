"""KECCAK-F: a circuit whose witness is a real computation with a published known answer.

zeth's guest hashes through the keccak accelerator (/root/reference/run-parallel.sh:70,
/root/reference/guests/stateless-client/Cargo.toml:39); upstream proves those batches with risc0-circuit-keccak 4.0.2
(un-vendored: /root/reference/Cargo.lock:5289).  Here every 25 active trace rows are one FIPS 202 keccak-f[1600]
permutation constrained bit by bit (zeth_amd/circuits/keccak_f.py).  Pinned OUTSIDE the repository: the witness generators
(oracle C, HIP) must reproduce `hashlib.sha3_256`."""
import hashlib

import numpy as np
import pytest

import zko
from zeth_amd.circuits import keccak_f as K
from zeth_amd.circuits.desc import Circuit

PO2, ZK = 13, 1994
MSG = b"zeth: keccak accelerator call, block 19000000"


def _pub(msg):
    return np.array([w for lane in K.sha3_256_block(msg) for w in (lane & 0xFFFFFFFF, lane >> 32)], dtype=np.uint32)


def _lanes(data, n, row, r1):
    col = data.reshape(K.WD, n)[:, row]
    assert ((col == 0) | (col == r1)).all(), "active rows of the data group hold bits"
    bits = (col == r1)
    return [int(sum(1 << z for z in range(64) if bits[64 * lane + z])) for lane in range(60)]


def test_reference_permutation_is_sha3():
    """The plain-Python statement of the trace rows (what both witness generators are compared with) is FIPS 202."""
    for msg in (b"", b"abc", MSG, bytes(range(135))):
        assert K.digest_of_state(K.keccak_f(K.sha3_256_block(msg))) == hashlib.sha3_256(msg).digest()
    assert K.RC[:3] == [0x1, 0x8082, 0x800000000000808A] and K.RC[23] == 0x8000000080008008
    c = Circuit.parse(K.keccak_f_circuit())
    assert c.group_sizes == (4, 15, 3840) and c.global_sizes == (200, 4) and c.kind == 2 and c.combos == [(0,), (0, 1)]


@pytest.fixture(scope="module")
def oracle_witness(oracle):
    oc = zko.OracleCircuit(oracle, K.keccak_f_circuit())
    return oc, oc.witgen(PO2, ZK, seed=0x5EED0000, noise_seed=0x2E80, pub=_pub(MSG))


def test_oracle_witness_is_keccak_f_row_by_row_and_its_output_is_the_sha3_digest(oracle, oracle_witness):
    oc, (code, data, out) = oracle_witness
    n, r1 = 1 << PO2, int(oracle.zko_fp_encode(1))
    k_perms = (n - ZK) // 25
    base = 25 * (k_perms - 1)                                  # the last permutation got the padded message block
    rows, final = K.keccak_round_rows(K.sha3_256_block(MSG))
    for r in (0, 1, 11, 23):
        got = _lanes(data, n, base + r, r1)
        a, t, c, b = rows[r]
        assert got[:25] == a and got[25:30] == t and got[30:35] == c and got[35:] == b, f"round row {r}"
    got = _lanes(data, n, base + 24, r1)
    assert got[:25] == final and not any(got[25:])
    assert K.digest_of_state(got[:25]) == hashlib.sha3_256(MSG).digest()
    # out = the last permutation's output limbs, then its INPUT limbs: the claim binds the pair (an output alone always has a preimage)
    assert [int(x) for x in out] == [int(oracle.zko_fp_encode(v)) for v in K.out_words(final, K.sha3_256_block(MSG))]
    # a seeded permutation elsewhere in the trace
    import ctypes as C
    oracle.zko_keccak_lane.restype, oracle.zko_keccak_lane.argtypes = C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint32]
    st = [int(oracle.zko_keccak_lane(0x5EED0000, 3, lane)) for lane in range(25)]
    assert _lanes(data, n, 75, r1)[:25] == st and _lanes(data, n, 99, r1)[:25] == K.keccak_f(st)
    # code group: selectors and the round-constant bits of the previous row's round
    cg = code.reshape(K.WC, n)
    assert [int((cg[i] == r1).sum()) for i in range(7)] == [n - ZK, 1, n - ZK - 1, 24 * k_perms, 24 * k_perms, k_perms, 1]
    for j, pos in enumerate(K.RC_POS):
        assert int((cg[7 + j] == r1).sum()) == k_perms * sum((rc >> pos) & 1 for rc in K.RC)
    assert cg[6, 25 * k_perms - 1] == r1
    assert int((cg[14] == r1).sum()) == 1 and cg[14, 25 * (k_perms - 1)] == r1          # bind: row 0 of the last block


def test_oracle_proves_the_permutations_and_both_verifiers_accept(oracle):
    """po2 13: 247 permutations, the last one = SHA3-256 of MSG; the seal's `out` words ARE the digest."""
    from zeth_amd.hal import HalError, HostCircuit, fp_decode
    desc = K.keccak_f_circuit()
    oc = zko.OracleCircuit(oracle, desc)
    seal = oc.prove(PO2, ZK, 0x5EED0000, 0x2E80, pub=_pub(MSG))
    assert oc.verify(seal) is None
    limbs = [fp_decode(int(w)) for w in seal[:100]]
    state = [sum(limbs[4 * lane + j] << (16 * j) for j in range(4)) for lane in range(25)]
    assert K.digest_of_state(state) == hashlib.sha3_256(MSG).digest()
    root = oc.control_root(PO2, ZK)
    HostCircuit(desc).verify_segment(seal, root)                       # the product's host verifier
    # the INPUT state is public too: out[100..200) is the padded message block
    in_limbs = [fp_decode(int(w)) for w in seal[100:200]]
    assert [sum(in_limbs[4 * lane + j] << (16 * j) for j in range(4)) for lane in range(25)] == K.sha3_256_block(MSG)
    # a forged digest: flip one output limb; a forged preimage claim: flip one INPUT limb — both refused by both verifiers
    for pos, cur in ((3, limbs[3]), (100 + 7, in_limbs[7])):
        bad = seal.copy()
        bad[pos] = oracle.zko_fp_encode((cur + 1) & 0xFFFF)
        assert oc.verify(bad) is not None
        with pytest.raises(HalError):
            HostCircuit(desc).verify_segment(bad, root)


def test_an_output_alone_no_longer_has_a_witness(oracle):
    """Round-3 advisor finding: with only the output bound, ANY `out` was provable (keccak-f is a bijection: invert the rounds to
    get a preimage).  Now the claim holds (input, output): a witness built for output y by inverting keccak-f proves
    (f^-1(y), y) — a true statement — and cannot be passed off as a claim about another input."""
    oc = zko.OracleCircuit(oracle, K.keccak_f_circuit())
    honest_in = K.sha3_256_block(MSG)
    other_in = K.sha3_256_block(b"another message")
    code, data, out = oc.witgen(PO2, ZK, seed=1, noise_seed=2, pub=_pub(MSG))
    n = 1 << PO2
    mix = np.array([5, 6, 7, 8], dtype=np.uint32)
    accum = np.zeros(4 * n, dtype=np.uint32)
    oracle.zko_syn_accum(oc.h, PO2, ZK, zko.key_words(2), data, mix, accum)
    claimed = out.copy()
    claimed[100:200] = [int(oracle.zko_fp_encode(v)) for v in K.out_words(other_in)]      # same output, a different claimed input
    assert oc.check_rows(PO2, accum, code, data, out, mix) == -1                          # the honest (input, output) pair: every row holds
    bind_row = 25 * ((n - ZK) // 25 - 1)
    assert oc.check_rows(PO2, accum, code, data, claimed, mix) == bind_row                # the forged input claim fails exactly on the bind row
    assert honest_in != other_in


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_witness_generator_equals_the_oracle_and_sha3(hal, oracle, oracle_witness):
    from zeth_amd.prover import Segment, SegmentProver
    oc, (code, data, out) = oracle_witness
    prover = SegmentProver(hal, K.keccak_f_circuit())
    seg = Segment(index=0, po2=PO2, seed=0x5EED0000, noise_seed=0x2E80, pub=tuple(int(x) for x in _pub(MSG)))
    gcode, gdata, gout = prover.witgen(seg)
    assert np.array_equal(gcode.to_vec(), code)
    assert np.array_equal(gdata.to_vec(), data)
    assert np.array_equal(gout, out)
    n, r1 = 1 << PO2, int(oracle.zko_fp_encode(1))
    last = 25 * ((n - ZK) // 25) - 1
    assert K.digest_of_state(_lanes(gdata.to_vec(), n, last, r1)[:25]) == hashlib.sha3_256(MSG).digest()


@pytest.mark.gpu
@pytest.mark.parametrize("po2", [13, 14])
def test_gpu_seal_of_keccak_permutations_is_byte_identical_to_the_oracles(hal, oracle, po2):
    from zeth_amd.hal import fp_decode
    from zeth_amd.prover import Segment, SegmentProver
    desc = K.keccak_f_circuit()
    prover = SegmentProver(hal, desc)
    assert prover.circuit.compiled_parts() > 1                         # the generated kernels, not the interpreter
    pub = tuple(int(x) for x in _pub(MSG))
    seg = Segment(index=0, po2=po2, seed=0x5EED0000 + po2, noise_seed=0x2E80, pub=pub)
    rec = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    want = oc.prove(po2, ZK, seg.seed, seg.noise_seed, pub=np.asarray(pub, dtype=np.uint32))
    assert np.array_equal(rec.seal, want)
    root = prover.control_root(po2)
    assert np.array_equal(root, oc.control_root(po2, ZK))
    assert oc.verify(rec.seal, root) is None
    rec.verify(desc, root)
    limbs = [fp_decode(int(w)) for w in rec.seal[:100]]
    state = [sum(limbs[4 * lane + j] << (16 * j) for j in range(4)) for lane in range(25)]
    assert K.digest_of_state(state) == hashlib.sha3_256(MSG).digest()


@pytest.mark.gpu
def test_keccak_eval_check_generated_kernels_equal_interpreter_and_oracle(hal, oracle):
    """The structured (non-random) constraint system through all three evaluators at random trace values."""
    import ctypes as C
    desc = K.keccak_f_circuit()
    circ = hal.load_circuit(desc)
    po2 = 6                                                             # 256 domain points, 3858 columns
    dom = 4 << po2
    rng = np.random.default_rng(5)
    P = 2013265921
    groups = [rng.integers(0, P, size=w * dom, dtype=np.uint64).astype(np.uint32) for w in (4, K.WC, 3840)]
    out_g = rng.integers(0, P, size=200, dtype=np.uint64).astype(np.uint32)
    mix_g = rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
    pm = rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
    dg = [hal.copy_from("g", g) for g in groups]
    dgl = [hal.copy_from("o", out_g), hal.copy_from("m", mix_g)]
    a, b = hal.alloc_elem("check", 4 * dom), hal.alloc_elem("check", 4 * dom)
    circ.eval_check(a, dg, dgl, pm, po2)
    circ.eval_check(b, dg, dgl, pm, po2, use_interpreter=True)
    got = a.to_vec()
    assert np.array_equal(got, b.to_vec())
    oc = zko.OracleCircuit(oracle, desc)
    want = np.zeros(4 * dom, np.uint32)
    gp = (C.c_void_p * 3)(*[g.ctypes.data for g in groups])
    glp = (C.c_void_p * 2)(out_g.ctypes.data, mix_g.ctypes.data)
    oracle.zko_eval_check(oc.h, want, gp, glp, pm, po2)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_smallest_segment_seeded_inputs_and_no_room_for_a_permutation(hal, oracle):
    """Edge cases: po2 11 (54 active rows = 2 permutations, both seeded, 4 idle active rows) seals byte-identically; a segment
    whose active rows cannot hold one permutation is refused loudly."""
    from zeth_amd.hal import HalError
    from zeth_amd.prover import Segment, SegmentProver
    desc = K.keccak_f_circuit()
    prover = SegmentProver(hal, desc)
    seg = Segment(index=0, po2=11, seed=77, noise_seed=5)
    rec = prover.prove_segment(seg)
    oc = zko.OracleCircuit(oracle, desc)
    assert np.array_equal(rec.seal, oc.prove(11, ZK, 77, 5))
    rec.verify(desc, prover.control_root(11))
    with pytest.raises(HalError, match="no room for a permutation"):
        prover.witgen(Segment(index=0, po2=11, seed=1, noise_seed=5, zk_cycles=2030))      # 18 active rows < 25
