"""RECURSION on the CPU: the circuit's gates and copy argument, the program assembler against its plain
Python semantics, and the in-circuit STARK verifier (lift) run on real seals of the CPU oracle - accepted, and refused as soon
as one word of the seal is forged.  The device twin is compared with all of this bit for bit in test_recursion_gpu.py."""
import numpy as np
import pytest

import zko
from zeth_amd.circuits import p2_join, rec_verify as V, recursion as R, syn_air
from zeth_amd.circuits.desc import Circuit, P

RM = (1 << 32) % P
RINV = pow(RM, -1, P)
MIX = np.array([(i * 7919 + 13) * RM % P for i in range(20)], dtype=np.uint32)


def enc(vals):
    return np.array([int(v) * RM % P for v in vals], dtype=np.uint32)


def dec(vals):
    """Montgomery words -> canonical residues"""
    rinv = pow((1 << 32) % P, -1, P)
    return [int(v) * rinv % P for v in vals]


def small_program():
    pr = R.Program()
    x, y = pr.input(0, 4), pr.input(4, 4)
    s, m = pr.add(x, y), pr.mul(x, y)
    iv = pr.inv(m)
    pr.eq(pr.mul(m, iv), pr.const(1))
    a, b, _, _ = pr.unpack(x)
    bits = pr.bits31(a, 12)
    sel = pr.mux(bits[0], s, m)
    h = pr.p2([x, y, s, m, pr.zero(), pr.zero()])
    h2 = pr.p2([h[0], h[1], sel, iv, h[4], h[5]])
    # conditional-swap blocks (one Merkle level each): bit 0 of a = 5 is set, bit 1 is clear
    z = pr.zero()
    sw1 = pr.p2([h2[0], h2[1], x, y, bits[0], z], swap=True)
    sw0 = pr.p2([h2[0], h2[1], x, y, bits[1], z], swap=True)
    pr.eq(sw1[0], pr.p2([x, y, h2[0], h2[1], z, z])[0])
    pr.eq(sw0[1], pr.p2([h2[0], h2[1], x, y, z, z])[1])
    pr.eq(pr.is_zero(b), pr.zero())
    pk = pr.pack(2, x, y, s, m)
    k = pr.const(3, 1, 4, 1)
    pr.public(h2[0], h2[1], pk, pr.mul(k, bits[12]))
    return pr, (h2[0], h2[1], pk)


@pytest.fixture(scope="module")
def rec(oracle):
    return zko.OracleCircuit(oracle, R.recursion_circuit())


def test_circuit_shape():
    c = Circuit.parse(R.recursion_circuit())
    assert c.kind == 4 and c.group_sizes == (R.WA, R.WC, R.WD) == (12, 58, 72) and c.global_sizes == (16, 20)
    assert max(b for _, _, b in c.taps) == 1                       # only the previous row is ever read


def test_program_semantics_trace_and_seal(rec):
    pr, (h0, h1, pk) = small_program()
    zk = 50
    po2 = pr.min_po2(zk)
    assert po2 == 7 and R.block_rows(list(range(24)))[-1][0] == p2_join.permute(list(range(24)))
    blob = pr.finish(po2, zk)
    words = [5, 6, 7, 8, 11, 12, 13, 14]
    vals = R.run_program(pr, words)
    code, data, out = rec.rec_witgen(blob, enc(words))
    assert list(out[:12]) == list(enc([w for v in (vals[h0], vals[h1], vals[pk]) for w in v]))
    # x = 5 + 6 X + 7 X^2 + 8 X^3: bits of 5, pack(2, ...) = the third components
    assert vals[pk] == (7, 13, (7 + 13) % P, vals[pk][3])
    accum = rec.rec_accum(po2, code, data, MIX, zk)
    assert rec.check_rows(po2, accum, code, data, out, MIX) == -1
    seal = rec.prove_traces(po2, code, data, out, zk)
    root = rec.root_of_code(po2, code)
    assert rec.verify(seal, root, zk) is None
    forged = seal.copy()
    forged[0] = (int(forged[0]) + 1) % P
    assert rec.verify(forged, root, zk) is not None
    # another program has another control root: the seal does not verify under it
    pr2, _ = small_program()
    pr2.add(pr2.const(1), pr2.const(2))
    code2, _, _ = rec.rec_witgen(pr2.finish(po2, zk), enc(words))
    assert rec.verify(seal, rec.root_of_code(po2, code2), zk) is not None


def test_every_kind_of_gate_and_the_copy_argument_bind_the_trace(rec):
    """flip one cell of a wire that a given constraint family reads: the row check must find it"""
    pr, _ = small_program()
    zk, po2 = 50, 7
    blob = pr.finish(po2, zk)
    n = 1 << po2
    code, data, out = rec.rec_witgen(blob, enc([5, 6, 7, 8, 11, 12, 13, 14]))
    cg = code.reshape(R.WC, n)

    def broken(col, row):
        d = data.copy()
        d[col * n + row] = (int(d[col * n + row]) + 1) % P
        accum = rec.rec_accum(po2, code, d, MIX, zk)
        return rec.check_rows(po2, accum, code, d, out, MIX)
    assert rec.check_rows(po2, rec.rec_accum(po2, code, data, MIX, zk), code, data, out, MIX) == -1
    for name, sel_col, wire in (("GEN", R.C_QM, 3), ("MUX", R.C_MUX, 3), ("BOOL", R.C_BOOL, 0), ("PACK", R.C_PACK + 2, 3),
                                ("EMB", R.C_EMB, 4), ("PUB", R.C_PUB, 1)):
        rows = np.nonzero(cg[sel_col])[0]
        assert rows.size, name
        assert broken(4 * wire + 1, int(rows[0])) >= 0, name
    # the copy argument alone: a wire nobody's gate reads, but which shares a variable with another position
    r_in = R.BLOCK                                                  # input row of the second permutation: its wire a is h[0]
    assert broken(0, r_in) >= 0 and broken(R.D_S + 5, 3) >= 0 and broken(R.D_Q + 2, R.BLOCK + 2) >= 0 and broken(R.D_Q + 5, 5) >= 0 and broken(R.D_Q + 17, R.BLOCK + 6) >= 0
    # conditional-swap blocks: the input state is the wires' digests in the order the bit says, and nothing else - a block whose S
    # cells hold the OTHER order (with its permutation rows recomputed, so that only the `pios` binding can object), a swap bit that
    # is not a bit, and a stray capacity cell are all found
    swaps = np.nonzero(cg[R.C_PIOS])[0]
    assert swaps.size == 2 and all(int(r) % R.BLOCK == 0 for r in swaps) and not cg[R.C_PIO][swaps].any()
    dm = data.reshape(R.WD, n)
    for r0 in (int(r) for r in swaps):
        wires = [int(dm[j, r0]) for j in range(R.T)]
        bit = wires[16]
        assert bit in (0, int(enc([1])[0]))
        want = wires[8:16] + wires[0:8] if bit else wires[0:16]
        assert [int(dm[R.D_S + j, r0]) for j in range(16)] == want and int(dm[R.D_S + 16, r0]) == 0
        d = data.copy().reshape(R.WD, n)
        other = wires[0:16] if bit else wires[8:16] + wires[0:8]                   # the order the bit does NOT say
        rows = R.block_rows([int(x) for x in dec(other)] + [0] + [int(x) for x in dec(wires[17:])])
        for k, (S_, Q_) in enumerate(rows):
            d[R.D_S:R.D_S + R.T, r0 + k] = enc(S_)
            d[R.D_Q:R.D_Q + R.T, r0 + k] = enc(Q_)
        d = d.reshape(-1)
        bad = rec.check_rows(po2, rec.rec_accum(po2, code, d, MIX, zk), code, d, out, MIX)
        assert bad == r0, (bad, r0)                                                 # exactly the input row objects
        assert broken(R.D_S + 16, r0) >= 0                                          # S[16] must be zero
    # the out globals are bound to the PUB row
    o2 = out.copy()
    o2[3] = (int(o2[3]) + 1) % P
    assert rec.check_rows(po2, rec.rec_accum(po2, code, data, MIX, zk), code, data, o2, MIX) >= 0


def test_witness_generator_refuses_what_the_program_forbids(rec):
    pr, _ = small_program()
    blob = pr.finish(7, 50)
    with pytest.raises(RuntimeError, match="tie"):
        rec.rec_witgen(blob, enc([5, 0, 7, 8, 11, 12, 13, 14]))              # is_zero(b) must be 0
    with pytest.raises(RuntimeError, match="reduced"):
        rec.rec_witgen(blob, np.array([P] * 8, dtype=np.uint32))
    with pytest.raises(RuntimeError, match="more input"):
        rec.rec_witgen(blob, enc([1, 2, 3]))
    with pytest.raises(RuntimeError, match="trailing"):
        rec.rec_witgen(blob, enc([5, 6, 7, 8, 11, 12, 13, 14, 15]))
    bad = blob.copy()
    bad[0] ^= 1
    with pytest.raises(RuntimeError, match="header"):
        rec.rec_witgen(bad, enc([5, 6, 7, 8, 11, 12, 13, 14]))
    with pytest.raises(AssertionError):
        R.run_program(pr, [5, 0, 7, 8, 11, 12, 13, 14])


def test_bits_are_canonical(rec):
    """31 boolean wires summing to x do not pin x's bits unless the value is forced below P"""
    for x in (0, 1, P - 1, (1 << 27) - 1, 15 << 27):
        pr = R.Program()
        v = pr.input(0, 1)
        bits = pr.bits31(v)
        pr.public(bits[0], bits[27], bits[30], v)
        vals = R.run_program(pr, [x])
        assert [vals[b][0] for b in bits] == [(x >> i) & 1 for i in range(31)]
        code, data, out = rec.rec_witgen(pr.finish(8, 50), enc([x]))
        assert rec.check_rows(8, rec.rec_accum(8, code, data, MIX, 50), code, data, out, MIX) == -1


@pytest.mark.parametrize("cpo2", [8, 10])
def test_lift_runs_the_verifier_in_circuit(oracle, rec, cpo2):
    """a real seal (CPU oracle, SYN-tiny): the lift program's witness exists, satisfies every constraint, carries the
    segment's claim; one forged word anywhere in the seal and no witness exists.  cpo2 10 has a FRI round, 8 has none."""
    desc = syn_air.syn_tiny()
    child = zko.OracleCircuit(oracle, desc)
    czk = 50
    seal = child.prove(cpo2, czk)
    croot = child.control_root(cpo2, czk)
    pr = V.build_lift(desc, cpo2, [int(w) * RINV % P for w in croot])
    assert pr.n_inputs == seal.size + 8                             # the program reads exactly the seal, then A
    po2 = pr.min_po2()
    blob = pr.finish(po2)
    A = np.arange(1, 9, dtype=np.uint32)
    inputs = np.concatenate([seal, A])
    code, data, out = rec.rec_witgen(blob, inputs)
    claim_in = np.concatenate([seal[:5], croot])                    # out (4) ‖ po2 ‖ control root
    want = np.zeros(8, np.uint32)
    oracle.zko_hash_elem_slice(np.ascontiguousarray(claim_in), claim_in.size, 1, want)
    # what the lift publishes is claim' = hash_pair(receipt claim, (pre, post, 0..)): SYN-tiny has no state words, so (0, 0)
    from zeth_amd import recursion as host_rec
    assert np.array_equal(out[:8], host_rec.wrap_claim(want, 0, 0)) and np.array_equal(out[8:], A)
    accum = rec.rec_accum(po2, code, data, MIX)
    assert rec.check_rows(po2, accum, code, data, out, MIX) == -1
    rng = np.random.default_rng(cpo2)
    for k in [0, 4, 5, seal.size - 1] + [int(x) for x in rng.integers(0, seal.size, 200)]:        # no word of the seal is unbound
        forged = inputs.copy()
        forged[k] = (int(forged[k]) + 1 + int(rng.integers(0, P - 1))) % P
        with pytest.raises(RuntimeError, match="tie|inverse"):
            rec.rec_witgen(blob, forged)
    # the same seal under a program that expects another control root
    other = V.build_lift(desc, cpo2, [(int(w) * RINV + 1) % P for w in croot])
    with pytest.raises(RuntimeError, match="tie"):
        rec.rec_witgen(other.finish(po2), inputs)


def test_join_program_shape():
    """lifts of po2-20 segments fit po2 17, join(17, 17) and join(18, 18) fit po2 18: the recursion closes on itself"""
    desc = R.recursion_circuit()
    assert V.build_lift(syn_air.syn_a(), 20, list(range(8))).min_po2() == 17
    for a, b in ((17, 17), (18, 18)):
        pr = V.build_join(desc, a, b)
        assert pr.min_po2() == 18
        c = Circuit.parse(desc)
        assert pr.n_inputs > 2 * (16 + 1 + 4 * (len(c.taps) + 16)) and pr.pub is not None


def test_join_verifies_two_lifts_in_circuit(oracle, rec):
    """the whole recursion on the CPU: two SYN-tiny segment seals, lifted (sealed by the oracle), joined - the join's witness
    exists, satisfies every constraint of its trace, and carries hash_pair of the two claims; it does not exist for a
    child under another allowed root or with a membership path for another program"""
    from zeth_amd import recursion as host_rec
    desc = syn_air.syn_tiny()
    child = zko.OracleCircuit(oracle, desc)
    cpo2, czk = 8, 50
    croot = child.control_root(cpo2, czk)
    lift = V.build_lift(desc, cpo2, [int(w) * RINV % P for w in croot])
    lpo2 = lift.min_po2()
    lblob = lift.finish(lpo2)
    join = V.build_join(R.recursion_circuit(), lpo2, lpo2)
    jpo2 = join.min_po2()
    jblob = join.finish(jpo2)
    lcode, jcode = np.zeros(R.WC << lpo2, np.uint32), np.zeros(R.WC << jpo2, np.uint32)
    assert oracle.zko_rec_code(lblob, lblob.size, lcode) is None and oracle.zko_rec_code(jblob, jblob.size, jcode) is None
    levels = host_rec.allowed_tree([rec.root_of_code(lpo2, lcode), rec.root_of_code(jpo2, jcode)])
    A = levels[-1][0]

    def lifted(seed, allowed):
        code, data, out = rec.rec_witgen(lblob, np.concatenate([child.prove(cpo2, czk, seed=seed), allowed]))
        return rec.prove_traces(lpo2, code, data, out)
    left, right = lifted(100, A), lifted(101, A)
    path = host_rec.membership_words(levels, 0)

    def leaf_claim(seed):
        claim_in = np.concatenate([child.prove(cpo2, czk, seed=seed)[:5], croot])
        c = np.zeros(8, np.uint32)
        oracle.zko_hash_elem_slice(np.ascontiguousarray(claim_in), claim_in.size, 1, c)
        return c
    # a join reads, per child: seal, membership path, then the OPENING of the child's claim' (core, pre, post) - checked in-circuit
    opening = lambda seed: np.concatenate([leaf_claim(seed), np.zeros(2, np.uint32)])
    code, data, out = rec.rec_witgen(jblob, np.concatenate([left, path, opening(100), right, path, opening(101)]))
    assert np.array_equal(out[:8], host_rec.wrap_claim(host_rec.hash_pair(left[:8], right[:8]), 0, 0)) and np.array_equal(out[8:], A)
    assert np.array_equal(out[:8], host_rec.fold_leaf_claims([leaf_claim(100), leaf_claim(101)]))
    # join3: THREE children in one program, out = what join(join(a, b), c) would publish (the inner claim' is computed in-circuit)
    j3 = V.build_join(R.recursion_circuit(), lpo2, lpo2, lpo2)
    j3po2 = j3.min_po2()
    third = lifted(102, A)
    c3, d3, o3 = rec.rec_witgen(j3.finish(j3po2), np.concatenate([left, path, opening(100), right, path, opening(101), third, path, opening(102)]))
    inner = host_rec.wrap_claim(host_rec.hash_pair(left[:8], right[:8]), 0, 0)
    assert np.array_equal(o3[:8], host_rec.wrap_claim(host_rec.hash_pair(inner, third[:8]), 0, 0)) and np.array_equal(o3[8:], A)
    nested = host_rec._parent_node(host_rec._parent_node((leaf_claim(100), 0, 0), (leaf_claim(101), 0, 0)), (leaf_claim(102), 0, 0))
    assert np.array_equal(o3[:8], host_rec.wrap_claim(*nested))
    assert rec.check_rows(j3po2, rec.rec_accum(j3po2, c3, d3, MIX), c3, d3, o3, MIX) == -1
    with pytest.raises(RuntimeError, match="tie"):                              # the third child's opening must be ITS claim'
        rec.rec_witgen(j3.finish(j3po2), np.concatenate([left, path, opening(100), right, path, opening(101), third, path, opening(100)]))
    with pytest.raises(RuntimeError, match="tie"):                              # an opening that is not the child's claim': no witness
        rec.rec_witgen(jblob, np.concatenate([left, path, opening(101), right, path, opening(101)]))
    with pytest.raises(RuntimeError, match="tie"):                              # a state range the child's claim' does not commit to
        rec.rec_witgen(jblob, np.concatenate([left, path, leaf_claim(100), np.array([0, 5], np.uint32), right, path, opening(101)]))
    assert rec.check_rows(jpo2, rec.rec_accum(jpo2, code, data, MIX), code, data, out, MIX) == -1
    # the product's host-side check of a recursion receipt (zeth_amd/recursion.py RecReceipt.verify: no GPU involved): the
    # oracle-sealed lift verifies under the allowed set it was made for, with the claim of its segment - and only so
    from zeth_amd.hal import HalError
    roots = [rec.root_of_code(lpo2, lcode), rec.root_of_code(jpo2, jcode)]
    receipt = host_rec.RecReceipt(left, lpo2, 0, roots[0])
    claim = leaf_claim(100)
    receipt.verify(roots, [claim])
    for bad_roots, bad_claims, what in ((roots[1:], [claim], "allowed set"), (roots[::-1], [claim], "allowed-programs root"),
                                        (roots, [claim[::-1].copy()], "claim tree")):
        with pytest.raises(HalError, match=what):
            receipt.verify(bad_roots, bad_claims)
    forged = host_rec.RecReceipt(left.copy(), lpo2, 0, roots[0])
    forged.seal[left.size // 2] ^= 1
    with pytest.raises(HalError):
        forged.verify(roots, [claim])
    # the same checks as ONE host-only library call (zkh_succinct_verify: what a non-Python verifier runs; no GPU, no session)
    host_rec.succinct_verify(left, roots, 0, [claim])
    host_rec.succinct_verify(left, roots, 0, [(claim, 0, 0)])
    for args, what in (((left, roots[::-1], 1, [claim]), "allowed-programs root"), ((left, roots, 1, [claim]), "root receipt"),
                       ((left, roots, 0, [claim[::-1].copy()]), "claim tree"), ((left, roots, 0, [(claim, 0, 7)]), "claim tree"),
                       ((forged.seal, roots, 0, [claim]), "root receipt"), ((left, roots, 0, [claim, claim]), "claim tree"),
                       ((left, roots, 2, [claim]), "not in the allowed set"), ((left, roots, 0, [claim, claim, claim], 2), "equal ranges")):
        with pytest.raises(HalError, match=what):
            host_rec.succinct_verify(*args)
    stranger = lifted(101, np.arange(8, dtype=np.uint32))                       # a valid lift, handed another allowed root
    with pytest.raises(RuntimeError, match="tie"):
        rec.rec_witgen(jblob, np.concatenate([left, path, opening(100), stranger, path, opening(101)]))
    with pytest.raises(RuntimeError, match="tie"):
        rec.rec_witgen(jblob, np.concatenate([left, path, opening(100), right, host_rec.membership_words(levels, 1), opening(101)]))


def test_union_sorts_the_pair_and_resolve_binds_the_assumption_receipt(oracle, rec):
    """`union` and `resolve` on the CPU (ProverServer::{union, resolve}, risc0-zkvm 3.0.3, un-vendored): three
    SYN-tiny seals lifted by the oracle; union(a, b) publishes wrap(hash_pair of the SORTED pair, 0, 0) whichever way round the two
    are handed over; resolve binds an assumption receipt (here a lift: any receipt of the allowed set will do) to a conditional
    receipt whose claim' it opens.  No witness for a swap word that is not a bit, a membership path of another program, an opening
    that is not the conditional's, a forged assumption receipt.  (Sealed union / resolve receipts, verified: test_recursion_gpu.py.)"""
    from zeth_amd import recursion as host_rec
    desc = syn_air.syn_tiny()
    child = zko.OracleCircuit(oracle, desc)
    cpo2, czk = 8, 50
    croot = child.control_root(cpo2, czk)
    rdesc = R.recursion_circuit()
    lift = V.build_lift(desc, cpo2, [int(w) * RINV % P for w in croot])
    lpo2 = lift.min_po2()
    union = V.build_union(rdesc, lpo2, lpo2)
    upo2 = union.min_po2()
    resolve = V.build_resolve(rdesc, lpo2, lpo2)
    rpo2 = resolve.min_po2()
    blobs = [lift.finish(lpo2), union.finish(upo2), resolve.finish(rpo2)]
    roots = []
    for blob, po2 in zip(blobs, (lpo2, upo2, rpo2)):
        code = np.zeros(R.WC << po2, np.uint32)
        assert oracle.zko_rec_code(blob, blob.size, code) is None
        roots.append(rec.root_of_code(po2, code))
    levels = host_rec.allowed_tree(roots)
    A = levels[-1][0]

    def lifted(seed, allowed=A):
        code, data, out = rec.rec_witgen(blobs[0], np.concatenate([child.prove(cpo2, czk, seed=seed), allowed]))
        return rec.prove_traces(lpo2, code, data, out)

    def leaf_claim(seed):
        claim_in = np.concatenate([child.prove(cpo2, czk, seed=seed)[:5], croot])
        c = np.zeros(8, np.uint32)
        oracle.zko_hash_elem_slice(np.ascontiguousarray(claim_in), claim_in.size, 1, c)
        return c
    a, b, sess = lifted(100), lifted(101), lifted(102)
    path0, other_path = host_rec.membership_words(levels, 0), host_rec.membership_words(levels, 1)
    bit = lambda on: np.array([RM if on else 0], dtype=np.uint32)
    want, swap = host_rec.union_node(a[:8], b[:8])
    lo, hi = (b, a) if swap else (a, b)
    assert tuple(dec(lo[:8])) <= tuple(dec(hi[:8])) and np.array_equal(want, host_rec.wrap_claim(host_rec.hash_pair(lo[:8], hi[:8]), 0, 0))
    code, data, out = rec.rec_witgen(blobs[1], np.concatenate([a, path0, b, path0, bit(swap)]))
    assert np.array_equal(out[:8], want) and np.array_equal(out[8:], A)
    assert rec.check_rows(upo2, rec.rec_accum(upo2, code, data, MIX), code, data, out, MIX) == -1
    # the other way round, the other bit: the SAME claim - and the host's tree does not depend on the order either
    _, _, out_ba = rec.rec_witgen(blobs[1], np.concatenate([b, path0, a, path0, bit(not swap)]))
    assert np.array_equal(out_ba[:8], want)
    assert np.array_equal(host_rec.union_claims([a[:8], b[:8]]), want) and np.array_equal(host_rec.union_claims([b[:8], a[:8]]), want)
    # the wrong bit has a witness too - for a claim no verifier recomputes
    _, _, out_wrong = rec.rec_witgen(blobs[1], np.concatenate([a, path0, b, path0, bit(not swap)]))
    assert not np.array_equal(out_wrong[:8], want)
    with pytest.raises(RuntimeError, match="tie|boolean|bit"):                      # a swap word that is not a bit
        rec.rec_witgen(blobs[1], np.concatenate([a, path0, b, path0, np.array([2 * RM % P], dtype=np.uint32)]))
    with pytest.raises(RuntimeError, match="tie"):                                  # a membership path for another program
        rec.rec_witgen(blobs[1], np.concatenate([a, path0, b, other_path, bit(swap)]))
    # three assumptions: (a u b) u c - an odd one moves up
    c3 = host_rec.wrap_claim(leaf_claim(103), 0, 0)
    assert np.array_equal(host_rec.union_claims([a[:8], b[:8], c3]), host_rec.union_node(want, c3)[0])
    # resolve: the conditional receipt (the third lift: core = its segment's claim, state (0, 0)) bound to an assumption receipt
    opening = np.concatenate([leaf_claim(102), np.zeros(2, np.uint32)])
    useal, path1 = a, path0
    rcode, rdata, rout = rec.rec_witgen(blobs[2], np.concatenate([sess, path0, opening, useal, path1]))
    assert np.array_equal(rout[:8], host_rec.resolved_claim(sess[:8], 0, 0, a[:8])) and np.array_equal(rout[8:], A)
    assert rec.check_rows(rpo2, rec.rec_accum(rpo2, rcode, rdata, MIX), rcode, rdata, rout, MIX) == -1
    with pytest.raises(RuntimeError, match="tie"):                                  # an opening that is not the conditional's claim'
        rec.rec_witgen(blobs[2], np.concatenate([sess, path0, leaf_claim(100), np.zeros(2, np.uint32), useal, path1]))
    with pytest.raises(RuntimeError, match="tie"):                                  # a state range the conditional does not commit to
        rec.rec_witgen(blobs[2], np.concatenate([sess, path0, leaf_claim(102), np.array([0, 9], np.uint32), useal, path1]))
    forged = useal.copy()
    forged[useal.size // 2] ^= 1
    with pytest.raises(RuntimeError):                                               # a forged assumption receipt: no witness
        rec.rec_witgen(blobs[2], np.concatenate([sess, path0, opening, forged, path1]))
    # the library's host-only check (zkh_succinct_verify_resolved): a union tree over ONE assumption is that receipt's lift itself
    from zeth_amd.hal import HalError
    host_rec.succinct_verify(a, roots, 0, [], assumption_claims=[leaf_claim(100)])
    with pytest.raises(HalError, match="union tree"):
        host_rec.succinct_verify(a, roots, 0, [], assumption_claims=[leaf_claim(100), leaf_claim(101)])
    with pytest.raises(HalError, match="resolved root"):                             # a lift is not a RESOLVED receipt
        host_rec.succinct_verify(sess, roots, 0, [leaf_claim(102)], assumption_claims=[leaf_claim(100)])
    # what the host recomputes for a resolved receipt: the join tree over the session's leaves AND the union tree over the assumptions
    core, pre, post = host_rec._fold_root([leaf_claim(102)])
    assert np.array_equal(host_rec.wrap_claim(core, pre, post), sess[:8])
    assert np.array_equal(rout[:8], host_rec.resolved_claim(sess[:8], pre, post, host_rec.union_claims([host_rec.wrap_claim(leaf_claim(100), 0, 0)])))


def test_committed_digests_of_circuit_programs_and_oracle_witnesses():
    """tests/golden/recursion_digests.json (make_golden_recursion.py): the circuit description, the assembler's blobs and the
    oracle's witnesses / seal have not moved since the fixture was committed"""
    import importlib.util
    import json
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_recursion", os.path.join(g, "make_golden_recursion.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(g, "recursion_digests.json")) as fh:
        want = json.load(fh)
    assert mod.digests() == want


@pytest.mark.parametrize("seed", range(8))
def test_random_programs_python_semantics_c_interpreter_and_constraints_agree(rec, seed):
    """seeded random programs over every gate kind: the plain-Python semantics, the C witness generator and the circuit's own
    constraints (row check over the whole trace incl. the copy argument) agree; a random wire flipped afterwards is caught"""
    from rec_programs import random_program
    pr, words, pub = random_program(seed)
    rng = np.random.default_rng(seed)
    zk = 50
    po2 = pr.min_po2(zk)
    blob = pr.finish(po2, zk)
    vals = R.run_program(pr, words)
    code, data, out = rec.rec_witgen(blob, enc(words))
    assert list(out) == list(enc([w for v in pub for w in vals[v]]))
    accum = rec.rec_accum(po2, code, data, MIX, zk)
    assert rec.check_rows(po2, accum, code, data, out, MIX) == -1
    n = 1 << po2
    used = np.nonzero(code.reshape(R.WC, n)[R.C_QD])[0]                     # rows whose gate writes wire d
    r = int(used[int(rng.integers(0, used.size))])
    d2 = data.copy()
    col = 12 + int(rng.integers(0, 4))
    d2[col * n + r] = (int(d2[col * n + r]) + 1 + int(rng.integers(0, P - 1))) % P
    assert rec.check_rows(po2, rec.rec_accum(po2, code, d2, MIX, zk), code, d2, out, MIX) >= 0


def test_program_set_of_a_block_closes_at_po2_18():
    """build_programs (host only): a SYN-A block with po2-20 segments and a po2-18 tail needs 2 lifts, 3 fused lift2, 4 joins and the
    join3 of the largest size (three po2-18 children in one po2-18 proof: conditional-swap blocks made the room); lifts fit po2 17,
    everything above po2 18, and the set fits the allowed tree"""
    from zeth_amd import recursion as host_rec
    r = np.arange(8, dtype=np.uint32)
    programs = host_rec.build_programs(syn_air.syn_a(), {20: r, 18: r + 1})
    kinds = [k for k, _ in programs]
    assert kinds == [("lift", 20, 0), ("lift", 18, 0), ("lift2", 20, 20), ("lift2", 20, 18), ("lift2", 18, 18),
                     ("join", 17, 17), ("join", 17, 18), ("join", 18, 17), ("join", 18, 18), ("join3", 18, 18, 18)]
    assert [int(b[2]) for _, b in programs] == [17, 17, 18, 18, 18, 18, 18, 18, 18, 18] and len(programs) <= host_rec.N_ALLOWED
    assert [k for k, _ in host_rec.build_programs(syn_air.syn_a(), {20: r, 18: r + 1}, ternary=False)] == kinds[:-1]
    j3 = programs[-1][1]
    assert int(j3[9]) <= ((1 << 18) - R.ZK_CYCLES) // R.BLOCK and int(j3[10]) <= ((1 << 18) - R.ZK_CYCLES) - 2 * (((1 << 18) - R.ZK_CYCLES) // R.BLOCK)
    levels = host_rec.allowed_tree([np.full(8, i + 1, np.uint32) for i in range(len(programs))])
    assert [len(lv) for lv in levels] == [16, 8, 4, 2, 1]
    w = host_rec.membership_words(levels, 5)
    assert w.size == 4 * 9 and [int(w[9 * i]) for i in range(4)] == [host_rec.R * b % P for b in (1, 0, 1, 0)]


def test_shipped_program_manifest_is_what_the_builder_emits():
    """examples/recursion_programs.manifest.json: the SHA-256 of every lift / lift2 / join program (and of the RECURSION circuit
    description) a non-Python host loads for a SYN-A block (`python -m zeth_amd.circuits.rec_verify DIR`, control roots from
    circuits/control_roots.json).  Rebuilding the set from the tree must give exactly these files: the g++ host
    (examples/prove_session --recursion-dir) is reproducible from the repository alone."""
    import hashlib
    import json
    import os
    from zeth_amd import recursion as host_rec
    from zeth_amd.prover import shipped_control_root
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    man = json.load(open(os.path.join(root, "examples", "recursion_programs.manifest.json")))
    desc = syn_air.syn_a()
    roots = {po2: shipped_control_root(desc, po2) for po2 in (20, 18)}
    assert all(r is not None for r in roots.values())
    assert {p: [int(w) for w in r] for p, r in roots.items()} == {int(k): v for k, v in man["segment_control_roots"].items()}
    built = {}
    for kind, blob in host_rec.build_programs(desc, roots):
        name = "-".join(str(x) for x in kind[:2 if kind[0] == "lift" else 4 if kind[0] == "join3" else 3]) + ".zkr1"
        built[name] = {"words": int(blob.size), "po2": int(blob[2]), "sha256": hashlib.sha256(np.asarray(blob, dtype="<u4").tobytes()).hexdigest()}
    rdesc = np.asarray(R.recursion_circuit(), dtype="<u4")
    built["recursion.desc"] = {"words": int(rdesc.size), "sha256": hashlib.sha256(rdesc.tobytes()).hexdigest()}
    assert built == man["files"]


def test_fold_plan_pairs_first_then_three_at_a_time():
    """zeth_amd/recursion.py fold_plan: THE shape every fold follows (Recursion.fold / fold_segments, csrc/session.hip, the claim
    tree): every node of a level is consumed exactly once, the first level pairs, the levels above take three (remainder two: a
    join, one: moves up); a group of three is join(join(a, b), c) claim-wise."""
    from zeth_amd import recursion as host_rec
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 64, 513, 1024):
        width = n
        for lv, groups in enumerate(host_rec.fold_plan(n)):
            assert [k for g in groups for k in g] == list(range(width))
            assert all(len(g) == (2 if lv == 0 else 3) for g in groups[:-1]) and 1 <= len(groups[-1]) <= (2 if lv == 0 else 3)
            width = len(groups)
        assert width == 1
    plan = host_rec.fold_plan(1024)
    assert [len(g) for g in plan] == [512, 171, 57, 19, 7, 3, 1]
    assert sum(1 for lv in plan for g in lv if len(g) > 1) == 768                    # 512 lift2 + 256 joins / join3s (binary: 1023)
    rng = np.random.default_rng(5)
    leaves = [rng.integers(0, P, 8, dtype=np.uint32) for _ in range(8)]
    node = lambda c: (c, 0, 0)
    par = host_rec._parent_node
    pairs = [par(node(leaves[2 * k]), node(leaves[2 * k + 1])) for k in range(4)]
    want = par(par(par(pairs[0], pairs[1]), pairs[2]), pairs[3])                        # level 2: (p0 p1 p2) + p3 carried; level 3: a join
    assert np.array_equal(host_rec.fold_leaf_claims(leaves), host_rec.wrap_claim(*want))
    # folded in two ranges of four (one per rank), whose roots rank 0 then joins
    half = lambda ps: par(ps[0], ps[1])
    want2 = par(half(pairs[:2]), half(pairs[2:]))
    assert np.array_equal(host_rec.fold_leaf_claims(leaves, ranks=2), host_rec.wrap_claim(*want2))
    # states must chain through every group
    chained = [(leaves[i], 10 + i, 11 + i) for i in range(8)]
    core, pre, post = host_rec._fold_nodes(chained)
    assert (pre, post) == (10, 18)
    broken = list(chained)
    broken[5] = (leaves[5], 99, 16)
    from zeth_amd.hal import HalError
    with pytest.raises(HalError, match="do not chain"):
        host_rec.fold_leaf_claims(broken)
