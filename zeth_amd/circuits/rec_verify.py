"""The STARK verifier as a RECURSION program: lift and join (BASELINE.json config 5).

`verify_seal` restates csrc/verifier.hip `zkh_verify_segment` (= risc0-zkp 3.0.2 src/verify/{mod.rs, merkle.rs, fri.rs,
read_iop.rs}, un-vendored: /root/reference/Cargo.lock:5393) statement by statement in the gate language of recursion.py: the
Fiat-Shamir sponge, every Merkle opening, the constraint polynomial at z against check(z), the DEEP combination and every
FRI fold of all 50 queries run INSIDE the circuit, on a child seal that enters as private witness words.  What upstream's
`lift` / `join` programs (risc0-circuit-recursion 4.0.2 `.zkr`, un-vendored: /root/reference/Cargo.lock:5305) do:

  lift(circuit, po2):  verify one segment seal;  out = claim(segment) ‖ A
  lift2(circuit, po2_l, po2_r):  verify two segment seals;  out = hash_pair(claim_l, claim_r) ‖ A   (lift + lift + join fused)
  join(po2_l, po2_r):  verify two recursion seals, require that both carry this A and that their programs' control roots are
                       members of the allowed set A;  out = hash_pair(claim_l, claim_r) ‖ A

`A` is the Merkle root of the control roots of the allowed programs (upstream: the control-id allow list): a program cannot
contain its own root, so membership is proven against a public root that every level hands down and the final verifier
checks (zeth_amd/recursion.py RecReceipt.verify).  claim(segment) is csrc/verifier.hip zkh_receipt_claim: Poseidon2 over (out globals, po2,
control root).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .desc import (GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, OP_ADD, OP_AND_COND, OP_AND_EQZ, OP_CONST, OP_CONST_EXT, OP_GET,
                   OP_GET_GLOBAL, OP_MUL, OP_SUB, OP_TRUE, Circuit, P)
from .recursion import NW, Program

QUERIES, INV_RATE, FRI_FOLD, FRI_MIN_DEGREE, CHECK_SIZE, EXT = 50, 4, 16, 256, 16, 4
ALLOWED_DEPTH = 4                                         # the allowed-programs tree holds 16 control roots
ROU_FWD = [pow(137, 1 << (27 - k), P) for k in range(28)]
ROU_REV = [pow(w, P - 2, P) for w in ROU_FWD]


def log2_ceil(x: int) -> int:
    return (x - 1).bit_length()


def bitrev4(i: int) -> int:
    return int(f"{i:04b}"[::-1], 2)


class Sponge:
    """ReadIop's copy of the Fiat-Shamir sponge (csrc/verifier.hip:50-77 = risc0-zkp verify/read_iop.rs + poseidon2/rng.rs):
    24 cells = 6 packed wires; commit adds a digest into cells 0..7 and mixes, elem squeezes the next rate cell"""

    def __init__(self, pr: Program):
        self.pr = pr
        z = pr.zero()
        self.cells = [z] * NW
        self.used = 0
        self._unpacked: Dict[int, Tuple[int, int, int, int]] = {}

    def mix(self):
        self.cells = self.pr.p2(self.cells)
        self.used = 0

    def commit(self, digest: Sequence[int]):
        if self.used:
            self.mix()
        self.cells = [self.pr.add(self.cells[0], digest[0]), self.pr.add(self.cells[1], digest[1])] + list(self.cells[2:])
        self.mix()

    def elem(self) -> int:
        if self.used == 16:
            self.mix()
        w = self.cells[self.used // 4]
        if w not in self._unpacked:
            self._unpacked[w] = self.pr.unpack(w)
        e = self._unpacked[w][self.used % 4]
        self.used += 1
        return e

    def ext(self) -> int:
        if self.used == 16:
            self.mix()
        if self.used % 4 == 0:
            w = self.cells[self.used // 4]
            self.used += 4
            return w
        return self.pr.pack(0, *(self.elem() for _ in range(4)))


class Verifier:
    def __init__(self, pr: Program, base: int = 0):
        self.pr = pr
        self.pos = base                        # next input word (ReadIop::read)
        self.io = Sponge(pr)

    # ---- seal words
    def read(self, count: int) -> List[int]:
        """`count` words as packed wires (4 words each, the last zero padded)"""
        out = [self.pr.input(self.pos + 4 * i, min(4, count - 4 * i)) for i in range((count + 3) // 4)]
        self.pos += count
        if count % 4:                          # the padding of a partial wire is part of what gets hashed: it must BE zero
            u = self.pr.unpack(out[-1])
            for k in range(count % 4, 4):
                self.pr.eq(u[k], self.pr.zero())
        return out

    # ---- hashing
    def elems(self, wires: Sequence[int]) -> List[int]:
        """hash_elem_slice (csrc/verifier.hip:28-42) over the words of the packed wires: whole sponge blocks of 4 wires, the rate is
        overwritten, the capacity carried; missing wires are zero"""
        pr, z = self.pr, self.pr.zero()
        cap = [z, z]
        blocks = max(1, (len(wires) + 3) // 4)
        for b in range(blocks):
            rate = list(wires[4 * b:4 * b + 4])
            rate += [z] * (4 - len(rate))
            o = pr.p2(rate + cap)
            cap = o[4:]
        return o[:2]

    def pair(self, a: Sequence[int], b: Sequence[int]) -> List[int]:
        z = self.pr.zero()
        return self.pr.p2([a[0], a[1], b[0], b[1], z, z])[:2]

    def pair_at(self, bit: int, cur: Sequence[int], sib: Sequence[int]) -> List[int]:
        """one Merkle level: hash_pair(cur, sib) if the index bit is clear, hash_pair(sib, cur) if it is set — a conditional-swap
        block (recursion.py `pios`), no gate"""
        return self.pr.p2([cur[0], cur[1], sib[0], sib[1], bit, self.pr.zero()], swap=True)[:2]

    # ---- MerkleTreeVerifier (csrc/verifier.hip:81-117 = risc0-zkp verify/merkle.rs)
    def tree_init(self, rows: int, cols: int) -> dict:
        layers = log2_ceil(rows)
        top_layer = 0
        for i in range(1, layers):
            if (1 << i) > QUERIES:
                break
            top_layer = i
        top_size = 1 << top_layer
        w = self.read(8 * top_size)
        top: List[Optional[List[int]]] = [None] * (2 * top_size)
        for i in range(top_size):
            top[top_size + i] = [w[2 * i], w[2 * i + 1]]
        for i in range(top_size - 1, 0, -1):
            top[i] = self.pair(top[2 * i], top[2 * i + 1])
        self.io.commit(top[1])
        return {"rows": rows, "cols": cols, "layers": layers, "top_layer": top_layer, "top": top}

    def tree_open(self, t: dict, idx_bits: Sequence[int]) -> List[int]:
        """idx_bits: the row index, least significant bit first (log2(rows) BOOL wires) -> the opened row as packed wires"""
        pr = self.pr
        assert len(idx_bits) == t["layers"]
        col = self.read(t["cols"])
        cur = self.elems(col)
        low = t["layers"] - t["top_layer"]
        for lvl in range(low):
            sib = self.read(8)
            cur = self.pair_at(idx_bits[lvl], cur, sib)
        cand = t["top"][1 << t["top_layer"]:]
        for b in idx_bits[low:]:
            cand = [[pr.mux(b, cand[2 * i][k], cand[2 * i + 1][k]) for k in range(2)] for i in range(len(cand) // 2)]
        pr.eq(cur[0], cand[0][0])
        pr.eq(cur[1], cand[0][1])
        return col

    # ---- arithmetic helpers
    def horner(self, coeffs: Sequence[int], x: int) -> int:
        pr = self.pr
        acc = coeffs[-1]
        for c in reversed(coeffs[:-1]):
            acc = pr.muladd(acc, x, c)
        return acc

    def pow_bits(self, g: int, bits: Sequence[int]) -> int:
        """g^(sum 2^i bit_i) for a constant g"""
        pr = self.pr
        acc = pr.const(1)
        for i, b in enumerate(bits):
            gi = pow(g, 1 << i, P)
            acc = pr.gen(b, acc, acc, qM=gi - 1, qC=1)
        return acc

    def poly_ext(self, c: Circuit, poly_mix: int, u: Sequence[int], out_words: Sequence[int], mix_words: Sequence[int]) -> int:
        """PolyExtStepDef::step over ExtElem (csrc/verifier.hip:125-144 = risc0-zkp adapter.rs): -> the constraint polynomial's value.
        A MixState's `mul` is a static power of poly_mix, so only `tot` needs wires."""
        pr = self.pr
        fv: List[int] = []
        mv: List[Tuple[Optional[int], int]] = []                 # (tot wire or None for zero, static exponent of poly_mix)
        pw = [pr.const(1), poly_mix]

        def power(e: int) -> int:
            while len(pw) <= e:
                pw.append(pr.mul(pw[-1], poly_mix))
            return pw[e]
        for op, a, b, cc, d in c.steps:
            if op == OP_CONST:
                fv.append(pr.const(a))
            elif op == OP_CONST_EXT:
                fv.append(pr.const(a, b, cc, d))
            elif op == OP_GET:
                fv.append(u[a])
            elif op == OP_GET_GLOBAL:
                fv.append((out_words if a == GLOBAL_OUT else mix_words)[b])
            elif op == OP_ADD:
                fv.append(pr.add(fv[a], fv[b]))
            elif op == OP_SUB:
                fv.append(pr.sub(fv[a], fv[b]))
            elif op == OP_MUL:
                fv.append(pr.mul(fv[a], fv[b]))
            elif op == OP_TRUE:
                mv.append((None, 0))
            elif op == OP_AND_EQZ:
                tot, e = mv[a]
                mv.append((pr.muladd(power(e), fv[b], tot) if tot is not None else pr.mul(power(e), fv[b]), e + 1))
            elif op == OP_AND_COND:
                (tot, e), (itot, ie) = mv[a], mv[cc]
                if itot is None:
                    mv.append((tot, e + ie))
                else:
                    t = pr.mul(fv[b], itot)
                    mv.append((pr.muladd(t, power(e), tot) if tot is not None else pr.mul(t, power(e)), e + ie))
            else:
                raise ValueError(op)
        tot = mv[c.ret][0]
        return tot if tot is not None else pr.zero()

    def fold_eval(self, v: List[int], mxn: Sequence[int], inv_wk: int) -> int:
        """fold_eval (csrc/verifier.hip:146-169 = risc0-zkp verify/fri.rs): 16 evaluations on a coset -> the folded polynomial's value:
        a 16-point inverse NTT with constant twiddles, then sum_i c_bitrev(i) (inv_wk mix)^i / 16.  mxn[i] = mix^i / 16."""
        pr = self.pr
        v = list(v)
        for N in (4, 3, 2, 1):
            ln, half = 1 << N, 1 << (N - 1)
            step, cur = ROU_REV[N], 1
            for i in range(half):
                for s in range(0, 16, ln):
                    a, b = v[s + i], v[s + i + half]
                    v[s + i] = pr.add(a, b)
                    v[s + i + half] = pr.lin(a, cur, b, P - cur)
                cur = cur * step % P
        tot, mw = None, None
        for i in range(16):
            ci = v[bitrev4(i)]
            if i == 0:
                tot = pr.mul(ci, mxn[0])
            else:
                mw = inv_wk if i == 1 else pr.mul(mw, inv_wk)
                tot = pr.muladd(pr.mul(ci, mw), mxn[i], tot)
        return tot

    # ---- csrc/verifier.hip:204-347 zkh_verify_segment = risc0-zkp verify/mod.rs, statement by statement
    def verify_seal(self, c: Circuit, po2: int) -> dict:
        """-> {'out': embedded out-global wires, 'out_packed': header wires (out ‖ po2), 'code_root': 2 wires}"""
        pr, io = self.pr, self.io
        regs = c.regs                                            # (group, offset, backs, combo_id) in tap order
        out_size, mix_size = c.global_sizes
        size, domain = 1 << po2, (1 << po2) * INV_RATE
        # header
        head = self.read(out_size + 1)
        head_words = [w for h in head for w in pr.unpack(h)][:out_size + 1]
        pr.eq(head_words[out_size], pr.const(po2))
        io.commit(self.elems(head))
        tg = [None, None, None]
        tg[GROUP_CODE] = self.tree_init(domain, c.group_sizes[GROUP_CODE])
        tg[GROUP_DATA] = self.tree_init(domain, c.group_sizes[GROUP_DATA])
        mix_words = [io.elem() for _ in range(mix_size)]
        tg[GROUP_ACCUM] = self.tree_init(domain, c.group_sizes[GROUP_ACCUM])
        poly_mix = io.ext()
        tcheck = self.tree_init(domain, CHECK_SIZE)
        z = io.ext()
        back_one = ROU_REV[po2]
        n_taps = len(c.taps)
        n_u = n_taps + CHECK_SIZE
        coeff_u = self.read(4 * n_u)                             # AoS ExtElems: one wire each
        io.commit(self.elems(coeff_u))
        zb: Dict[int, int] = {}                                  # z * w^-back

        def z_back(back: int) -> int:
            if back not in zb:
                zb[back] = z if back == 0 else pr.scale(z, pow(back_one, back, P))
            return zb[back]
        eval_u: List[int] = []
        at = 0
        for (_, _, backs, _) in regs:
            for bk in backs:
                eval_u.append(self.horner(coeff_u[at:at + len(backs)], z_back(bk)))
            at += len(backs)
        result = self.poly_ext(c, poly_mix, eval_u, head_words, mix_words)
        # check(z) from its 16 coefficient planes, times Z(z) = (3 z)^size - 1
        zi = [pr.const(1), z, pr.mul(z, z)]
        zi.append(pr.mul(zi[2], z))
        basis = [pr.const(*[1 if t == k else 0 for t in range(4)]) for k in range(4)]
        remap = (0, 2, 1, 3)
        check = None
        for i in range(4):
            for k in range(4):
                zb_ik = zi[i] if k == 0 else pr.mul(zi[i], basis[k])
                cu = coeff_u[n_taps + remap[i] + 4 * k]
                check = pr.mul(cu, zb_ik) if check is None else pr.muladd(cu, zb_ik, check)
        t3 = pr.scale(z, 3)
        for _ in range(po2):
            t3 = pr.mul(t3, t3)
        zv = pr.scale(t3, 1, plus=P - 1)
        pr.eq(pr.mul(check, zv), result)
        # DEEP: combine the U coefficients per combo with powers of mix
        mix = io.ext()
        combo_begin = [0]
        for cb in c.combos:
            combo_begin.append(combo_begin[-1] + len(cb))
        tot_backs = combo_begin[-1]
        combo_u: List[Optional[int]] = [None] * (tot_backs + 1)
        mix_pows: List[int] = []
        cur = pr.const(1)
        at = 0

        def acc_into(slot: int, term_a: int, term_b: int):
            combo_u[slot] = pr.mul(term_a, term_b) if combo_u[slot] is None else pr.muladd(term_a, term_b, combo_u[slot])
        for (_, _, backs, cid) in regs:
            for i in range(len(backs)):
                acc_into(combo_begin[cid] + i, cur, coeff_u[at + i])
            mix_pows.append(cur)
            cur = pr.mul(cur, mix)
            at += len(backs)
        for i in range(CHECK_SIZE):
            acc_into(tot_backs, cur, coeff_u[at])
            at += 1
            mix_pows.append(cur)
            cur = pr.mul(cur, mix)
        combo_u = [x if x is not None else pr.zero() for x in combo_u]
        # FRI commitments
        rounds = []
        degree, dom = size, domain
        while degree > FRI_MIN_DEGREE:
            t = self.tree_init(dom // FRI_FOLD, FRI_FOLD * EXT)
            rmix = io.ext()
            mxn = [pr.const(pow(16, P - 2, P))]
            for _ in range(15):
                mxn.append(pr.mul(mxn[-1], rmix))
            rounds.append({"domain": dom, "tree": t, "mxn": mxn})
            dom //= FRI_FOLD
            degree //= FRI_FOLD
        fin = self.read(EXT * degree)                            # component planes: word j * degree + i = coefficient i, component j
        io.commit(self.elems(fin))
        q4 = degree // 4
        final_poly = [pr.pack(i % 4, fin[i // 4], fin[q4 + i // 4], fin[2 * q4 + i // 4], fin[3 * q4 + i // 4]) for i in range(degree)] \
            if degree >= 4 else None
        assert final_poly is not None, "segments below 4 rows are not supported"
        gen_final, gen0 = ROU_FWD[log2_ceil(dom)], ROU_FWD[log2_ceil(domain)]
        z4 = pr.mul(zi[2], zi[2])
        L = log2_ceil(domain)
        # queries
        for _ in range(QUERIES):
            v = io.elem()
            for _ in range(3):
                nv = io.elem()
                v = pr.mux(pr.is_zero(v), v, nv)
            bits = pr.bits31(v, L)[:L]
            x = self.pow_bits(gen0, bits)
            rows = [None, None, None]
            for g in range(3):
                rows[g] = [w for h in self.tree_open(tg[g], bits) for w in pr.unpack(h)]
            check_row = [w for h in self.tree_open(tcheck, bits) for w in pr.unpack(h)]
            tot: List[Optional[int]] = [None] * (len(c.combos) + 1)

            def acc_tot(slot: int, a: int, b: int):
                tot[slot] = pr.mul(a, b) if tot[slot] is None else pr.muladd(a, b, tot[slot])
            for r, (g, off, _, cid) in enumerate(regs):
                acc_tot(cid, mix_pows[r], rows[g][off])
            for i in range(CHECK_SIZE):
                acc_tot(len(c.combos), mix_pows[len(regs) + i], check_row[i])
            goal = None
            for i, cb in enumerate(c.combos):
                divisor = None
                for bk in cb:
                    f = pr.sub(x, z_back(bk))
                    divisor = f if divisor is None else pr.mul(divisor, f)
                num = pr.sub(tot[i] if tot[i] is not None else pr.zero(), self.horner(combo_u[combo_begin[i]:combo_begin[i + 1]], x))
                goal = pr.mul(num, pr.inv(divisor)) if goal is None else pr.muladd(num, pr.inv(divisor), goal)
            num = pr.sub(tot[len(c.combos)], combo_u[tot_backs])
            goal = pr.muladd(num, pr.inv(pr.sub(x, z4)), goal)
            pbits = bits
            for r in rounds:
                lg = log2_ceil(r["domain"] // FRI_FOLD)
                group_bits, quot_bits = pbits[:lg], pbits[lg:lg + 4]
                data = self.tree_open(r["tree"], group_bits)
                vals = [pr.pack(i % 4, data[i // 4], data[4 + i // 4], data[8 + i // 4], data[12 + i // 4]) for i in range(16)]
                cand = vals
                for b in quot_bits:
                    cand = [pr.mux(b, cand[2 * i], cand[2 * i + 1]) for i in range(len(cand) // 2)]
                pr.eq(cand[0], goal)
                inv_wk = self.pow_bits(ROU_REV[log2_ceil(r["domain"])], group_bits)
                goal = self.fold_eval(vals, r["mxn"], inv_wk)
                pbits = group_bits
            xf = self.pow_bits(gen_final, pbits)
            pr.eq(self.horner(final_poly, xf), goal)
        return {"out": head_words[:out_size], "head": head, "code_root": tg[GROUP_CODE]["top"][1], "end": self.pos}

    # ---- claims
    def repack(self, words: Sequence[int]) -> List[int]:
        """embedded word wires -> packed wires (4 words each, zero padded)"""
        pr, z = self.pr, self.pr.zero()
        out = []
        for i in range(0, len(words), 4):
            w = list(words[i:i + 4]) + [z] * (4 - len(words[i:i + 4]))
            out.append(pr.pack(0, *w, embedded=True))
        return out

    def digest_words(self, d: Sequence[int]) -> List[int]:
        return [w for h in d for w in self.pr.unpack(h)]

    def allowed_member(self, root: Sequence[int], allowed: Sequence[int]):
        """the program whose control root is `root` is a leaf of the allowed-programs tree `allowed` (witness: index bits + path)"""
        pr = self.pr
        cur = list(root)
        for _ in range(ALLOWED_DEPTH):
            b = pr.unpack(self.read(1)[0])[0]
            pr.boolean(b)
            sib = self.read(8)
            cur = self.pair_at(b, cur, sib)
        pr.eq(cur[0], allowed[0])
        pr.eq(cur[1], allowed[1])


def chain_words(c: Circuit):
    """(index of the pre-state, index of the post-state) in the `out` globals of a circuit whose segments chain (SYN-C: kind 1 with
    one public input, out = (post, 0, 0, 0, pre)), or None for circuits without a state"""
    return (4, 0) if c.kind == 1 and c.global_sizes[0] in (5, 23) else None      # SYN-C, SYN-S (syn_air.py syn_session)


def _wrap(v: Verifier, core: Sequence[int], pre: int, post: int) -> List[int]:
    """claim' = hash_pair(core, (pre, post, 0, 0, 0, 0, 0, 0)): the claim every recursion receipt publishes — its core claim (a
    segment's receipt claim, or hash_pair of the two children's claim') bound to the state range [pre, post] it covers"""
    pr, z = v.pr, v.pr.zero()
    return v.pair(core, [pr.pack(0, pre, post, z, z, embedded=True), z])


def _segment_claim(v: Verifier, c: Circuit, po2: int, control_root: Sequence[int]):
    """verify one segment seal (fresh transcript), pin its code root to `control_root` -> (its claim digest: 2 wires, pre, post):
    pre / post are the segment's state words (embedded wires of its `out` header; zero for circuits without a state)"""
    pr = v.pr
    v.io = Sponge(pr)
    s = v.verify_seal(c, po2)
    root_words = v.digest_words(s["code_root"])
    for w, k in zip(root_words, control_root):
        pr.eq(w, pr.const(int(k)))
    cw = chain_words(c)
    pre, post = (s["out"][cw[0]], s["out"][cw[1]]) if cw else (pr.zero(), pr.zero())
    return v.elems(v.repack(s["out"] + [pr.const(po2)] + root_words)), pre, post


def build_lift(circuit_desc: np.ndarray, po2: int, control_root: Sequence[int]) -> Program:
    """Inputs: the segment seal, then A (8 words).  control_root: canonical residues of the segment circuit's code root.
    out = claim' ‖ A with claim' = hash_pair(receipt claim, (pre, post, 0..)): the segment's state words ride in the claim."""
    pr = Program()
    v = Verifier(pr)
    claim, pre, post = _segment_claim(v, Circuit.parse(circuit_desc), po2, control_root)
    wrapped = _wrap(v, claim, pre, post)
    allowed = v.read(8)
    pr.public(wrapped[0], wrapped[1], allowed[0], allowed[1])
    return pr


def build_lift2(circuit_desc: np.ndarray, po2_left: int, root_left: Sequence[int], po2_right: int, root_right: Sequence[int]) -> Program:
    """lift + lift + join as ONE program: verify two SEGMENT seals, ASSERT post(left) = pre(right), out = claim' of the node two lifts
    and a join would produce ‖ A.  The bottom level of the join tree then costs one recursion proof per pair of segments instead
    of three.  Inputs: left seal, right seal, A."""
    c = Circuit.parse(circuit_desc)
    pr = Program()
    v = Verifier(pr)
    left, pre_l, post_l = _segment_claim(v, c, po2_left, root_left)
    right, pre_r, post_r = _segment_claim(v, c, po2_right, root_right)
    pr.eq(post_l, pre_r)                                          # continuity: the right segment starts where the left one ended
    parent = v.pair(_wrap(v, left, pre_l, post_l), _wrap(v, right, pre_r, post_r))
    wrapped = _wrap(v, parent, pre_l, post_r)
    allowed = v.read(8)
    pr.public(wrapped[0], wrapped[1], allowed[0], allowed[1])
    return pr


def build_join(recursion_desc: np.ndarray, *po2s: int) -> Program:
    """join of two — or THREE — recursion seals.  Inputs per child: its seal, its membership path (ALLOWED_DEPTH x (bit, 8 sibling
    words)), then the OPENING of its claim': core (8 words) and (pre, post).  Each child's out is claim' ‖ A; every A must be this
    program's A (= its own public output); hash_pair(core, (pre, post, 0..)) must BE the child's claim'; and post(k) = pre(k + 1)
    for neighbours — upstream's join asserts the same continuity between the `ReceiptClaim`s it merges.
    Two children: out = wrap(hash_pair(claim'_l, claim'_r), pre_l, post_r) ‖ A.
    Three children (a, b, c): out = what join(join(a, b), c) would publish — the inner node's claim' is computed in-circuit (two
    permutations) instead of proven on its own: one proof where the binary tree needs two."""
    assert len(po2s) in (2, 3)
    c = Circuit.parse(recursion_desc)
    pr = Program()
    v = Verifier(pr)
    claims, states, allowed = [], [], None
    for po2 in po2s:
        v.io = Sponge(pr)
        s = v.verify_seal(c, po2)
        out_packed = s["head"][:4]                               # out = 16 words: claim' (2 wires) ‖ A (2 wires)
        if allowed is None:
            allowed = out_packed[2:4]
        else:
            pr.eq(out_packed[2], allowed[0])
            pr.eq(out_packed[3], allowed[1])
        v.allowed_member(s["code_root"], allowed)
        core = v.read(8)
        st = pr.unpack(v.read(2)[0])                             # (pre, post, 0, 0): the padding is constrained by read()
        opened = _wrap(v, core, st[0], st[1])
        pr.eq(opened[0], out_packed[0])
        pr.eq(opened[1], out_packed[1])
        claims.append(out_packed[:2])
        states.append((st[0], st[1]))
    node, pre, post = claims[0], states[0][0], states[0][1]
    for k in range(1, len(po2s)):
        pr.eq(post, states[k][0])                                 # continuity: child k starts where the node so far ended
        node = _wrap(v, v.pair(node, claims[k]), pre, states[k][1])
        post = states[k][1]
    pr.public(node[0], node[1], allowed[0], allowed[1])
    return pr


def _child(v: Verifier, c: Circuit, po2: int, allowed):
    """verify one child RECURSION seal (fresh transcript) and its program's membership in the allowed set -> (claim' wires, A wires)"""
    pr = v.pr
    v.io = Sponge(pr)
    s = v.verify_seal(c, po2)
    out_packed = s["head"][:4]
    if allowed is None:
        allowed = out_packed[2:4]
    else:
        pr.eq(out_packed[2], allowed[0])
        pr.eq(out_packed[3], allowed[1])
    v.allowed_member(s["code_root"], allowed)
    return out_packed[:2], allowed


def build_union(recursion_desc: np.ndarray, po2_left: int, po2_right: int) -> Program:
    """`union` (risc0-zkvm 3.0.3 ProverServer::union -> risc0-circuit-recursion's union program: two receipts of ANY claims merged
    into one whose claim is the digest of the SORTED pair, /root/reference/Cargo.lock:5418, :5305): how a session's assumption
    receipts (keccak batches) become one receipt before `resolve`.  Inputs per child: its seal, its membership path; then ONE word,
    the swap bit.  No state is opened and nothing chains: out = wrap(hash_pair(lo, hi), 0, 0) ‖ A with (lo, hi) = (claim'_l, claim'_r),
    or the two swapped when the bit is set.  The ORDER is the host's rule (zeth_amd/recursion.py union_node: canonical words,
    lexicographic): a verifier recomputes the sorted tree from the leaves, so a prover that swaps the wrong way proves a claim nobody
    asks for."""
    c = Circuit.parse(recursion_desc)
    pr = Program()
    v = Verifier(pr)
    left, allowed = _child(v, c, po2_left, None)
    right, allowed = _child(v, c, po2_right, allowed)
    b = pr.unpack(v.read(1)[0])[0]
    pr.boolean(b)
    node = _wrap(v, v.pair_at(b, left, right), pr.zero(), pr.zero())
    pr.public(node[0], node[1], allowed[0], allowed[1])
    return pr


def build_resolve(recursion_desc: np.ndarray, po2_cond: int, po2_assum: int) -> Program:
    """`resolve` (ProverServer::resolve, same crates): the CONDITIONAL receipt (a session's join-tree root) bound to the receipt of
    what it assumed (the union-tree root of its assumption receipts).  Inputs: the conditional seal, its membership path, the
    OPENING of its claim' (core, (pre, post)); the assumption seal, its membership path.  out = wrap(hash_pair(claim'_cond,
    claim'_assum), pre, post) ‖ A: the session's state range, unchanged, over a core that commits to both — the verifier recomputes
    it from the segment leaves AND the assumption leaves, so a session resolved against another assumption set has another claim."""
    c = Circuit.parse(recursion_desc)
    pr = Program()
    v = Verifier(pr)
    cond, allowed = _child(v, c, po2_cond, None)
    core = v.read(8)
    st = pr.unpack(v.read(2)[0])
    opened = _wrap(v, core, st[0], st[1])
    pr.eq(opened[0], cond[0])
    pr.eq(opened[1], cond[1])
    assum, allowed = _child(v, c, po2_assum, allowed)
    node = _wrap(v, v.pair(cond, assum), st[0], st[1])
    pr.public(node[0], node[1], allowed[0], allowed[1])
    return pr


if __name__ == "__main__":      # python -m zeth_amd.circuits.rec_verify out_dir [segment.desc po2:root_hex8 ...]
    # writes the program set of a block (zeth_amd/recursion.py build_programs) as u32 blobs for non-Python hosts; without a
    # segment description: SYN-A at po2 20 and 18 with the control roots shipped in circuits/control_roots.json
    import os
    import sys
    from . import syn_air
    from .. import recursion as host_rec
    from ..prover import shipped_control_root
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    desc = syn_air.syn_a()
    roots = {po2: shipped_control_root(desc, po2) for po2 in (20, 18)}
    if any(r is None for r in roots.values()):
        raise SystemExit("no shipped control root for SYN-A at po2 20 / 18 (python -m zeth_amd.prover on a GPU box)")
    import hashlib
    import json
    manifest = {}
    for kind, blob in host_rec.build_programs(desc, roots):
        name = "-".join(str(x) for x in kind[:2 if kind[0] == "lift" else 4 if kind[0] == "join3" else 3]) + ".zkr1"
        path = os.path.join(out_dir, name)
        np.asarray(blob, dtype="<u4").tofile(path)
        manifest[name] = {"words": int(blob.size), "po2": int(blob[2]), "sha256": hashlib.sha256(np.asarray(blob, dtype="<u4").tobytes()).hexdigest()}
        print(f"{path}: {blob.size} words, po2 {int(blob[2])}")
    # the manifest a non-Python host checks its program files against (committed: examples/recursion_programs.manifest.json;
    # tests/test_recursion.py rebuilds the set and compares) — upstream ships its lift / join programs as hashed .zkr files too
    from . import recursion as rcirc
    rdesc = np.asarray(rcirc.recursion_circuit(), dtype="<u4")
    manifest["recursion.desc"] = {"words": int(rdesc.size), "sha256": hashlib.sha256(rdesc.tobytes()).hexdigest()}
    with open(os.path.join(out_dir, "manifest.json"), "w") as fh:
        json.dump({"generator": "python -m zeth_amd.circuits.rec_verify <dir> (SYN-A segments at po2 20 / 18, control roots from circuits/control_roots.json)",
                   "segment_control_roots": {str(p): [int(w) for w in r] for p, r in roots.items()}, "files": manifest}, fh, indent=1)
