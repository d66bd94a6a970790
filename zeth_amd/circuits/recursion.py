"""RECURSION: the circuit that verifies STARK seals in-circuit — lift and join.

Upstream's recursion circuit (risc0-circuit-recursion 4.0.2, un-vendored: /root/reference/Cargo.lock:5305; BASELINE.json
config 5 "lift/join recursion to a single succinct receipt") is a small machine whose *program* lives in the code (control)
group: one micro-op per row over Fp4 registers, a Poseidon2 accelerator, and a memory argument in the accum group; `lift`
and `join` are programs that run the STARK verifier on one / two child seals.  That Zirgen-generated circuit and its
`.zkr` programs cannot be obtained offline.  This module states a circuit of the same kind from public material only:

  * a row holds six Fp4 WIRES (a b c d e f = 24 data columns) and one GATE whose coefficients are code columns:
        GEN    qM a*b + qA a + qB b + qC c + qD d + qK = 0        (Fp4 arithmetic; every add / mul / inverse / assertion)
        MUX    d = b + a_0 (c - b)                                  (bit select)
        BOOL   a_0 (a_0 - 1) = 0, a_1 = a_2 = a_3 = 0
        EMB    a, b, c, e are base-field elements (components 1..3 zero)
        PACKj  d = (a_j, b_j, c_j, e_j)                            (4 x 4 transposes: hash words <-> Fp4 values)
        PUB    (a b c d) = the 16 output globals
  * every 12 rows are one Poseidon2 permutation: S[24], Q[24].  Row 0 holds the input state (bound to the row's six wires),
    rows 1..4 and 7..10 one FULL round each (Q_j = (S_j + rc_j)^3, as in P2-JOIN), rows 5 and 6 TWELVE and NINE partial rounds:
    a partial round touches one s-box, so its row stores (Q_i, X_i) = ((u_i)^3, (u_i)^7) per round in the Q columns and every
    later state cell is a LINEAR form of (S, X_1 .. X_i) with constant coefficients (the powers of the internal matrix);
    row 11 holds the output (bound to the wires).  All constraints have degree <= 3 (+ selector).  The gates of the 10 rows
    between input and output are free for arithmetic, so hashing and arithmetic run side by side;
  * a block may take its first two digests CONDITIONALLY SWAPPED (`pios` instead of `pio` on its input row): with the bit
    t = e_0 of the row's fifth wire, S[0..8) = (a b) + t ((c d) - (a b)), S[8..16) = (c d) + t ((a b) - (c d)), S[16] = 0 —
    one Merkle level (hash_pair(cur, sibling) or hash_pair(sibling, cur), chosen by the index bit, which sits where hash_pair's
    capacity is zero) costs one block and no gate; as four MUX gates per level it was 19 % of a verifier program's rows;
  * equal wires are tied by a PLONK-style COPY argument in the accum group: position (row, wire) has the id 6 row + wire,
    the code columns sigma_w hold the id of the next position of the same variable, and three running products
        Z_k(row) = Z_k(row - 1) * prod_{w in {2k, 2k+1}} F(id_w, W_w) / F(sigma_w, W_w),
        F(t, w) = gamma + t + beta_1 w_0 + beta_2 w_1 + beta_3 w_2 + beta_4 w_3                (beta_i, gamma: mix globals)
    must multiply to 1 on the last active row.
The program (gate coefficients, sigma) is the code group, hence the control root identifies the program, as upstream.
rec_verify.py compiles this repository's verifier (csrc/verifier.hip = risc0-zkp src/verify/mod.rs) into such programs.

Columns:  data 72 = W[6][4] | S[24] | Q[24];  accum 12 = Z[3][4];  code 58:
   0 active 1 first 2 body 3 last   4 rowid   5..10 sigma[6]   11 qM 12 qA 13 qB 14 qC 15 qD 16 qK
   17 qMux 18 qBool 19 qEmb 20..23 qP[4]   24 pio 25 pub   26 lin 27 fullr 28 partA 29 partB 30 lf 31 lpA 32 lpB   33..56 rc[24]
   57 pios
Globals: out = 16 words (program-defined: rec_verify puts claim (8) ‖ allowed-programs root (8));  mix = 20 words.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import p2_join
from .desc import GLOBAL_MIX, GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, P, CircuitBuilder

KIND_RECURSION = 4
T, BLOCK, NW = 24, 12, 6
WD, WA, WC, OUT_WORDS, MIX_WORDS = 72, 12, 58, 16, 20
PART_A, PART_B = 12, 9                 # partial rounds held by rows 5 and 6 of a block (rounds 4..15, 16..24)
C_ACTIVE, C_FIRST, C_BODY, C_LAST, C_ROWID, C_SIGMA = 0, 1, 2, 3, 4, 5
C_QM, C_QA, C_QB, C_QC, C_QD, C_QK = 11, 12, 13, 14, 15, 16
C_MUX, C_BOOL, C_EMB, C_PACK, C_PIO, C_PUB = 17, 18, 19, 20, 24, 25
C_LIN, C_FULLR, C_PARTA, C_PARTB, C_LF, C_LPA, C_LPB, C_RC = 26, 27, 28, 29, 30, 31, 32, 33
C_PIOS = 57
D_S, D_Q = 24, 48
NBETA = P - 11
M4, DIAG, RC = p2_join.M4, p2_join.DIAG, p2_join.RC
ZK_CYCLES = 1994


def partial_forms(m: int):
    """m consecutive partial rounds as linear algebra: with X_i = (s0 + rc_i)^7 of round i taken as a variable, every state
    cell is a linear form over (S_0..S_23, X_1..X_m) with constant coefficients.  -> {'s0': [form of cell 0 BEFORE round i
    (without rc)], 'out': [forms of the 24 cells after round m]}, a form = list of 24 + m canonical coefficients."""
    nv = T + m
    f = [[1 if v == j else 0 for v in range(nv)] for j in range(T)]
    s0 = []
    for i in range(m):
        s0.append(list(f[0]))
        x = [1 if v == T + i else 0 for v in range(nv)]
        tot = [(x[v] + sum(f[j][v] for j in range(1, T))) % P for v in range(nv)]
        f = [[(tot[v] + DIAG[0] * x[v]) % P for v in range(nv)]] + [[(tot[v] + DIAG[j] * f[j][v]) % P for v in range(nv)] for j in range(1, T)]
    return {"s0": s0, "out": f}


def block_rows(inp):
    """One permutation as the 12 trace rows of a block, over canonical residues: [(S[24], Q[24])].  The plain-Python statement
    of what oracle/recursion.c and csrc/recursion.hip write; rows[11][0] is the output."""
    rows = [([x % P for x in inp], [0] * T)]
    s = p2_join.m_ext(rows[0][0])
    rnd = 0

    def full(s, rnd):
        rc = RC[rnd * T:(rnd + 1) * T]
        q = [pow((s[i] + rc[i]) % P, 3, P) for i in range(T)]
        return q, p2_join.m_ext([q[i] * q[i] % P * ((s[i] + rc[i]) % P) % P for i in range(T)])
    for _ in range(4):
        q, nxt = full(s, rnd)
        rows.append((s, q))
        s, rnd = nxt, rnd + 1
    for m in (PART_A, PART_B):
        q = [0] * T
        s_in = s
        for i in range(m):
            u = (s[0] + RC[rnd * T]) % P
            q[2 * i] = pow(u, 3, P)
            x7 = q[2 * i] * q[2 * i] % P * u % P
            q[2 * i + 1] = x7
            tot = (x7 + sum(s[1:])) % P
            s = [(tot + DIAG[0] * x7) % P] + [(tot + DIAG[j] * s[j]) % P for j in range(1, T)]
            rnd += 1
        rows.append((s_in, q))
    for _ in range(4):
        q, nxt = full(s, rnd)
        rows.append((s, q))
        s, rnd = nxt, rnd + 1
    rows.append((s, [0] * T))
    assert rnd == 29 and len(rows) == BLOCK
    return rows


# ---------------------------------------------------------------------------------------------------------------------
# the constraint system
# ---------------------------------------------------------------------------------------------------------------------
def build_recursion() -> np.ndarray:
    b = CircuitBuilder((WA, WC, WD), (OUT_WORDS, MIX_WORDS), kind=KIND_RECURSION)
    code = lambda c, back=0: b.get(GROUP_CODE, c, back)
    W = lambda w, c, back=0: b.get(GROUP_DATA, 4 * w + c, back)
    S = lambda j, back=0: b.get(GROUP_DATA, D_S + j, back)
    Q = lambda j, back=0: b.get(GROUP_DATA, D_Q + j, back)
    Z = lambda k, c, back=0: b.get(GROUP_ACCUM, 4 * k + c, back)
    out = lambda i: b.get_global(GLOBAL_OUT, i)
    mixg = lambda i: b.get_global(GLOBAL_MIX, i)
    one, nbeta = b.const(1), b.const(NBETA)
    cst = {v: b.const(v) for v in (2, 3, 4, 5, 6, 7)}
    active, first, body, last = code(C_ACTIVE), code(C_FIRST), code(C_BODY), code(C_LAST)

    def ext_mul(x, y):
        mm = lambda i, j: b.mul(x[i], y[j])
        return [b.add(mm(0, 0), b.mul(nbeta, b.add(b.add(mm(1, 3), mm(2, 2)), mm(3, 1)))),
                b.add(b.add(mm(0, 1), mm(1, 0)), b.mul(nbeta, b.add(mm(2, 3), mm(3, 2)))),
                b.add(b.add(b.add(mm(0, 2), mm(1, 1)), mm(2, 0)), b.mul(nbeta, mm(3, 3))),
                b.add(b.add(mm(0, 3), mm(1, 2)), b.add(mm(2, 1), mm(3, 0)))]

    def gated(chain, sel, constraints):
        inner = b.true()
        for c in constraints:
            inner = b.and_eqz(inner, c)
        return b.and_cond(chain, sel, inner)

    wa, wb, wc_, wd_, we = ([W(w, c) for c in range(4)] for w in range(5))
    chain = b.true()
    # GEN: the q's are zero wherever no gate sits (and on the blinding rows), so the constraint needs no other selector
    ab = ext_mul(wa, wb)
    qm, qa, qb, qc, qd, qk = (code(C_QM + i) for i in range(6))
    for c in range(4):
        e = b.add(b.add(b.mul(qm, ab[c]), b.mul(qa, wa[c])), b.add(b.mul(qb, wb[c]), b.add(b.mul(qc, wc_[c]), b.mul(qd, wd_[c]))))
        chain = b.and_eqz(chain, b.add(e, qk) if c == 0 else e)
    chain = gated(chain, code(C_MUX), [b.sub(b.add(wb[c], b.mul(wa[0], b.sub(wc_[c], wb[c]))), wd_[c]) for c in range(4)])
    chain = gated(chain, code(C_BOOL), [b.mul(wa[0], b.sub(wa[0], one)), wa[1], wa[2], wa[3]])
    chain = gated(chain, code(C_EMB), [x[c] for x in (wa, wb, wc_, we) for c in (1, 2, 3)])
    for j in range(4):
        chain = gated(chain, code(C_PACK + j), [b.sub(wd_[0], wa[j]), b.sub(wd_[1], wb[j]), b.sub(wd_[2], wc_[j]), b.sub(wd_[3], we[j])])
    chain = gated(chain, code(C_PUB), [b.sub(W(i // 4, i % 4), out(i)) for i in range(OUT_WORDS)])
    # Poseidon2 blocks (p2_join.py): wires <-> state on the input / output rows, the rounds in between
    chain = gated(chain, code(C_PIO), [b.sub(W(j // 4, j % 4), S(j)) for j in range(T)])
    # ... or, on an input row, with the first two digests swapped when the bit t = e_0 is set (t sits in the capacity cell that
    # hash_pair leaves at zero, so S[16] = 0 either way); t is a bit here too, whatever gate produced the wire
    flat = lambda j: W(j // 4, j % 4)
    t_bit = we[0]
    swap = [b.sub(S(j), b.add(flat(j), b.mul(t_bit, b.sub(flat(j + 8), flat(j))))) for j in range(8)]
    swap += [b.sub(S(j + 8), b.add(flat(j + 8), b.mul(t_bit, b.sub(flat(j), flat(j + 8))))) for j in range(8)]
    swap += [S(16)] + [b.sub(flat(j), S(j)) for j in range(17, T)]
    swap += [b.mul(t_bit, b.sub(t_bit, one))]
    chain = gated(chain, code(C_PIOS), swap)

    def times(c, x):
        return x if c == 1 else b.mul(cst[c], x)

    def lin_m_ext(x):
        y = []
        for blk in range(0, T, 4):
            for i in range(4):
                e = times(M4[i][0], x[blk])
                for j in range(1, 4):
                    e = b.add(e, times(M4[i][j], x[blk + j]))
                y.append(e)
        tot = []
        for i in range(4):
            e = y[i]
            for blk in range(4, T, 4):
                e = b.add(e, y[blk + i])
            tot.append(e)
        return [b.add(y[k], tot[k % 4]) for k in range(T)]

    u = [b.add(S(j), code(C_RC + j)) for j in range(T)]
    cube = lambda x: b.mul(b.mul(x, x), x)
    chain = gated(chain, code(C_FULLR), [b.sub(Q(j), cube(u[j])) for j in range(T)])
    prev = [S(j, 1) for j in range(T)]
    chain = gated(chain, code(C_LIN), [b.sub(S(k), e) for k, e in enumerate(lin_m_ext(prev))])
    x7 = [b.mul(b.mul(Q(j, 1), Q(j, 1)), b.add(S(j, 1), code(C_RC + j, 1))) for j in range(T)]
    chain = gated(chain, code(C_LF), [b.sub(S(k), e) for k, e in enumerate(lin_m_ext(x7))])
    # partial rows: m rounds, (Q_i, X_i) in Q columns 2i, 2i + 1; states as linear forms over (S[24], X_1 .. X_m)
    for m, sel_here, sel_next in ((PART_A, C_PARTA, C_LPA), (PART_B, C_PARTB, C_LPB)):
        forms = partial_forms(m)
        for back, sel in ((0, sel_here), (1, sel_next)):
            var = [S(j, back) for j in range(T)] + [Q(2 * i + 1, back) for i in range(m)]

            def lin_form(coeffs):
                e = None
                for v, cf in zip(var, coeffs):
                    if cf:
                        t = v if cf == 1 else b.mul(b.const(cf), v)
                        e = t if e is None else b.add(e, t)
                return e if e is not None else b.const(0)
            if back == 0:
                cons = []
                for i in range(m):
                    ui = b.add(lin_form(forms["s0"][i]), code(C_RC + i))
                    qi, xi = Q(2 * i), Q(2 * i + 1)
                    cons += [b.sub(qi, cube(ui)), b.sub(xi, b.mul(b.mul(qi, qi), ui))]
                chain = gated(chain, code(sel), cons)
            else:
                chain = gated(chain, code(sel), [b.sub(S(j), lin_form(forms["out"][j])) for j in range(T)])
    # the copy argument
    beta = [[mixg(4 * i + c) for c in range(4)] for i in range(4)]
    gamma = [mixg(16 + c) for c in range(4)]
    six, rowid = cst[6], code(C_ROWID)

    def F(tag, w):               # gamma + tag + sum_i beta_i * W_w[i]
        f = []
        for c in range(4):
            e = gamma[c]
            for i in range(4):
                e = b.add(e, b.mul(beta[i][c], W(w, i)))
            f.append(e)
        f[0] = b.add(f[0], tag)
        return f
    ids = [b.add(b.mul(six, rowid), b.const(w)) if w else b.mul(six, rowid) for w in range(NW)]
    fin = None
    for k in range(WA // 4):
        num = ext_mul(F(ids[2 * k], 2 * k), F(ids[2 * k + 1], 2 * k + 1))
        den = ext_mul(F(code(C_SIGMA + 2 * k), 2 * k), F(code(C_SIGMA + 2 * k + 1), 2 * k + 1))
        z = [Z(k, c) for c in range(4)]
        zp = [Z(k, c, 1) for c in range(4)]
        lhs = ext_mul(z, den)
        chain = gated(chain, first, [b.sub(lhs[c], num[c]) for c in range(4)])
        rhs = ext_mul(zp, num)
        chain = gated(chain, body, [b.sub(lhs[c], rhs[c]) for c in range(4)])
        fin = z if fin is None else ext_mul(fin, z)
    chain = gated(chain, last, [b.sub(fin[0], one), fin[1], fin[2], fin[3]])
    # selector sanity
    chain = b.and_eqz(chain, b.mul(active, b.sub(one, active)))
    chain = b.and_eqz(chain, b.mul(first, b.sub(one, first)))
    chain = b.and_eqz(chain, b.sub(b.sub(active, first), body))
    return b.finish(chain)


_cached = None


def recursion_circuit() -> np.ndarray:
    global _cached
    if _cached is None:
        _cached = build_recursion()
    return _cached.copy()


# ---------------------------------------------------------------------------------------------------------------------
# programs
# ---------------------------------------------------------------------------------------------------------------------
# witness ops (executed in order by oracle/recursion.c and csrc/recursion.hip): [op | aux << 8, out, i0 .. i5]
OP_INPUT, OP_GEN, OP_MUX, OP_PACK, OP_UNPACK, OP_INV, OP_BITS, OP_P2, OP_EQ, OP_ISZ = range(1, 11)
OP_WORDS = 8
G_MUX, G_BOOL, G_EMB, G_PACK0, G_PUB, G_SWAP = 1, 2, 4, 8, 128, 256   # gate flag bits (G_PACK0 << j); G_SWAP marks a block's input row
PROG_MAGIC = 0x5a4b5231                                             # 'ZKR1'
PROG_HEADER = 16
PROG_VERSION = 2                                                    # 2: conditional-swap blocks (OP_P2 aux bit 0, G_SWAP rows, 58 code columns)


@dataclass
class Gate:
    pos: List[int]                      # variable at positions a b c d e f (-1: unused)
    q: Tuple[int, int, int, int, int, int] = (0, 0, 0, 0, 0, 0)
    flags: int = 0


class Program:
    """A recursion program under construction: variables (Fp4 wires), gates (one row each), Poseidon2 blocks and the witness
    schedule.  `finish(po2)` places it into rows and returns the blob oracle/recursion.c and csrc/recursion.hip consume."""

    def __init__(self):
        self.n_vars = 0
        self.ops: List[Tuple[int, ...]] = []
        self.gates: List[Gate] = []
        self.p2s: List[Tuple[List[int], int, bool]] = []    # (6 input vars, first of 6 output vars, conditional swap)
        self.consts: List[int] = []                         # pool of canonical residues (GEN coefficients)
        self._const_at: Dict[Tuple[int, ...], int] = {}
        self._const_var: Dict[Tuple[int, ...], int] = {}
        self.parent: List[int] = []                         # union-find over variables (copy classes)
        self.n_inputs = 0
        self.pub: Optional[List[int]] = None

    # -- variables
    def var(self, count: int = 1) -> int:
        v = self.n_vars
        self.n_vars += count
        self.parent.extend(range(v, v + count))
        return v

    def find(self, v: int) -> int:
        p = self.parent
        while p[v] != v:
            p[v] = p[p[v]]
            v = p[v]
        return v

    def eq(self, x: int, y: int) -> None:
        """x and y are the same wire (checked by the copy argument; the witness generator refuses a witness where they differ)"""
        self.ops.append((OP_EQ, 0, x, y, 0, 0, 0, 0))
        rx, ry = self.find(x), self.find(y)
        if rx != ry:
            self.parent[max(rx, ry)] = min(rx, ry)

    def _op(self, op, out, *ins, aux=0):
        ins = list(ins) + [0] * (6 - len(ins))
        self.ops.append((op | (aux << 8), out, *ins))

    def _kidx(self, q) -> int:
        q = tuple(int(x) % P for x in q)
        at = self._const_at.get(q)
        if at is None:
            at = len(self.consts)
            self.consts.extend(q)
            self._const_at[q] = at
        return at

    # -- witness inputs: `count` (1..4) consecutive words of the input vector as one wire, zero padded
    def input(self, off: int, count: int = 4) -> int:
        v = self.var()
        self._op(OP_INPUT, v, off, aux=count)
        self.n_inputs = max(self.n_inputs, off + count)
        return v

    # -- gates
    def gen(self, a: int, b: int, c: int, qM=0, qA=0, qB=0, qC=0, qK=0) -> int:
        """d = qM a*b + qA a + qB b + qC c + qK"""
        d = self.var()
        self.gates.append(Gate([a, b, c, d, -1, -1], (qM % P, qA % P, qB % P, qC % P, P - 1, qK % P)))
        self._op(OP_GEN, d, a, b, c, self._kidx((qM, qA, qB, qC, qK)))
        return d

    def require(self, a: int, b: int, c: int, qM=0, qA=0, qB=0, qC=0, qK=0) -> None:
        """qM a*b + qA a + qB b + qC c + qK = 0"""
        self.gates.append(Gate([a, b, c, -1, -1, -1], (qM % P, qA % P, qB % P, qC % P, 0, qK % P)))
        z = self.var()                                      # the witness generator evaluates the left side into a scratch wire ...
        self._op(OP_GEN, z, a, b, c, self._kidx((qM, qA, qB, qC, qK)))
        self.ops.append((OP_EQ, 0, z, self.zero(), 0, 0, 0, 0))     # ... and refuses unless it is zero (no copy class: z sits nowhere)

    def const(self, k0: int, k1: int = 0, k2: int = 0, k3: int = 0) -> int:
        key = (k0 % P, k1 % P, k2 % P, k3 % P)
        v = self._const_var.get(key)
        if v is None:
            if key[1:] == (0, 0, 0):
                v = self.var()
                self.gates.append(Gate([-1, -1, -1, v, -1, -1], (0, 0, 0, 0, P - 1, key[0])))
                self._op(OP_GEN, v, v, v, v, self._kidx((0, 0, 0, 0, key[0])))
            else:
                v = self.pack(0, *(self.const(k) for k in key))
            self._const_var[key] = v
        return v

    def zero(self) -> int:
        return self.const(0)

    def add(self, a, b):
        return self.gen(a, b, a, qA=1, qB=1)

    def sub(self, a, b):
        return self.gen(a, b, a, qA=1, qB=P - 1)

    def mul(self, a, b):
        return self.gen(a, b, a, qM=1)

    def muladd(self, a, b, c, k=1):
        """c + k a*b"""
        return self.gen(a, b, c, qM=k, qC=1)

    def scale(self, a, k, plus=0):
        return self.gen(a, a, a, qA=k, qK=plus)

    def lin(self, a, ka, b, kb, k=0):
        """ka a + kb b + k"""
        return self.gen(a, b, a, qA=ka, qB=kb, qK=k)

    def mux(self, bit: int, b: int, c: int) -> int:
        """bit ? c : b   (bit must be a BOOL wire)"""
        d = self.var()
        self.gates.append(Gate([bit, b, c, d, -1, -1], flags=G_MUX))
        self._op(OP_MUX, d, bit, b, c)
        return d

    def boolean(self, a: int) -> None:
        """a is a BOOL wire: a_0 in {0, 1}, a_1 = a_2 = a_3 = 0 (the witness generator checks a*a = a, which says the same)"""
        self.gates.append(Gate([a, -1, -1, -1, -1, -1], flags=G_BOOL))
        z = self.var()
        self._op(OP_GEN, z, a, a, a, self._kidx((1, P - 1, 0, 0, 0)))
        self.ops.append((OP_EQ, 0, z, self.zero(), 0, 0, 0, 0))

    def pack(self, j: int, a: int, b: int, c: int, e: int, embedded: bool = False) -> int:
        """d = (a_j, b_j, c_j, e_j)"""
        d = self.var()
        self.gates.append(Gate([a, b, c, d, e, -1], flags=(G_PACK0 << j) | (G_EMB if embedded else 0)))
        self._op(OP_PACK, d, a, b, c, e, aux=j)
        return d

    def unpack(self, d: int) -> Tuple[int, int, int, int]:
        """four base-field wires (d_0,0,0,0) .. (d_3,0,0,0)"""
        v = self.var(4)
        self.gates.append(Gate([v, v + 1, v + 2, d, v + 3, -1], flags=G_PACK0 | G_EMB))
        self._op(OP_UNPACK, v, d)
        return v, v + 1, v + 2, v + 3

    def inv(self, a: int) -> int:
        b = self.var()
        self._op(OP_INV, b, a)
        self.require(a, b, a, qM=1, qK=P - 1)
        return b

    def is_zero(self, a: int) -> int:
        """1 if the base-field wire a is zero else 0 (a must be embedded)"""
        h = self.var()
        self._op(OP_ISZ, h, a)                          # h = 1 / a_0, or 0
        z = self.gen(a, h, a, qM=P - 1, qK=1)           # z = 1 - a h
        self.require(a, z, a, qM=1)                     # a z = 0
        return z

    def bits31(self, x: int, want: int = 31) -> List[int]:
        """canonical bits (least significant first) of the base-field wire x; returns the first `want`, and with
        want < 31 also the wire holding sum_{i < want} 2^i bit_i as the last element."""
        base = self.var(31)
        self._op(OP_BITS, base, x)
        acc, partial = None, None
        for i in range(31):
            bit = base + i
            prev = acc if acc is not None else self.zero()
            acc = self.var()
            self.gates.append(Gate([bit, -1, prev, acc, -1, -1], (0, pow(2, i, P), 0, 1, P - 1, 0), flags=G_BOOL))
            self._op(OP_GEN, acc, bit, bit, prev, self._kidx((0, pow(2, i, P), 0, 1, 0)))
            if i + 1 == 27:
                low27 = acc
            if i + 1 == want:
                partial = acc
        self.eq(acc, x)
        t = self.mul(self.mul(base + 27, base + 28), self.mul(base + 29, base + 30))
        self.require(t, low27, t, qM=1)                 # bits 27..30 all set => the low 27 bits are zero (value < P)
        out = [base + i for i in range(want)]
        return out + ([partial] if want < 31 else [])

    def p2(self, ins: Sequence[int], swap: bool = False) -> List[int]:
        """one permutation of the six wires; swap: wire 4 is a bit t (a BOOL wire) and the state is (t ? ins[2:4] ‖ ins[0:2] :
        ins[0:4]) ‖ (0, ins[4]_1..3) ‖ ins[5] — hash_pair with its two digests in the order an index bit says"""
        assert len(ins) == NW
        out = self.var(NW)
        self.p2s.append((list(ins), out, bool(swap)))
        self._op(OP_P2, out, *ins, aux=1 if swap else 0)
        return list(range(out, out + NW))

    def public(self, a: int, b: int, c: int, d: int) -> None:
        assert self.pub is None
        self.pub = [a, b, c, d]
        self.gates.append(Gate([a, b, c, d, -1, -1], flags=G_PUB))

    # -- placement
    def rows_needed(self) -> Tuple[int, int]:
        return len(self.p2s), len(self.gates)

    def min_po2(self, zk: int = ZK_CYCLES) -> int:
        for po2 in range(1, 25):
            A = (1 << po2) - zk
            if A <= 0:
                continue
            K = A // BLOCK
            if K >= len(self.p2s) and A - 2 * K >= len(self.gates):
                return po2
        raise ValueError("program too large")

    def finish(self, po2: int, zk: int = ZK_CYCLES) -> np.ndarray:
        """-> program blob (u32): header | gate table [rows][13] | position table [rows][6] | consts | ops"""
        n = 1 << po2
        A = n - zk
        K = A // BLOCK
        assert A > 1 and K >= len(self.p2s), f"{len(self.p2s)} permutations need more than {K} blocks (po2 {po2})"
        free = A - 2 * K
        assert free >= len(self.gates), f"{len(self.gates)} gates need more than {free} rows (po2 {po2})"
        pos = np.full((A, NW), -1, dtype=np.int64)
        gate = np.zeros((A, 7), dtype=np.int64)                     # qM qA qB qC qD qK flags
        rows = np.arange(A)
        k = rows % BLOCK
        is_free = (rows >= BLOCK * K) | ((k != 0) & (k != BLOCK - 1))
        free_rows = rows[is_free][:len(self.gates)]
        if self.gates:
            gp = np.array([g.pos for g in self.gates], dtype=np.int64)
            gq = np.array([list(g.q) + [g.flags] for g in self.gates], dtype=np.int64)
            pos[free_rows] = gp
            gate[free_rows] = gq
        for p, (ins, out, swap) in enumerate(self.p2s):
            pos[BLOCK * p] = ins
            pos[BLOCK * p + BLOCK - 1] = np.arange(out, out + NW)
            if swap:
                gate[BLOCK * p, 6] = G_SWAP
        # copy classes -> sigma
        roots = np.array(self.parent, dtype=np.int64)                  # union-find resolved by pointer jumping (parents point downwards)
        while True:
            nxt = roots[roots]
            if np.array_equal(nxt, roots):
                break
            roots = nxt
        flat = pos.reshape(-1)
        used = np.nonzero(flat >= 0)[0]
        cls = roots[flat[used]]
        order = np.argsort(cls, kind="stable")
        su, sc = used[order], cls[order]
        nxt = np.roll(su, -1)
        starts = np.nonzero(np.r_[True, sc[1:] != sc[:-1]])[0]
        ends = np.r_[starts[1:], len(su)] - 1
        nxt[ends] = su[starts]
        sigma = np.arange(A * NW, dtype=np.int64)
        sigma[su] = nxt
        posr = flat.copy()
        posr[used] = cls                                               # the trace is filled from the class representative
        ops = np.array(self.ops, dtype=np.int64).reshape(-1, OP_WORDS) if self.ops else np.zeros((0, OP_WORDS), dtype=np.int64)
        head = np.zeros(PROG_HEADER, dtype=np.int64)
        head[:9] = [PROG_MAGIC, PROG_VERSION, po2, zk, A, self.n_vars, len(self.consts), len(ops), self.n_inputs]
        head[9] = len(self.p2s)
        head[10] = len(self.gates)
        table = np.concatenate([gate, sigma.reshape(A, NW)], axis=1)           # 13 words per row
        blob = np.concatenate([head, table.reshape(-1), (posr + 1).reshape(-1), np.array(self.consts, dtype=np.int64), ops.reshape(-1)])
        assert blob.min() >= 0 and blob.max() < (1 << 32)
        return blob.astype(np.uint32)


# ---------------------------------------------------------------------------------------------------------------------
# plain-Python semantics of a program (the statement the C / HIP witness generators are tested against; small programs only)
# ---------------------------------------------------------------------------------------------------------------------
def f4mul(x, y):
    a0, a1, a2, a3 = x
    b0, b1, b2, b3 = y
    return ((a0 * b0 + NBETA * (a1 * b3 + a2 * b2 + a3 * b1)) % P, (a0 * b1 + a1 * b0 + NBETA * (a2 * b3 + a3 * b2)) % P,
            (a0 * b2 + a1 * b1 + a2 * b0 + NBETA * a3 * b3) % P, (a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0) % P)


def f4inv(x):
    # x^-1 = x^(P^4 - 2): fine for tests
    e, r, b = P ** 4 - 2, (1, 0, 0, 0), tuple(x)
    while e:
        if e & 1:
            r = f4mul(r, b)
        b = f4mul(b, b)
        e >>= 1
    return r


def swap_state(st: Sequence[int]) -> List[int]:
    """the input state of a conditional-swap block from the row's 24 wire cells: cell 16 is the bit"""
    st = list(st)
    assert st[16] in (0, 1), "conditional swap: the selector is not a bit"
    if st[16]:
        st[0:8], st[8:16] = st[8:16], st[0:8]
    st[16] = 0
    return st


def run_program(prog: Program, inputs: Sequence[int]) -> List[Tuple[int, int, int, int]]:
    """Evaluate the witness schedule on canonical residues: -> value of every variable; raises on a failed assertion."""
    val: List[Tuple[int, int, int, int]] = [(0, 0, 0, 0)] * prog.n_vars
    K = prog.consts
    for n_op, (opw, out, i0, i1, i2, i3, i4, i5) in enumerate(prog.ops):
        op, aux = opw & 0xFF, opw >> 8
        if op == OP_INPUT:
            w = [int(x) for x in inputs[i0:i0 + aux]]
            assert len(w) == aux and all(0 <= x < P for x in w), f"op {n_op}: input word out of range"
            val[out] = tuple(w + [0] * (4 - aux))
        elif op == OP_GEN:
            qM, qA, qB, qC, qK = K[i3:i3 + 5]
            a, b, c = val[i0], val[i1], val[i2]
            ab = f4mul(a, b) if qM else (0, 0, 0, 0)
            val[out] = tuple((qM * ab[t] + qA * a[t] + qB * b[t] + qC * c[t] + (qK if t == 0 else 0)) % P for t in range(4))
        elif op == OP_MUX:
            a, b, c = val[i0], val[i1], val[i2]
            val[out] = tuple((b[t] + a[0] * (c[t] - b[t])) % P for t in range(4))
        elif op == OP_PACK:
            val[out] = (val[i0][aux], val[i1][aux], val[i2][aux], val[i3][aux])
        elif op == OP_UNPACK:
            for t in range(4):
                val[out + t] = (val[i0][t], 0, 0, 0)
        elif op == OP_INV:
            assert any(val[i0]), f"op {n_op}: inverse of zero"
            val[out] = f4inv(val[i0])
        elif op == OP_ISZ:
            val[out] = (pow(val[i0][0], P - 2, P), 0, 0, 0)
        elif op == OP_BITS:
            for t in range(31):
                val[out + t] = ((val[i0][0] >> t) & 1, 0, 0, 0)
        elif op == OP_P2:
            st = swap_state([val[v][t] for v in (i0, i1, i2, i3, i4, i5) for t in range(4)]) if aux & 1 else \
                [val[v][t] for v in (i0, i1, i2, i3, i4, i5) for t in range(4)]
            o = p2_join.permute(st)
            for w in range(NW):
                val[out + w] = tuple(o[4 * w:4 * w + 4])
        elif op == OP_EQ:
            assert val[i0] == val[i1], f"op {n_op}: wires {i0} and {i1} differ ({val[i0]} != {val[i1]})"
        else:
            raise ValueError(f"op {n_op}: unknown opcode {op}")
    return val


if __name__ == "__main__":      # python -m zeth_amd.circuits.recursion out.desc
    import sys
    blob = recursion_circuit()
    np.asarray(blob, dtype="<u4").tofile(sys.argv[1])
    print(f"{sys.argv[1]}: {blob.size} words")
