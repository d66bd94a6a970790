"""eval_check code generator: circuit desc (PolyExtStep list) -> straight-line HIP for gfx950.

Upstream ships machine-generated CUDA for `CircuitHal::eval_check` (Zirgen output inside
risc0-circuit-rv32im-sys, un-vendored: /root/reference/Cargo.lock:5320), split over many translation units because
one domain point evaluates O(10^4 - 10^5) field operations.  Here the generator is part of the build:
`zeth_amd.build` runs it for the shipped circuit shapes and compiles the result into libzkhal_mi355x.so; at run time
`zkh_circuit_load` picks the kernels whose desc hash matches, a desc that is not shipped gets the same treatment at
load time (circuits/jit.py), and the on-device interpreter (circuit.hip) is the cross-check / last resort.

MI355X-first choices (all result-preserving — field arithmetic is exact, results equal the literal step interpreter):
  * value registers are Fp (the prover evaluates on the base-field coset); a value is Fp4 only downstream of a
    ConstExt; mix totals are Fp4;
  * every MixState's `mul` is a *static* power of poly_mix (True = mix^0, AndEqz adds 1, AndCond adds the inner
    exponent), so the per-step Fp4 x Fp4 product of the literal algorithm disappears: powers come from a precomputed
    table through scalar loads, AndEqz is 4 multiply-adds into 64-bit accumulators reduced once per 4 constraints;
  * VALUE NUMBERING: structurally identical sub-expressions (same op on the same canonical operands, commutative ops
    normalised) get one canonical value;
  * a software REGISTER CACHE bounds the register pressure: canonical values (taps and intermediates) are kept in an
    LRU of REG_BUDGET C++ variables; what drops out is recomputed / re-loaded when a later constraint needs it, so the
    taps a neighbourhood of constraints keeps reading stay in VGPRs and nothing else does.  Offset "epochs" (opaque
    copies of the lane offsets every EPOCH_LOADS loads) stop LLVM's GVN from merging all loads of a tap into one
    kernel-long live range, which is what turns a 50 k-step circuit into spills;
  * SPLITTING: the constraint leaves (in depth-first order of the mix tree) are cut into parts of roughly equal weight;
    each part is its own kernel (own translation unit when shipped, own code object when compiled at load time, all
    compiled in parallel) that evaluates only its leaves — partial sums of an AndCond's inner chain are multiplied by
    the condition in every part that holds some of its leaves (distributivity) — and ADDS its share into `check`;
  * one lane per domain point, every tap read is a coalesced column read; 1/((3x)^n - 1) takes only 4 values on the
    coset (3^n * i^(idx mod 4)): a 4-entry kernel argument.
"""
from __future__ import annotations

import bisect
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import syn_air
from .desc import (OP_ADD, OP_AND_COND, OP_AND_EQZ, OP_CONST, OP_CONST_EXT, OP_GET, OP_GET_GLOBAL, OP_MUL, OP_SUB,
                   OP_TRUE, Circuit, P)

R2 = pow(2, 64, P)
R1 = pow(2, 32, P)           # Montgomery one
USE_SOP = os.environ.get("ZKH_CODEGEN_SOP", "1") != "0"     # sums of products as one 64-bit chain + one reduction
REG_BUDGET = int(os.environ.get("ZKH_CODEGEN_REGS", "96"))      # values (taps + intermediates) the register cache of a kernel holds
EPOCH_LOADS = int(os.environ.get("ZKH_CODEGEN_EPOCH", "48"))    # tap loads per offset epoch
PART_WEIGHT = int(os.environ.get("ZKH_CODEGEN_PART", "6400"))   # value steps per generated kernel (~ one translation unit / code object)
USE_LAZY = int(os.environ.get("ZKH_CODEGEN_LAZY", "1"))
SOP_MIN = int(os.environ.get("ZKH_CODEGEN_SOPMIN", "3"))       # fewest terms an ADD/SUB tree needs to become a sum of products
SOP_SLACK = int(os.environ.get("ZKH_CODEGEN_SOPSLACK", "4"))   # cost-model margin a sum of products must win by
PREFETCH = int(os.environ.get("ZKH_CODEGEN_PREFETCH", "8"))    # tap loads issued this many constraints ahead of their first use
# Order the constraints of a chain by tap-set locality (Plan.order_by_locality): 23 % (SYN-HEAVY) / 37 % (KECCAK-F) fewer tap
# loads per point.  On its own it LOST (round 3, profiles/r03_eval_check_locality.txt: 12.41 -> 12.87 ms): it scatters the
# constraints' mix-power exponents, whose scalar loads then no longer merge (s_load_dwordx16 -> x4) and spill SGPRs (+6.2 k
# v_readlane / v_writelane per point).  With GATHER (below) the powers are consecutive in ANY emission order and it pays
# (round 4, profiles/r04_eval_check_gather.txt): SYN-HEAVY 13.8 -> 13.2 ms, KECCAK-F 1.29 -> 1.14 ms, bit-exact.
LOCALITY = int(os.environ.get("ZKH_CODEGEN_LOCALITY", "1"))
LOCALITY_WINDOW = int(os.environ.get("ZKH_CODEGEN_LOCWIN", "0")) or max(16, REG_BUDGET * 2 // 3)   # taps assumed resident when the next constraint is chosen
# The mix powers a part reads, GATHERED into emission order (slot k of the part's table = the k-th power its code touches): the
# scalar loads of consecutive constraints are consecutive again whatever order the constraints are emitted in, so they merge
# (s_load_dwordx16: 587 -> 783 over SYN-HEAVY's 13 kernels) and no SGPR spills into VGPR lanes (v_readlane + v_writelane
# 1 464 -> 0).  The kernel exports its exponent list (`exps_<kernel>` / `<kernel>_exps`, first word = count); circuit.hip computes
# the per-call power table through it (mix^exps[i], ONE small launch, as before) and hands every part its own slice.
GATHER = int(os.environ.get("ZKH_CODEGEN_GATHER", "1"))
# FACTOR (round 5): constraints of one chain that are products with a COMMON factor,  mix^e_i * (f * q_i),  are accumulated as
# f * sum_i mix^e_i * q_i  (distributivity: the same field element, hence the same canonical words in `check`).  A component of a
# real circuit multiplies every one of its constraints by the same selector / vanishing term (SYN-HEAVY: 44 constraints per triple
# share Z_j); the per-constraint product f * q_i (a multiply + Montgomery step) disappears and the factor is applied once per group
# as four 64-bit multiply-adds into the outer sums.  Groups of at least FACTOR_MIN members; 0 disables.
FACTOR_MIN = int(os.environ.get("ZKH_CODEGEN_FACTOR", "3"))
# DISTRIBUTE (round 5): a product term  o * (x +- y)  of a sum of products, with o a single-use arithmetic value and x, y values
# that are canonical anyway (taps, constants, globals, values with other consumers), is emitted as  o*x +- o*y: one more 64-bit
# multiply-add, but the addition disappears and o may stay in [0, 2P) — its conditional subtraction (v_subrev_co + v_cndmask, and
# the s_nop between them) is what the lazy sum x +- y used to force.
# LINFORM (round 6): an Fp4-valued constraint whose value is LINEAR over Fp4 CONSTANTS —  x = sum_j C_j * b_j  with C_j constants
# (ConstExt operands, products of them, the unit) and b_j base-field values — contributes  mix^e * x = sum_j (mix^e * C_j) * b_j:
# every term is an ordinary BASE leaf (four 64-bit multiply-adds) against a table slot that holds mix^e * C_j instead of mix^e.  The
# slot's constant travels with the kernel (`<kernel>_pwc` / `pwc_<kernel>`), the library multiplies it in when it builds the gathered
# table (one small launch more per call).  The Fp4 value itself — its four component products, the sixteen products + three
# reductions of ext_accumulate — is never computed.  Same field element, same canonical words in `check`.  SYN-HEAVY: the 371
# constraints (c + L) * s cost ~58 VALU instructions each before, ~12 after.  Needs GATHER; at most LIN_MAX terms, else the Fp4 path.
LINFORM = int(os.environ.get("ZKH_CODEGEN_LINFORM", "1"))
LIN_MAX = 4
# SIGNED (round 6): the running constraint sums are SIGNED 64-bit.  A base leaf multiplies the CENTRED mix power (|p| <= (P-1)/2: the
# slot is flagged in bit 31 of its exponent word, the library centres it when it builds the table) by its operand as an int32 — a
# canonical word as it is, a lazy one ([0, 2P)) as x - P in [-P, P), the same residue, one subtraction — so EVERY leaf is at most
# (P-1)/2 * P ~ P^2 / 2: four leaves between folds whether their operands were reduced or not (2^63 = 2.27 P^2), where the unsigned
# sums held two lazy ones.  Folding was 17 % of a kernel; SYN-HEAVY 84.1 k -> 77 k VALU instructions per point.  A non-linear Fp4 leaf
# (a product of two non-constant Fp4 values: none in the shipped circuits) goes through the reduced total instead of ext_accumulate.
SIGNED = int(os.environ.get("ZKH_CODEGEN_SIGNED", "1"))


def signed_sums() -> bool:
    """the signed running sums need the gathered tables (a centred slot is a slot of a kernel's OWN table): read when code is emitted"""
    return bool(SIGNED and GATHER)

DISTRIBUTE = int(os.environ.get("ZKH_CODEGEN_DISTRIBUTE", "0"))    # measured on the static opcode table and REJECTED as the default (profiles/r05_eval_check_static.txt)
# compile flags of the generated translation units (build.py and jit.py use the same list)
KERNEL_FLAGS = [f for f in os.environ.get("ZKH_CODEGEN_FLAGS", "").split() if f]
GENERATOR_VERSION = 13


def desc_hash64(desc: np.ndarray) -> int:
    """FNV-1a over the little-endian bytes of the desc words (must match circuit.hip desc_hash64)."""
    h = 0xCBF29CE484222325
    for b in np.asarray(desc, dtype="<u4").tobytes():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def mont(x: int) -> int:
    return (x % P) * pow(2, 32, P) % P


def fp4_const_mul(a: Tuple[int, int, int, int], b: Tuple[int, int, int, int]) -> Tuple[int, int, int, int]:
    """product of two Fp4 constants (canonical residues), x^4 = -11"""
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += a[i] * b[j]
    return tuple((r[k] - 11 * (r[k + 4] if k + 4 < 7 else 0)) % P for k in range(4))


UNIT = (1, 0, 0, 0)


def analyse(c: Circuit):
    """-> (kinds, mix_exp, used_fp, used_mix, n_pows): static mix exponents and liveness of the step list."""
    mix_exp: List[int] = []
    nf = nm = 0
    kinds = []
    for op, a, b, cc, d in c.steps:
        if op >= OP_TRUE:
            if op == OP_TRUE:
                mix_exp.append(0)
            elif op == OP_AND_EQZ:
                mix_exp.append(mix_exp[a] + 1)
            else:
                mix_exp.append(mix_exp[a] + mix_exp[cc])
            kinds.append(("m", nm))
            nm += 1
        else:
            kinds.append(("f", nf))
            nf += 1
    # liveness from ret backwards
    used_f = [False] * nf
    used_m = [False] * nm
    used_m[c.ret] = True
    for i in range(len(c.steps) - 1, -1, -1):
        op, a, b, cc, d = c.steps[i]
        k, vid = kinds[i]
        live = used_m[vid] if k == "m" else used_f[vid]
        if not live:
            continue
        if op in (OP_ADD, OP_SUB, OP_MUL):
            used_f[a] = used_f[b] = True
        elif op == OP_AND_EQZ:
            used_m[a] = True
            used_f[b] = True
        elif op == OP_AND_COND:
            used_m[a] = used_m[cc] = True
            used_f[b] = True
    max_pow = 0
    for i, (op, a, b, cc, d) in enumerate(c.steps):
        k, vid = kinds[i]
        if k == "m" and used_m[vid] and op in (OP_AND_EQZ, OP_AND_COND):
            max_pow = max(max_pow, mix_exp[a])
    return kinds, mix_exp, used_f, used_m, max_pow + 1


# -------------------------------------------------------------------------------------------------------------------
# Plan: value numbering + the mix tree in chain form
# -------------------------------------------------------------------------------------------------------------------
@dataclass
class Plan:
    c: Circuit
    fp: List[Tuple[int, int, int, int, int]] = field(default_factory=list)     # per fp var: (op, a, b, c, d), operands canonical
    canon: List[int] = field(default_factory=list)                             # fp var -> canonical fp var
    ext: List[bool] = field(default_factory=list)                              # fp var is Fp4-valued
    mix: List[Tuple] = field(default_factory=list)                             # per mix var: ('t',) | ('e', x, v) | ('c', x, cond, y)
    mix_exp: List[int] = field(default_factory=list)
    n_leaves: List[int] = field(default_factory=list)                          # AndEqz leaves below each mix var
    cone_size: Dict[int, int] = field(default_factory=dict)
    n_pows: int = 1
    n_unique: int = 0                                                          # canonical value steps reachable from ret
    sop: Dict[int, List[Tuple]] = field(default_factory=dict)                  # root value -> [(sign, 'p', a, b) | (sign, 'v', x)]
    absorbed: set = field(default_factory=set)                                 # values that only exist inside a root's sum of products
    n_sop_terms: int = 0
    lazy: set = field(default_factory=set)                                     # values kept in [0, 2P): every consumer multiplies
    _chains: Dict[int, List[Tuple]] = field(default_factory=dict)
    _tapsets: Dict[int, frozenset] = field(default_factory=dict)
    _vn: Dict[Tuple, int] = field(default_factory=dict)                        # the value-numbering table (kept: LINFORM adds products)
    _lin: Dict[int, Optional[List[Tuple]]] = field(default_factory=dict)       # Fp4 value -> [(Fp4 constant, base value)] or None
    n_synth: int = 0                                                           # values LINFORM added to the step list's own

    @staticmethod
    def build(c: Circuit) -> "Plan":
        p = Plan(c)
        table = p._vn
        for op, a, b, cc, d in c.steps:
            if op >= OP_TRUE:
                if op == OP_TRUE:
                    p.mix.append(("t",)); p.mix_exp.append(0); p.n_leaves.append(0)
                elif op == OP_AND_EQZ:
                    p.mix.append(("e", a, p.canon[b])); p.mix_exp.append(p.mix_exp[a] + 1); p.n_leaves.append(p.n_leaves[a] + 1)
                else:
                    p.mix.append(("c", a, p.canon[b], cc)); p.mix_exp.append(p.mix_exp[a] + p.mix_exp[cc])
                    p.n_leaves.append(p.n_leaves[a] + p.n_leaves[cc])
                continue
            vid = len(p.fp)
            if op in (OP_ADD, OP_SUB, OP_MUL):
                x, y = p.canon[a], p.canon[b]
                if op != OP_SUB and x > y:
                    x, y = y, x                                  # commutative: one canonical operand order
                key = (op, x, y)
                is_ext = p.ext[x] or p.ext[y]
                step = (op, x, y, 0, 0)
            elif op == OP_CONST:
                key, is_ext, step = (op, a % P), False, (op, a % P, 0, 0, 0)
            elif op == OP_CONST_EXT:
                key, is_ext, step = (op, a % P, b % P, cc % P, d % P), True, (op, a % P, b % P, cc % P, d % P)
            elif op == OP_GET:
                key, is_ext, step = (op, a), False, (op, a, 0, 0, 0)
            elif op == OP_GET_GLOBAL:
                key, is_ext, step = (op, a, b), False, (op, a, b, 0, 0)
            else:
                raise ValueError(f"unknown step op {op}")
            rep = table.setdefault(key, vid)
            p.fp.append(step); p.canon.append(rep); p.ext.append(is_ext)
        # reachable canonical values + max power used
        seen = set()
        max_pow = 0
        stack = [c.ret]
        seen_m = set()
        roots: List[int] = []
        while stack:
            m = stack.pop()
            if m in seen_m:
                continue
            seen_m.add(m)
            node = p.mix[m]
            if node[0] == "e":
                stack.append(node[1]); roots.append(node[2]); max_pow = max(max_pow, p.mix_exp[node[1]])
            elif node[0] == "c":
                stack.append(node[1]); stack.append(node[3]); roots.append(node[2]); max_pow = max(max_pow, p.mix_exp[node[1]])
        p.n_pows = max_pow + 1
        work = list(roots)
        while work:
            v = work.pop()
            if v in seen:
                continue
            seen.add(v)
            op, x, y, _, _ = p.fp[v]
            if op in (OP_ADD, OP_SUB, OP_MUL):
                work.append(x); work.append(y)
        p.n_unique = sum(1 for v in seen if p.fp[v][0] in (OP_ADD, OP_SUB, OP_MUL))
        # The values the emitter will actually ask for.  With FACTOR a grouped constraint f * q_i is never computed — its members q_i
        # and the factor f are; with LINFORM an Fp4 leaf that is linear over constants is never computed — the base values b_j of its
        # terms are.  Both analyses below (sums of products, lazy representatives) must see THOSE as the consumers' operands (round 5
        # shipped the lazy analysis on the old roots once: tools/check_bounds.py is the net under that).
        eff: List[int] = []

        def leaf(v: int):
            terms = p.linform(v)
            if terms is None:
                eff.append(v)
            else:
                eff.extend(b for _, b in terms)

        def walk(m: int):
            for it in p.chain(m):
                if it[0] == "e":
                    leaf(it[1])
                elif it[0] == "g":
                    eff.append(it[1])
                    for q, _ in it[2]:
                        leaf(q)
                else:
                    eff.append(it[1])
                    walk(it[2])
        walk(c.ret)
        reach0 = set()
        for r in eff:
            reach0.update(p.cone(r, reach0))
        if USE_SOP:
            p.find_sums_of_products(reach0, eff)
        if USE_LAZY:
            reach = set()
            for r in eff:
                reach.update(p.cone(r, reach))
            p.find_lazy(reach, eff)
        return p

    def value(self, op: int, x: int, y: int) -> int:
        """the canonical value  x op y  of two BASE values: the step list's own if it has one, else a new one (LINFORM's products)"""
        if op != OP_SUB and x > y:
            x, y = y, x
        key = (op, x, y)
        v = self._vn.get(key)
        if v is None:
            v = len(self.fp)
            self._vn[key] = v
            self.fp.append((op, x, y, 0, 0)); self.canon.append(v); self.ext.append(False)
            self.n_synth += 1
        return v

    def one(self) -> int:
        v = self._vn.get((OP_CONST, 1))
        if v is None:
            v = len(self.fp)
            self._vn[(OP_CONST, 1)] = v
            self.fp.append((OP_CONST, 1, 0, 0, 0)); self.canon.append(v); self.ext.append(False)
        return v

    def linform(self, v: int) -> Optional[List[Tuple]]:
        """An Fp4 value as  sum_j C_j * b_j  (C_j Fp4 constants as canonical residues, b_j base values; b_j = the constant 1 for a
        pure constant term), or None: not Fp4, not linear over constants (a product of two non-constant Fp4 values), more than
        LIN_MAX terms, or LINFORM off."""
        if not (LINFORM and GATHER) or not self.ext[v]:
            return None
        if v in self._lin:
            return self._lin[v]

        def form(x: int):
            if not self.ext[x]:
                return [(UNIT, x)]
            if x in self._lin and self._lin[x] is not None:
                return self._lin[x]
            op, a, b, cc, d = self.fp[x]
            if op == OP_CONST_EXT:
                return [((a, b, cc, d), self.one())]
            if op in (OP_ADD, OP_SUB):
                fa, fb = form(a), form(b)
                if fa is None or fb is None:
                    return None
                if op == OP_SUB:
                    fb = [(tuple((P - k) % P for k in C), t) for C, t in fb]
                acc: Dict[int, Tuple] = {}
                for C, t in fa + fb:
                    acc[t] = tuple((p_ + q_) % P for p_, q_ in zip(acc[t], C)) if t in acc else C
                out = [(C, t) for t, C in acc.items() if any(C)]
                return out if len(out) <= LIN_MAX else None
            if op == OP_MUL:
                if self.ext[a] and self.ext[b]:
                    fa, fb = form(a), form(b)
                    if fa is None or fb is None:
                        return None
                    one = self.one()
                    for k, o in ((fa, fb), (fb, fa)):
                        if len(k) == 1 and k[0][1] == one:                    # an Fp4 constant times a linear form
                            return [(fp4_const_mul(k[0][0], C), t) for C, t in o]
                    return None
                e_, s_ = (a, b) if self.ext[a] else (b, a)
                fe = form(e_)
                if fe is None:
                    return None
                one = self.one()
                return [(C, s_ if t == one else self.value(OP_MUL, t, s_)) for C, t in fe]
            return None
        out = form(v)
        self._lin[v] = out
        return out

    def find_lazy(self, reachable, mix_ext=()) -> None:
        """Which arithmetic values may skip their final conditional subtraction and live in [0, 2P).
        A product a * b with a < 2P, b < P is below 2 P^2 < P 2^32, the domain of the Montgomery step, whose uncorrected
        output is again below 2P; so a value whose every consumer MULTIPLIES it (a product, a term of a sum of products, a
        constraint's mix-power accumulation, the base factor of an Fp4 * Fp product) never needs the canonical
        representative, provided the other factor is canonical.  Operands of additions and subtractions must be canonical
        (a + b < 2P has to fit 32 bits, a - b + P has to stay positive)."""
        def ext_by_base(v):
            """An Fp4 value computed as Fp4 * Fp: four base products, so its components can be lazy in the same way."""
            return self.ext[v] and self.fp[v][0] == OP_MUL and self.ext[self.fp[v][1]] != self.ext[self.fp[v][2]]

        def arith(v):
            return v in self.sop or (self.fp[v][0] in (OP_ADD, OP_SUB, OP_MUL) and not self.ext[v]) or ext_by_base(v)
        live = sorted(v for v in reachable if v not in self.absorbed)
        need_c = set(v for v in mix_ext if self.ext[v])        # Fp4 constraints and Fp4 conditions are consumed whole
        pairs: List[Tuple[int, int]] = []
        for r in live:
            if r in self.sop:
                pairs.extend((t[2], t[3]) for t in self.sop[r] if t[1] == "p")
                continue
            op, a, b, _, _ = self.fp[r]
            if op not in (OP_ADD, OP_SUB, OP_MUL):
                continue
            if self.ext[r]:
                if op == OP_MUL and self.ext[a] != self.ext[b]:
                    pairs.append((a, b))                       # Fp4 * Fp: four products; one side canonical is enough
                    continue
                need_c.update((a, b))                          # Fp4 +- anything, Fp4 * Fp4: canonical components
            elif op == OP_MUL:
                pairs.append((a, b))
            else:
                need_c.update((a, b))
        for a, b in pairs:
            if a == b:
                need_c.add(a)
        # two lazy factors would reach 4 P^2: one of each pair pays.  Greedy vertex cover of the conflict graph, highest
        # degree first — the common factor of many products (a selector, say) is the one to keep canonical.
        adj: Dict[int, set] = {}
        for a, b in pairs:
            if a != b and arith(a) and arith(b) and a not in need_c and b not in need_c:
                adj.setdefault(a, set()).add(b); adj.setdefault(b, set()).add(a)
        import heapq
        heap = [(-len(n), v) for v, n in adj.items()]
        heapq.heapify(heap)
        while heap:
            negdeg, v = heapq.heappop(heap)
            if v in need_c or not adj.get(v):
                continue
            if -negdeg != len(adj[v]):
                heapq.heappush(heap, (-len(adj[v]), v))        # stale entry: re-queue with the current degree
                continue
            need_c.add(v)
            for o in adj.pop(v):
                adj[o].discard(v)
                if adj[o]:
                    heapq.heappush(heap, (-len(adj[o]), o))
        self.lazy = {v for v in live if arith(v) and v not in need_c}

    def find_sums_of_products(self, reachable, mix_roots) -> None:
        """Lazy arithmetic: an ADD/SUB tree whose inner nodes have no other consumer is one expression
               v = sum_i (+-) a_i * b_i  +  sum_j (+-) x_j
        and is evaluated as a 64-bit multiply-add chain with ONE Montgomery reduction (products as they are, plain terms as
        x * R so that they pick up the same 2^-32) instead of a full modular multiply per product and a modular add per
        node.  Exact: the 64-bit sum is the integer sum of the terms (bounded below 2 P 2^32, the domain of
        mont_reduce_wide; negative terms enter as (P - a) * b), so the reduced value is the same field element."""
        uses: Dict[int, int] = {}
        for v in reachable:
            op, a, b, _, _ = self.fp[v]
            if op in (OP_ADD, OP_SUB, OP_MUL):
                uses[a] = uses.get(a, 0) + 1
                uses[b] = uses.get(b, 0) + 1
        for r in mix_roots:
            uses[r] = uses.get(r, 0) + 1
        PROD, PLAIN, LIMIT = float(P) * P, float(P) * pow(2, 32, P), 1.72e19

        def is_sum(v):
            return self.fp[v][0] in (OP_ADD, OP_SUB) and not self.ext[v]

        for r in sorted(reachable, reverse=True):
            if not is_sum(r) or r in self.absorbed:
                continue
            terms: List[Tuple] = []
            inner: List[int] = []
            stack = [(r, 1)]
            while stack:
                v, sg = stack.pop()
                op, a, b, _, _ = self.fp[v]
                for child, csg in ((b, sg if op == OP_ADD else -sg), (a, sg)):      # a is pushed last: emitted first
                    cop = self.fp[child][0]
                    if uses.get(child, 0) == 1 and not self.ext[child] and child not in self.absorbed:
                        if cop in (OP_ADD, OP_SUB):
                            inner.append(child); stack.append((child, csg)); continue
                        if cop == OP_MUL:
                            fa, fb = self.fp[child][1], self.fp[child][2]
                            split = None
                            if DISTRIBUTE:
                                for o, g in ((fa, fb), (fb, fa)):
                                    if (self.fp[g][0] in (OP_ADD, OP_SUB) and not self.ext[g] and uses.get(g, 0) == 1 and g not in self.absorbed
                                            and not self.ext[o] and uses.get(o, 0) == 1 and self.fp[o][0] in (OP_ADD, OP_SUB, OP_MUL)
                                            and all(self.fp[t][0] in (OP_GET, OP_CONST, OP_GET_GLOBAL) or uses.get(t, 0) > 1 for t in self.fp[g][1:3])
                                            and not any(self.ext[t] for t in self.fp[g][1:3])):
                                        split = (o, g)
                                        break
                            if split:
                                o, g = split
                                inner.extend((child, g))
                                terms.append((csg, "p", o, self.fp[g][1]))
                                terms.append((csg if self.fp[g][0] == OP_ADD else -csg, "p", o, self.fp[g][2]))
                                continue
                            inner.append(child); terms.append((csg, "p", fa, fb)); continue
                    terms.append((csg, "v", child))
            n_prod = sum(1 for t in terms if t[1] == "p")
            n_plain = len(terms) - n_prod
            n_add = sum(1 for v in inner if self.fp[v][0] in (OP_ADD, OP_SUB)) + 1
            plain_cost = 16 * n_prod + 8 * n_add
            n_red = 1
            acc = 0.0
            for t in terms:                                   # reductions needed to stay inside the 64-bit bound
                wgt = PROD if t[1] == "p" else PLAIN
                if acc + wgt > LIMIT:
                    n_red += 1; acc = PLAIN
                acc += wgt
            neg = sum(1 for t in terms if t[0] < 0 and not (t[1] == "v" and self.fp[t[2]][0] == OP_CONST)
                      and not (t[1] == "p" and OP_CONST in (self.fp[t[2]][0], self.fp[t[3]][0])))
            sop_cost = 4 * len(terms) + 2 * neg + 20 * n_red
            if n_prod + n_plain >= SOP_MIN and sop_cost + SOP_SLACK <= plain_cost:
                self.sop[r] = terms
                self.absorbed.update(inner)
                self.n_sop_terms += len(terms)

    def deps(self, v: int) -> List[int]:
        """Values that must be available to compute v (the leaves of its sum of products, or its two operands)."""
        if v in self.sop:
            out = []
            for t in self.sop[v]:
                out.extend(t[2:] if t[1] == "p" else (t[2],))
            return out
        op, a, b, _, _ = self.fp[v]
        return [a, b] if op in (OP_ADD, OP_SUB, OP_MUL) else []

    def chain(self, m: int) -> List[Tuple]:
        """Items of the chain ending in mix var m, in EMISSION order: ('e', value, exp) | ('c', cond, inner var, exp).
        Every item carries its own static exponent, and a chain's total is the plain sum of its items' contributions, so the
        items may be emitted in any order: they are ordered by tap-set locality (below) instead of the order the circuit
        happened to list them in.  Deterministic and cached: the splitter, the dry run and the emitter all see one order."""
        cached = self._chains.get(m)
        if cached is not None:
            return cached
        items = []
        k = m
        while self.mix[k][0] != "t":
            node = self.mix[k]
            if node[0] == "e":
                items.append(("e", node[2], self.mix_exp[node[1]]))
            else:
                items.append(("c", node[2], node[3], self.mix_exp[node[1]]))
            k = node[1]
        items.reverse()
        if FACTOR_MIN:
            items = self.group_by_factor(items)
        if LOCALITY and len(items) > 2:
            items = self.order_by_locality(items)
        self._chains[m] = items
        return items

    def group_by_factor(self, items: List[Tuple]) -> List[Tuple]:
        """('e', v, e) items whose value is a plain product v = f * q with a factor f shared by >= FACTOR_MIN items of this chain
        become ONE item ('g', f, ((q, e), ...)) at the position of the group's first member: contribution f * sum mix^e q."""
        cand: Dict[int, List[int]] = {}
        ops: Dict[int, Tuple[int, int]] = {}
        for i, it in enumerate(items):
            if it[0] != "e":
                continue
            v = it[1]
            if v in self.sop or self.fp[v][0] != OP_MUL:
                continue
            a, b = self.fp[v][1], self.fp[v][2]
            if self.ext[a] and self.ext[b]:
                continue                                   # Fp4 * Fp4: no scalar factor to pull out
            ops[i] = (a, b)
            for f in {a, b}:
                if not self.ext[f] and self.fp[f][0] not in (OP_CONST,):
                    cand.setdefault(f, []).append(i)
        taken: Dict[int, int] = {}                         # item index -> factor
        for f in sorted(cand, key=lambda f: (-len(cand[f]), f)):
            free = [i for i in cand[f] if i not in taken]
            if len(free) >= FACTOR_MIN:
                for i in free:
                    taken[i] = f
        if not taken:
            return items
        groups: Dict[int, List[Tuple[int, int]]] = {}
        for i in sorted(taken):
            f = taken[i]
            a, b = ops[i]
            groups.setdefault(f, []).append((b if a == f else a, items[i][2]))
        out, done = [], set()
        for i, it in enumerate(items):
            f = taken.get(i)
            if f is None:
                out.append(it)
            elif f not in done:
                done.add(f)
                out.append(("g", f, tuple(groups[f])))
        return out

    def tapset(self, v: int) -> frozenset:
        """taps (canonical OP_GET values) below value v"""
        ts = self._tapsets.get(v)
        if ts is None:
            ts = frozenset(x for x in self.cone(v, ()) if self.fp[x][0] == OP_GET)
            self._tapsets[v] = ts
        return ts

    def item_taps(self, it) -> frozenset:
        if it[0] == "e":
            return self.tapset(it[1])
        if it[0] == "g":
            acc = set(self.tapset(it[1]))
            for q, _ in it[2]:
                acc |= self.tapset(q)
            return frozenset(acc)
        acc = set(self.tapset(it[1]))
        for sub in self.chain(it[2]):
            acc |= self.item_taps(sub)
        return frozenset(acc)

    def order_by_locality(self, items: List[Tuple]) -> List[Tuple]:
        """Greedy clustering: after an item, emit next the unplaced item with the largest FRACTION of its taps among the
        last LOCALITY_WINDOW distinct taps emitted (what the register cache plausibly still holds), then the fewest new taps;
        ties and cold starts fall back to the circuit's own order.  A tap that far-apart constraints read is then loaded
        once per cluster instead of once per neighbourhood: tap loads per domain point SYN-HEAVY 2 910 -> 2 230, KECCAK-F
        16 584 -> 10 497 (and 33.4 k -> 29.6 k arithmetic steps: fewer evicted intermediates are recomputed)."""
        taps = [self.item_taps(it) for it in items]
        n = len(items)
        users: Dict[int, List[int]] = {}
        for i, ts in enumerate(taps):
            for t in ts:
                users.setdefault(t, []).append(i)
        placed = [False] * n
        score = [0] * n                      # taps of item i currently in the window
        window: "OrderedDict[int, None]" = OrderedDict()
        order: List[int] = []
        next_cold = 0
        hot: set = set()                     # unplaced items with score > 0

        def enter(t):
            if t in window:
                window.move_to_end(t)
                return
            window[t] = None
            for i in users.get(t, ()):
                if not placed[i]:
                    score[i] += 1
                    hot.add(i)
            while len(window) > LOCALITY_WINDOW:
                old, _ = window.popitem(last=False)
                for i in users.get(old, ()):
                    if not placed[i]:
                        score[i] -= 1
                        if score[i] <= 0:
                            hot.discard(i)
        for _ in range(n):
            best = -1
            if hot:
                # most taps already resident, then fewest NEW taps to load, then the circuit's order
                best = min(hot, key=lambda i: (-(score[i] / len(taps[i])), len(taps[i]) - score[i], i))
            if best < 0:
                while placed[next_cold]:
                    next_cold += 1
                best = next_cold
            placed[best] = True
            hot.discard(best)
            order.append(best)
            for t in sorted(taps[best]):
                enter(t)
        return [items[i] for i in order]

    def cone(self, v: int, have) -> List[int]:
        """Canonical values needed to compute v that are not in `have`, in dependency order (operands first)."""
        out: List[int] = []
        mark = set()
        stack = [(v, False)]
        while stack:
            x, done = stack.pop()
            if done:
                out.append(x)
                continue
            if x in have or x in mark:
                continue
            mark.add(x)
            stack.append((x, True))
            for o in reversed(self.deps(x)):
                stack.append((o, False))
        return out

    def leaf_weights(self) -> List[int]:
        """Cost estimate per AndEqz leaf in depth-first order (arithmetic steps in its cone, no cross-leaf sharing)."""
        w: List[int] = []

        def cost(v: int) -> int:
            terms = self.linform(v)
            vals = [v] if terms is None else [b for _, b in terms]
            return 4 * len(vals) + sum(1 for r in vals for x in self.cone(r, set()) if self.fp[x][0] in (OP_ADD, OP_SUB, OP_MUL))

        def walk(m: int):
            for it in self.chain(m):
                if it[0] == "e":
                    w.append(cost(it[1]))
                elif it[0] == "g":
                    for q, _ in it[2]:
                        w.append(cost(q))
                else:
                    walk(it[2])
        walk(self.c.ret)
        return w


def split_points(weights: List[int], part_weight: Optional[int] = None) -> List[Tuple[int, int]]:
    """Cut the leaf sequence into contiguous [lo, hi) ranges of roughly part_weight each (default: ZKH_CODEGEN_PART as it is
    NOW, else PART_WEIGHT — read at call time so that a test can ask for a multi-part split of a small circuit)."""
    if part_weight is None:
        part_weight = int(os.environ.get("ZKH_CODEGEN_PART", PART_WEIGHT))
    total = sum(weights)
    n_parts = max(1, -(-total // part_weight))
    target = total / n_parts
    cuts, acc, lo = [], 0.0, 0
    for i, w in enumerate(weights):
        acc += w
        if acc >= target * (len(cuts) + 1) and len(cuts) + 1 < n_parts:
            cuts.append((lo, i + 1)); lo = i + 1
    cuts.append((lo, len(weights)))
    return [c for c in cuts if c[1] > c[0]] or [(0, len(weights))]


# -------------------------------------------------------------------------------------------------------------------
# Emission of one part
# -------------------------------------------------------------------------------------------------------------------
class _Emitter:
    """Straight-line code for the constraint leaves [lo, hi) of the mix tree.

    Values live in a software-managed register cache: `cache` maps a canonical value to the C++ variable that currently
    holds it, in LRU order, and is capped at REG_BUDGET entries.  A value that falls out of the cache is simply no longer
    referenced (its live range ends at its last use); if a later constraint needs it again it is recomputed — a tap is
    re-loaded — under a fresh variable name.  Taps that a neighbourhood of constraints keeps reading therefore stay in
    registers, cold ones do not pin any, and the VGPR demand of the kernel is the budget plus a constant."""

    def __init__(self, plan: Plan, lo: int, hi: int, future: Optional[Dict[int, List[int]]] = None):
        self.p, self.lo, self.hi = plan, lo, hi
        self.lines: List[str] = []
        # `future` = for every canonical value the (sorted) request numbers at which some constraint of this part needs it,
        # recorded by a dry run over the same leaves.  With it the cache evicts the value whose next use is FARTHEST away
        # (Belady's rule: the code is generated offline, so the future is known) instead of the least recently used one.
        self.future = future
        self.dry = future is None
        self.request = 0
        self.requests: List[int] = []       # dry run: the root of every request, in order
        self.upcoming: List[int] = []       # real run: the dry run's request list (for prefetching)
        self.cache: "OrderedDict[int, str]" = OrderedDict()
        self.gen: Dict[int, int] = {}       # canonical value -> how often it has been (re)defined
        self.pinned: set = set()
        self.depth_used = 0
        self.pend: Dict[int, int] = {}      # depth -> lazy products pending in s{d}_*
        self.tzero: Dict[int, bool] = {}    # depth -> t{d}_* statically known to be zero
        self.folded: Dict[int, bool] = {}   # depth -> s{d}_* carries a folded (unreduced) part of the total
        self.globals_used: set = set()
        self.leaf = 0                       # depth-first index of the next leaf
        self.pw_exps: List[int] = []        # GATHER: the exponent behind every slot of this part's power table, in emission order
        self.pw_consts: List[Tuple[int, Tuple[int, int, int, int]]] = []    # LINFORM: (slot, Fp4 constant the slot's power is multiplied by)
        self.epoch = -1
        self.epoch_loads = EPOCH_LOADS      # forces an epoch before the first load
        self.epoch_backs: List[set] = []
        self.n_loads = 0
        self.n_arith = 0
        # BOUNDS TRACE (round 6): the generator's own worst cases in exact Python integers, emitted as `// BOUND name <= N` comments
        # next to the statements they describe — every lazy value (< 2P), every 64-bit sum of products before its reduction, the four
        # running constraint sums of a level after every accumulation.  tools/check_bounds.py re-derives each from the emitted text
        # alone (it shares no code with this file) and fails if the text allows more than is claimed here, or than a word holds.
        self.sb: Dict[int, int] = {}        # depth -> worst case of s{d}_k

    def w(self, s: str) -> None:
        self.lines.append(s)

    def claim(self, name: str, hi: int) -> None:
        if not self.dry:
            self.lines.append(f"    // BOUND {name} <= {hi}")

    def hi_of(self, v: int) -> int:
        """worst case of the word that holds value v: a constant is itself, a lazy value is below 2P, anything else canonical"""
        if self.p.fp[v][0] == OP_CONST:
            return mont(self.p.fp[v][1])
        return 2 * P - 1 if v in self.p.lazy else P - 1

    # ---- offset epochs ----
    def new_epoch(self) -> None:
        # Epoch-local copies of the lane's byte offsets and of the domain size, made opaque to the optimiser: tap loads and
        # column-base computations of different epochs are different SSA values, so GVN cannot undo the cache policy by
        # merging every load of a tap into one kernel-long live range (hundreds of VGPRs and SGPR spills at realistic circuit
        # sizes).  Not `volatile` and no memory clobber: the mix-power reads must stay provably invariant to remain scalar
        # loads; the epoch number in the asm text keeps identical-looking statements from being merged.
        self.epoch += 1
        self.epoch_backs.append(set())
        self.epoch_loads = 0
        self.w(f"    OPAQUE_EPOCH_{self.epoch}")

    # ---- values ----
    def ref(self, v: int) -> str:
        if self.dry:
            return "0"
        op, a, b, _, _ = self.p.fp[v]
        if op == OP_CONST:
            return f"{mont(a)}u"
        if op == OP_GET_GLOBAL:
            self.globals_used.add((a, b))
            return f"q{a}_{b}"
        return self.cache[v]

    def ext_ref(self, v: int) -> str:
        """An Fp4 expression for value v (promoting Fp values)."""
        if self.dry:
            return "0"
        if self.p.ext[v]:
            return self.cache[v]
        return f"Fp4(Fp::raw({self.ref(v)}))"

    def touch(self, v: int) -> None:
        if v in self.cache:
            self.cache.move_to_end(v)

    def need(self, roots: List[int]) -> None:
        """Make every root available in the cache (computing what is missing), pinned until release()."""
        for r in roots:
            self.request += 1
            if self.dry:
                self.requests.append(r)
                continue
            if PREFETCH and self.upcoming:
                # software prefetch: the taps of the next constraints are loaded now, so their latency overlaps this
                # constraint's arithmetic (their next use is near, so the eviction rule keeps them)
                for nxt in self.upcoming[self.request: self.request + PREFETCH]:
                    if self.p.fp[nxt][0] in (OP_CONST, OP_GET_GLOBAL):
                        continue
                    for x in self.p.cone(nxt, self.cache):
                        if self.p.fp[x][0] == OP_GET and x not in self.cache:
                            self.define(x)
            if self.p.fp[r][0] in (OP_CONST, OP_GET_GLOBAL):
                continue
            missing = self.p.cone(r, self.cache)
            # operands that are already cached must survive the evictions the new definitions cause
            for x in missing:
                for o in self.p.deps(x):
                    if o in self.cache:
                        self.pinned.add(o)
            for x in missing:
                self.define(x)
                self.pinned.add(x)
            self.pinned.add(r)
            self.touch(r)

    def release(self) -> None:
        self.pinned.clear()
        if self.dry:
            return
        self.request += 0
        self.drop_dead()
        while len(self.cache) > REG_BUDGET:
            self.evict_for_one_forced()

    def evict_for_one_forced(self) -> None:
        victim, far = None, -1
        for k in self.cache:
            nu = self.next_use(k)
            if nu > far:
                victim, far = k, nu
        del self.cache[victim]

    def next_use(self, v: int) -> int:
        uses = self.future.get(v)
        if not uses:
            return 1 << 60
        i = bisect.bisect_left(uses, self.request)      # the current request still counts as a use
        return uses[i] if i < len(uses) else 1 << 60

    def evict_for_one(self) -> None:
        if len(self.cache) < REG_BUDGET:
            return
        victim, far = None, -1
        for k in self.cache:
            if k in self.pinned:
                continue
            nu = self.next_use(k)
            if nu > far:
                victim, far = k, nu
        if victim is not None:
            del self.cache[victim]

    def drop_dead(self) -> None:
        """After a request: values that no later constraint of this part needs leave the cache at once."""
        for k in [k for k in self.cache if k not in self.pinned and self.next_use(k) >= (1 << 60)]:
            del self.cache[k]

    def define(self, v: int) -> None:
        op, a, b, cc, d = self.p.fp[v]
        if op in (OP_CONST, OP_GET_GLOBAL):
            self.ref(v)
            return
        for o in self.p.deps(v):
            self.touch(o)
        g = self.gen.get(v, 0)
        self.gen[v] = g + 1
        self.evict_for_one()
        if v in self.p.sop:
            self.define_sop(v, g)
            return
        if op == OP_GET:
            grp, off, back = self.p.c.taps[a]
            if self.epoch_loads >= EPOCH_LOADS:
                self.new_epoch()
            self.epoch_backs[-1].add(back)
            self.epoch_loads += 1
            self.n_loads += 1
            name = f"v{v}" + (f"_{g}" if g else "")
            self.w(f"    const uint32_t {name} = tap_load(g{grp}, (size_t){off} * dw{self.epoch}, o{back}_{self.epoch});")
        elif op == OP_CONST_EXT:
            name = f"x{v}" + (f"_{g}" if g else "")
            self.w(f"    const Fp4 {name}(Fp::raw({mont(a)}u), Fp::raw({mont(b)}u), Fp::raw({mont(cc)}u), Fp::raw({mont(d)}u));")
        elif self.p.ext[v]:
            name = f"x{v}" + (f"_{g}" if g else "")
            sym = {OP_ADD: "+", OP_SUB: "-", OP_MUL: "*"}[op]
            if op == OP_MUL and not self.p.ext[a]:
                a, b = b, a                            # the Fp4 operand first
            if op == OP_MUL and not self.p.ext[b]:
                if v in self.p.lazy:                   # every consumer multiplies the components by a canonical factor
                    self.w(f"    const Fp4 {name} = ext_mul_base_lazy({self.cache[a]}, {self.ref(b)});")
                    self.claim(name, 2 * P - 1)
                else:
                    self.w(f"    const Fp4 {name} = {self.cache[a]} * Fp::raw({self.ref(b)});")
            elif op in (OP_ADD, OP_SUB) and not self.p.ext[b]:
                self.w(f"    const Fp4 {name} = ext_{'add' if op == OP_ADD else 'sub'}_base({self.cache[a]}, {self.ref(b)});")
            elif op == OP_ADD and not self.p.ext[a]:
                self.w(f"    const Fp4 {name} = ext_add_base({self.cache[b]}, {self.ref(a)});")
            else:
                self.w(f"    const Fp4 {name} = {self.ext_ref(a)} {sym} {self.ext_ref(b)};")
            self.n_arith += 1
        else:
            name = f"v{v}" + (f"_{g}" if g else "")
            if v in self.p.lazy:                       # every consumer multiplies: [0, 2P) will do (Plan.find_lazy)
                expr = {OP_ADD: f"{self.ref(a)} + {self.ref(b)}", OP_SUB: f"{self.ref(a)} - {self.ref(b)} + {P}u",
                        OP_MUL: f"mul_lazy({self.ref(a)}, {self.ref(b)})"}[op]
                self.w(f"    const uint32_t {name} = {expr};")
                self.claim(name, 2 * P - 1)
            else:
                fn = {OP_ADD: "add_mod", OP_SUB: "sub_mod", OP_MUL: "mul_mod"}[op]
                self.w(f"    const uint32_t {name} = {fn}({self.ref(a)}, {self.ref(b)});")
            self.n_arith += 1
        self.cache[v] = name

    def define_sop(self, v: int, g: int) -> None:
        """v = sum of (+-) products and (+-) plain terms: 64-bit multiply-add chain, one reduction (see Plan.find_sums_of_products).
        A lazy operand (< 2P) doubles its term's bound and is negated as 2P - x."""
        PROD, PLAIN, LIMIT = float(P) * P, float(P) * R1, 1.84e19
        name = f"v{v}" + (f"_{g}" if g else "")
        lz = self.p.lazy
        const_sum = 0                                  # constant plain terms fold into the accumulator's initial value
        dyn: List[Tuple[float, str, int]] = []      # (float weight the decisions use, expression, exact worst case of the term)

        def neg(x: int) -> str:
            return f"({2 * P if x in lz else P}u - {self.ref(x)})"

        def nhi(x: int, sg: int) -> int:               # worst case of the factor as it is written: x, or (2P | P) - x
            return self.hi_of(x) if sg > 0 else (2 * P if x in lz else P)

        for t in self.p.sop[v]:
            sg = t[0]
            if t[1] == "v":
                x = t[2]
                if self.p.fp[x][0] == OP_CONST:
                    c = mont(self.p.fp[x][1])
                    const_sum += (c if sg > 0 else (P - c) % P) * R1
                else:
                    dyn.append((PLAIN * (2 if x in lz else 1), f"(uint64_t){self.ref(x) if sg > 0 else neg(x)} * {R1}u", nhi(x, sg) * R1))
            else:
                a, b = t[2], t[3]
                if self.p.fp[a][0] == OP_CONST:
                    a, b = b, a                        # constant (if any) second
                wgt = PROD * (2 if (a in lz or b in lz) else 1)
                if self.p.fp[b][0] == OP_CONST:
                    c = mont(self.p.fp[b][1])
                    dyn.append((wgt, f"(uint64_t){self.ref(a)} * {c if sg > 0 else (P - c) % P}u", self.hi_of(a) * (c if sg > 0 else (P - c) % P)))
                else:
                    dyn.append((wgt, f"(uint64_t){self.ref(a) if sg > 0 else neg(a)} * {self.ref(b)}", nhi(a, sg) * self.hi_of(b)))
        tmp = f"u{v}" + (f"_{g}" if g else "")
        const_sum %= P                                 # only the residue matters to the Montgomery step
        acc = float(const_sum)
        exact = const_sum                               # the same sum in exact integers: the bounds trace

        def folded(x: int) -> int:
            return x if x < (1 << 32) else (x >> 32) * R1 + (1 << 32) - 1
        self.w(f"    uint64_t {tmp} = {const_sum}ull;")
        for wgt, expr, hi in dyn:
            if acc + wgt > LIMIT:                       # out of room: hi 2^32 + lo = hi R + lo (mod P), below 2^60 + 2^32
                self.w(f"    {tmp} = fold_acc({tmp});")
                acc = 4294967296.0 * R1 + 4294967296.0
                exact = folded(exact)
            self.w(f"    {tmp} += {expr};")
            acc += wgt
            exact += hi
        # below P 2^32 the plain reduction is enough (one correction instead of two); a lazy root skips the last correction
        wide = acc >= float(P) * 4294967296.0
        if wide and acc >= 2.0 * float(P) * 4294967296.0:
            self.w(f"    {tmp} = fold_acc({tmp});")
            exact = folded(exact)
            wide = False
        self.claim(tmp, exact)
        fn = ("mont_reduce_wide" if wide else "mont_reduce") + ("_lazy" if v in lz else "")
        self.w(f"    const uint32_t {name} = {fn}({tmp});")
        if v in lz:
            self.claim(name, 2 * P - 1)
        self.n_arith += 1
        self.cache[v] = name

    def pw(self, e: int, const: Optional[Tuple[int, int, int, int]] = None, centered: bool = False) -> int:
        """Index of mix^e in the table this kernel reads: e itself, or (GATHER) the next slot of the part's own table — which may
        hold mix^e times an Fp4 constant (LINFORM)."""
        if not GATHER:
            assert const is None or const == UNIT
            return e
        self.pw_exps.append(e | (1 << 31) if centered else e)
        if const is not None and const != UNIT:
            self.pw_consts.append((len(self.pw_exps) - 1, const))
        return len(self.pw_exps) - 1

    # ---- accumulators ----
    def use_depth(self, d: int) -> None:
        self.depth_used = max(self.depth_used, d + 1)

    def reset(self, d: int) -> None:
        self.use_depth(d)
        self.w(f"    t{d}_0 = t{d}_1 = t{d}_2 = t{d}_3 = 0; s{d}_0 = s{d}_1 = s{d}_2 = s{d}_3 = 0;")
        self.pend[d], self.tzero[d], self.folded[d] = 0, True, False
        self.sb[d] = 0

    def fold(self, d: int) -> None:
        """Make room in the 64-bit constraint sums without reducing them: s = hi 2^32 + lo = hi R + lo (mod P), which is
        below 2^60 + 2^32 and leaves room for four more products (4 P^2 + 2^60 + 2^32 < 2^64)."""
        for k in range(4):
            self.w(f"    s{d}_{k} = fold_acc{'_s' if signed_sums() else ''}(s{d}_{k});")
        self.pend[d] = 0
        self.folded[d] = True
        x = self.sb.get(d, 0)
        if signed_sums():                   # |hi| R + lo with |hi| <= (|s| >> 32) + 1
            self.sb[d] = ((x >> 32) + 1) * R1 + (1 << 32) - 1
        else:
            self.sb[d] = x if x < (1 << 32) else (x >> 32) * R1 + (1 << 32) - 1

    def flush(self, d: int) -> None:
        """t{d} = everything accumulated at depth d so far, as canonical words (needed before an Fp4 contribution is
        added, before the level is multiplied by its condition, and at the end of the kernel)."""
        if self.pend.get(d, 0) == 0 and not self.folded.get(d, False):
            return
        # The running total re-enters as t * R and the plain reduction wants the sum below P 2^32.  Two units of pending
        # products alone qualify: (P-1)(2P-1) + (P-1) R = (P-1)(2^32 - 1); so does a folded part (< 2^60 + 2^32) with one
        # unit on top; anything more is folded first.
        if signed_sums():
            # the signed Montgomery step wants |s + t R| < P 2^31 = 1.07 P^2: one unit (P^2 / 2) + a folded part (0.14 P^2) + t R (0.13 P^2)
            if self.pend.get(d, 0) > 1:
                self.fold(d)
            for k in range(4):
                if self.tzero[d]:
                    self.w(f"    t{d}_{k} = smont_canon(s{d}_{k}); s{d}_{k} = 0;")
                else:
                    self.w(f"    t{d}_{k} = smont_canon(mad_i64_k((int32_t)t{d}_{k}, (int32_t){R1}u, s{d}_{k})); s{d}_{k} = 0;")
            self.pend[d], self.tzero[d], self.folded[d] = 0, False, False
            self.sb[d] = 0
            return
        if self.pend.get(d, 0) > 2 or (self.pend.get(d, 0) > 1 and self.folded.get(d, False)):
            self.fold(d)
        for k in range(4):
            if self.tzero[d]:
                self.w(f"    t{d}_{k} = mont_reduce(s{d}_{k}); s{d}_{k} = 0;")
            else:
                self.w(f"    t{d}_{k} = mont_reduce(s{d}_{k} + (uint64_t)t{d}_{k} * {R1}u); s{d}_{k} = 0;")
        self.pend[d], self.tzero[d], self.folded[d] = 0, False, False
        self.sb[d] = 0

    def add_fp4(self, d: int, expr: str) -> None:
        """t{d} += Fp4 expression (non-lazy path: ConstExt-valued constraints and AndCond contributions)."""
        self.flush(d)
        self.w(f"    {{ const Fp4 c_ = {expr};")
        for k in range(4):
            self.w(f"      t{d}_{k} = add_mod(t{d}_{k}, c_.c[{k}].v);")
        self.w("    }")
        self.tzero[d] = False

    def acc_leaf(self, d: int, v: int, e: int, const: Optional[Tuple[int, int, int, int]] = None) -> None:
        """depth d += mix^e * v for one constraint value (`const`: the slot holds mix^e * const — a term of an Fp4 leaf's linear form)"""
        terms = self.p.linform(v) if const is None else None
        if terms is not None:                   # mix^e * sum_j C_j b_j = sum_j (mix^e C_j) b_j: base leaves against constant-scaled slots
            for C, b in terms:
                self.acc_leaf(d, b, e, C)
            return
        self.need([v])
        if self.p.ext[v] and signed_sums():
            # an Fp4 leaf that is not linear over constants: mix^e * x as an Fp4 product into the REDUCED total (the signed sums read centred
            # powers; this slot is a canonical one)
            k = self.pw(e)
            self.add_fp4(d, f"Fp4(Fp::raw(pwp[{k}].x), Fp::raw(pwp[{k}].y), Fp::raw(pwp[{k}].z), Fp::raw(pwp[{k}].w)) * {self.ext_ref(v)}")
            self.release()
            return
        if self.p.ext[v]:
            # tot += mix^e * x for an Fp4 x: sixteen products straight into the unreduced sums (four units of room)
            if self.pend.get(d, 0) > 0:
                self.fold(d)
            self.w(f"    ext_accumulate(s{d}_0, s{d}_1, s{d}_2, s{d}_3, pwp[{self.pw(e)}], {self.ext_ref(v)});")
            self.release()
            self.pend[d] = 4
            self.sb[d] = self.sb.get(d, 0) + 4 * (P - 1) * (P - 1)      # at most three products + (-11) x one reduced word per component
            for k in range(4):
                self.claim(f"s{d}_{k}", self.sb[d])
            return
        # tot += mix^e * v: four 64-bit multiply-adds (scalar-loaded power words); the sums are folded (not reduced)
        # when four units of products are pending and reduced once where the total is needed
        r = self.ref(v)
        if signed_sums():
            # tot += centred(mix^e) * (v as int32): one unit whether v is lazy (v - P in [-P, P)) or canonical
            if self.pend.get(d, 0) + 1 > 4:
                self.fold(d)
            rs = f"(int32_t)({r} - {P}u)" if (v in self.p.lazy and self.p.fp[v][0] != OP_CONST) else f"(int32_t){r}"
            # (plain C, not the pinned v_mad_i64_i32 of fp.h: hipcc picks that instruction here by itself, and an asm operand tied to an
            # SGPR keeps it from merging the scalar loads of the powers — s_load 313 -> 188 per part, 12 VGPRs fewer)
            self.w(f"    {{ const uint4 p_ = pwp[{self.pw(e, const, centered=True)}]; const int32_t r_ = {rs}; "
                   f"s{d}_0 += (int64_t)r_ * (int64_t)(int32_t)p_.x; s{d}_1 += (int64_t)r_ * (int64_t)(int32_t)p_.y; "
                   f"s{d}_2 += (int64_t)r_ * (int64_t)(int32_t)p_.z; s{d}_3 += (int64_t)r_ * (int64_t)(int32_t)p_.w; }}")
            self.release()
            self.pend[d] = self.pend.get(d, 0) + 1
            self.sb[d] = self.sb.get(d, 0) + ((P - 1) // 2) * (P if v in self.p.lazy else self.hi_of(v))
            self.claim(f"s{d}_0", self.sb[d])
            return
        wgt = 2 if v in self.p.lazy else 1      # a lazy value (< 2P) makes a product below 2 P^2
        if self.pend.get(d, 0) + wgt > 4:
            self.fold(d)
        self.w(f"    {{ const uint4 p_ = pwp[{self.pw(e, const)}]; s{d}_0 += (uint64_t)p_.x * {r}; s{d}_1 += (uint64_t)p_.y * {r}; "
               f"s{d}_2 += (uint64_t)p_.z * {r}; s{d}_3 += (uint64_t)p_.w * {r}; }}")
        self.release()
        self.pend[d] = self.pend.get(d, 0) + wgt
        self.sb[d] = self.sb.get(d, 0) + (P - 1) * self.hi_of(v)
        self.claim(f"s{d}_0", self.sb[d])

    # ---- the mix tree ----
    def emit_chain(self, m: int, d: int) -> bool:
        """Accumulate into depth d the leaves of chain m that fall into [lo, hi).  Returns whether anything was emitted."""
        any_emitted = False
        for it in self.p.chain(m):
            if it[0] == "e":
                idx = self.leaf
                self.leaf += 1
                if not (self.lo <= idx < self.hi):
                    continue
                _, v, e = it
                any_emitted = True
                self.acc_leaf(d, v, e)
            elif it[0] == "g":
                # f * sum_i mix^e_i q_i (Plan.group_by_factor): the members accumulate one level down with their OWN exponents, the
                # factor multiplies the reduced level total straight into this level's unreduced sums (four multiply-adds)
                _, f, members = it
                first = self.leaf
                if first + len(members) <= self.lo or first >= self.hi:
                    self.leaf += len(members)
                    continue
                self.reset(d + 1)
                got = False
                for q, e in members:
                    idx = self.leaf
                    self.leaf += 1
                    if self.lo <= idx < self.hi:
                        self.acc_leaf(d + 1, q, e)
                        got = True
                if not got:
                    continue
                self.flush(d + 1)
                self.need([f])
                any_emitted = True
                r = self.ref(f)
                if signed_sums():
                    # the level total is a canonical word (< P: a positive int32), the factor enters like a leaf operand: |t f| <= (P-1) P, two units
                    if self.pend.get(d, 0) + 2 > 4:
                        self.fold(d)
                    rs = f"(int32_t)({r} - {P}u)" if (f in self.p.lazy and self.p.fp[f][0] != OP_CONST) else f"(int32_t){r}"
                    self.w(f"    {{ const int32_t f_ = {rs}; s{d}_0 += (int64_t)(int32_t)t{d + 1}_0 * (int64_t)f_; s{d}_1 += (int64_t)(int32_t)t{d + 1}_1 * (int64_t)f_; "
                           f"s{d}_2 += (int64_t)(int32_t)t{d + 1}_2 * (int64_t)f_; s{d}_3 += (int64_t)(int32_t)t{d + 1}_3 * (int64_t)f_; }}")
                    self.release()
                    self.pend[d] = self.pend.get(d, 0) + 2
                    self.sb[d] = self.sb.get(d, 0) + (P - 1) * (P if f in self.p.lazy else self.hi_of(f))
                    self.claim(f"s{d}_0", self.sb[d])
                    continue
                wgt = 2 if f in self.p.lazy else 1
                if self.pend.get(d, 0) + wgt > 4:
                    self.fold(d)
                self.w(f"    s{d}_0 += (uint64_t)t{d + 1}_0 * {r}; s{d}_1 += (uint64_t)t{d + 1}_1 * {r}; "
                       f"s{d}_2 += (uint64_t)t{d + 1}_2 * {r}; s{d}_3 += (uint64_t)t{d + 1}_3 * {r};")
                self.release()
                self.pend[d] = self.pend.get(d, 0) + wgt
                self.sb[d] = self.sb.get(d, 0) + (P - 1) * self.hi_of(f)
                self.claim(f"s{d}_0", self.sb[d])
            else:
                _, cond, inner, e = it
                first = self.leaf
                n_in = self.p.n_leaves[inner]
                if first + n_in <= self.lo or first >= self.hi or n_in == 0:
                    self.leaf += n_in
                    continue
                self.reset(d + 1)
                got = self.emit_chain(inner, d + 1)
                if not got:
                    continue
                self.flush(d + 1)
                self.need([cond])
                any_emitted = True
                tin = f"Fp4(Fp::raw(t{d + 1}_0), Fp::raw(t{d + 1}_1), Fp::raw(t{d + 1}_2), Fp::raw(t{d + 1}_3))"
                prod = f"{tin} * {self.ext_ref(cond)}" if self.p.ext[cond] else f"{tin} * Fp::raw({self.ref(cond)})"
                if e != 0:
                    k = self.pw(e)
                    prod = f"({prod}) * Fp4(Fp::raw(pwp[{k}].x), Fp::raw(pwp[{k}].y), Fp::raw(pwp[{k}].z), Fp::raw(pwp[{k}].w))"
                self.add_fp4(d, prod)
                self.release()
        return any_emitted


def emit_part(kernel: str, plan: Plan, lo: int, hi: int, standalone: bool, header: str) -> str:
    # dry run: which root is requested when -> for every value in those roots' cones, the requests that need it
    dry = _Emitter(plan, lo, hi)
    dry.use_depth(0)
    dry.pend[0], dry.tzero[0] = 0, True
    dry.emit_chain(plan.c.ret, 0)
    future: Dict[int, List[int]] = {}
    cone_cache: Dict[int, List[int]] = {}
    for req, r in enumerate(dry.requests, start=1):
        if r not in cone_cache:
            cone_cache[r] = plan.cone(r, ())
        for x in cone_cache[r]:
            lst = future.setdefault(x, [])
            if not lst or lst[-1] != req:
                lst.append(req)
    em = _Emitter(plan, lo, hi, future)
    em.upcoming = dry.requests
    em.use_depth(0)
    em.pend[0], em.tzero[0] = 0, True
    em.sb[0] = 0
    em.emit_chain(plan.c.ret, 0)
    em.flush(0)
    body = []
    backs_used = set()
    for ln in em.lines:
        if ln.startswith("    OPAQUE_EPOCH_"):
            k = int(ln.rsplit("_", 1)[1])
            bks = sorted(em.epoch_backs[k])
            backs_used |= set(bks)
            decl = " ".join(f"uint32_t o{bk}_{k} = b{bk};" for bk in bks)
            outs = ", ".join(f'"+v"(o{bk}_{k})' for bk in bks)
            # the domain size gets its own statement: sharing one with the lane offsets makes LLVM treat it as divergent,
            # and every column base (col * dom) is then computed in the VALU and moved back with two v_readfirstlane
            body.append(f"    {decl} uint32_t dw{k} = a.dom; asm(\"; epoch {k}\" : {outs}); asm(\"; epoch {k} dom\" : \"+s\"(dw{k}));")
        else:
            body.append(ln)
    L: List[str] = []
    w = L.append
    w(header)
    w(f"//   {em.n_loads} tap loads, {em.n_arith} arithmetic steps emitted (register cache of {REG_BUDGET} values, offset epochs of {EPOCH_LOADS} loads)")
    linkage = 'extern "C" ' if standalone else ""
    w(f"{linkage}__global__ __launch_bounds__(256) void {kernel}(EvalCheckArgs a) {{")
    w("    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;")
    w("    if (idx >= a.dom) return;")
    w("    const uint32_t mask = a.dom - 1;")
    w("    const size_t dom = a.dom;")
    w("    const uint4* __restrict__ pwp = (const uint4*)a.mix_pows;")
    for g in range(3):
        w(f"    const uint32_t* __restrict__ g{g} = a.groups[{g}];")
    for bk in sorted(backs_used):               # byte offset of this lane's row at each back (32-bit: 4n words < 2^26)
        w(f"    const uint32_t b{bk} = " + ("idx * 4u;" if bk == 0 else f"((idx - {4 * bk}u) & mask) * 4u;"))
    for (x, y) in sorted(em.globals_used):
        w(f"    const uint32_t q{x}_{y} = a.globals[{x}][{y}];")
    for d in range(em.depth_used):
        w(f"    uint32_t t{d}_0 = 0, t{d}_1 = 0, t{d}_2 = 0, t{d}_3 = 0; {'int64_t' if signed_sums() else 'uint64_t'} s{d}_0 = 0, s{d}_1 = 0, s{d}_2 = 0, s{d}_3 = 0;")
    L.extend(body)
    w("    const uint32_t zi = a.zinv[idx & 3];")
    w("    if (a.accumulate) {")
    for kk in range(4):
        off = "" if kk == 0 else f"{kk} * dom + "
        w(f"        a.check[{off}idx] = add_mod(a.check[{off}idx], mul_mod(t0_{kk}, zi));")
    w("    } else {")
    for kk in range(4):
        off = "" if kk == 0 else f"{kk} * dom + "
        w(f"        a.check[{off}idx] = mul_mod(t0_{kk}, zi);")
    w("    }")
    w("}")
    if not standalone:
        w(f"void launch_{kernel}(const EvalCheckArgs& a, hipStream_t s) {{")
        w(f"    {kernel}<<<(a.dom + 255u) / 256u, 256, 0, s>>>(a);")
        w("}")
    if GATHER:          # the exponent behind every slot of this part's power table (first word: how many)
        words = ", ".join(str(x) for x in [len(em.pw_exps)] + em.pw_exps)
        # ... and the slots whose power is multiplied by an Fp4 constant (LINFORM): {count, then (slot, c0, c1, c2, c3) each, Montgomery words}
        cw = [len(em.pw_consts)]
        for slot, C in em.pw_consts:
            cw += [slot] + [mont(k) for k in C]
        cwords = ", ".join(str(x) for x in cw)
        if standalone:
            w(f'extern "C" __device__ __attribute__((used)) const uint32_t {kernel}_exps[] = {{{words}}};')
            w(f'extern "C" __device__ __attribute__((used)) const uint32_t {kernel}_pwc[] = {{{cwords}}};')
        else:
            w(f"extern const uint32_t exps_{kernel}[] = {{{words}}};")
            w(f"extern const uint32_t pwc_{kernel}[] = {{{cwords}}};")
    return "\n".join(L)


def part_kernel_names(name: str, n_parts: int) -> List[str]:
    return [f"k_eval_check_{name}"] if n_parts == 1 else [f"k_eval_check_{name}_p{i:02d}" for i in range(n_parts)]


def emit_parts(name: str, desc: np.ndarray, standalone: bool = False) -> Tuple[List[Tuple[str, str]], int, int]:
    """-> ([(kernel name, HIP source of that kernel)], desc hash, number of mix powers the kernels read).
    standalone=True: `extern "C"` kernels with no host-side launcher, for code objects attached at run time (circuits/jit.py)."""
    c = Circuit.parse(desc)
    plan = Plan.build(c)
    cuts = split_points(plan.leaf_weights())
    names = part_kernel_names(name, len(cuts))
    out = []
    for k, (lo, hi) in zip(names, cuts):
        header = (f"// {name}: groups (accum, code, data) = {c.group_sizes}, {len(c.taps)} taps, {len(c.steps)} steps "
                  f"({plan.n_unique} distinct arithmetic values after value numbering); constraints [{lo}, {hi}) of {plan.n_leaves[c.ret]}")
        out.append((k, emit_part(k, plan, lo, hi, standalone, header)))
    return out, desc_hash64(desc), plan.n_pows


def emit_kernel(name: str, desc: np.ndarray, standalone: bool = False) -> Tuple[str, int, int]:
    """All kernels of a circuit as one source text (small circuits: exactly one kernel `k_eval_check_<name>`)."""
    parts, h, n_pows = emit_parts(name, desc, standalone)
    return "\n\n".join(src for _, src in parts), h, n_pows


# -------------------------------------------------------------------------------------------------------------------
# Shipped circuits: kernels compiled into libzkhal_mi355x.so
# -------------------------------------------------------------------------------------------------------------------
SHIPPED: Dict[str, np.ndarray] = {}


def shipped() -> Dict[str, np.ndarray]:
    if not SHIPPED:
        from . import syn_heavy
        SHIPPED["syn_a"] = syn_air.syn_a()
        SHIPPED["syn_small"] = syn_air.syn_small()
        SHIPPED["syn_tiny"] = syn_air.syn_tiny()
        SHIPPED["syn_join"] = syn_air.syn_join()
        SHIPPED["syn_chain"] = syn_air.syn_chain()
        SHIPPED["syn_session"] = syn_air.syn_session()
        SHIPPED["syn_heavy"] = syn_heavy.syn_heavy()
        from . import keccak_f
        SHIPPED["keccak_f"] = keccak_f.keccak_f_circuit()
        from . import p2_join
        SHIPPED["p2_join"] = p2_join.p2_join_circuit()
        from . import recursion
        SHIPPED["recursion"] = recursion.recursion_circuit()
    return SHIPPED


PREAMBLE = ["// GENERATED by zeth_amd/circuits/codegen.py — do not edit.  Straight-line eval_check kernels (gfx950) for the",
            "// shipped circuit descriptions; selected at run time by desc hash (circuit.hip).",
            '#include "circuit.h"', "", "using namespace zkh;", ""]


def generate_sources() -> Dict[str, str]:
    """-> {file name: source}: `eval_check_gen.hip` (registry + the single-kernel circuits) and one
    `eval_check_gen_<circuit>_pNN.hip` per part of a split circuit, so that the build compiles the parts in parallel."""
    files: Dict[str, str] = {}
    main = list(PREAMBLE)
    table, externs = [], []
    for name, desc in shipped().items():
        parts, h, n_pows = emit_parts(name, desc)
        launchers = []
        if len(parts) == 1:
            main += [parts[0][1], ""]
            launchers.append(f"launch_{parts[0][0]}")
        else:
            for k, src in parts:
                files[f"eval_check_gen_{name}_{k.rsplit('_', 1)[1]}.hip"] = "\n".join(PREAMBLE + [src, ""])
                externs.append(f"void launch_{k}(const EvalCheckArgs&, hipStream_t);")
                launchers.append(f"launch_{k}")
        main.append(f"static const eval_check_launch_fn parts_{name}[] = {{{', '.join(launchers)}}};")
        if GATHER:
            if len(parts) > 1:
                externs.extend(f"extern const uint32_t exps_{k}[];" for k, _ in parts)
                externs.extend(f"extern const uint32_t pwc_{k}[];" for k, _ in parts)
            main.append(f"static const uint32_t* const gexps_{name}[] = {{{', '.join('exps_' + k for k, _ in parts)}}};")
            main.append(f"static const uint32_t* const gpwc_{name}[] = {{{', '.join('pwc_' + k for k, _ in parts)}}};")
            gather = f"gexps_{name}, gpwc_{name}"
        else:
            gather = "nullptr, nullptr"
        table.append(f'    {{0x{h:016x}ull, "{name}", parts_{name}, {len(launchers)}u, {n_pows}u, {gather}}},')
    # the extern declarations must precede the arrays that reference them
    src = "\n".join(PREAMBLE + externs + [""] + main[len(PREAMBLE):] + ["", "static const CompiledEvalCheck k_table[] = {", *table, "};", "",
                    "namespace zkh {", "const CompiledEvalCheck* find_compiled_eval_check(uint64_t h) {",
                    "    for (const auto& e : k_table) if (e.desc_hash == h) return &e;", "    return nullptr;", "}",
                    "}  // namespace zkh", ""])
    files["eval_check_gen.hip"] = src
    return files


def write_generated(csrc_dir: str) -> List[str]:
    """Write the generated translation units into csrc_dir (only files whose content changed are touched) and remove stale
    generated files.  Returns the file names, `eval_check_gen.hip` first."""
    import os
    files = generate_sources()
    for fn in os.listdir(csrc_dir):
        if fn.startswith("eval_check_gen") and fn.endswith(".hip") and fn not in files:
            os.remove(os.path.join(csrc_dir, fn))
    for fn, src in files.items():
        path = os.path.join(csrc_dir, fn)
        try:
            with open(path) as fh:
                if fh.read() == src:
                    continue
        except FileNotFoundError:
            pass
        with open(path, "w") as fh:
            fh.write(src)
    return ["eval_check_gen.hip"] + sorted(f for f in files if f != "eval_check_gen.hip")
