"""SYN-HEAVY: SYN-A's trace shape with a constraint system of realistic WEIGHT.

SYN-A's polynomial is ~2.6 k steps over 289 taps: its eval_check is an HBM-bound 1.1 ms of a 34 ms seal, whereas upstream
ranks the real rv32im `eval_check` (O(10^4 - 10^5) field operations per domain point, thousands of taps reused across
constraints, VGPR-bound generated code split over many translation units) as the largest kernel of the prover
(SURVEY.md §3.2 HOT LOOP B; [UP risc0-circuit-rv32im 4.0.2 > eval_check], un-vendored: /root/reference/Cargo.lock:5320).
SYN-HEAVY keeps SYN-A's columns and witness (W_code 16, W_data 208, W_accum 32; same witgen kernels) and adds what the
real thing has and SYN-A lacks:

  * ~1.1 k taps: data columns are read at backs drawn from {0..4} in several patterns, code columns at {0..3}, accum
    columns at {0,1,2}: 7 distinct back-sets = 7 tap combos for DEEP / FRI batching (SYN-A: 2);
  * ~50 k PolyExtSteps, degree 5: ~3 k constraints  active * Z_j * Q  where Z_j = d[3j] d[3j+1] - d[3j+2] vanishes on
    every active row by construction of the SYN witness and Q is a pseudo-random degree-2 expression (~12 steps) over
    taps at backs 0..4, drawn mostly from the ~30 registers of the constraint's own component (locality, as in a real
    circuit) with sub-terms shared between constraints (the generator's value numbering finds them) — the seal's check
    polynomial still depends on every coefficient of Q at the random point z;
  * nested conditionals (AndCond inside AndCond) and ConstExt operands (extension-field valued sub-expressions).

Everything stays data: the same desc blob format, consumed by the oracle, the interpreter and the code generator.
"""
from __future__ import annotations

import numpy as np

from .desc import GLOBAL_MIX, GLOBAL_OUT, GROUP_ACCUM, GROUP_CODE, GROUP_DATA, P, CircuitBuilder

NBETA = P - 11
# back-sets of the data columns (column c uses DATA_BACKS[c % len]); the running-sum column keeps {0, 1}
_FULL = (0, 1, 2, 3, 4)
DATA_BACKS = [_FULL, _FULL, (0, 1, 2), _FULL, _FULL, (0, 2, 4), _FULL, _FULL, (0, 4), _FULL, _FULL, _FULL]


def _rng(seed: int):
    """Small deterministic generator (the circuit is a fixed artefact: same blob on every machine)."""
    state = [seed & 0xFFFFFFFFFFFFFFFF]

    def nxt(mod: int) -> int:
        state[0] = (state[0] + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state[0]
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) % mod
    return nxt


# SYN-HUGE (round 6): the same generator at a real circuit's SCALE — backs 0 .. 7 on every column, ~2 k taps, ~14 k constraints,
# > 250 k PolyExtSteps — so that the code generator, hipcc and the code-object loader meet the size upstream's rv32im `eval_check`
# has (tens of generated translation units) before the real tables are ever supplied.
_FULL8 = (0, 1, 2, 3, 4, 5, 6, 7)
DATA_BACKS_HUGE = [_FULL8, _FULL8, _FULL8, _FULL8, _FULL8, (0, 2, 3, 4, 5, 6, 7), _FULL8, _FULL8, (0, 1, 2, 3, 4, 5, 6), _FULL8, _FULL8, _FULL8]


def build_syn_heavy(wc: int = 16, wd: int = 208, wa: int = 32, per_triple: int = 44, seed: int = 0x48454156, huge: bool = False) -> np.ndarray:
    """per_triple heavy constraints for each of the T = (wd-2)//3 multiplicative triples (68 x 44 ~ 3 k constraints)."""
    assert wc >= 5 and wd >= 8 and wa >= 4 and wa % 4 == 0
    data_backs = DATA_BACKS_HUGE if huge else DATA_BACKS
    code_backs, accum_backs = ((0, 1, 2, 3, 4, 5, 6, 7), (0, 1, 2, 3, 4, 5, 6, 7)) if huge else ((0, 1, 2, 3), (0, 1, 2))
    b = CircuitBuilder((wa, wc, wd), (4, wa))
    rnd = _rng(seed)
    code = lambda c, back=0: b.get(GROUP_CODE, c, back)
    data = lambda c, back=0: b.get(GROUP_DATA, c, back)
    acc = lambda c, back=0: b.get(GROUP_ACCUM, c, back)
    one = b.const(1)
    nbeta = b.const(NBETA)
    active, first, body, rowidx, last = (code(i) for i in range(5))
    T = (wd - 2) // 3
    s_col = wd - 1

    # every (column, back) pair the circuit is allowed to read; handed out round-robin so that each becomes a tap
    tap_pool = [(GROUP_DATA, c, bk) for c in range(wd - 1) for bk in data_backs[c % len(data_backs)]]
    tap_pool += [(GROUP_CODE, c, bk) for c in range(5, wc) for bk in code_backs]
    if huge:                                             # the selector columns at earlier rows too
        tap_pool += [(GROUP_CODE, c, bk) for c in range(5) for bk in code_backs[1:]]
    tap_pool += [(GROUP_ACCUM, c, bk) for c in range(wa) for bk in accum_backs]
    pool_pos = [0]

    # Locality, as in a real circuit: the constraints of one component (here: one triple) read a small set of related
    # registers over and over — LOCAL taps, handed out round-robin so that every (column, back) pair of the pool becomes a
    # tap — and only now and then something from elsewhere in the trace.
    local: list = []

    def new_local_pool(size: int = 32 if huge else 28):
        local.clear()
        for _ in range(size):
            local.append(tap_pool[pool_pos[0] % len(tap_pool)])
            pool_pos[0] += 1
        local.append(tap_pool[rnd(len(tap_pool))])
        local.append(tap_pool[rnd(len(tap_pool))])

    next_pos = [0]

    def next_tap():
        g, c, bk = local[next_pos[0] % len(local)]
        next_pos[0] += 1
        return b.get(g, c, bk)

    def rand_tap():
        g, c, bk = tap_pool[rnd(len(tap_pool))] if rnd(16) == 0 else local[rnd(len(local))]
        return b.get(g, c, bk)

    def linear(k: int):
        """c0 + sum_i c_i * tap_i over k taps (degree 1); written the way a circuit compiler would emit it."""
        e = b.const(1 + rnd(1 << 20))
        for i in range(k):
            t = next_tap() if i == 0 else rand_tap()
            e = b.add(e, b.mul(b.const(2 + rnd(1 << 16)), t))
        return e

    # (A) triples + degree-4 product + the heavy constraints, gated by `active`
    inner = b.true()
    for j in range(T):
        x, y, p = data(3 * j), data(3 * j + 1), data(3 * j + 2)
        inner = b.and_eqz(inner, b.sub(b.mul(x, y), p))
        new_local_pool()
        # sub-terms shared by the constraints of this triple (re-emitted every time: the generator's CSE finds them)
        shared = [(1 + rnd(1 << 12), rnd(len(local)), rnd(len(local))) for _ in range(6)]

        def shared_term(i):
            c, ta, tb = shared[i]
            ga, ca, ba = local[ta]
            gb, cb, bb = local[tb]
            return b.add(b.mul(b.const(c), b.get(ga, ca, ba)), b.get(gb, cb, bb))      # degree 1

        nested = b.true()
        for k in range(per_triple):
            z = b.sub(b.mul(data(3 * j), data(3 * j + 1)), data(3 * j + 2))            # Z_j, recomputed (degree 2)
            kind = rnd(8)
            if kind < 5:        # Q = L1 * L2 + s  (degree 2)
                q = b.add(b.mul(linear(2), b.add(shared_term(rnd(6)), rand_tap())), shared_term(rnd(6)))
                inner = b.and_eqz(inner, b.mul(z, q))                                   # active * Z * Q: degree 5
            elif kind < 7:      # nested conditional: active * cond * (Z * L): cond a data tap, L degree 1
                q = b.add(linear(3), shared_term(rnd(6)))
                nested = b.and_eqz(nested, b.mul(z, q))                                 # degree 3 (+1 cond +1 active)
            else:               # ConstExt operand: Q = (L + ext) * s  -> Fp4-valued constraint
                ext = b.const_ext(1 + rnd(P - 1), rnd(P), rnd(P), 1 + rnd(P - 1))
                q = b.mul(b.add(linear(2), ext), shared_term(rnd(6)))
                inner = b.and_eqz(inner, b.mul(z, q))
        cond = rand_tap()
        inner = b.and_cond(inner, cond, nested)
    prod4 = b.mul(b.mul(data(0), data(1)), b.mul(data(3), data(4)))
    inner = b.and_eqz(inner, b.sub(prod4, data(wd - 2)))
    chain = b.and_cond(b.true(), active, inner)

    def term(e):          # mix_e + d (as Fp4 components)
        m = [b.get_global(GLOBAL_MIX, 4 * e + i) for i in range(4)]
        return [b.add(m[0], data(e % wd)), m[1], m[2], m[3]]

    def ext_mul(x, y):    # Fp4 product written out in Fp steps (x^4 = -11)
        m = lambda i, j: b.mul(x[i], y[j])
        c0 = b.add(m(0, 0), b.mul(nbeta, b.add(b.add(m(1, 3), m(2, 2)), m(3, 1))))
        c1 = b.add(b.add(m(0, 1), m(1, 0)), b.mul(nbeta, b.add(m(2, 3), m(3, 2))))
        c2 = b.add(b.add(b.add(m(0, 2), m(1, 1)), m(2, 0)), b.mul(nbeta, m(3, 3)))
        c3 = b.add(b.add(m(0, 3), m(1, 2)), b.add(m(2, 1), m(3, 0)))
        return [c0, c1, c2, c3]

    # (first) s = d0 ; a_e = term_e
    inner = b.and_eqz(b.true(), b.sub(data(s_col), data(0)))
    for e in range(wa // 4):
        t = term(e)
        for i in range(4):
            inner = b.and_eqz(inner, b.sub(acc(4 * e + i), t[i]))
    chain = b.and_cond(chain, first, inner)

    # (body) s = s@1 + d0 + row*d1 ; a_e = a_e@1 * term_e
    rhs = b.add(b.add(data(s_col, 1), data(0)), b.mul(rowidx, data(1)))
    inner = b.and_eqz(b.true(), b.sub(data(s_col), rhs))
    for e in range(wa // 4):
        prev = [acc(4 * e + i, 1) for i in range(4)]
        pr = ext_mul(prev, term(e))
        for i in range(4):
            inner = b.and_eqz(inner, b.sub(acc(4 * e + i), pr[i]))
    chain = b.and_cond(chain, body, inner)

    # (last) s = out[0]
    inner = b.and_eqz(b.true(), b.sub(data(s_col), b.get_global(GLOBAL_OUT, 0)))
    chain = b.and_cond(chain, last, inner)

    # selector sanity (ungated)
    chain = b.and_eqz(chain, b.mul(active, b.sub(one, active)))
    chain = b.and_eqz(chain, b.mul(first, b.sub(one, first)))
    chain = b.and_eqz(chain, b.sub(b.sub(active, first), body))
    return b.finish(chain)


_CACHE = {}


def syn_heavy() -> np.ndarray:
    """SYN-HEAVY at SYN-A's widths (the bench's `--circuit syn_heavy`)."""
    if "heavy" not in _CACHE:
        _CACHE["heavy"] = build_syn_heavy()
    return _CACHE["heavy"]


def syn_huge() -> np.ndarray:
    """SYN-HUGE: SYN-A's trace under a constraint system of a real circuit's size (> 250 k steps, > 2 k taps at backs 0 .. 7,
    > 14 k degree-5 constraints).  Not compiled into the library: it is DATA, loaded like any circuit the library has never seen
    (generated + compiled at load time, parts in parallel: circuits/jit.py) — `tests/soak/syn_huge_report.py` measures that path."""
    if "huge" not in _CACHE:
        _CACHE["huge"] = build_syn_heavy(per_triple=218, seed=0x48554745, huge=True)
    return _CACHE["huge"]


def syn_heavy_small() -> np.ndarray:
    """A small instance for byte-exact tests against the oracle (same generator, fewer columns and constraints)."""
    if "small" not in _CACHE:
        _CACHE["small"] = build_syn_heavy(8, 32, 8, per_triple=24, seed=0x534D4C)
    return _CACHE["small"]


if __name__ == "__main__":
    import sys
    from .desc import Circuit
    for name, d in (("syn_heavy", syn_heavy()), ("syn_heavy_small", syn_heavy_small()), ("syn_huge", syn_huge())):
        c = Circuit.parse(d)
        print(name, "groups", c.group_sizes, "taps", len(c.taps), "combos", c.combos, "steps", len(c.steps))
    if len(sys.argv) > 1:
        np.asarray(syn_heavy(), dtype="<u4").tofile(sys.argv[1])
