"""Random constraint systems for cross-checking the three evaluators of a circuit description (generated kernels,
on-device step interpreter, the oracle's literal interpreter).  Not a provable circuit: nothing makes the constraints
vanish on a witness — `eval_check` is a pure function of (description, evaluated groups, globals, mix), which is what is
compared.  The shapes are chosen to reach the generator's corner cases: squares and repeated factors, long add/sub chains
(sums of products with negative terms), values used both as factors and as addends (canonical vs lazy representatives),
base-field values flowing into Fp4 expressions from either side, Fp4-valued constraints, nested AndCond with base and
Fp4 conditions, constraints on bare taps / constants / globals, runs of constraints sharing one factor."""
from __future__ import annotations

import numpy as np

from .desc import GLOBAL_MIX, GLOBAL_OUT, P, CircuitBuilder


def random_circuit(seed: int, groups=(4, 6, 12), n_values: int = 220, n_constraints: int = 60, max_back: int = 3) -> np.ndarray:
    rng = np.random.default_rng(seed)
    b = CircuitBuilder(tuple(groups), (4, groups[0]), kind=0)
    base, ext = [], []
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 11, P - 11]

    def leaf():
        k = int(rng.integers(0, 10))
        if k < 7:
            g = int(rng.integers(0, 3))
            return b.get(g, int(rng.integers(0, groups[g])), int(rng.integers(0, max_back + 1)))
        if k < 9:
            return b.const(int(edge[rng.integers(0, len(edge))]) if rng.integers(0, 2) else int(rng.integers(0, P)))
        base_id = GLOBAL_OUT if rng.integers(0, 2) else GLOBAL_MIX
        return b.get_global(base_id, int(rng.integers(0, 4 if base_id == GLOBAL_OUT else groups[0])))

    def pick(pool):
        # recent values more often (chains), old ones sometimes (sharing)
        if not pool or rng.integers(0, 6) == 0:
            return leaf()
        n = len(pool)
        i = n - 1 - int(min(n - 1, rng.geometric(0.15) - 1)) if rng.integers(0, 3) else int(rng.integers(0, n))
        return pool[i]

    for _ in range(n_values):
        k = int(rng.integers(0, 100))
        if k < 8 and base:                                   # square / cube of one value
            x = pick(base)
            v = b.mul(x, x)
            if rng.integers(0, 2):
                v = b.mul(v, x)
            base.append(v)
        elif k < 40:
            base.append(b.mul(pick(base), pick(base)))
        elif k < 60:
            base.append(b.add(pick(base), pick(base)))
        elif k < 80:
            base.append(b.sub(pick(base), pick(base)))
        elif k < 86:                                         # a longer +/- chain of products (one sum of products)
            acc = b.mul(pick(base), pick(base))
            for _ in range(int(rng.integers(2, 7))):
                t = b.mul(pick(base), pick(base)) if rng.integers(0, 3) else pick(base)
                acc = b.add(acc, t) if rng.integers(0, 2) else b.sub(acc, t)
            base.append(acc)
        elif k < 90 or not ext:
            c = b.const_ext(*(int(rng.integers(0, P)) for _ in range(4)))
            x = pick(base)
            ext.append([b.mul(c, x), b.mul(x, c), b.add(c, x), b.add(x, c), b.sub(c, x), b.sub(x, c)][int(rng.integers(0, 6))])
        elif k < 94:                                         # Fp4 * Fp * Fp ...: products whose only consumer multiplies again
            x = ext[int(rng.integers(0, len(ext)))]
            for _ in range(int(rng.integers(2, 5))):
                y = pick(base)
                x = b.mul(x, y) if rng.integers(0, 2) else b.mul(y, x)
            ext.append(x)
        else:
            x = ext[int(rng.integers(0, len(ext)))]
            y = ext[int(rng.integers(0, len(ext)))] if rng.integers(0, 2) else pick(base)
            op = [b.mul, b.add, b.sub][int(rng.integers(0, 3))]
            ext.append(op(x, y) if rng.integers(0, 2) else op(y, x))

    def constraint_value():
        k = int(rng.integers(0, 10))
        if k < 6 and base:
            return pick(base)
        if k < 8 and ext:
            return ext[int(rng.integers(0, len(ext)))]
        return leaf()

    def chain(n: int, depth: int):
        m = b.true()
        for _ in range(n):
            if rng.integers(0, 7) == 0 and base:
                # a COMPONENT: several constraints that share one factor (a selector / vanishing term), sometimes with an Fp4 member —
                # what the generator's factor grouping (codegen.py Plan.group_by_factor) pulls out of the sum
                f = pick(base)
                for _ in range(int(rng.integers(3, 7))):
                    q = ext[int(rng.integers(0, len(ext)))] if (ext and rng.integers(0, 5) == 0) else pick(base)
                    m = b.and_eqz(m, b.mul(f, q) if rng.integers(0, 2) else b.mul(q, f))
            elif depth < 3 and rng.integers(0, 5) == 0:
                inner = chain(int(rng.integers(0, 5)), depth + 1)      # may be empty
                cond = ext[int(rng.integers(0, len(ext)))] if (ext and rng.integers(0, 4) == 0) else pick(base)
                m = b.and_cond(m, cond, inner)
            else:
                m = b.and_eqz(m, constraint_value())
        return m

    return b.finish(chain(n_constraints, 0))
