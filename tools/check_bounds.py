#!/usr/bin/env python3
"""Exact integer bound verifier for the GENERATED eval_check kernels (zeth_amd/circuits/codegen.py).

The generator keeps values in lazy representatives ([0, 2P) instead of [0, P)), sums products in 64-bit accumulators that are folded
or reduced "when there is no more room", and decides all of that with its own accounting (Plan.find_lazy, find_sums_of_products,
_Emitter.pend / fold / flush).  A mistake there does not crash anything: a sum wraps past 2^64 for SOME inputs and `check` is wrong —
round 5 shipped exactly that once (lazy Fp4 operands under the factor grouping), 197 random circuits did not notice, one po2-20
stage digest did.  This tool is the net under that accounting.  It shares NO code with it: it reads only the EMITTED SOURCE TEXT
(what hipcc compiles) and the contracts of the primitives the text calls (csrc/fp.h, csrc/circuit.h, restated below with the
preconditions their comments give), and runs the straight-line code once over exact Python integer INTERVALS:

  * every tap word, global, mix-power word, zinv and check word enters as [0, P-1] (canonical Montgomery words: the inputs' contract);
  * every 32-bit expression must end inside [0, 2^32), every 64-bit one inside [0, 2^64) — wherever a value is stored, cast,
    multiplied or passed on (intermediate wrap-around of a u32 `a - b + P` is modular arithmetic and allowed, the END value is not);
  * every primitive's precondition is checked with the worst-case operand: mont_reduce(t) needs t < P 2^32, mont_reduce_wide
    t < 2 P 2^32, add_mod a + b < 2P, sub_mod canonical operands, mul_mod / mul_lazy a b < P 2^32, Fp4 x Fp4 and ext_accumulate the
    sums inside them; the result interval is the primitive's exact worst case (mont_reduce_lazy: (t + (2^32 - 1) P) >> 32);
  * a `// BOUND name <= N` comment is the generator's own claim (its bounds trace): the derived worst case must not exceed it.

Worst-case intervals are sound for straight-line code over non-negative integers (every operation here is monotone in each operand),
so "no violation" means: for EVERY input that honours the input contract no accumulator wraps and no reduction sees an operand
outside its domain.  Exit code 1 and the offending source line otherwise.

    python tools/check_bounds.py                       # every shipped circuit (codegen.shipped())
    python tools/check_bounds.py --circuit syn_heavy
    python tools/check_bounds.py --fuzz 197            # the random constraint systems of tests/test_fuzz_gpu.py
    python tools/check_bounds.py --files zeth_amd/csrc/eval_check_gen*.hip     # the generated translation units on disk
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Tuple

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeth_amd.circuits.bounds import *          # noqa: E402,F401,F403  (the checker itself: zeth_amd/circuits/bounds.py)
from zeth_amd.circuits.bounds import P, Violation, check_source, execute_source, kernel_tables  # noqa: E402,F401


def check_desc(name: str, desc) -> Tuple[List[str], dict]:
    """the kernels the generator emits for one circuit description (what build.py / jit.py compile)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from zeth_amd.circuits import codegen
    parts, _, _ = codegen.emit_parts(name, desc)
    violations: List[str] = []
    total = {"kernels": 0, "statements": 0, "reductions": 0, "claims": 0, "max_acc_bits": 0.0}
    for _, src in parts:
        v, st = check_source(src, name)
        violations += v
        for k in ("kernels", "statements", "reductions", "claims"):
            total[k] += st[k]
        total["max_acc_bits"] = max(total["max_acc_bits"], st["max_acc_bits"])
    return violations, total


def fuzz_descs(count: int, first: int = 0):
    """the random constraint systems tests/test_fuzz_gpu.py runs through three evaluators (same generator arguments), and more"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from zeth_amd.circuits import syn_random
    for seed in range(first, first + count):
        groups = [(4, 6, 12), (8, 5, 20), (4, 16, 33)][seed % 3]
        yield f"fuzz{seed}", syn_random.random_circuit(seed, groups=groups, n_values=160 + 40 * (seed % 4), n_constraints=30 + 15 * (seed % 3),
                                                       max_back=1 + seed % 4)


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--circuit", action="append", help="a shipped circuit by name (default: all of codegen.shipped())")
    ap.add_argument("--fuzz", type=int, default=0, help="also the first N random constraint systems")
    ap.add_argument("--fuzz-from", type=int, default=0, help="the first random seed of --fuzz (a soak over seeds the test suite does not use)")
    ap.add_argument("--files", nargs="*", help="generated translation units on disk instead of running the generator")
    a = ap.parse_args()
    bad: List[str] = []
    if a.files:
        for f in a.files:
            v, st = check_source(open(f).read(), os.path.basename(f))
            print(f"{os.path.basename(f):44s} kernels {st['kernels']:3d}  statements {st['statements']:7d}  reductions {st['reductions']:7d}  "
                  f"largest 64-bit sum 2^{st['max_acc_bits']:.3f}  {'OK' if not v else 'VIOLATIONS: ' + str(len(v))}")
            bad += v
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from zeth_amd.circuits import codegen
        todo = [(n, d) for n, d in codegen.shipped().items() if not a.circuit or n in a.circuit]
        todo += list(fuzz_descs(a.fuzz, a.fuzz_from))
        for name, desc in todo:
            v, st = check_desc(name, desc)
            print(f"{name:14s} kernels {st['kernels']:3d}  statements {st['statements']:7d}  reductions {st['reductions']:7d}  claims {st['claims']:6d}  "
                  f"largest 64-bit sum 2^{st['max_acc_bits']:.3f}  {'OK' if not v else 'VIOLATIONS: ' + str(len(v))}")
            bad += v
    for v in bad[:40]:
        print("VIOLATION", v)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
