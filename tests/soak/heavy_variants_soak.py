#!/usr/bin/env python3
"""Soak of the eval_check generator over SYN-HEAVY-like circuits it has never seen: other seeds, widths, sizes and tap depths of
circuits/syn_heavy.py's builder (nested conditions, shared factors, Fp4 constraints linear over constants — the structures the
factor grouping, LINFORM and the signed sums of DESIGN.md §4b act on).  The shipped circuits are three points of that family.

  --cpu   per variant: the exact bound verifier over the emitted kernels (tools/check_bounds.py) and the emitted text EXECUTED on
          the CPU against the oracle's literal interpreter on random / all-maximal / half-range inputs (no GPU, no hipcc)
  (GPU)   per variant: generated (hipcc, attached at load time) == on-device interpreter == oracle on five input families at po2 6,
          and one po2-9 seal byte for byte against the oracle's

    python tests/soak/heavy_variants_soak.py --cpu --first 0 --count 40
    python tests/soak/heavy_variants_soak.py --first 100 --count 25          # on an MI355X
One JSON line at the end; a problem prints its variant's parameters (they reproduce it) and the exit code is 1.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
P = 2013265921


def variant(rng, s: int):
    from zeth_amd.circuits import syn_heavy
    wd = int(rng.choice([32, 64, 104, 208]))
    wc = int(rng.choice([8, 16]))
    wa = int(rng.choice([8, 16, 32]))
    per = int(rng.integers(8, 70))
    huge = bool(rng.integers(0, 2))
    return dict(wc=wc, wd=wd, wa=wa, per_triple=per, seed=0x1000 + s, huge=huge), syn_heavy.build_syn_heavy(wc, wd, wa, per_triple=per, seed=0x1000 + s, huge=huge)


def cpu_leg(first: int, count: int) -> dict:
    os.chdir(os.path.join(ROOT, "tests"))
    import check_bounds
    import test_bounds_checker as T
    import zko
    oracle = zko.load()
    po2, problems, n, t0 = 5, [], 0, time.time()
    dom = 4 << po2
    rng = np.random.default_rng(first)
    worst = 0.0
    for s in range(first, first + count):
        par, desc = variant(rng, s)
        v, st = check_bounds.check_desc(f"heavy{s}", desc)
        worst = max(worst, st["max_acc_bits"])
        if v:
            problems.append({"variant": par, "violations": v[:2]})
        for pattern, idx in (("random", 7), ("max", 0), ("half_lo", dom - 2)):
            gs, out, mix, pm = T._inputs(desc, po2, pattern, seed=n)
            want = T._oracle_check(oracle, desc, po2, gs, out, mix, pm)
            got = T._executed(f"heavy{s}", desc, po2, gs, out, mix, pm, idx)
            if got != [int(want[k * dom + idx]) for k in range(4)]:
                problems.append({"variant": par, "pattern": pattern, "point": idx})
            n += 1
        print(f"heavy{s} {par}: {st['kernels']} kernels, {st['statements']} statements ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    return {"leg": "cpu", "variants": count, "first": first, "executions": n, "largest_64bit_sum_log2": round(worst, 3), "problems": problems,
            "seconds": round(time.time() - t0)}


def gpu_leg(first: int, count: int) -> dict:
    import ctypes as C
    import zko
    from zeth_amd.circuits.desc import Circuit
    from zeth_amd.hal import HipHal
    from zeth_amd.prover import Segment, SegmentProver
    hal = HipHal(0)
    oracle = zko.load()
    problems, cases, t0 = [], 0, time.time()
    rng = np.random.default_rng(first)
    po2s = 6
    dom = 4 << po2s
    kernels = 0
    for s in range(first, first + count):
        par, desc = variant(rng, s)
        with tempfile.TemporaryDirectory(prefix="zkh_jit_soak_") as cache:
            os.environ["ZKH_JIT_CACHE"] = cache
            prover = SegmentProver(hal, desc)              # generate -> bound verifier -> hipcc -> attach
        circ = prover.circuit
        if circ.kernel_kind() != "attached":
            problems.append({"variant": par, "what": f"kernel kind {circ.kernel_kind()}"})
            continue
        kernels += circ.compiled_parts()
        oc = zko.OracleCircuit(oracle, desc)
        widths = list(Circuit.parse(desc).group_sizes)      # accum, code, data
        vr = np.random.default_rng(1000 + s)
        fam = {"random": [vr.integers(0, P, size=w * dom, dtype=np.uint64).astype(np.uint32) for w in widths]}
        for name, word in (("all_P-1", P - 1), ("all_(P-1)/2", (P - 1) // 2), ("all_(P+1)/2", (P + 1) // 2)):
            fam[name] = [np.full(w * dom, word, np.uint32) for w in widths]
        alt = [np.full((w, dom), P - 1, np.uint32) for w in widths]
        for g in alt:
            g[:, 1::2] = 0
        fam["rows_alternate"] = [np.ascontiguousarray(g.reshape(-1)) for g in alt]
        for name, gs in fam.items():
            word = int(gs[0][0]) if name != "random" else None
            out = np.full(4, P - 1, np.uint32) if word is not None else vr.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
            mix = np.full(widths[0], P - 1, np.uint32) if word is not None else vr.integers(0, P, size=widths[0], dtype=np.uint64).astype(np.uint32)
            pm = np.full(4, word if word is not None else 12345, np.uint32)
            want = np.zeros(4 * dom, np.uint32)
            gp = (C.c_void_p * 3)(*[x.ctypes.data for x in gs])
            glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
            oracle.zko_eval_check(oc.h, want, gp, glp, pm, po2s)
            dev = [hal.copy_from("g", g) for g in gs]
            g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
            for interp in (False, True):
                check = hal.alloc_elem("check", 4 * dom)
                try:
                    circ.eval_check(check, dev, [g_out, g_mix], pm, po2s, use_interpreter=interp)
                except Exception as e:
                    if interp:                              # the interpreter keeps live values in LDS: a large variant may not fit
                        continue
                    problems.append({"variant": par, "family": name, "what": str(e)[:160]})
                    continue
                cases += 1
                if not np.array_equal(check.to_vec(), want):
                    problems.append({"variant": par, "family": name, "evaluator": "interpreter" if interp else "generated"})
        seg = Segment(index=0, po2=9, seed=77 + s, noise_seed=78 + s, zk_cycles=200)
        got = prover.prove_segment(seg)
        cases += 1
        if not np.array_equal(got.seal, oc.prove(9, 200, seg.seed, seg.noise_seed)):
            problems.append({"variant": par, "what": "po2-9 seal differs from the oracle's"})
        print(f"heavy{s} {par}: {circ.compiled_parts()} kernels ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
        del prover, circ
    return {"leg": "gpu", "variants": count, "first": first, "kernels_compiled": kernels, "comparisons": cases, "problems": problems,
            "seconds": round(time.time() - t0)}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=20)
    a = ap.parse_args()
    rep = cpu_leg(a.first, a.count) if a.cpu else gpu_leg(a.first, a.count)
    print(json.dumps(rep))
    return 1 if rep["problems"] else 0


if __name__ == "__main__":
    raise SystemExit(main())
