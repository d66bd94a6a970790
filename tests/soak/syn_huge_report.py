#!/usr/bin/env python3
"""SYN-HUGE: `eval_check` at a real circuit's SCALE (round-5 verdict, item 3) — one JSON report.

The generator had never seen more than 54 k steps / 1 061 taps (SYN-HEAVY).  SYN-HUGE (circuits/syn_heavy.py syn_huge) is the same
trace shape under > 250 k PolyExtSteps, > 2 000 taps at backs 0 .. 7 and ~15 k degree-5 constraints, loaded the way a circuit the
library has never seen is loaded: as DATA, generated + compiled at load time (circuits/jit.py).  The report answers what was unknown:

  CPU (any host, hipcc cross-compiles):  generator seconds, the exact bound verifier over the emitted text (tools/check_bounds.py),
       hipcc wall seconds for all parts in parallel + the slowest part, code-object MB, kernels, VGPRs / scratch (spills), the
       static opcode table = VALU instructions per domain point;
  GPU (when one is there):  load + attach seconds, `eval_check` ms per po2-20 call (HIP events of the library, all parts), VALU
       instructions per point per ms, one whole po2-20 seal (ms, verified by the host verifier), and generated == interpreter ==
       oracle on a small segment + the extreme vectors (tests/test_maximal_vectors_gpu.py's patterns).

    python tests/soak/syn_huge_report.py [--circuit syn_huge|syn_heavy] [--no-gpu] [--po2 20] > report.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuit", default="syn_huge", choices=("syn_huge", "syn_heavy"))
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--no-static", action="store_true", help="skip the hipcc -S opcode table (a second compile of every part)")
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--jobs", type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    from zeth_amd.circuits import codegen, jit, syn_heavy
    from zeth_amd.circuits.desc import Circuit
    rep = {"circuit": a.circuit, "cpu_count": os.cpu_count()}
    t0 = time.perf_counter()
    desc = syn_heavy.syn_huge() if a.circuit == "syn_huge" else syn_heavy.syn_heavy()
    rep["build_desc_s"] = round(time.perf_counter() - t0, 2)
    c = Circuit.parse(desc)
    rep.update(steps=len(c.steps), taps=len(c.taps), combos=[list(x) for x in c.combos], group_sizes=list(c.group_sizes), desc_words=int(desc.size),
               max_back=max(t[2] for t in c.taps))
    # ---- generator ----
    t0 = time.perf_counter()
    srcs = jit.eval_check_sources(desc)
    rep["generator_s"] = round(time.perf_counter() - t0, 2)
    plan = codegen.Plan.build(c)
    rep.update(kernels=len(srcs), constraints=int(plan.n_leaves[c.ret]), distinct_arithmetic_values=int(plan.n_unique), mix_powers=int(plan.n_pows),
               source_MB=round(sum(len(s) for _, s in srcs) / 1e6, 2), lazy_values=len(plan.lazy), sums_of_products=len(plan.sop))
    # ---- the exact bound verifier over the emitted text ----
    import check_bounds
    t0 = time.perf_counter()
    viol, stats = [], {"statements": 0, "reductions": 0, "claims": 0, "max_acc_bits": 0.0}
    for _, src in srcs:
        v, st = check_bounds.check_source(src, a.circuit)
        viol += v
        for k in ("statements", "reductions", "claims"):
            stats[k] += st[k]
        stats["max_acc_bits"] = max(stats["max_acc_bits"], st["max_acc_bits"])
    rep["bounds"] = {"violations": len(viol), "seconds": round(time.perf_counter() - t0, 2), **stats, "first": viol[:2]}
    # ---- hipcc: every part, in parallel ----
    jobs = a.jobs or min(len(srcs), os.cpu_count() or 8)
    times = {}

    os.environ["ZKH_JIT_NO_BOUNDS_CHECK"] = "1"       # the verifier ran above and is reported on its own: time hipcc alone here

    def timed(ns):
        t = time.perf_counter()
        img = jit._compile_one(ns[0], ns[1], False)
        times[ns[0]] = time.perf_counter() - t
        return img
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        images = list(ex.map(timed, srcs))
    del os.environ["ZKH_JIT_NO_BOUNDS_CHECK"]
    rep["hipcc"] = {"wall_s": round(time.perf_counter() - t0, 1), "jobs": jobs, "cpu_s_sum": round(sum(times.values()), 1),
                    "slowest_part_s": round(max(times.values()), 1), "code_object_MB": round(sum(len(i) for i in images) / 1e6, 2),
                    "largest_code_object_MB": round(max(len(i) for i in images) / 1e6, 2), "flags": " ".join(jit.FLAGS)}
    if not a.no_static:
        import collections
        import subprocess
        import static_valu
        tmp = tempfile.mkdtemp(prefix="zkh_huge_")

        def asm(ns):
            p = os.path.join(tmp, ns[0] + ".hip")
            open(p, "w").write(ns[1])
            out = os.path.join(tmp, ns[0] + ".s")
            subprocess.run([jit.hipcc_path(), *jit.FLAGS, "--cuda-device-only", "-S", "-I", jit.CSRC, "-I", jit.INCLUDE, p, "-o", out], check=True, capture_output=True)
            return static_valu.count(open(out).read(), ns[0])
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            rows = list(ex.map(asm, srcs))
        tot = collections.Counter()
        for r in rows:
            for k, v in r.items():
                if k not in ("vgprs", "scratch"):
                    tot[k] += v
        rep["static"] = {"valu_per_point": tot["valu"], "mad64": tot["mad64"], "mul32": tot["mul32"], "add_sub": tot["add_sub"], "min_select": tot["min_select"],
                         "mov_logic": tot["mov_logic"], "global_load": tot["global_load"], "s_load": tot["s_load"], "s_nop": tot["s_nop"],
                         "vgprs_max": max(r["vgprs"] for r in rows), "vgprs_min": min(r["vgprs"] for r in rows), "parts_over_128_vgprs": sum(1 for r in rows if r["vgprs"] > 128),
                         "scratch_bytes_max": max(r["scratch"] for r in rows), "valu_per_constraint": round(tot["valu"] / max(1, rep["constraints"]), 1)}
    # ---- the GPU ----
    gpu = None
    if not a.no_gpu:
        try:
            from zeth_amd.hal import HipHal
            hal = HipHal(0)
            gpu = {}
        except Exception as e:
            rep["gpu"] = {"skipped": repr(e)}
    if gpu is not None:
        import ctypes as C
        import zko
        from zeth_amd.prover import Segment, SegmentProver
        P = 2013265921
        with tempfile.TemporaryDirectory(prefix="zkh_jit_huge_") as cache:
            os.environ["ZKH_JIT_CACHE"] = cache
            t0 = time.perf_counter()
            prover = SegmentProver(hal, desc)                       # generates, compiles (parallel), attaches
            gpu["load_generate_compile_attach_s"] = round(time.perf_counter() - t0, 1)
            t0 = time.perf_counter()
            prover2 = SegmentProver(hal, desc)                      # second load: code objects from the cache
            gpu["load_from_cache_s"] = round(time.perf_counter() - t0, 2)
            del prover2
        circ = prover.circuit
        gpu["kernel_kind"], gpu["parts_attached"] = circ.kernel_kind(), circ.compiled_parts()
        # (1) three evaluators on a small segment's real evaluations + the extreme vectors
        oracle = zko.load()
        oc = zko.OracleCircuit(oracle, desc)
        po2s = 6
        dom = 4 << po2s
        widths = list(c.group_sizes)
        rng = np.random.default_rng(6)
        cases = {"random": [rng.integers(0, P, size=w * dom, dtype=np.uint64).astype(np.uint32) for w in widths]}
        for name, word in (("all_P-1", P - 1), ("all_(P-1)/2", (P - 1) // 2), ("all_(P+1)/2", (P + 1) // 2)):
            cases[name] = [np.full(w * dom, word, np.uint32) for w in widths]
        alt = [np.full((w, dom), P - 1, np.uint32) for w in widths]
        for g in alt:
            g[:, 1::2] = 0
        cases["rows_alternate"] = [np.ascontiguousarray(g.reshape(-1)) for g in alt]
        agree = {}
        for name, gs in cases.items():
            word = int(gs[0][0]) if name != "random" else None
            out = np.full(4, P - 1, np.uint32) if word is not None else rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
            mix = np.full(widths[0], P - 1, np.uint32) if word is not None else rng.integers(0, P, size=widths[0], dtype=np.uint64).astype(np.uint32)
            pm = np.full(4, word if word is not None else 12345, np.uint32)
            want = np.zeros(4 * dom, np.uint32)
            gp = (C.c_void_p * 3)(*[x.ctypes.data for x in gs])
            glp = (C.c_void_p * 2)(out.ctypes.data, mix.ctypes.data)
            oracle.zko_eval_check(oc.h, want, gp, glp, pm, po2s)
            dev = [hal.copy_from("g", g) for g in gs]
            g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
            res = {}
            for interp in (False, True):
                check = hal.alloc_elem("check", 4 * dom)
                try:
                    circ.eval_check(check, dev, [g_out, g_mix], pm, po2s, use_interpreter=interp)
                    res["interpreter" if interp else "generated"] = bool(np.array_equal(check.to_vec(), want))
                except Exception as e:                              # the interpreter keeps live values in LDS: a circuit this size may not fit
                    res["interpreter" if interp else "generated"] = f"unavailable: {str(e)[:120]}"
            agree[name] = res
        gpu["equals_oracle_po2_6"] = agree
        gpu["generated_equals_oracle"] = all(v["generated"] is True for v in agree.values())
        # (2) a small whole seal byte for byte against the oracle
        t0 = time.perf_counter()
        small = Segment(index=0, po2=10, seed=77, noise_seed=78, zk_cycles=300)
        got = prover.prove_segment(small)
        want_seal = oc.prove(10, 300, 77, 78)
        gpu["po2_10_seal_equals_oracle"] = bool(np.array_equal(got.seal, want_seal))
        gpu["po2_10_seal_and_oracle_s"] = round(time.perf_counter() - t0, 1)
        # (3) the full-size seal: eval_check ms (HIP events, all parts), seal ms, verified
        seg = Segment(index=0, po2=a.po2, seed=0x5EED0000, noise_seed=0x2E80)
        code, data, out = prover.witgen(seg)
        prover.seal(seg, code, data, out)
        hal.sync()
        reps = []
        for _ in range(3):
            hal.prof_reset(); hal.prof_enable(True)
            t0 = time.perf_counter()
            rec = prover.seal(seg, code, data, out)
            hal.sync()
            wall = time.perf_counter() - t0
            prof = {p["name"]: p for p in hal.prof_get()}
            hal.prof_enable(False)
            reps.append((wall, prof))
        wall, prof = min(reps, key=lambda r: r[0])
        ec = prof.get("eval_check", {"total_ms": 0.0, "calls": 0})
        rec.verify(desc, prover.control_root(a.po2))
        gpu.update(po2=a.po2, seal_ms=round(1e3 * wall, 2), seal_ms_all=[round(1e3 * w, 2) for w, _ in reps], seal_verified=True,
                   eval_check_ms=round(ec["total_ms"], 3), eval_check_launches=int(ec["calls"]),
                   kernels_ms={k: round(v["total_ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]},
                   seal_words=int(rec.seal.size))
        if "static" in rep and ec["total_ms"]:
            n_pts = 4 << a.po2
            vi = rep["static"]["valu_per_point"] * n_pts / 64.0                      # wave-instructions
            gpu["eval_check_valu_wave_instr_static"] = vi
            gpu["eval_check_G_wave_instr_per_s"] = round(vi / (ec["total_ms"] * 1e-3) / 1e9, 1)
            gpu["eval_check_issue_frac_at_2p0GHz"] = round(vi / (ec["total_ms"] * 1e-3 * 2.0e9 * 1024), 4)
        rep["gpu"] = gpu
    print(json.dumps(rep))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
