#!/usr/bin/env python3
"""GPU experiment: eval_check time of a circuit as a function of the generator's knobs (register-cache size, offset-epoch
length, part weight).  Every variant is generated, cross-compiled (hipcc --genco, parts in parallel), attached and timed
at po2 20 on the SYN-A-shaped evaluated groups; results are checked against the first variant (bit-exact).

    python tools/exp_codegen.py syn_heavy REGS=72 REGS=96,SOP=0 REGS=128,EPOCH=96 > gpurun_out/exp_codegen.jsonl
(every argument after the circuit is one variant: comma-separated ZKH_CODEGEN_<KNOB>=value assignments)
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "syn_heavy"
    variants = sys.argv[2:] or ["REGS=72", "REGS=96", "REGS=128"]
    knobs = ("REGS", "EPOCH", "PART", "SOP", "PREFETCH", "LAZY", "FLAGS", "GATHER", "LOCALITY", "LOCWIN")
    po2 = int(os.environ.get("EXP_PO2", "20"))
    from zeth_amd.circuits import codegen, jit
    from zeth_amd.hal import HipHal
    from zeth_amd.prover import Segment, SegmentProver
    desc = codegen.shipped()[name]
    hal = HipHal(0)
    prover = SegmentProver(hal, desc)
    circ = prover.circuit
    wa, wc, wd = (int(x) for x in desc[3:6])
    n, dom = 1 << po2, 4 << po2
    seg = Segment(index=0, po2=po2, seed=1, noise_seed=2)
    code, data, out = prover.witgen(seg)
    mix = np.arange(1, wa + 1, dtype=np.uint32)
    accum = hal.alloc_elem("accum", wa * n)
    hal.syn_accum(circ, po2, seg.zk_cycles, 2, data, mix, accum)
    ev = []
    for buf, w in ((accum, wa), (code, wc), (data, wd)):
        co = hal.alloc_elem("co", w * n)
        hal.batch_interpolate_ntt_from(co, buf, w, True)
        e = hal.alloc_elem("ev", w * dom)
        hal.batch_expand_into_evaluate_ntt(e, co, w, 2)
        ev.append(e)
    g_out, g_mix = hal.copy_from("out", out), hal.copy_from("mix", mix)
    poly_mix = np.array([5, 6, 7, 8], dtype=np.uint32)
    check = hal.alloc_elem("check", 4 * dom)

    def timed(reps=10):
        circ.eval_check(check, ev, [g_out, g_mix], poly_mix, po2)
        hal.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            circ.eval_check(check, ev, [g_out, g_mix], poly_mix, po2)
        hal.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    base = timed()
    ref = check.to_vec()
    print(json.dumps({"variant": "built-in", "circuit": name, "parts": circ.compiled_parts(), "ms": round(base, 3)}), flush=True)
    for var in variants:
        for k in knobs:
            os.environ.pop("ZKH_CODEGEN_" + k, None)
        for kv in var.split(","):
            k, v = kv.split("=", 1)
            assert k in knobs, k
            os.environ["ZKH_CODEGEN_" + k] = v.replace(";", " ")          # FLAGS=-mllvm;-enable-misched=0
        importlib.reload(codegen)
        importlib.reload(jit)
        t0 = time.perf_counter()
        objs = jit.compile_code_objects(desc, use_cache=False)
        t_c = time.perf_counter() - t0
        for i, (img, kn) in enumerate(objs):
            circ.attach_code_object(img, kn, i, len(objs))
        ms = timed()
        same = bool(np.array_equal(check.to_vec(), ref))
        print(json.dumps({"variant": var, "circuit": name, "parts": len(objs), "compile_s": round(t_c, 1),
                          "ms": round(ms, 3), "bit_exact_vs_builtin": same}), flush=True)


if __name__ == "__main__":
    main()
