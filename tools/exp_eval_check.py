#!/usr/bin/env python3
"""eval_check timing of a shipped circuit on one MI355X (HIP events, kernels summed): python tools/exp_eval_check.py syn_heavy [po2]
Inputs are zero-filled evaluated groups (straight-line kernels: timing does not depend on the values)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeth_amd.circuits import codegen  # noqa: E402
from zeth_amd.circuits.desc import Circuit  # noqa: E402
from zeth_amd.hal import HipHal  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "syn_heavy"
po2 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
desc = codegen.shipped()[name]
c = Circuit.parse(desc)
hal = HipHal(0)
circ = hal.load_circuit(desc)
dom = 4 << po2
groups = [hal.alloc(f"g{i}", w * dom, zero=True) for i, w in enumerate(c.group_sizes)]
gl = [hal.alloc("out", max(1, c.global_sizes[0]), zero=True), hal.alloc("mix", max(1, c.global_sizes[1]), zero=True)]
check = hal.alloc_elem("check", 4 * dom)
pm = np.array([5, 6, 7, 8], dtype=np.uint32)
circ.eval_check(check, groups, gl, pm, po2)
hal.sync()
hal.prof_reset(); hal.prof_enable(True)
reps = 3
for _ in range(reps):
    circ.eval_check(check, groups, gl, pm, po2)
hal.sync()
rec = {p["name"]: p["total_ms"] / reps for p in hal.prof_get()}
print(json.dumps({"circuit": name, "po2": po2, "kernels": circ.compiled_parts(), "taps": len(c.taps), "steps": len(c.steps),
                  "eval_check_ms": round(sum(rec.values()), 3), "lib": os.environ.get("ZKH_LIBRARY", "default"),
                  "alg_GB": round(4 * dom * (sum(c.group_sizes) + 4) / 1e9, 3)}), flush=True)
