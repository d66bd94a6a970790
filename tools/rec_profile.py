#!/usr/bin/env python3
"""Where a lift / a join spends its time: per-op HIP-event brackets of one lane running alone (python tools/rec_profile.py)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from zeth_amd import recursion as rec
from zeth_amd.circuits import syn_air
from zeth_amd.hal import HipHal
from zeth_amd.prover import Segment, SegmentProver

hal = HipHal(0)
desc = syn_air.syn_a()
sp = SegmentProver(hal, desc, resident_code_group=True)
leaves = [sp.prove_segment(Segment(i, 20, seed=1 + i, noise_seed=9)) for i in range(2)]
roots = {20: sp.control_root(20)}
rx = rec.Recursion(hal, rec.build_programs(desc, roots))
l = [rx.lift(r, 7) for r in leaves]
j = rx.join(l[0], l[1], 7)
jj = rx.join(j, j, 7)
hal.sync()
out = {}
for name, fn in (("lift", lambda: rx.lift(leaves[0], 7)), ("join_18_18", lambda: rx.join(l[0], l[1], 7)), ("join_19_19", lambda: rx.join(j, j, 7))):
    hal.sync()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    hal.sync()
    wall = (time.perf_counter() - t) / 5
    hal.prof_reset(); hal.prof_enable(True)
    fn(); hal.sync()
    prof = sorted(hal.prof_get(), key=lambda p: -p["total_ms"])
    hal.prof_enable(False)
    out[name] = {"wall_ms": 1e3 * wall, "kernels_ms": sum(p["total_ms"] for p in prof),
                 "ops": [{"name": p["name"], "calls": p["calls"], "ms": round(p["total_ms"], 3)} for p in prof[:14]]}
    print(name, json.dumps(out[name]))
