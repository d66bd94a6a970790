#!/usr/bin/env python3
"""Forward / inverse NTT A/B on one MI355X: per-pass HIP-event times of the data-group shape (W x 2^po2 -> 2^(po2+2)) plus a
SHA-256 of the first 8 output columns so that variants (another build through ZKH_LIBRARY; the round-3 env switches are gone) can be checked for identical results.   python tools/exp_ntt.py [--po2 20] [--width 208]"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeth_amd.hal import HipHal  # noqa: E402

P = 2013265921
ap = argparse.ArgumentParser()
ap.add_argument("--po2", type=int, default=20)
ap.add_argument("--width", type=int, default=208)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--tag", default="")
a = ap.parse_args()
hal = HipHal(0)
rng = np.random.default_rng(1)
n, w = 1 << a.po2, a.width
co = hal.alloc_elem("co", w * n)
for off in range(0, w * n, 1 << 26):
    co.write(rng.integers(0, P, size=min(1 << 26, w * n - off), dtype=np.uint64).astype(np.uint32), off)
ev = hal.alloc_elem("ev", 4 * w * n)
hal.batch_expand_into_evaluate_ntt(ev, co, w, 2)
hal.sync()
digest = hashlib.sha256(ev.to_vec()[: 8 * 4 * n].tobytes()).hexdigest()
hal.prof_reset(); hal.prof_enable(True)
for _ in range(a.reps):
    hal.batch_expand_into_evaluate_ntt(ev, co, w, 2)
hal.sync()
rec = {p["name"]: round(p["total_ms"] / p["calls"], 4) for p in hal.prof_get()}
hal.prof_enable(False)
env = {k: v for k, v in os.environ.items() if k.startswith("ZKH_")}
print(json.dumps({"tag": a.tag, "po2": a.po2, "width": w, "env": env, "ms_per_call": rec, "sum_ms": round(sum(rec.values()), 4),
                  "sha256_first8cols": digest}), flush=True)
