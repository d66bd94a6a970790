#!/usr/bin/env python3
"""Regenerate tests/golden/large_digests.json: the CPU oracle's po2-20 SYN-A seal (BASELINE config 2's exact shape:
W_code 16, W_data 208, W_accum 32, n = 2^20, 4n = 2^22) with a SHA-256 of EVERY intermediate buffer of the seal
(traces, coefficient columns, evaluated groups, Merkle node arrays, check polynomial, mixed / divided combos, final
polynomial) and of the seal itself.  The GPU test replays the same pipeline op by op through the C ABI and compares
each buffer's digest: byte-exactness of hash_rows at 208 x 2^22, of both NTTs at 208 x 2^20 -> 2^22, of eval_check
at 2^22 points, ... without shipping gigabytes of fixtures.

Takes a few minutes of CPU (the oracle is a literal restatement, not tuned).  Run from the repo root:
    python tests/golden/make_golden_large.py [po2 ...]
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import zko  # noqa: E402
from zeth_amd.circuits import syn_air  # noqa: E402

SEED, NOISE, ZK = 0x5EED0000, 0x2E80, 1994


def stage_digests(lib, desc, po2, seed=SEED, noise=NOISE, zk=ZK):
    stages = {}

    @C.CFUNCTYPE(None, C.c_char_p, C.POINTER(C.c_uint32), C.c_size_t)
    def hook(name, ptr, words):
        buf = (C.c_uint32 * words).from_address(C.addressof(ptr.contents))
        rec = {"words": int(words), "sha256": hashlib.sha256(memoryview(buf)).hexdigest()}
        if words <= 2048:                      # challenges, globals, coeff_u: small enough to carry verbatim
            rec["values"] = [int(x) for x in buf]
        stages[name.decode()] = rec

    lib.zko_set_stage_hook.argtypes = [C.c_void_p]
    lib.zko_set_stage_hook(C.cast(hook, C.c_void_p))
    try:
        oc = zko.OracleCircuit(lib, desc)
        t0 = time.time()
        seal = oc.prove(po2, zk, seed, noise)
        dt = time.time() - t0
        assert oc.verify(seal) is None
    finally:
        lib.zko_set_stage_hook(None)
    return seal, stages, dt


def main():
    """usage: make_golden_large.py [shape:po2 ...]   (default syn_a:20 syn_heavy:20; shapes: syn_a, syn_heavy)"""
    from zeth_amd.circuits import syn_heavy
    shapes = {"syn_a": syn_air.syn_a, "syn_heavy": syn_heavy.syn_heavy}
    todo = [a.split(":") for a in sys.argv[1:]] or [["syn_a", "20"], ["syn_heavy", "20"]]
    lib = zko.load()
    path = os.path.join(HERE, "large_digests.json")
    out = {"generator": "tests/golden/make_golden_large.py (CPU oracle, stage hook)", "cases": []}
    try:                                      # keep the cases that are not being regenerated
        with open(path) as fh:
            keep = [c for c in json.load(fh)["cases"] if [c["shape"], str(c["po2"])] not in todo]
        out["cases"].extend(keep)
    except (OSError, ValueError, KeyError):
        pass
    for shape, po2 in todo:
        po2 = int(po2)
        seal, stages, dt = stage_digests(lib, shapes[shape](), po2)
        out["cases"].append({"shape": shape, "po2": po2, "zk_cycles": ZK, "seed": SEED, "noise_seed": NOISE,
                             "oracle_seconds": round(dt, 1), "threads": int(lib.zko_num_threads()),
                             "seal_words": int(seal.size), "seal_sha256": hashlib.sha256(seal.astype("<u4").tobytes()).hexdigest(),
                             "seal_head": [int(x) for x in seal[:8]], "stages": stages})
        print(f"{shape} po2 {po2}: {dt:.1f} s, {len(stages)} stages, seal {seal.size} words", flush=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
