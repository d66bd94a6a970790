#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

There is no reference implementation to import or build here (risc0-zkp is an un-vendored Rust crate), so these
vectors pin the ORACLE (and the Poseidon2 constant tables in include/zkh_poseidon2_consts.h) against drift; the HIP
path is compared with them on the GPU box, where /root/reference does not exist either."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import zko  # noqa: E402
from zeth_amd.circuits import syn_air  # noqa: E402

lib = zko.load()
seals = []
for shape, po2, zk, seed, noise in [("syn_tiny", 9, 100, 0x5EED0000, 0x2E80), ("syn_tiny", 13, 1994, 0x5EED0001, 0x2E80),
                                    ("syn_small", 12, 1994, 0x5EED0002, 0x2E81), ("syn_a", 13, 1994, 0x5EED0003, 0x2E80)]:
    oc = zko.OracleCircuit(lib, getattr(syn_air, shape)())
    seal = oc.prove(po2, zk, seed, noise)
    assert oc.verify(seal, zk_cycles=zk) is None
    seals.append({"shape": shape, "po2": po2, "zk_cycles": zk, "seed": seed, "noise_seed": noise, "words": int(seal.size),
                  "control_root": [int(x) for x in oc.control_root(po2, zk)],
                  "sha256": hashlib.sha256(seal.astype("<u4").tobytes()).hexdigest(), "head": [int(x) for x in seal[:8]]})
with open(os.path.join(HERE, "seal_digests.json"), "w") as fh:
    json.dump({"generator": "tests/golden/make_golden.py (CPU oracle)", "seals": seals}, fh, indent=1)

# Poseidon2 / hash / NTT known answers of the oracle for the HIP kernels
rng = np.random.default_rng(2024)
P = 2013265921
state = rng.integers(0, P, size=24, dtype=np.uint64).astype(np.uint32)
mixed = state.copy()
lib.zko_poseidon2_mix(mixed)
col = rng.integers(0, P, size=64, dtype=np.uint64).astype(np.uint32)
intt = col.copy()
lib.zko_batch_interpolate_ntt(intt, 64, 1)
rows = rng.integers(0, P, size=8 * 19, dtype=np.uint64).astype(np.uint32)
dig = np.zeros(8 * 8, np.uint32)
lib.zko_hash_rows(dig, 8, rows, rows.size)
with open(os.path.join(HERE, "kernel_kats.json"), "w") as fh:
    json.dump({"generator": "tests/golden/make_golden.py (CPU oracle)",
               "poseidon2_mix": {"in": state.tolist(), "out": mixed.tolist()},
               "interpolate_ntt_64": {"in": col.tolist(), "out": intt.tolist()},
               "hash_rows_8x19": {"matrix": rows.tolist(), "digests": dig.tolist()}}, fh)
print("wrote golden vectors")
