"""HIP path vs the committed golden fixtures (tests/golden/*.json, made by make_golden.py from the CPU oracle)."""
import hashlib
import json
import os

import numpy as np
import pytest

from zeth_amd.circuits import syn_air
from zeth_amd.prover import Segment, SegmentProver

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_kernel_kats(hal):
    k = json.load(open(os.path.join(G, "kernel_kats.json")))
    # poseidon2 permutation through hash_fold: a 16-word block hashed from the zero state is NOT the raw permutation, so
    # check the permutation through hash_rows on a 24-column... rate is 16: use the sponge KAT instead
    m = np.array(k["hash_rows_8x19"]["matrix"], dtype=np.uint32)
    out = hal.alloc_digest("d", 8)
    hal.hash_rows(out, hal.copy_from("m", m))
    assert out.to_vec().tolist() == k["hash_rows_8x19"]["digests"]
    col = np.array(k["interpolate_ntt_64"]["in"], dtype=np.uint32)
    buf = hal.copy_from("c", col)
    hal.batch_interpolate_ntt(buf, 1)
    assert buf.to_vec().tolist() == k["interpolate_ntt_64"]["out"]


def test_golden_seal_digests(hal):
    gold = json.load(open(os.path.join(G, "seal_digests.json")))
    for g in gold["seals"]:
        prover = SegmentProver(hal, getattr(syn_air, g["shape"])())
        seg = Segment(index=0, po2=g["po2"], seed=g["seed"], noise_seed=g["noise_seed"], zk_cycles=g["zk_cycles"])
        seal = prover.prove_segment(seg).seal
        assert seal.size == g["words"] and seal[:8].tolist() == g["head"]
        assert hashlib.sha256(seal.astype("<u4").tobytes()).hexdigest() == g["sha256"]
        assert prover.control_root(g["po2"], g["zk_cycles"]).tolist() == g["control_root"]


def test_error_behaviour(hal):
    """Shape errors come back as HalError strings (risc0-sys convention: NULL ok / heap error string), never crashes."""
    from zeth_amd.hal import HalError
    a = hal.alloc_elem("a", 100)
    with pytest.raises(HalError, match="power of two"):
        hal.batch_interpolate_ntt(a, 1)
    with pytest.raises(HalError, match="multiple"):
        hal.batch_interpolate_ntt(a, 3)
    with pytest.raises(HalError, match="out of range"):
        a.slice(50, 51)
    with pytest.raises(HalError, match="16x"):
        hal.fri_fold(hal.alloc_elem("o", 4), hal.alloc_elem("i", 60), np.zeros(4, np.uint32))
    with pytest.raises(HalError, match="bad header"):
        hal.load_circuit(np.zeros(32, np.uint32))
    with pytest.raises(HalError, match="only poseidon2"):
        type(hal)(0, "sha-256")
