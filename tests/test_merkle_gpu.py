"""MerkleTreeProver::new as one call (zkh_merkle_build) == hash_rows + every hash_fold layer == the oracle's tree."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import zko
from conftest import rand_fp
from zeth_amd.circuits import syn_air
from zeth_amd.circuits.desc import Circuit
from zeth_amd.circuits.desc import Circuit as Desc
from zeth_amd.hal import HalError, HipHal
from zeth_amd.prover import Segment, SegmentProver, shipped_control_root

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
P = 2013265921



@pytest.mark.parametrize("log_rows,cols", [(17, 0), (17, 5), (17, 16), (17, 17), (18, 33), (17, 208), (12, 40)])
def test_merkle_build_equals_hash_rows_plus_fold_all(oracle, log_rows, cols):
    """zkh_merkle_build gives the nodes zkh_hash_rows + zkh_merkle_fold_all give, and the oracle's whole tree."""
    from zeth_amd.hal import HipHal
    hal = HipHal(0)
    rng = np.random.default_rng(400 + cols + log_rows)
    rows = 1 << log_rows
    mat = rand_fp(rng, cols * rows) if cols else np.zeros(0, np.uint32)
    m = hal.copy_from("m", mat) if cols else hal.alloc("m", 0)
    fused = hal.alloc_digest("nodes", 2 * rows)
    hal.merkle_build(fused, m, rows)
    plain = hal.alloc_digest("nodes2", 2 * rows)
    hal.hash_rows(plain.slice(rows * 8, rows * 8), m)
    hal.merkle_fold_all(plain, rows)
    a, b = fused.to_vec(), plain.to_vec()
    assert np.array_equal(a[8:], b[8:])                      # nodes[1 .. 2 rows): node 0 is unused
    # oracle: a few leaves, their parent, and the whole tree's root
    want = np.zeros(rows * 8, dtype=np.uint32)
    oracle.zko_hash_rows(want, rows, np.ascontiguousarray(mat) if cols else np.zeros(1, np.uint32), rows * cols)
    assert np.array_equal(a[rows * 8:], want)
    nodes = np.zeros(2 * rows * 8, dtype=np.uint32)
    nodes[rows * 8:] = want
    size = rows
    while size > 1:
        oracle.zko_hash_fold(nodes, size, size // 2)
        size //= 2
    assert np.array_equal(a[8:], nodes[8:])
