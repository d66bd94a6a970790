import os
import sys

# The oracle's OpenMP loops are sized for a handful of cores; on a 128-core GPU host the fork/join cost of tiny
# parallel regions dominates the small test shapes (minutes instead of seconds).  Must be set before libgomp starts.
os.environ.setdefault("OMP_NUM_THREADS", "16")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

P = 2013265921


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout / the GPU box's source-only snapshot has no built artefacts: build them (incremental; a current tree
    # costs milliseconds; hipcc cross-compiles gfx950 without a GPU); xdist workers serialise on the build lock
    from zeth_amd import build as _build
    _build.ensure_built(oracle=True)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    import zko
    return zko.load()


@pytest.fixture(scope="session")
def hal():
    """The product HAL on cuda:0 — fails loudly (no fallback) if the HIP library or the GPU is missing."""
    from zeth_amd.hal import HipHal
    h = HipHal(0)
    yield h
    h.close()


def rand_fp(rng, *shape):
    """Uniform field elements in Montgomery form (every u32 < P is a valid Montgomery word)."""
    return rng.integers(0, P, size=shape, dtype=np.uint64).astype(np.uint32)
