"""examples/verify_receipts — the verifier side of a block as a g++-only host with NO GPU (`receipt.verify`,
/root/reference/crates/host/src/bin/cli.rs:103): receipt containers written by a prover (here: seals of the CPU oracle, wrapped by
the product's zkh_receipt_encode) are checked envelope, order, seal and — for a chained session — continuity; anything that is
not what the expected control root admits is refused with the segment's index."""
import json
import os
import subprocess

import numpy as np
import pytest

import zko
from zeth_amd import build
from zeth_amd.circuits import syn_air
from zeth_amd.host import chain_segments
from zeth_amd.prover import Segment, SegmentReceipt


def _hex(root):
    return "".join(f"{int(w):08x}" for w in root)


@pytest.fixture(scope="module")
def exe():
    return os.path.join(os.path.dirname(build.build_examples()), "verify_receipts")


def _run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)


def test_verifier_cli_accepts_a_block_and_names_the_receipt_it_refuses(oracle, exe, tmp_path):
    desc = syn_air.syn_tiny()
    oc = zko.OracleCircuit(oracle, desc)
    zk = 1994
    sizes = [12, 12, 11]
    roots = {p: oc.control_root(p, zk) for p in set(sizes)}
    seals = [oc.prove(p, zk, seed=40 + i, noise_seed=7) for i, p in enumerate(sizes)]
    for i, (p, seal) in enumerate(zip(sizes, seals)):
        SegmentReceipt(seal=seal, index=i, po2=p).to_words(desc, roots[p]).astype("<u4").tofile(tmp_path / f"segment_{i}.zkr")
    common = ["--circuit", "syn_tiny", "--receipts-dir", str(tmp_path)] + [x for p, r in roots.items() for x in ("--control-root", f"{p}:{_hex(r)}")]
    r = _run(exe, *common)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"driver": "verify_receipts", "library": out["library"], "verified": 3, "chained": False, "gpu": False}
    # the description from a file instead of the compiled-in one
    dpath = tmp_path / "c.desc"
    np.asarray(desc, dtype="<u4").tofile(dpath)
    assert _run(exe, "--desc", str(dpath), *common[2:]).returncode == 0
    # the EXPECTED root decides: a container wrapped around another root (its claim digest then commits to that root) is refused,
    # and so is the honest container when the verifier expects another root (the seal's code tree does not hash to it)
    other = roots[12].copy()
    other[0] ^= 1
    SegmentReceipt(seal=seals[1], index=1, po2=12).to_words(desc, other).astype("<u4").tofile(tmp_path / "segment_1.zkr")
    r = _run(exe, *common)
    assert r.returncode == 1 and "segment 1" in r.stderr and "another control root" in r.stderr
    r = _run(exe, "--circuit", "syn_tiny", "--receipts-dir", str(tmp_path), "--control-root", f"11:{_hex(roots[11])}", "--control-root", f"12:{_hex(other)}")
    assert r.returncode == 1 and "segment 0" in r.stderr and "another control root" in r.stderr
    # a flipped seal word: the envelope's checksum notices; with the checksum repaired, the seal verification does
    blob = SegmentReceipt(seal=seals[1], index=1, po2=12).to_words(desc, roots[12])
    bad = blob.copy()
    bad[26 + seals[1].size // 2] ^= 4
    bad.astype("<u4").tofile(tmp_path / "segment_1.zkr")
    r = _run(exe, *common)
    assert r.returncode == 1 and "segment 1: container" in r.stderr and "checksum" in r.stderr
    forged_seal = seals[1].copy()
    forged_seal[seals[1].size // 2] ^= 4
    SegmentReceipt(seal=forged_seal, index=1, po2=12).to_words(desc, roots[12]).astype("<u4").tofile(tmp_path / "segment_1.zkr")
    r = _run(exe, *common)
    assert r.returncode == 1 and "segment 1: seal" in r.stderr
    # segments out of order, and a size nobody gave a root for
    SegmentReceipt(seal=seals[1], index=2, po2=12).to_words(desc, roots[12]).astype("<u4").tofile(tmp_path / "segment_1.zkr")
    r = _run(exe, *common)
    assert r.returncode == 1 and "out of order" in r.stderr
    blob.astype("<u4").tofile(tmp_path / "segment_1.zkr")
    r = _run(exe, "--circuit", "syn_tiny", "--receipts-dir", str(tmp_path), "--control-root", f"12:{_hex(roots[12])}")
    assert r.returncode == 1 and "segment 2" in r.stderr and "no expected control root" in r.stderr
    assert _run(exe, "--circuit", "syn_small", *common[2:]).returncode == 1          # receipts of another circuit


def test_verifier_cli_checks_the_continuity_of_a_chained_session(oracle, exe, tmp_path):
    desc = syn_air.syn_chain_small()
    oc = zko.OracleCircuit(oracle, desc)
    po2, zk = 11, 1994
    root = oc.control_root(po2, zk)
    base = [Segment(index=i, po2=po2, seed=500 + i, noise_seed=0x33) for i in range(3)]
    segs = chain_segments(base, lambda s: int(oc.witgen(s.po2, zk, s.seed, s.noise_seed, pub=np.zeros(1, np.uint32))[2][0]), initial_state=9)
    seals = [oc.prove(s.po2, zk, s.seed, s.noise_seed, pub=np.asarray(s.pub, dtype=np.uint32)) for s in segs]
    dpath = tmp_path / "chain.desc"
    np.asarray(desc, dtype="<u4").tofile(dpath)

    def write(order):
        for i, k in enumerate(order):
            SegmentReceipt(seal=seals[k], index=i, po2=po2).to_words(desc, root).astype("<u4").tofile(tmp_path / f"segment_{i}.zkr")
    args = ["--desc", str(dpath), "--receipts-dir", str(tmp_path), "--control-root", f"{po2}:{_hex(root)}", "--segments", "3", "--chained"]
    write([0, 1, 2])
    # SYN-C seals bind no exit code: --chained without the expected segment count is refused outright (a truncated session would pass)
    r = _run(exe, *[a for a in args if a not in ("--segments", "3")], "--initial-state", "9")
    assert r.returncode == 1 and "needs --segments" in r.stderr
    r = _run(exe, *args[:-3], "--segments", "4", "--chained", "--initial-state", "9")
    assert r.returncode == 1 and "3 segment receipts found, the session has 4" in r.stderr
    r = _run(exe, *args, "--initial-state", "9")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["chained"] is True, r.stderr
    r = _run(exe, *args, "--initial-state", "8")                    # another starting state
    assert r.returncode == 1 and "segment 0" in r.stderr and "not continuous" in r.stderr
    write([0, 2, 1])                                                # every seal valid, the chain broken
    r = _run(exe, *args, "--initial-state", "9")
    assert r.returncode == 1 and "segment 1" in r.stderr and "not continuous" in r.stderr
    assert _run(exe, *args[:-1]).returncode == 0                    # without --chained the same receipts are three valid seals
