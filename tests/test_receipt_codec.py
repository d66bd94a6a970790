"""bincode codec for upstream's `Receipt` containers (zeth_amd/receipt_codec.py; row f3): round trips, hand-computed byte
layouts of the bincode rules it relies on, error reporting with offsets, and tools/check_upstream_receipt.py on a file."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from zeth_amd import receipt_codec as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _segment(i, n_words=1000):
    rng = np.random.default_rng(70 + i)
    return rng.integers(0, 2013265921, n_words, dtype=np.uint64).astype(np.uint32), i, rng.integers(0, 2013265921, 8, dtype=np.uint64)


def test_bincode_rules_byte_for_byte():
    # u32 / u64 little-endian, Vec and String with u64 lengths, Option tag byte, enum variant index as u32, arrays bare
    assert rc.encode(rc.U32, 0x01020304) == bytes([4, 3, 2, 1])
    assert rc.encode(rc.Vec(rc.U32), [1, 2]) == struct.pack("<QII", 2, 1, 2)
    assert rc.encode(rc.String(), "poseidon2") == struct.pack("<Q", 9) + b"poseidon2"
    assert rc.encode(rc.Opt(rc.U32), None) == b"\x00" and rc.encode(rc.Opt(rc.U32), 7) == b"\x01" + struct.pack("<I", 7)
    assert rc.encode(rc.Digest, list(range(8))) == struct.pack("<8I", *range(8))
    assert rc.encode(rc.ExitCode, ("Halted", 0)) == struct.pack("<II", 0, 0)
    assert rc.encode(rc.ExitCode, ("SystemSplit", None)) == struct.pack("<I", 2)
    assert rc.encode(rc.MaybePruned(rc.SystemState), ("Value", {"pc": 0x200000, "merkle_root": [9] * 8})) == struct.pack("<II8I", 0, 0x200000, *[9] * 8)
    assert rc.encode(rc.MaybePruned(rc.Opt(rc.Input)), ("Value", None)) == struct.pack("<I", 0) + b"\x00"
    # a SegmentReceipt: seal, index, hashfn, verifier_parameters, claim — in that order
    v = rc.segment_receipt_value([5, 6, 7], 3, [1] * 8, [2] * 8)
    b = rc.encode(rc.SegmentReceipt, v)
    assert b[:8 + 12] == struct.pack("<QIII", 3, 5, 6, 7) and b[20:24] == struct.pack("<I", 3)
    assert b[24:24 + 8 + 9] == struct.pack("<Q", 9) + b"poseidon2" and b[41:73] == struct.pack("<8I", *[2] * 8)
    assert rc.decode(rc.SegmentReceipt, b) == v


def test_receipts_round_trip_and_reject_corruption():
    segs = [_segment(i) for i in range(3)]
    blob = rc.composite_receipt_bytes(segs, journal=bytes(range(32)), verifier_parameters=list(range(8)))
    val = rc.decode(rc.Receipt, blob)
    assert val["inner"][0] == "Composite" and len(val["inner"][1]["segments"]) == 3 and val["journal"]["bytes"] == bytes(range(32))
    for (seal, idx, claim), got in zip(segs, val["inner"][1]["segments"]):
        assert got["index"] == idx and got["seal"] == [int(w) for w in seal] and got["claim"]["post"] == ("Pruned", [int(w) for w in claim])
    assert rc.encode(rc.Receipt, val) == blob
    # a succinct receipt with an inclusion proof, and a composite that carries an assumption receipt (recursive type)
    sb = rc.succinct_receipt_bytes(segs[0][0], [3] * 8, segs[0][2], b"\x01\x02", 5, [[k] * 8 for k in range(4)])
    sv = rc.decode(rc.Receipt, sb)
    assert sv["inner"][0] == "Succinct" and sv["inner"][1]["control_inclusion_proof"]["index"] == 5 and rc.encode(rc.Receipt, sv) == sb
    val["inner"][1]["assumption_receipts"].append(("Succinct", dict(sv["inner"][1], claim=("Pruned", [4] * 8))))
    val["inner"][1]["assumption_receipts"].append(("Composite", val["inner"][1] | {"assumption_receipts": []}))
    nested = rc.encode(rc.Receipt, val)
    assert rc.decode(rc.Receipt, nested) == val
    # corruption is reported with a path and an offset, never as a crash or a silent success
    with pytest.raises(rc.CodecError, match="trailing"):
        rc.decode(rc.Receipt, blob + b"\x00")
    with pytest.raises(rc.CodecError, match=r"needs \d+ bytes"):
        rc.decode(rc.Receipt, blob[:-5])
    bad = bytearray(blob)
    bad[0:4] = struct.pack("<I", 9)                                  # InnerReceipt variant 9
    with pytest.raises(rc.CodecError, match="variant index 9"):
        rc.decode(rc.Receipt, bytes(bad))
    bad = bytearray(blob)
    bad[4:12] = struct.pack("<Q", 1 << 60)                           # segments: absurd length
    with pytest.raises(rc.CodecError, match="exceeds the input"):
        rc.decode(rc.Receipt, bytes(bad))
    with pytest.raises(rc.CodecError, match="uninhabited"):
        rc.encode(rc.ReceiptClaim, dict(rc.claim_placeholder([0] * 8), input=("Value", {"x": 1})))
    with pytest.raises(rc.CodecError, match="missing field"):
        rc.encode(rc.SegmentReceipt, {"seal": []})


def test_check_upstream_receipt_tool(tmp_path):
    segs = [_segment(i, 300) for i in range(2)]
    p = tmp_path / "receipt.bin"
    p.write_bytes(rc.composite_receipt_bytes(segs, journal=b"\xaa" * 32))
    tool = os.path.join(ROOT, "tools", "check_upstream_receipt.py")
    r = subprocess.run([sys.executable, tool, str(p), "--dump-seals", str(tmp_path / "seals")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "re-encoded byte for byte" in r.stdout and "segments" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "seals" / "segment_1.seal.bin", dtype="<u4")
    assert np.array_equal(got, segs[1][0])
    # a file that is NOT this layout: reported with the field it stopped at, exit 1
    q = tmp_path / "other.bin"
    q.write_bytes(b"\x07\x00\x00\x00" + bytes(64))
    r = subprocess.run([sys.executable, tool, str(q)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "DECODE FAILED" in r.stdout and "variant index 7" in r.stdout


def test_this_repositorys_receipts_travel_in_upstreams_containers(oracle):
    """A composite of oracle-sealed segments -> bincode `Receipt{inner: Composite}` -> back: the seals verify again; a recursion
    receipt -> `Receipt{inner: Succinct}` carries its control id and the membership path of its program as the inclusion proof."""
    import zko
    from zeth_amd import recursion as host_rec
    from zeth_amd.circuits import syn_air
    from zeth_amd.host import CompositeReceipt, hash_pair
    from zeth_amd.prover import SegmentReceipt
    desc = syn_air.syn_tiny()
    oc = zko.OracleCircuit(oracle, desc)
    po2, zk = 9, 100
    root = oc.control_root(po2, zk)
    recs = []
    for i in range(3):
        seal = oc.prove(po2, zk, 40 + i, 0x2E80)
        recs.append(SegmentReceipt(seal=seal, index=i, po2=po2, output=seal[:4].copy()))
    blob = CompositeReceipt(recs).to_upstream_bytes(desc, root, journal=b"\x01\x02\x03\x04")
    val = rc.decode(rc.Receipt, blob)
    assert val["journal"]["bytes"] == b"\x01\x02\x03\x04" and [s["hashfn"] for s in val["inner"][1]["segments"]] == ["poseidon2"] * 3
    back = CompositeReceipt.from_upstream_bytes(blob, out_size=4)
    assert [r.po2 for r in back.segments] == [po2] * 3
    for a, b in zip(recs, back.segments):
        assert np.array_equal(a.seal, b.seal) and a.index == b.index
        assert oc.verify(b.seal, root) is None
    # a recursion receipt as a SuccinctReceipt: the inclusion proof is the program's path in the allowed-programs tree
    roots = [np.full(8, i + 1, np.uint32) for i in range(3)]
    levels = host_rec.allowed_tree(roots)
    fake = host_rec.RecReceipt(np.arange(40, dtype=np.uint32), 16, 2, roots[2])
    sv = rc.decode(rc.Receipt, fake.to_upstream_bytes(levels))["inner"]
    assert sv[0] == "Succinct" and sv[1]["control_id"] == [3] * 8 and sv[1]["control_inclusion_proof"]["index"] == 2
    cur, idx = roots[2], 2
    for d in sv[1]["control_inclusion_proof"]["digests"]:                       # the proof folds to the allowed root
        d = np.asarray(d, dtype=np.uint32)
        cur = hash_pair(d, cur) if idx & 1 else hash_pair(cur, d)
        idx >>= 1
    assert np.array_equal(cur, levels[-1][0]) and sv[1]["verifier_parameters"] == [int(w) for w in levels[-1][0]]
