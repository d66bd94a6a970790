#!/usr/bin/env python3
"""One command for an upstream `Receipt` (bincode, as `default_prover().prove` returns it / as `r0vm` writes it):

    python tools/check_upstream_receipt.py receipt.bin [--type Receipt|SegmentReceipt|CompositeReceipt|SuccinctReceipt]
                                           [--dump-seals DIR] [--desc circuit.desc.npy]

Decodes it with the RECALLED type table of zeth_amd/receipt_codec.py (risc0-zkvm 3.0.3, un-vendored:
/root/reference/Cargo.lock:5418), prints the structure, re-encodes it and compares byte for byte.  Exit 0 = the table reproduces
the file exactly (field order, variant order, integer widths and length prefixes all agree); otherwise the first differing
offset and the path of the field being decoded there are named.  --dump-seals writes every segment seal as little-endian u32
words (what tools/check_upstream_seal.py takes); with --desc each one is walked through that tool's layout check.
Reference call sites: /root/reference/crates/host/src/lib.rs:137-141 (ProveInfo.receipt), bin/cli.rs:103-106 (verify, journal).
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeth_amd import receipt_codec as rc    # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--type", default="Receipt", choices=sorted(rc.SCHEMAS))
    ap.add_argument("--dump-seals")
    ap.add_argument("--desc")
    a = ap.parse_args()
    data = open(a.path, "rb").read()
    schema = rc.SCHEMAS[a.type]
    try:
        value = rc.decode(schema, data)
    except rc.CodecError as e:
        print(f"DECODE FAILED as {a.type}: {e}")
        print("  -> the recalled layout disagrees with this file at that field (field order / variant order / integer width)")
        return 1
    print(f"{a.path}: {len(data)} bytes decode as {a.type}")
    print(rc.describe(value))
    back = rc.encode(schema, value)
    if back != data:
        k = next((i for i, (x, y) in enumerate(zip(back, data)) if x != y), min(len(back), len(data)))
        print(f"RE-ENCODE DIFFERS at byte {k} ({len(back)} vs {len(data)} bytes)")
        return 1
    print("re-encoded byte for byte: the type table reproduces this file")
    segs = []
    if a.type == "Receipt" and value["inner"][0] == "Composite":
        segs = value["inner"][1]["segments"]
    elif a.type == "CompositeReceipt":
        segs = value["segments"]
    elif a.type == "SegmentReceipt":
        segs = [value]
    if a.dump_seals:
        import numpy as np
        os.makedirs(a.dump_seals, exist_ok=True)
        for s in segs:
            p = os.path.join(a.dump_seals, f"segment_{s['index']}.seal.bin")
            np.asarray(s["seal"], dtype="<u4").tofile(p)
            print(f"  wrote {p}: {len(s['seal'])} words, hashfn {s['hashfn']}")
            if a.desc:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_upstream_seal.py"), p, a.desc], capture_output=True, text=True)
                print("    " + "\n    ".join((r.stdout + r.stderr).strip().splitlines()[-6:]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
