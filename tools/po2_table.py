#!/usr/bin/env python3
"""Mcycles/s of SYN-A segment seals at --segment-po2 20 .. 24 in one table (one GPU; witness resident before the clock, every
seal verified after it).  zeth passes any po2 up to upstream's MAX_CYCLES_PO2 = 24 through
/root/reference/crates/host/src/bin/cli.rs:61-66 -> /root/reference/crates/host/src/lib.rs:132-135.

    python tools/po2_table.py [po2 ...] > profiles/r05_po2_table.json
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zeth_amd import build as _build  # noqa: E402

_build.ensure_built()
from zeth_amd.circuits import syn_air  # noqa: E402
from zeth_amd.hal import HipHal  # noqa: E402
from zeth_amd.prover import Segment, SegmentProver  # noqa: E402

NOISE = 0x2E80
# seals in flight per size: what 288 GB holds (one po2-24 seal keeps ~110 GB resident: data group 56 GB evaluated + 14 GB coefficients
# + 14 GB witness, four Merkle trees of 4.3 GB, code / accum / check groups)
INFLIGHT = {20: 3, 21: 3, 22: 3, 23: 2, 24: 1}


def run(po2: int, steps: int):
    desc = syn_air.syn_a()
    k = INFLIGHT.get(po2, 1)
    lanes = []
    for i in range(k):
        hal = HipHal(0)
        pv = SegmentProver(hal, desc)
        seg = Segment(index=i, po2=po2, seed=0x5EED0000 + i, noise_seed=NOISE)
        wit = pv.witgen(seg)
        pv.seal(seg, *wit)                        # warm: pools, tables, clocks
        hal.sync()
        lanes.append((hal, pv, seg, wit))
    # one seal alone on the GPU: the latency
    t0 = time.perf_counter()
    rec = lanes[0][1].seal(lanes[0][2], *lanes[0][3])
    lanes[0][0].sync()
    alone = time.perf_counter() - t0
    nxt, lock, out = [0], threading.Lock(), []

    def work(lane):
        hal, pv, seg, wit = lane
        while True:
            with lock:
                if nxt[0] >= steps:
                    break
                nxt[0] += 1
            r = pv.seal(seg, *wit)
            with lock:
                out.append((seg, r))
        hal.sync()
    ths = [threading.Thread(target=work, args=(ln,)) for ln in lanes]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    root = lanes[0][1].control_root(po2)
    t_v = time.perf_counter()
    for seg, r in out[:2] + [(lanes[0][2], rec)]:
        r.verify(desc, root)
    verify_s = (time.perf_counter() - t_v) / 3
    peak = lanes[0][0].memory()["peak"]
    row = {"po2": po2, "cycles": 1 << po2, "in_flight": k, "steps": steps, "ms_per_seal": 1e3 * dt / steps, "segments_per_s": steps / dt,
           "Mcycles_per_s": steps * (1 << po2) / dt / 1e6, "seal_alone_ms": 1e3 * alone, "Mcycles_per_s_alone": (1 << po2) / alone / 1e6,
           "seal_words": int(rec.seal.size), "verified": True, "verify_s_host": verify_s, "peak_live_GB_lane0": peak / 1e9}
    for hal, pv, _, wit in lanes:
        del wit
    lanes.clear()
    return row


def main():
    po2s = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24]
    rows = []
    for p in po2s:
        rows.append(run(p, steps={20: 30, 21: 18, 22: 12, 23: 6, 24: 4}.get(p, 4)))
        sys.stderr.write(json.dumps(rows[-1]) + "\n")
    print(json.dumps({"workload": "SYN-A segment seals (W_code 16, W_data 208, W_accum 32), witness resident in HBM, every group re-committed per segment",
                      "library": HipHal.version(), "rows": rows}))


if __name__ == "__main__":
    main()
