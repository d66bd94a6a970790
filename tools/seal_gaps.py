#!/usr/bin/env python3
"""Where one UNLOADED seal's wall-clock goes that is not kernel time (round-5 verdict, item 5: seal wall-clock).

Runs one serial seal (bench.py --pmc-child: one lane, warm seal + ONE more) under `rocprofv3 --kernel-trace --memory-copy-trace`
(no counters) and reads the timeline of the LAST seal: first dispatch .. last dispatch = the span; the kernels' own durations; every
idle gap between consecutive dispatches, attributed to what sits in it (a device-to-host copy = a Fiat-Shamir round trip: tree top /
tap evaluations / remainders / final coefficients / openings -> host sponge -> the next challenge goes back up).  The sum of the gaps
is the MOST an on-device transcript (no host round trips) plus perfect back-to-back launches could take off a seal; the kernel sum is
the floor no such change can go below.

    python tools/seal_gaps.py [--po2 20] [--circuit syn_a] > profiles/r06_seal_gaps.json
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kname(full: str) -> str:
    m = re.search(r"(k_[A-Za-z0-9_]+)", full)
    return m.group(1) if m else full[:40]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--circuit", default="syn_a")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="zkh_gaps_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", d, "-o", "seal", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--po2", str(a.po2), "--circuit", a.circuit]
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"])) for r in csv.DictReader(open(kf))]
    ks.sort()
    copies = []
    if mf:
        for r in csv.DictReader(open(mf[0])):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), int(r.get("Bytes", 0) or 0)))
    # the LAST seal: it begins at the last dispatch of the first kernel a seal launches (the code group's inverse NTT follows the
    # witness generators; the witness is generated once, so the last k_ntt_high<8> ... take the dispatches after the midpoint marker:
    # the child runs exactly two seals, so the second half of the k_hash_rows dispatches belongs to the last one
    rows = [i for i, k in enumerate(ks) if k[2] == "k_hash_rows"]
    first_of_last = rows[len(rows) // 2]
    # walk back to the first dispatch of that seal: the nearest preceding gap > 200 us (the host verifies nothing between seals, but the
    # previous seal ends with openings + a D2H and the next begins with uploads) — or simply the dispatch after the previous seal's merkle_open
    start = first_of_last
    while start > 0 and ks[start - 1][2] not in ("k_merkle_open",):
        start -= 1
    seal = ks[start:]
    span = seal[-1][1] - seal[0][0]
    busy = sum(e - s for s, e, _ in seal)
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(seal, seal[1:]):
        g = s1 - e0
        if g > 0:
            inside = [c for c in copies if c[0] >= e0 - 2000 and c[1] <= s1 + 2000]
            gaps.append({"us": g / 1e3, "after": n0, "before": n1, "copies": [{"dir": c[2], "bytes": c[3], "us": (c[1] - c[0]) / 1e3} for c in inside]})
    big = sorted(gaps, key=lambda g: -g["us"])
    d2h = [g for g in gaps if any("DEVICE_TO_HOST" in c["dir"].upper() or "D2H" in c["dir"].upper() for c in g["copies"])]
    out = {"circuit": a.circuit, "po2": a.po2, "dispatches": len(seal), "span_ms": span / 1e6, "kernel_sum_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6,
           "idle_frac": (span - busy) / span, "gaps": len(gaps), "gaps_over_20us": sum(1 for g in gaps if g["us"] > 20),
           "idle_in_gaps_over_20us_ms": sum(g["us"] for g in gaps if g["us"] > 20) / 1e3,
           "gaps_with_a_device_to_host_copy": len(d2h), "idle_in_those_ms": sum(g["us"] for g in d2h) / 1e3,
           "median_small_gap_us": sorted(g["us"] for g in gaps)[len(gaps) // 2] if gaps else 0.0,
           "largest_gaps": big[:16],
           "reading": "span = first .. last dispatch of one seal alone on the GPU; idle = span - kernel durations (one in-order stream: kernels never overlap); "
                      "a gap holding a device-to-host copy is a Fiat-Shamir round trip (copy + host sponge + the next upload + launch); the idle total is the "
                      "ceiling of what an on-device transcript and back-to-back launches could remove, the kernel sum the floor they cannot touch"}
    print(json.dumps(out))
    shutil.rmtree(d, ignore_errors=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
