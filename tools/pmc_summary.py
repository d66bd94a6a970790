#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (KB units -> bytes), largest launch per kernel.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> [traffic.json]   (each dir holds *_counter_collection.csv)
FETCH_SIZE on gfx950 reports half of the bytes of a wide (16 B/lane) coalesced stream (MI355X_MICROARCH.md §HBM); both
the raw number and the x2-corrected one are printed.  WRITE_SIZE is uncalibrated (raw).
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def load(d):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
if len(sys.argv) > 3:      # machine-readable copy for bench.py's roofline.traffic
    import json
    json.dump({k: {"launches": len(fetch[k]), "fetch_x2_bytes": 2 * sum(fetch[k]), "write_bytes": sum(write.get(k, [0.0]))}
               for k in fetch}, open(sys.argv[3], "w"), indent=1)
print(f"{'kernel':34s} {'launches':>8s} {'max FETCH raw GB':>17s} {'x2 GB':>8s} {'max WRITE GB':>13s} {'sum FETCH x2 GB':>16s} {'sum WRITE GB':>13s}")
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    w = write.get(k, [0.0])
    print(f"{k:34s} {len(fetch[k]):8d} {max(fetch[k]) / 1e9:17.3f} {2 * max(fetch[k]) / 1e9:8.3f} {max(w) / 1e9:13.3f} "
          f"{2 * sum(fetch[k]) / 1e9:16.3f} {sum(w) / 1e9:13.3f}")
