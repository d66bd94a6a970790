#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* pass: for the largest dispatch of every kernel print duration, engine clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration), VALU wave-instructions and wave/wait cycles.

usage: python tools/sq_summary.py <dir with *_counter_collection.csv> [top N]
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


rows = collections.defaultdict(dict)
meta = {}
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])):
    d = int(r["Dispatch_Id"])
    rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[d] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
best = {}
for d, (name, ms) in meta.items():
    if name not in best or ms > meta[best[name]][1]:
        best[name] = d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; "
      "largest dispatch per kernel (po2-20 SYN-A seal, inflight 1)")
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration")
for name, d in sorted(best.items(), key=lambda kv: -meta[kv[1]][1])[:top]:
    c, ms = rows[d], meta[d][1]
    clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8 / (ms * 1e-3) / 1e9 if ms else 0.0
    print(f"{name:28s} dur {ms:7.3f} ms  clock {clk:4.2f} GHz  VALU wave-instr {c.get('SQ_INSTS_VALU', 0) / 1e6:9.1f} M  "
          f"wave_cycles {c.get('SQ_WAVE_CYCLES', 0) / 1e6:9.1f} M  wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / 1e6:9.1f} M  "
          f"active_inst {c.get('SQ_ACTIVE_INST_ANY', 0) / 1e6:9.1f} M")
