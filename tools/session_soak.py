#!/usr/bin/env python3
"""Soak of the native session executor (csrc/session.hip + scheduler.h) through the g++ host: random session shapes — 1 .. 40 segments,
sizes 13 .. 15 with a shorter tail (SYN-A: the BASELINE widths), 1 .. 4 lanes, 0 .. 5 assumed keccak receipts, join3 on or off — each proven TWICE with the same noise
seed, once as one streamed pipeline and once in two phases (seal everything, then fold): both runs must verify inside the library
(zkh_session_verify: every leaf seal, the root seal, the claim tree recomputed from the leaves, the resolved assumptions) and must end in
the SAME root receipt output (claim root || allowed-programs root), whatever order the lanes happened to prove the nodes in.

    python tools/session_soak.py --first 0 --count 30          # on an MI355X; one JSON line, exit code 1 on any problem
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=30)
    ap.add_argument("--circuit", default="syn_a", help="a shipped circuit; a narrow one (syn_small) has lift / join programs of three sizes, and a session of it that "
                    "assumes receipts is REFUSED when the programs are built (the allowed set holds 16): counted as `refused`, not as a problem")
    a = ap.parse_args()
    from zeth_amd import build
    exe = os.path.join(os.path.dirname(build.build_examples()), "prove_session")
    problems, runs, lifts, joins, refused, t0 = [], 0, 0, 0, 0, time.time()
    for s in range(a.first, a.first + a.count):
        rng = np.random.default_rng(90000 + s)
        po2 = int(rng.integers(13, 16))
        shape = {"segments": int(rng.integers(1, 41)), "po2": po2, "tail_po2": int(rng.integers(13, po2 + 1)), "inflight": int(rng.integers(1, 5)),
                 "keccak_batches": int(rng.integers(0, 6)) if rng.integers(0, 2) else 0, "no_join3": bool(rng.integers(0, 2))}
        base = [exe, "--circuit", a.circuit, "--build-recursion", "--segments", str(shape["segments"]), "--po2", str(po2), "--tail-po2", str(shape["tail_po2"]),
                "--inflight", str(shape["inflight"]), "--noise-seed", str(0x700 + s)]
        if shape["keccak_batches"]:
            base += ["--keccak-batches", str(shape["keccak_batches"])]
        if shape["no_join3"]:
            base.append("--no-join3")
        outs = []
        for extra in ([], ["--two-phase"]):
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
            runs += 1
            if r.returncode != 0 and "the allowed set (16 programs) has no room for resolve" in r.stderr:
                refused += 1
                outs.append(None)
                continue
            if r.returncode != 0:
                problems.append({"shape": shape, "two_phase": bool(extra), "rc": r.returncode, "stderr": r.stderr[-400:]})
                outs.append(None)
                continue
            l = json.loads(r.stdout.strip().splitlines()[-1])
            if l.get("verified") is not True or (shape["keccak_batches"] and l.get("resolved") is not True):
                problems.append({"shape": shape, "two_phase": bool(extra), "line": l})
            lifts += l["lifts"]
            joins += l["joins"]
            outs.append(l["root_out"])
        if outs[0] is not None and outs[1] is not None and outs[0] != outs[1]:
            problems.append({"shape": shape, "what": "streamed and two-phase folds end in different root outputs", "roots": outs})
        print(f"session {s} {shape}: {'ok' if not problems else 'PROBLEMS'} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    print(json.dumps({"sessions": a.count, "first": a.first, "runs": runs, "lift_proofs": lifts, "join_proofs": joins, "refused": refused, "circuit": a.circuit, "problems": problems,
                      "seconds": round(time.time() - t0)}))
    return 1 if problems else 0


if __name__ == "__main__":
    raise SystemExit(main())
