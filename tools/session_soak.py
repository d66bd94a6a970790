#!/usr/bin/env python3
"""Soak of the native session executor (csrc/session.hip + scheduler.h) through the g++ host: random session shapes — 1 .. 40 segments,
sizes 13 .. 15 with a shorter tail (SYN-A: the BASELINE widths), 1 .. 4 lanes, 0 .. 5 assumed keccak receipts, join3 on or off — each proven TWICE with the same noise
seed, once as one streamed pipeline and once in two phases (seal everything, then fold): both runs must verify inside the library
(zkh_session_verify: every leaf seal, the root seal, the claim tree recomputed from the leaves, the resolved assumptions) and must end in
the SAME root receipt output (claim root || allowed-programs root), whatever order the lanes happened to prove the nodes in.

--chained: the same on the SYN-S circuit (syn_session) as a CHAINED session with a random initial state and a random 32-byte journal
(zkh_session_set_chained + zkh_session_set_journal): the joins assert continuity in-circuit, the last seal binds the journal; the receipts
are written out and examples/verify_receipts — no GPU — must accept them for that journal and initial state and refuse another journal.

    python tools/session_soak.py --first 0 --count 30          # on an MI355X; one JSON line, exit code 1 on any problem
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
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
    ap.add_argument("--chained", action="store_true")
    a = ap.parse_args()
    if a.chained:
        a.circuit = "syn_session"
    from zeth_amd import build
    exe = os.path.join(os.path.dirname(build.build_examples()), "prove_session")
    verifier = os.path.join(os.path.dirname(exe), "verify_receipts")
    checked = 0
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
        tmp = tempfile.TemporaryDirectory(prefix="zkh_session_soak_") if a.chained else None
        if a.chained:
            shape["initial_state"], shape["journal"] = int(rng.integers(0, 2013265921)), bytes(rng.integers(0, 256, size=32, dtype=np.uint8)).hex()
            base += ["--chained", "--initial-state", str(shape["initial_state"]), "--journal", shape["journal"], "--receipts-dir", tmp.name]
        for extra in ([], ["--two-phase"]):
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
            if a.chained and r.returncode == 0 and not extra and not shape["keccak_batches"]:
                # the verifier's side, no GPU: these receipts, that journal, that initial state — and not another journal (a session that
                # assumes receipts binds them too; the library's own verifier has checked that list above, this one is not handed it)
                roots = re.findall(r"^control-root (\d+):([0-9a-f]{64})$", r.stderr, re.M)
                vb = [verifier, "--circuit", a.circuit, "--receipts-dir", tmp.name, "--initial-state", str(shape["initial_state"])] + [x for p, h in roots for x in ("--control-root", f"{p}:{h}")]
                ok = subprocess.run(vb + ["--journal", shape["journal"]], capture_output=True, text=True, timeout=600)
                other = subprocess.run(vb + ["--journal", shape["journal"][2:] + "00"], capture_output=True, text=True, timeout=600)
                checked += 1
                if ok.returncode != 0 or other.returncode == 0 or "journal does not hash" not in other.stderr:
                    problems.append({"shape": shape, "what": "verify_receipts", "ok_rc": ok.returncode, "other_rc": other.returncode, "stderr": (ok.stderr + other.stderr)[-400:]})
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
    print(json.dumps({"sessions": a.count, "first": a.first, "runs": runs, "lift_proofs": lifts, "join_proofs": joins, "refused": refused, "circuit": a.circuit, "chained": a.chained, "verified_without_gpu": checked, "problems": problems,
                      "seconds": round(time.time() - t0)}))
    return 1 if problems else 0


if __name__ == "__main__":
    raise SystemExit(main())
