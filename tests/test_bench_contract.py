"""The driver's contract for bench.py, checked on the committed line of the last GPU run (profiles/r03_bench_default.json): the
one JSON line carries BASELINE.json's metric with every field the contract names, a roofline object for the dominant kernel and
a CPU baseline; and the command line still parses the driver's flags.  (No GPU: the line is a committed measurement.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_default_line_has_every_contract_field():
    l = _line("r03_bench_default.json")
    assert l["metric"] == "segments/sec" and l["unit"] == "segments/s" and l["higher_is_better"] is True
    assert l["n_gpus"] == 1 and l["steps"] == 30 and l["warmup"] == 2 and l["scaling"] == "weak" and l["data"] == "synthetic"
    assert l["vs_baseline"] is None and l["dtype"] == "u32"                      # BASELINE.md publishes no number for this metric
    assert abs(l["value"] - 1e3 / l["ms_per_step"]) / l["value"] < 1e-6 and "workload" in l["config"] and "model" not in l["config"]
    r = l["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0 and r["kernel"] == "hash_rows"
    c = l["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "segments/s" and c["sample"]
    # the line certifies its own work, and folds a block to one receipt both ways
    assert l["timed_seals_verified"] == 30 and l["seal_matches_golden"] is True
    b = l["block"]
    assert b["verified_after_clock"] == b["segments"] == 64 and b["succinct"]["compact_receipt_verified"] is True
    assert b["recursive"]["root_verified_against_leaf_claims"] is True and b["recursive"]["proofs"] == 63


def test_config5_line_is_the_in_circuit_fold():
    l = _line("r03_bench_succinct_recursion.json")
    r = l["recursion"]
    assert l["steps"] == 1024 and l["succinct_root_follows_from_leaf_claims"] is True and "recursion" in l["config"]["join_circuit"]
    assert r["fused_lift2"] == 512 and r["joins"] == 511 and r["proofs"] == 1023 and l["verified_after_clock"] >= 1025
    assert abs(l["block_wall_clock_s"] - (l["leaf_phase_s"] + r["fold_s"])) < 0.5


def test_bench_accepts_the_drivers_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--join-circuit", "--fold-inflight", "--no-fused-lift"):
        assert flag in out.stdout, flag
