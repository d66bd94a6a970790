"""The session executor's plan and state machine (zeth_amd/csrc/scheduler.h) as a unit: plain C++, driven without a GPU by
tests/cpp/scheduler_test.cpp — fold-plan shapes against zeth_amd/recursion.py fold_plan, every node proven once and only after its
children under random completion orders (streamed and two-phase), retries on another lane, lane retirement, the fatal path.  The
same plan must come out of the Python side (what the verifiers recompute)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scheduler_transitions_and_plan_shapes(tmp_path):
    exe = tmp_path / "scheduler_test"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "zeth_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "scheduler_test.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "scheduler ok" in r.stdout, r.stdout + r.stderr


def test_python_fold_plan_has_the_same_proof_counts():
    """the counts scheduler_test.cpp's helper assumes ARE the Python plan's (groups of three above the pairing level)"""
    from zeth_amd.recursion import fold_plan
    for n in (1, 2, 3, 4, 5, 6, 7, 9, 10, 27, 64, 100, 1024):
        levels = fold_plan(n)
        above = sum(1 for groups in levels[1:] for g in groups if len(g) > 1)
        cur, want = (n + 1) // 2, 0
        while cur > 1:
            g3, rem = divmod(cur, 3)
            want += g3 + (1 if rem == 2 else 0)
            cur = g3 + (1 if rem else 0)
        assert above == want, n
    assert sum(1 for groups in fold_plan(1024) for g in groups if len(g) > 1) == 768
