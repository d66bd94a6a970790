#!/usr/bin/env python3
"""The success path of bench.py's RCCL health probe (benchlib/control.py rccl_probe) on a box with ONE GPU: a world-1 gloo group, then
the probe on a group of its own (RCCL with one rank is a real communicator: init, one all_reduce, the sum checked).  The N-rank form
cannot run here (RCCL refuses two ranks on one device: tools/gpu.sh devices records that path as "unavailable (...)")."""
import json
import os
import sys
import time
from datetime import timedelta

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist  # noqa: E402

from benchlib.control import rccl_probe  # noqa: E402


class FakeRun:
    device, world, devices_distinct, rccl_hung = 0, 1, True, False


saved = os.dup(1)
os.dup2(2, 1)
dist.init_process_group("gloo", timeout=timedelta(seconds=60))
os.dup2(saved, 1)
run = FakeRun()
t0 = time.perf_counter()
res = rccl_probe(run, timeout_s=60.0)
print(json.dumps({"rccl_probe": res, "rccl_world": 1 if res == "ok" else None, "seconds": round(time.perf_counter() - t0, 2), "hung": run.rccl_hung}))
if not run.rccl_hung:
    dist.destroy_process_group()
