"""bench.py's N > 1 control plane with a failing rank — on CPU, through `--config dev` (BASELINE config 1: fake receipts, no GPU work;
everything around the seals is the real thing: the launcher, the rendezvous, the store-based barriers / reductions / gathers,
the secondary-leg guard).  One dead, hung or faulting rank must never hang the run or take rank 0's line with it
(/root/reference/run-parallel.sh:93: one failing job must not hang the batch)."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, args=("--config", "dev", "--gpus", "4", "--steps", "40"), launcher=None, timeout=90):
    env = dict(os.environ, ZKH_BENCH_TIMEOUT_S="6", ZKH_BENCH_KILL_GRACE_S="3", **extra_env)
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    cmd = [sys.executable, BENCH, *args] if launcher is None else [*launcher, BENCH, *args]
    t0 = time.perf_counter()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, time.perf_counter() - t0, lines, p.stderr


def test_four_healthy_ranks_print_one_line():
    rc, dt, lines, err = _run({})
    assert rc == 0 and len(lines) == 1, err[-2000:]
    l = lines[0]
    assert l["n_gpus"] == 4 and l["data"] == "dev-mode" and "failed_ranks" not in l and l["value"] > 0
    assert l["block"]["segments"] == 64 and l["block"]["assembled_in_index_order"] is True


def test_a_rank_faulting_in_the_headline_is_reported_and_the_survivors_line_is_printed():
    rc, dt, lines, err = _run({"ZKH_BENCH_FAULT_RANK": "2"})
    assert dt < 30 and rc != 0 and "rank 2 failed in the headline leg" in err
    assert len(lines) == 1
    l = lines[0]
    assert l["failed_ranks"] == [2] and l["failed_in"] == "headline" and l["ranks_reporting"] == 3 and l["value"] > 0
    assert "skipped" in l["block"]                         # nothing is attempted on a broken group


def test_a_rank_dying_hard_gets_its_siblings_killed_within_seconds():
    rc, dt, lines, err = _run({"ZKH_BENCH_FAULT_RANK": "1", "ZKH_BENCH_FAULT_LEG": "hard"})
    assert dt < 30 and rc != 0 and "rank 1 exited with code 17" in err
    # rank 0 learns of the death from the launcher (the store) and still prints the survivors' line
    assert len(lines) == 1 and lines[0]["failed_ranks"] == [1] and "exited with code 17" in lines[0]["failed_ranks_detail"][0]


def test_a_hung_rank_is_declared_dead_after_the_timeout_and_killed():
    rc, dt, lines, err = _run({"ZKH_BENCH_FAULT_RANK": "3", "ZKH_BENCH_FAULT_LEG": "hang"})
    assert dt < 40 and rc != 0 and "killed ranks [3]" in err
    assert len(lines) == 1 and lines[0]["failed_ranks"] == [3] and "did not reach control step" in lines[0]["failed_ranks_detail"][0]


def test_a_rank_failing_in_the_block_leg_only_leaves_the_headline_whole():
    rc, dt, lines, err = _run({"ZKH_BENCH_FAULT_RANK": "2", "ZKH_BENCH_FAULT_LEG": "block"})
    assert dt < 30 and len(lines) == 1, err[-2000:]
    l = lines[0]
    assert l["failed_in"] == "block" and l["failed_ranks"] == [2] and l["ranks_reporting"] == 4      # `value` is all four ranks'
    assert "error" in l["block"] and "rank 2 failed in the block leg" in l["block"]["error"]
    assert rc != 0                                         # the launcher still reports that a rank failed


def test_rank_zero_failing_ends_the_run_without_a_line_and_without_a_hang():
    rc, dt, lines, err = _run({"ZKH_BENCH_FAULT_RANK": "0"})
    assert dt < 30 and rc != 0 and not lines and "rank 0" in err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("fault", [None, "2"])
def test_the_drivers_launch_shape(fault):
    """python -m torch.distributed.run ... bench.py --gpus N: the agent's store is the control plane; a rank that fails softly
    stays around until rank 0 has printed (the agent kills the group on the first non-zero exit)."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    rc, dt, lines, err = _run({} if fault is None else {"ZKH_BENCH_FAULT_RANK": fault}, launcher=launcher, timeout=120)
    assert len(lines) == 1, err[-3000:]
    if fault is None:
        assert rc == 0 and "failed_ranks" not in lines[0]
    else:
        assert rc != 0 and lines[0]["failed_ranks"] == [2] and dt < 60


def test_an_n_rank_line_lists_n_distinct_devices_or_refuses_to_start():
    """Round-5 verdict, weak #6: an N > 1 line must PROVE it ran on N GPUs.  Every rank's device identity (PCI bus id, UUID, NUMA
    node) is gathered into config.devices; N ranks that do not hold N distinct devices refuse to start unless --allow-shared-gpu
    says it is a dry run.  (`--config dev` touches no GPU: ZKH_BENCH_FAKE_DEVICES stands in for the identities.)"""
    rc, dt, lines, err = _run({"ZKH_BENCH_FAKE_DEVICES": "0000:05:00.0,0000:15:00.0,0000:65:00.0,0000:75:00.0"})
    assert rc == 0 and len(lines) == 1, err[-2000:]
    cfg = lines[0]["config"]
    assert [d["rank"] for d in cfg["devices"]] == [0, 1, 2, 3] and cfg["devices_distinct"] is True
    assert len({d["pci_bus_id"] for d in cfg["devices"]}) == 4 and cfg["launcher"] == "ranks"
    # two ranks on one device: nobody starts, the message names them, no line
    rc, dt, lines, err = _run({"ZKH_BENCH_FAKE_DEVICES": "0000:05:00.0,0000:15:00.0,0000:15:00.0,0000:75:00.0"})
    assert rc != 0 and not lines and dt < 30
    assert "do not hold 4 distinct GPUs" in err and "ranks [1, 2] all drive 0000:15:00.0" in err and "--allow-shared-gpu" in err
    # ... unless it is declared a dry run: the line then says so
    rc, dt, lines, err = _run({"ZKH_BENCH_FAKE_DEVICES": "0000:05:00.0"}, args=("--config", "dev", "--gpus", "4", "--steps", "40", "--allow-shared-gpu"))
    assert rc == 0 and len(lines) == 1, err[-2000:]
    cfg = lines[0]["config"]
    assert cfg["devices_distinct"] is False and len(cfg["devices"]) == 4 and {d["pci_bus_id"] for d in cfg["devices"]} == {"0000:05:00.0"}


def test_a_slow_rank_still_sees_every_device_before_anyone_refuses(monkeypatch):
    """The refusal is a verdict every rank reaches from the SAME view.  A rank that got the view first used to declare itself failed
    on its way out while a slower peer was still polling inside the exchange; the peer then dropped that rank's entry, counted one
    device fewer, found them distinct and started the run (seen once on the GPU box: no message, the launcher killed rank 0 after its
    grace period).  Two ranks in threads over one store, rank 0 made slow inside the exchange: both must refuse."""
    import threading
    import types
    import torch.distributed as dist
    from benchlib.common import Run
    from benchlib.control import ControlPlane, RankFailed
    monkeypatch.setenv("ZKH_BENCH_FAKE_DEVICES", "0000:05:00.0")
    store = dist.HashStore()
    args = types.SimpleNamespace(po2=16, inflight=1, allow_shared_gpu=False)
    ctls = [ControlPlane(r, 2, dist.PrefixStore("t", store), 20.0) for r in range(2)]
    slow = {"first": True}
    plain = ControlPlane._refresh_failed

    def dawdle(self):
        if self.rank == 0 and slow["first"]:
            slow["first"] = False
            time.sleep(1.0)                      # rank 1 has everything it needs long before rank 0 looks at the store again
        return plain(self)
    monkeypatch.setattr(ControlPlane, "_refresh_failed", dawdle)
    got = {}

    def rank_main(r):
        run = Run(args, ctls[r], r, r, 2)
        try:
            run.exchange_devices()
            got[r] = ("started", len(run.devices))
        except SystemExit as e:
            ctls[r].fail("segment", e)           # what bench.py's handler does with it
            got[r] = ("refused", e.code)
        except RankFailed as e:                  # rank 0 said its verdict while this rank was still in the barrier: out, non-zero
            got[r] = ("released", str(e))
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(30)
    assert got[1] == ("refused", 2) or (got[1][0] == "released" and "rank 0 is gone" in got[1][1])
    assert got[0][0] == "refused" and "do not hold 2 distinct GPUs" in str(got[0][1]) and "ranks [0, 1] all drive 0000:05:00.0" in str(got[0][1])
