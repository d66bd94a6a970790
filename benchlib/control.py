"""The N > 1 control plane of bench.py, built so that ONE dead rank cannot hang the run.

The data path has no exchange step (segments are independent: SURVEY.md §8e), so everything the ranks say to each other is
control: "I am at step k", a float to reduce, a handful of receipts to gather.  All of it goes through a key/value store
(`torch.distributed.TCPStore`: the launcher's own store under `python -m torch.distributed.run`, the self-launcher's otherwise),
not through collectives — a collective with a dead peer blocks until its timeout, a store does not:

  * every exchange (barrier, gather, reduction) waits only for the ranks that are still ALIVE;
  * a rank that fails inside a leg says so (`fail`): the others stop waiting for it at once, rank 0 still prints the line with
    `failed_ranks` and the surviving ranks' throughput, and the failed rank stays around (idle) until rank 0 is done, so that a
    launcher which kills the whole group on the first non-zero exit does not take the line with it;
  * a rank that hangs is declared dead by whoever waits for it longer than the timeout (ZKH_BENCH_TIMEOUT_S, default 120 s);
  * the self-launcher (`python bench.py --gpus N`) polls its children: on a hard exit it flags the rank in the store, gives the
    survivors a few seconds to finish their line and kills what is left (/root/reference/run-parallel.sh:93 — one failing job
    must not hang the batch).
"""
from __future__ import annotations

import os
import pickle
import socket
import subprocess
import sys
import time
from datetime import timedelta

TIMEOUT_S = float(os.environ.get("ZKH_BENCH_TIMEOUT_S", "120"))
KILL_GRACE_S = float(os.environ.get("ZKH_BENCH_KILL_GRACE_S", "5"))
PREFIX = "zkhbench/"


class RankFailed(RuntimeError):
    """this rank was declared dead by its peers (it was too slow to arrive), or marked itself failed"""


class LegAborted(RuntimeError):
    """a peer gave up on the current leg (something a failed rank left behind broke it): everybody leaves the leg together"""


class ControlPlane:
    """barrier / gather / reduce over a store, counting live ranks only.  world == 1: everything is local."""

    def __init__(self, rank: int, world: int, store=None, timeout_s: float = TIMEOUT_S):
        self.rank, self.world, self.store, self.timeout_s = rank, world, store, timeout_s
        self.seq = 0
        self._failed = {}            # rank -> message
        self._nfailed_seen = 0
        self.i_failed = None
        self._p2p = {}               # (src, dst) -> messages so far
        self.leg, self.leg_no = None, 0

    # ---- construction ----
    @staticmethod
    def connect(rank: int, world: int, timeout_s: float = TIMEOUT_S) -> "ControlPlane":
        if world == 1:
            return ControlPlane(0, 1)
        import torch.distributed as dist
        host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
        store = dist.TCPStore(host, port, None, False, timedelta(seconds=timeout_s))
        return ControlPlane(rank, world, dist.PrefixStore(PREFIX, store), timeout_s)

    # ---- failure bookkeeping ----
    def _refresh_failed(self):
        n = self.store.add("nfailed", 0)
        if n != self._nfailed_seen:
            for r in range(self.world):
                if r not in self._failed and self.store.check([f"failed/{r}"]):
                    self._failed[r] = self.store.get(f"failed/{r}").decode(errors="replace")
            self._nfailed_seen = n
            if self.rank in self._failed and self.i_failed is None:
                self.i_failed = self._failed[self.rank]
                raise RankFailed(self._failed[self.rank])
            if 0 in self._failed and self.rank != 0:          # nobody is left to print the line
                raise RankFailed(f"rank 0 is gone ({self._failed[0]}): no line will be printed")

    def _declare(self, r: int, msg: str):
        if not self.store.check([f"failed/{r}"]):
            self.store.set(f"failed/{r}", msg)
            self.store.add("nfailed", 1)

    def fail(self, where: str, exc=None) -> str:
        """mark THIS rank failed; the others stop waiting for it"""
        msg = f"rank {self.rank} failed in {where}: {exc!r}" if exc is not None else f"rank {self.rank} failed in {where}"
        self.i_failed = msg
        if self.world > 1:
            self._declare(self.rank, msg)
        return msg

    def failed(self) -> dict:
        if self.world > 1:
            try:
                self._refresh_failed()
            except RankFailed:
                pass
        return dict(self._failed)

    def alive(self) -> list:
        return [r for r in range(self.world) if r not in self._failed]

    # ---- the one primitive: everybody posts a payload, everybody waits for the live ranks' payloads ----
    def _exchange(self, payload: bytes, readers=None, key=None) -> dict:
        self.seq += 1
        if self.world == 1:
            return {0: payload}
        if self.i_failed is not None:
            raise RankFailed(self.i_failed)
        in_leg = key is None and self.leg is not None
        key = key or f"x/{self.seq}/"
        self.store.set(key + str(self.rank), payload)
        missing = set(range(self.world)) - {self.rank}
        t0 = time.perf_counter()
        polls = 0
        while True:
            self._refresh_failed()
            missing -= set(self._failed)
            for r in list(missing):
                if self.store.check([key + str(r)]):
                    missing.discard(r)
            if not missing:
                break
            if in_leg and self.store.check([f"abort/{self.leg_no}"]):
                raise LegAborted(self.store.get(f"abort/{self.leg_no}").decode(errors="replace"))
            polls += 1
            if time.perf_counter() - t0 > self.timeout_s:
                for r in missing:
                    self._declare(r, f"rank {r} did not reach control step {self.seq} within {self.timeout_s:.0f} s (declared dead by rank {self.rank})")
                continue
            time.sleep(0.0002 if polls < 5000 else 0.002)
        if readers is not None and self.rank not in readers:
            return {}
        out = {self.rank: payload}
        for r in range(self.world):
            if r != self.rank and r not in self._failed:
                out[r] = self.store.get(key + str(r))
        return out

    def barrier(self) -> None:
        self._exchange(b"")

    def allgather(self, obj) -> dict:
        return {r: pickle.loads(b) for r, b in self._exchange(pickle.dumps(obj)).items()}

    def gather(self, obj, dst: int = 0):
        """-> {rank: obj} of the live ranks on `dst`, None elsewhere"""
        got = self._exchange(pickle.dumps(obj), readers=(dst,))
        return {r: pickle.loads(b) for r, b in got.items()} if self.rank == dst else None

    def broadcast(self, obj, src: int = 0):
        got = self._exchange(pickle.dumps(obj) if self.rank == src else b"")
        if src not in got:
            raise RankFailed(f"the source rank {src} of a broadcast is dead")
        return pickle.loads(got[src])

    def max(self, x: float) -> float:
        return max(self.allgather(float(x)).values())

    def min(self, x: float) -> float:
        return min(self.allgather(float(x)).values())

    def sum(self, xs):
        """element-wise sum of a list of floats over the live ranks"""
        parts = list(self.allgather([float(x) for x in xs]).values())
        return [sum(p[i] for p in parts) for i in range(len(xs))]

    # ---- legs: a stretch of exchanges that the live ranks enter and leave TOGETHER, whatever happens inside ----
    def begin_leg(self, name: str) -> None:
        self.leg_no += 1
        self.leg = name
        self.seq = self.leg_no * 100000            # the ranks may leave a leg at different exchanges: numbering restarts per leg

    def abort_leg(self, why: str) -> None:
        """this rank cannot finish the leg (and is not itself broken): the others' exchanges inside the leg raise LegAborted"""
        if self.world > 1 and self.leg is not None:
            self.store.set(f"abort/{self.leg_no}", f"rank {self.rank}: {why}")

    def end_leg(self, obj) -> dict:
        """the exchange every live rank reaches, by the leg's normal end or by an abort -> {rank: obj}"""
        name, self.leg = self.leg, None
        return {r: pickle.loads(b) for r, b in self._exchange(pickle.dumps(obj), key=f"leg/{self.leg_no}/").items()}

    # ---- point to point (the Python join tree: a right child's receipt to the rank that joins it) ----
    def send(self, obj, dst: int) -> None:
        n = self._p2p.get((self.rank, dst), 0)
        self._p2p[(self.rank, dst)] = n + 1
        self.store.set(f"p/{self.rank}/{dst}/{n}", pickle.dumps(obj))

    def recv(self, src: int):
        n = self._p2p.get((src, self.rank), 0)
        self._p2p[(src, self.rank)] = n + 1
        key, t0 = f"p/{src}/{self.rank}/{n}", time.perf_counter()
        while not self.store.check([key]):
            self._refresh_failed()
            if src in self._failed or time.perf_counter() - t0 > self.timeout_s:
                raise RankFailed(f"rank {src} never sent what rank {self.rank} waits for: {self._failed.get(src, 'timeout')}")
            time.sleep(0.0005)
        return pickle.loads(self.store.get(key))

    # ---- the end of the run ----
    def finish(self) -> None:
        """rank 0, after its line is out: release the ranks that are waiting to exit"""
        if self.world > 1:
            self.store.set("done", b"1")

    def wait_done(self) -> None:
        """a failed rank idles here until rank 0 is done (or dead), then exits non-zero: a launcher that kills the group on the
        first non-zero exit must not take rank 0's line with it"""
        if self.world == 1:
            return
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < self.timeout_s:
            try:
                if self.store.check(["done"]) or self.store.check(["failed/0"]):
                    return
            except Exception:
                return
            time.sleep(0.05)


def rccl_probe(run, timeout_s: float = 60.0):
    """Bring RCCL up on a process group OF ITS OWN and run ONE all_reduce over xGMI before anything is timed: a health probe (the
    data path has no collective and never will: SURVEY.md §8e).  Default: whenever the ranks hold N distinct GPUs
    (ZKH_DIST_BACKEND=gloo switches it off, =nccl forces it).  Every rank takes the same decision (it depends only on the device
    identities all ranks have just exchanged), so the collective `new_group` is entered by all or none.  The probe runs on a helper
    thread with a deadline: a hung RCCL bring-up costs 70 s and the string "timeout", never the run.
    -> "ok" | "wrong sum" | "unavailable (...)" | "timeout" | None (not attempted)"""
    import threading
    want = os.environ.get("ZKH_DIST_BACKEND", "auto")
    if want == "gloo" or (want == "auto" and not run.devices_distinct):
        return None
    # the watchdog of a wedged communicator must not abort the process: the seals never needed RCCL
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
    out = {}

    def work():
        try:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(run.device)
            grp = dist.new_group(backend="nccl", timeout=timedelta(seconds=timeout_s))
            probe = torch.ones(1, device=f"cuda:{run.device}")
            dist.all_reduce(probe, group=grp)
            torch.cuda.synchronize(run.device)
            out["v"] = "ok" if int(probe.item()) == run.world else "wrong sum"
        except Exception as e:                       # the seals never needed RCCL
            out["v"] = f"unavailable ({type(e).__name__}: {str(e).splitlines()[0][:120] if str(e) else ''})"

    sys.stdout.flush()
    saved = os.dup(1)                                # RCCL / gloo may announce themselves on stdout: keep it for the ONE JSON line
    os.dup2(2, 1)
    try:
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout_s + 10.0)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    if th.is_alive():
        run.rccl_hung = True
        return "timeout"
    if out.get("v") != "ok":
        run.rccl_hung = True                         # a communicator in an unknown state: do not tear the groups down, just exit
    return out.get("v")


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, script: str, argv) -> int:
    """`python bench.py --gpus N` without a launcher: host the store, spawn the N ranks (one process per GPU), pass rank 0's stdout
    (the ONE JSON line) through, and never hang: when a rank exits non-zero the survivors get KILL_GRACE_S seconds (rank 0 may
    still print its line with `failed_ranks`), then whatever is left is killed.  -> exit code (non-zero if any rank failed)."""
    import torch.distributed as dist
    port = free_port()
    store = dist.TCPStore("127.0.0.1", port, n, True, timedelta(seconds=TIMEOUT_S), wait_for_workers=False)
    ctl = dist.PrefixStore(PREFIX, store)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ZKH_BENCH_CHILD="1", TORCHELASTIC_USE_AGENT_STORE="True")      # the ranks are clients of OUR store, as under torchrun
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, first_fail = 0, None
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, abs(code) or 1)
                if first_fail is None:
                    first_fail = time.perf_counter()
                    sys.stderr.write(f"bench: rank {r} exited with code {code}; the remaining ranks have {KILL_GRACE_S:.0f} s to finish\n")
                try:                       # a hard exit never reached ControlPlane.fail: say it for the rank
                    if not ctl.check([f"failed/{r}"]):
                        ctl.set(f"failed/{r}", f"rank {r} exited with code {code}")
                        ctl.add("nfailed", 1)
                except Exception:
                    pass
        if first_fail is not None and live and time.perf_counter() - first_fail > KILL_GRACE_S:
            for r in live:
                procs[r].kill()
            for r in live:
                procs[r].wait()
            sys.stderr.write(f"bench: killed ranks {sorted(live)} after rank failure\n")
            live.clear()
            rc = rc or 1
        time.sleep(0.05)
    return rc
