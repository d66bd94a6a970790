"""--launcher session: the N-GPU headline as ONE process — `zkh_session_create` with N devices x K lanes, the library's own executor
(csrc/session.hip + csrc/scheduler.h: C++ threads, one shared work index, retry on another lane) instead of one Python rank per GPU.

What a Rust host gets when it calls the session entry point once per block (`ProverServer::prove_session`,
/root/reference/crates/host/src/lib.rs:137): no launcher, no rendezvous, no control plane — segment i goes to whichever lane is
free, on whichever device.  The line has the contract's shape; two things differ from the `ranks` launcher and are said in
`config.workload`: witness generation runs INSIDE the clock (the session owns its witnesses; the `ranks` headline keeps them resident
in HBM before the clock starts), and there is one host process, so `config.devices` are the devices of that one process."""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

from .common import BASE_SEED, BENCH_NOISE, HBM_PEAK_GBPS, ROOT, Run, config_common, seal_algorithmic_bytes
from .control import ControlPlane


def run_session_launcher(args, t_start: float, build_s: float) -> int:
    from zeth_amd import hal as zhal
    from zeth_amd.host import Session
    from zeth_amd.prover import Segment
    N = max(1, args.gpus)
    visible = zhal.device_identity(0)["visible_devices"]
    if N > visible and not args.allow_shared_gpu:
        raise SystemExit(f"bench: --launcher session --gpus {N} but this process sees {visible} GPU(s).  A dry run of the N-device shape "
                         f"on fewer GPUs needs --allow-shared-gpu; its line then says `devices_distinct: false`.")
    devs = [i % visible for i in range(N)]
    run = Run(args, ControlPlane(0, 1), 0, 0, 1)
    run.load_circuit()
    run.devices = [dict(zhal.device_identity(d), rank=i, hip_device=d, hip_visible_devices=os.environ.get("HIP_VISIBLE_DEVICES")) for i, d in enumerate(devs)]
    keys = [d["uuid"] or d["pci_bus_id"] for d in run.devices]
    run.devices_distinct = len(set(keys)) == len(keys)
    run.ranks_per_gpu = -(-N // visible)
    inflight = run.inflight = max(1, min(args.inflight, args.steps))
    wa, wc, wd = run.widths
    n = run.n

    sess = Session(run.desc, devices=devs, lanes_per_device=inflight)
    sess.set_resident_code(False)                  # like `value` of the ranks launcher: the code group is re-committed per segment
    lanes = N * inflight

    def segs(first, count):
        return [Segment(index=first + k, po2=args.po2, seed=BASE_SEED + first + k, noise_seed=BENCH_NOISE) for k in range(count)]
    # warm-up: at least two seals per lane (code objects, pool blocks, clocks); the work index hands them out, so ask for three
    sess.prove(segs(1 << 20, max(args.warmup, 3) * lanes))
    steps_total = args.steps * N
    t0 = time.perf_counter()
    comp, _, st = sess.prove(segs(0, steps_total))
    dt = time.perf_counter() - t0
    # after the clock: every timed seal through the host verifier, segment 0 against the oracle's golden digest
    t_v = time.perf_counter()
    lane0 = run.lane()
    croot = lane0.prover.control_root(args.po2)
    for r in comp.segments:
        r.verify(run.desc, croot)
    verify_s = time.perf_counter() - t_v
    matches = None
    try:
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "large_digests.json")))["cases"]
        golden = next((c for c in cases if c["shape"] == args.circuit and c["po2"] == args.po2 and c["seed"] == BASE_SEED
                       and c["noise_seed"] == BENCH_NOISE and c["zk_cycles"] == 1994), None)
        if golden is not None:
            matches = hashlib.sha256(comp.segments[0].seal_bytes()).hexdigest() == golden["seal_sha256"]
            if not matches:
                raise SystemExit("bench: the timed seal of segment 0 differs from the CPU oracle's golden seal (tests/golden/large_digests.json)")
    except (OSError, ValueError, KeyError):
        pass
    sess.close()

    cfg = config_common(run)
    cfg.update({"launcher": "session",
                "workload": (f"single 2^{args.po2}-cycle segment seal per step per GPU, {run.workload}; ONE process, zkh_session_create({N} device(s) x "
                             f"{inflight} lanes): the library's executor hands segment i to the next free lane; the code group is re-committed per "
                             f"segment; witness generation runs INSIDE the clock here ({1e3 * st['witgen_s_sum'] / max(1, steps_total):.2f} ms per segment "
                             f"on its lane) — the `ranks` launcher's `value` keeps the witnesses resident before the clock starts"),
                "parallelism": f"{steps_total} segments over {N} device(s) x {inflight} lanes through one work index, no collectives",
                "value_recomputes_code_group": True, "witgen_in_clock": True, "seal_words": int(comp.segments[0].seal.size)})
    line = {"metric": "segments/sec", "value": steps_total / dt, "unit": "segments/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic", "config": cfg,
            "timed_seals_verified": len(comp.segments), "seal_matches_golden": matches, "verify_ms_per_seal_host": 1e3 * verify_s / max(1, len(comp.segments)),
            "session": {"wall_s_library": st["wall_s"], "witgen_ms_per_segment": 1e3 * st["witgen_s_sum"] / max(1, steps_total),
                        "seal_call_ms_per_segment": 1e3 * st["seal_s_sum"] / max(1, steps_total), "retries": st["n_retries"]}}
    alg = seal_algorithmic_bytes(wa, wc, wd, len(run.circ.taps), len(run.circ.combos), n)
    per_gpu_step_s = dt / args.steps / max(1, run.ranks_per_gpu)
    line["seal_roofline"] = {"alg_bytes": alg, "achieved": alg / per_gpu_step_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": alg / per_gpu_step_s / 1e9 / HBM_PEAK_GBPS}
    # roofline{}: one more seal alone on device 0 under the library's HIP-event brackets (+ the rocprofv3 --pmc child runs)
    if not args.no_prof:
        from .roofline import add_roofline
        seg = Segment(index=0, po2=args.po2, seed=BASE_SEED, noise_seed=BENCH_NOISE)
        code, data, out = lane0.prover.witgen(seg)
        lane0.prover.seal(seg, code, data, out)
        lane0.hal.sync()
        lane0.hal.prof_reset(); lane0.hal.prof_enable(True)
        t_s = time.perf_counter()
        lane0.prover.seal(seg, code, data, out)
        lane0.hal.sync()
        line["seal_wall_clock_unloaded_s"] = time.perf_counter() - t_s
        ref = lane0.hal.prof_get()
        lane0.hal.prof_enable(False)
        del code, data
        lane0.hal.trim()
        add_roofline(line, ref, ref, args, inflight, (wa, wc, wd), n, 0, live=True)
    if not args.no_cpu_baseline:
        from .cpu_baseline import cpu_baseline
        try:
            line["cpu_baseline"] = cpu_baseline(run.desc, args.circuit, None, full_host=args.cpu_full_host, all_cores=not args.no_cpu_all_cores)
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)}
    line["build_s"] = round(build_s, 2)
    line["command_wall_s"] = round(time.perf_counter() - t_start, 1)
    print(json.dumps(line))
    sys.stdout.flush()
    return 0
