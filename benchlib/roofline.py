"""roofline{} for the dominant op of a seal: live HIP-event durations (the library's own brackets on the stream the kernels run
on), HBM traffic and VALU issue from `rocprofv3 --pmc` child runs of this command (separate passes, kernel trace only, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes)."""
from __future__ import annotations

import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

from .common import HBM_PEAK_GBPS, N_SIMDS, PO2, ROOT

# kernels (rocprofv3 names) behind the ops whose counters bench.py can measure on itself
TRAFFIC_KERNELS = {"hash_rows": ("k_hash_rows",), "hash_fold": ("k_hash_fold",), "eval_check": ("k_eval_check_",)}
# witness generators that run BEFORE a seal (not part of the unit of work): excluded from the whole-seal VALU sum
WITGEN_KERNELS = ("k_syn_code", "k_syn_data", "k_syn_rowfill", "k_keccak_", "k_p2join_")
# Share of k_hash_rows' VALU wave-instructions that are of the HALF-RATE class on gfx950 (32-bit multiplies, v_mad_*64*, fp64,
# v_cvt_f64, v_min/v_add3/v_lshl_add: 4 issue cycles per wave64 instruction against 2.46 for plain add / logic / move —
# tools/ubench_valu.hip), from the static opcode table of the steady-state absorb block (tools/isa_histogram.py ->
# profiles/r03_hash_rows_isa_histogram.txt).  A kernel made only of that class issues at most 0.25 wave-instr per SIMD-cycle.
HALF_RATE_SHARE = {"hash_rows": 0.85}
HALF_RATE_ISSUE_PEAK = 0.25


def _kname(full: str) -> str:
    m = re.search(r"(k_[A-Za-z0-9_]+)", full)
    return m.group(1) if m else full[:48]


def _matches(name: str, kernels) -> bool:
    return any(name == k or (k.endswith("_") and name.startswith(k)) for k in kernels)


def pmc_child(counters, circuit: str, po2: int, device: int, timeout_s: float):
    """One child run of bench.py (`--pmc-child`: one warm seal + ONE seal on one lane, nothing else) under
    `rocprofv3 --pmc <counters> --kernel-trace` -> rows [(kernel, dispatch id, duration ms, {counter: value})] of the LAST seal, or
    None.  The child sees this rank's GPU as device 0."""
    if shutil.which("rocprofv3") is None:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ZKH_BENCH_CHILD", "ZKH_SHARE_GPUS", "GROUP_RANK", "ROLE_RANK",
                "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE",
                "ZKH_BENCH_FAULT_RANK", "ZKH_BENCH_FAULT_LEG"):
        env.pop(var, None)
    visible = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
    env["HIP_VISIBLE_DEVICES"] = visible[device] if device < len(visible) else str(device)
    d = tempfile.mkdtemp(prefix="zkh_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "bench", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--po2", str(po2), "--circuit", circuit]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        disp = {}
        for r in csv.DictReader(open(files[0])):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"kernel": _kname(r["Kernel_Name"]), "ms": 0.0, "c": {}})
            if "End_Timestamp" in r and r.get("End_Timestamp"):
                e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        # the LAST seal: per kernel name the last floor(count / 2) dispatches (the child ran two seals; one-off table builds of the
        # first one have an odd count and drop out)
        by_name = {}
        for k in sorted(disp):
            by_name.setdefault(disp[k]["kernel"], []).append(k)
        rows = []
        for name, ids in by_name.items():
            if _matches(name, WITGEN_KERNELS):
                continue
            for k in ids[len(ids) - len(ids) // 2:]:
                rows.append((name, k, disp[k]["ms"], disp[k]["c"]))
        return rows or None
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def valu_pass(kernels, circuit: str, po2: int, device: int, timeout_s: float):
    """One child run under `--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE` -> (figures of the kernels named `kernels`, per launch AND per seal;
    figures of the whole seal), either None when the pass gave nothing."""
    rows = pmc_child(("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"), circuit, po2, device, timeout_s)
    if not rows:
        return None, None
    valu = None
    dom = [(ms, c) for name, _, ms, c in rows if _matches(name, kernels)]
    if dom:
        wi = sum(c.get("SQ_INSTS_VALU", 0.0) for _, c in dom) / len(dom)
        gui = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for _, c in dom) / len(dom)        # summed over the 8 XCDs
        ms = sum(m for m, _ in dom) / len(dom)
        cyc = gui / 8.0 * N_SIMDS
        valu = {"wave_instr": wi, "simd_cycles": cyc, "issue_frac": wi / cyc if cyc else None, "launches": len(dom),
                "clock_GHz": gui / 8.0 / (ms * 1e-3) / 1e9 if ms else None, "launch_ms_under_counters": ms,
                "wave_instr_per_seal": wi * len(dom), "ms_per_seal_under_counters": ms * len(dom)}
    tot_wi = sum(c.get("SQ_INSTS_VALU", 0.0) for _, _, _, c in rows)
    tot_gui = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for _, _, _, c in rows)
    tot_ms = sum(ms for _, _, ms, _ in rows)
    seal = {"wave_instr": tot_wi, "kernel_ms_serial": tot_ms, "dispatches": len(rows),
            "clock_GHz": tot_gui / 8.0 / (tot_ms * 1e-3) / 1e9 if tot_ms else None,
            "issue_frac_serial": tot_wi / (tot_gui / 8.0 * N_SIMDS) if tot_gui else None}
    return valu, seal


def add_heavy_valu(line, args, n, device=0):
    """`syn_heavy.roofline`: the VALU roofline of SYN-HEAVY's eval_check (the kernel that makes the heavy seal heavy), live — one more
    child run of this command with `--circuit syn_heavy` under `--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE` (its own pass, kernel trace
    only).  instr_per_point = wave-instructions x 64 lanes / 4n domain points: the figure the generator's static table predicts."""
    h = line.get("syn_heavy")
    if not isinstance(h, dict) or "segments_per_s" not in h or getattr(args, "no_live_traffic", False):
        return
    t0 = time.perf_counter()
    valu, seal = valu_pass(("k_eval_check_",), "syn_heavy", args.po2, device, 120.0)
    if not valu:
        h["roofline"] = None
        return
    clock = (seal or {}).get("clock_GHz") or 0.0
    step_s = h["ms_per_step"] * 1e-3 / max(1, int((line.get("config") or {}).get("ranks_per_gpu", 1)))
    h["roofline"] = {"kernel": "eval_check", "bound": "valu",
                     "valu": {"wave_instr_per_seal": valu["wave_instr_per_seal"], "launches_per_seal": valu["launches"],
                              "simd_cycles_per_seal": valu["simd_cycles"] * valu["launches"], "issue_frac": valu["issue_frac"],
                              "issue_peak_half_rate_class": HALF_RATE_ISSUE_PEAK, "clock_GHz": valu["clock_GHz"],
                              "instr_per_point": valu["wave_instr_per_seal"] * 64.0 / (4.0 * n),
                              "ms_per_seal_under_counters": valu["ms_per_seal_under_counters"]},
                     "seal_valu_wave_instr": seal["wave_instr"] if seal else None,
                     "seal_valu_issue_frac": seal["wave_instr"] / (step_s * clock * 1e9 * N_SIMDS) if seal and clock else None,
                     "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE around a child run of this command with "
                               "--circuit syn_heavy (one serial seal); the generated eval_check kernels (k_eval_check_*) summed over their parts",
                     "child_run_s": time.perf_counter() - t0}
    cfg = line.get("config")
    if isinstance(cfg, dict):
        cfg["syn_heavy_eval_instr_per_point"] = round(h["roofline"]["valu"]["instr_per_point"], 0)
        if h["roofline"]["valu"]["issue_frac"] is not None:
            cfg["syn_heavy_eval_valu_issue_frac"] = round(h["roofline"]["valu"]["issue_frac"], 4)
        if h["roofline"]["seal_valu_issue_frac"] is not None:
            cfg["syn_heavy_seal_valu_issue_frac"] = round(h["roofline"]["seal_valu_issue_frac"], 4)


def live_counters(kernels, circuit: str, po2: int, budget_s: float = 120.0, device: int = 0):
    """HBM bytes per launch of the kernels named `kernels` and VALU issue figures (theirs, and the whole seal's), measured NOW:
    three child runs — `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (separate passes; FETCH_SIZE doubled per the guide's gfx950
    correction; rocprofv3 reports KB) and `--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE`.  -> dict (missing parts are None)."""
    if isinstance(kernels, str):
        kernels = (kernels,)
    t0 = time.perf_counter()
    out = {"traffic": None, "traffic_source": None, "valu": None, "seal_valu": None}
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = budget_s - (time.perf_counter() - t0)
        rows = pmc_child((counter,), circuit, po2, device, left) if left > 15 else None
        vals = [c.get(counter, 0.0) * 1024.0 for name, _, _, c in (rows or []) if _matches(name, kernels)]
        if not vals:
            sums = None
            break
        sums[counter] = vals
    if sums:
        nl = len(sums["FETCH_SIZE"])
        out["traffic"] = (2.0 * sum(sums["FETCH_SIZE"]) + sum(sums["WRITE_SIZE"])) / nl
        out["traffic_launches"] = nl
        out["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel trace only) around "
                                 f"two child runs of this command with one serial seal each; FETCH x2 per the gfx950 correction; mean over {nl} launches")
    left = budget_s - (time.perf_counter() - t0)
    if left > 15:
        out["valu"], out["seal_valu"] = valu_pass(kernels, circuit, po2, device, left)
    out["child_runs_s"] = time.perf_counter() - t0
    return out


def by_op(records):
    """HIP-event records -> per Hal op.  NTT records are "<op>:<kernel>" per pass and the op's §8d bytes are charged to exactly ONE
    pass of every invocation, so: op time = sum over its passes, op bytes = sum, op invocations = calls of the passes that carry
    bytes.  A sub-kernel bracket is never a roofline candidate on its own (its bytes live with the parent op)."""
    ops = {}
    for p in records:
        o = ops.setdefault(p["name"].split(":")[0], {"name": p["name"].split(":")[0], "total_ms": 0.0, "alg_bytes": 0.0, "calls": 0, "launches": 0})
        o["total_ms"] += p["total_ms"]; o["alg_bytes"] += p["alg_bytes"]; o["launches"] += p["calls"]
        if p["alg_bytes"] > 0 or ":" not in p["name"]:
            o["calls"] += p["calls"]
    for o in ops.values():
        o["calls"] = max(1, o["calls"])
    return ops


def add_roofline(line, prof, ref, args, inflight, widths, n, device=0, live=True):
    """roofline{} for the dominant op + the per-kernel table, from the HIP-event brackets of the timed region (prof)
    and of one extra seal that ran alone on the GPU (ref)."""
    wa, wc, wd = widths
    unshared = {p["name"]: p for p in (ref or prof)}
    ops_unshared, ops_timed = by_op(unshared.values()), by_op(prof)
    # the dominant OP among those with algorithmic bytes (every Hal op has them; witness-generator brackets may not)
    cands = [o for o in ops_unshared.values() if o["alg_bytes"] > 0 and o["name"] in ops_timed] or list(ops_unshared.values())
    dom_u = max(cands, key=lambda o: o["total_ms"])
    dom_name = dom_u["name"]
    dom = ops_timed.get(dom_name, dom_u)
    per_launch_ms = dom["total_ms"] / dom["calls"]
    per_launch_ms_unshared = dom_u["total_ms"] / dom_u["calls"]
    per_launch_bytes = dom_u["alg_bytes"] / dom_u["calls"]
    ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    # HBM bytes per launch and VALU issue: measured now (child runs under rocprofv3 --pmc, separate passes) on this rank's GPU, else
    # from the committed PMC passes of an earlier run of this command (tools/pmc_summary.py)
    traffic, traffic_source, counters = None, None, None
    knames = TRAFFIC_KERNELS.get(dom_name)
    if knames and live and not getattr(args, "no_live_traffic", False):
        counters = live_counters(knames, args.circuit, args.po2, device=device)
        if counters.get("traffic") is not None:
            # an op invocation = dom_u["launches"] / dom_u["calls"] kernel launches (eval_check of a split circuit: one per part)
            traffic = counters["traffic"] * dom_u["launches"] / dom_u["calls"]
            traffic_source = counters["traffic_source"]
    for fn in () if traffic is not None else ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            kname = (knames or ("",))[0]
            if kname == "k_eval_check_":
                kname = "k_eval_check_" + args.circuit
            if kname in tj and args.po2 == PO2 and args.circuit == "syn_a":
                traffic = (tj[kname]["fetch_x2_bytes"] + tj[kname]["write_bytes"]) / tj[kname]["launches"]
                traffic_source = f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, not measured in this run)"
                break
        except Exception:
            continue
    # Primary figures = the kernel's own launch duration (HIP-event brackets of one seal that ran ALONE right after the timed
    # region: what `rocprofv3 --kernel-trace --stats` reports per dispatch, profiles/r0N_kernel_stats*.csv).  With several
    # seals in flight the brackets of the timed region also contain the time a launch spent queued behind the other streams'
    # kernels; those are kept as *_timed_region.
    ach_unshared = per_launch_bytes / (per_launch_ms_unshared * 1e-3) / 1e9
    line["roofline"] = {"bound": "hbm", "kernel": dom_name, "achieved": ach_unshared, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ach_unshared / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                        "avg_launch_ms": per_launch_ms_unshared,
                        "avg_launch_ms_timed_region": per_launch_ms, "achieved_timed_region": ach,
                        "alg_bytes_per_launch": per_launch_bytes,
                        "share_of_kernel_time": dom_u["total_ms"] / sum(p["total_ms"] for p in unshared.values()),
                        "launches_overlap": inflight > 1,
                        "measured_on": f"rank 0's GPU (device {device}), one seal alone after the timed region" if ref else "the timed region",
                        "note": "the dominant kernel is integer-VALU-bound by construction (Poseidon2: ~21 Montgomery products per absorbed "
                                "byte): its HBM fraction is reported as the contract asks, the roofline that BINDS it is VALU issue — "
                                "`valu.issue_frac` (wave-instructions per SIMD-cycle, live from SQ_INSTS_VALU / GRBM_GUI_ACTIVE) against "
                                "`valu.issue_peak_half_rate_class` — and `seal_valu_issue_frac` says the same for the whole seal at the "
                                "headline's rate; avg_launch_ms is the kernel's own duration (one seal alone on the GPU, live, after the timed region)"}
    r = line["roofline"]
    if counters and counters.get("valu"):
        v = counters["valu"]
        share = HALF_RATE_SHARE.get(dom_name)
        r["valu"] = {"wave_instr": v["wave_instr"], "simd_cycles": v["simd_cycles"], "issue_frac": v["issue_frac"],
                     "half_rate_share": share, "issue_peak_half_rate_class": HALF_RATE_ISSUE_PEAK,
                     "frac_of_half_rate_peak": v["issue_frac"] / HALF_RATE_ISSUE_PEAK if v["issue_frac"] else None,
                     "clock_GHz": v["clock_GHz"], "launches": v["launches"], "launch_ms_under_counters": v["launch_ms_under_counters"],
                     "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE (its own pass, kernel trace only) around a child run "
                               "with one serial seal; wave_instr and simd_cycles (= GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) are per launch; "
                               "half_rate_share is static (the kernel's opcode table: profiles/r03_hash_rows_isa_histogram.txt)"}
        if dom_name == "hash_rows" and v.get("simd_cycles") and v.get("launches"):
            # ... and against the ALGORITHMIC work of SURVEY.md §8d: a Poseidon2 permutation of width 24 is 1 356 Montgomery products
            # (8 full rounds x 24 x 4 for x^7, 21 partial rounds x (4 + 24 diagonal), M_ext in f64 counted as its multiplies) = 3
            # half-rate instructions each (4 issue cycles per wave64) + ~3 k full-rate additions (2 cycles): 22.3 k SIMD-cycles per
            # 64 permutations.  What the launches of one seal really spent per wave-permutation says how close the kernel is to THAT.
            perms = sum(-(-w // 16) for w in (wc, wd, wa, 16)) * 4 * n
            deg = n
            while deg > 256:
                perms += 4 * (4 * deg // 16)
                deg //= 16
            floor = 1356 * 3 * 4 + 3000 * 2
            cpw = v["simd_cycles"] * v["launches"] / (perms / 64.0)
            r["valu"].update(permutations_per_seal=perms, cycles_per_wave_permutation=cpw, algorithmic_floor_cycles=floor,
                             frac_of_algorithmic_floor=floor / cpw,
                             frac_of_algorithmic_floor_at_2p4GHz=(floor / cpw) * (v["clock_GHz"] / 2.4) if v.get("clock_GHz") else None)
    if counters and counters.get("seal_valu"):
        s = counters["seal_valu"]
        clock = s["clock_GHz"] or 0.0
        # (ranks that SHARE one GPU in a dry run each see 1 / ranks_per_gpu of it: the chip's rate is what counts)
        step_s = line["ms_per_step"] * 1e-3 / max(1, int((line.get("config") or {}).get("ranks_per_gpu", 1)))
        r["seal_valu_wave_instr"] = s["wave_instr"]
        r["seal_valu_issue_frac"] = s["wave_instr"] / (step_s * clock * 1e9 * N_SIMDS) if clock else None
        r["seal_valu_issue_frac_serial"] = s["issue_frac_serial"]
        r["seal_valu_note"] = (f"whole seal: {s['wave_instr'] / 1e9:.2f} G VALU wave-instructions ({s['dispatches']} dispatches, one serial seal under counters) over "
                               f"ms_per_step x {clock:.2f} GHz (the duration-weighted engine clock of that seal) x {N_SIMDS} SIMDs; "
                               f"{HALF_RATE_ISSUE_PEAK} per SIMD-cycle is the half-rate class's ceiling: at the headline's rate ({inflight} seals in flight) the chip "
                               f"issues VALU at about the rate the dominant kernel reaches alone")
    elif dom_name == "hash_rows":
        # no live counters (rocprofv3 unavailable / N > 1 with a failed rank): the modelled VALU view of the same kernel —
        # permutations per launch x modelled issue cycles per 64-lane permutation
        perms = sum(-(-w // 16) for w in (wc, wd, wa, 16)) * 4 * n
        deg = n
        while deg > 256:
            perms += 4 * (4 * deg // 16)
            deg //= 16
        cyc = 8 * 1990 + 7 * 1259 + 711 + 480
        per_seal_ms = dom_u["total_ms"] / (1 if ref else args.steps)
        r["valu_model"] = {"permutations_per_seal": perms, "model_cycles_per_wave_permutation": cyc,
                           "issue_utilisation_at_2p4GHz": (perms / 64.0) * cyc / (N_SIMDS * 2.4e9 * per_seal_ms * 1e-3)}
    if counters:
        r["counter_child_runs_s"] = counters.get("child_runs_s")
    # the same two numbers as scalars of `config` (a record that drops nested objects keeps them)
    cfg = line.get("config")
    if isinstance(cfg, dict):
        if isinstance(r.get("valu"), dict) and r["valu"].get("issue_frac") is not None:
            cfg["dominant_kernel_valu_issue_frac"] = round(r["valu"]["issue_frac"], 4)
        if r.get("seal_valu_issue_frac") is not None:
            cfg["seal_valu_issue_frac"] = round(r["seal_valu_issue_frac"], 4)
        if isinstance(r.get("valu"), dict) and r["valu"].get("frac_of_algorithmic_floor") is not None:
            cfg["dominant_valu_algorithmic_frac"] = round(r["valu"]["frac_of_algorithmic_floor"], 4)
        cfg["valu_issue_peak_half_rate_class"] = HALF_RATE_ISSUE_PEAK
    div = 1 if ref else args.steps
    line["kernels"] = [{"name": p["name"], "calls_per_seal": p["calls"] / div,
                        "ms_per_seal": p["total_ms"] / div,          # unshared (one seal alone on the GPU)
                        "ms_per_seal_timed_region": next((q["total_ms"] / args.steps for q in prof if q["name"] == p["name"]), None),
                        # §8d algorithmic bytes (operands once in, once out) / time; 0 for the later passes of a multi-pass op
                        "alg_GBps": (p["alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9) if p["total_ms"] > 0 else 0.0}
                       for p in sorted(unshared.values(), key=lambda p: -p["total_ms"])]
    # per Hal op (NTT records are "<op>:<kernel>" per pass): op totals with §8d bytes over the sum of the passes
    line["ops"] = [{"op": o["name"], "ms_per_seal": o["total_ms"] / div,
                    "alg_GBps": o["alg_bytes"] / (o["total_ms"] * 1e-3) / 1e9 if o["total_ms"] > 0 else 0.0}
                   for o in sorted(ops_unshared.values(), key=lambda o: -o["total_ms"])]
