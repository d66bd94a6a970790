#!/usr/bin/env python3
"""Per-op micro-benchmarks M1..M8 of SURVEY.md §8d (+ M9: the trace-driven witness, row f1) on one MI355X, through the C ABI (HipHal).

Each line of output is one JSON object: the op, its shape, the average wall time of one call (stream drained on both
sides of `reps` back-to-back calls), its ALGORITHMIC bytes (SURVEY.md §8a "B_alg": inputs read once + outputs written
once) and what fraction of the 8.0 TB/s HBM3E peak (and of the 6.29 TB/s measured copy ceiling) that is.  The
Poseidon2 ops and the NTTs are integer-VALU-bound on gfx950, so they also carry `valu_frac`: modelled
issue cycles (DESIGN.md's per-permutation / per-butterfly counts, from tools/ubench_valu.hip) / (CUs*4 SIMDs*clock*t).

GPU only; inputs are random field elements generated on the host and uploaded before the clock starts.
    python tools/microbench.py [--po2 20] [--reps 5] [--only M3,M4]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from zeth_amd.circuits import syn_air  # noqa: E402
from zeth_amd.circuits.desc import Circuit as Desc  # noqa: E402
from zeth_amd.hal import HipHal  # noqa: E402

P = 2013265921
HBM_PEAK, HBM_COPY = 8.0e12, 6.29e12
SIMDS, CLOCK = 256 * 4, 2.4e9                 # 256 CUs x 4 SIMDs, peak engine clock
PERM_CYC = 8 * 1990 + 7 * 1259 + 711 + 480   # issue cycles of one wave64 over 64 Poseidon2 permutations
BFLY_CYC = 34                                 # mul_mod 18 + add_mod 8 + sub_mod 8


def rand_fp(rng, size):
    return rng.integers(0, P, size=size, dtype=np.uint64).astype(np.uint32)


def upload(hal, rng, name, words):
    """Random field elements, uploaded in 64M-word pieces (keeps the host staging small)."""
    buf = hal.alloc_elem(name, words)
    step = 1 << 26
    for off in range(0, words, step):
        buf.write(rand_fp(rng, min(step, words - off)), off)
    return buf


def timed(hal, fn, reps):
    fn()
    hal.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    hal.sync()
    return (time.perf_counter() - t0) / reps


def line(tag, op, shape, dt, alg_bytes, valu_cycles=None):
    rec = {"bench": tag, "op": op, "shape": shape, "ms": round(dt * 1e3, 4), "alg_GB": round(alg_bytes / 1e9, 4),
           "GB_per_s": round(alg_bytes / dt / 1e9, 1), "hbm_frac_8TBs": round(alg_bytes / dt / HBM_PEAK, 4),
           "hbm_frac_6.29TBs": round(alg_bytes / dt / HBM_COPY, 4)}
    if valu_cycles is not None:
        rec["valu_frac"] = round(valu_cycles / (SIMDS * CLOCK * dt), 4)
        rec["bound"] = "valu"
    else:
        rec["bound"] = "hbm"
    print(json.dumps(rec), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--width", type=int, default=208, help="columns of the group under test (SYN-A data group)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    want = lambda m: not only or m in only  # noqa: E731

    hal = HipHal(0)
    rng = np.random.default_rng(0xB0B)
    n, w = 1 << args.po2, args.width
    dom = 4 * n
    wave_perms = lambda perms: perms / 64 * PERM_CYC  # noqa: E731

    if want("M1") or want("M2"):
        io = upload(hal, rng, "m1", w * n)
        if want("M1"):
            dt = timed(hal, lambda: hal.batch_interpolate_ntt(io, w), args.reps)
            line("M1", "batch_interpolate_ntt", f"{w} x 2^{args.po2}", dt, 8 * w * n,
                 w * (n // 2) * args.po2 / 64 * BFLY_CYC)
            dt = timed(hal, lambda: hal.batch_interpolate_ntt_zk_shift(io, w), args.reps)
            line("M1b", "batch_interpolate_ntt+zk_shift (fused)", f"{w} x 2^{args.po2}", dt, 8 * w * n,
                 w * (n // 2) * args.po2 / 64 * BFLY_CYC)
        if want("M2"):
            ev = hal.alloc_elem("m2", w * dom)
            dt = timed(hal, lambda: hal.batch_expand_into_evaluate_ntt(ev, io, w, 2), args.reps)
            line("M2", "batch_expand_into_evaluate_ntt", f"{w} x 2^{args.po2} -> 2^{args.po2 + 2}", dt, 20 * w * n,
                 w * (dom // 2) * args.po2 / 64 * BFLY_CYC)
            del ev
        del io

    if want("M3") or want("M4"):
        nodes = hal.alloc_digest("nodes", 2 * dom)
        if want("M3"):
            mat = upload(hal, rng, "m3", w * dom)
            leaves = nodes.slice(8 * dom, 8 * dom)
            dt = timed(hal, lambda: hal.hash_rows(leaves, mat), args.reps)
            line("M3", "hash_rows", f"{w} cols x 2^{args.po2 + 2} rows", dt, 4 * w * dom + 32 * dom,
                 wave_perms(dom * -(-w // 16)))
            del mat, leaves
        if want("M4"):
            nodes.slice(8 * dom, 8 * dom).write(rand_fp(rng, 8 * dom))
            dt = timed(hal, lambda: hal.merkle_fold_all(nodes, dom), args.reps)
            line("M4", "hash_fold 2^%d -> 1 (merkle_fold_all)" % (args.po2 + 2), f"{dom} leaves", dt, 96 * dom,
                 wave_perms(dom - 1))
        del nodes

    if want("M5"):
        for m in (n, n >> 4, n >> 8):
            if m < 16:
                continue
            src = upload(hal, rng, "m5", 4 * m)
            dst = hal.alloc_elem("m5o", 4 * (m // 16))
            mix = rand_fp(rng, 4)
            dt = timed(hal, lambda: hal.fri_fold(dst, src, mix), args.reps)
            line("M5", "fri_fold", f"2^{m.bit_length() - 1} -> 2^{m.bit_length() - 5} ext", dt, 16 * m + m)
            del src, dst

    if want("M6") or want("M7"):
        coeffs = upload(hal, rng, "m6", w * n)
        if want("M6"):
            ncombo = 12
            # columns of one combo are adjacent, as a TapSet lays registers out (the accumulator is flushed on a change)
            combos = hal.copy_from("combos", (np.arange(w, dtype=np.uint32) * ncombo // w))
            out = hal.alloc_extelem("m6o", ncombo * n)
            ms, mx = rand_fp(rng, 4), rand_fp(rng, 4)
            dt = timed(hal, lambda: hal.mix_poly_coeffs(out, ms, mx, coeffs, combos, w, n), args.reps)
            line("M6", "mix_poly_coeffs", f"{w} x 2^{args.po2} -> {ncombo} combos", dt, 4 * w * n + 32 * ncombo * n)
            del out, combos
        if want("M7"):
            k = 256
            which = hal.copy_from("which", (np.arange(k, dtype=np.uint32) * 13) % w)
            xs = hal.copy_from("xs", rand_fp(rng, 4 * k))
            out = hal.alloc_extelem("m7o", k)
            dt = timed(hal, lambda: hal.batch_evaluate_any(coeffs, w, which, xs, out), args.reps)
            line("M7", "batch_evaluate_any", f"{k} taps over {w} x 2^{args.po2}", dt, 4 * k * n)
            del which, xs, out
            # a real tap set: every column at backs 0..4 (runs of 5 equal `which`): the column is streamed once per run
            k5 = 5 * w
            which = hal.copy_from("which", np.repeat(np.arange(w, dtype=np.uint32), 5))
            xs = hal.copy_from("xs", rand_fp(rng, 4 * k5))
            out = hal.alloc_extelem("m7o", k5)
            dt = timed(hal, lambda: hal.batch_evaluate_any(coeffs, w, which, xs, out), args.reps)
            line("M7t", "batch_evaluate_any, 5 taps per column", f"{k5} taps over {w} x 2^{args.po2}", dt, 4 * w * n)
            del which, xs, out
        del coeffs

    if want("M8"):
        desc = syn_air.syn_a()
        d = Desc.parse(desc)
        circ = hal.load_circuit(desc)
        groups = [upload(hal, rng, f"g{i}", gw * dom) for i, gw in enumerate(d.group_sizes)]
        globals_ = [hal.copy_from(f"gl{i}", rand_fp(rng, max(1, gs))) for i, gs in enumerate(d.global_sizes)]
        check = hal.alloc_elem("check", 4 * dom)
        mix = rand_fp(rng, 4)
        dt = timed(hal, lambda: circ.eval_check(check, groups, globals_, mix, args.po2), args.reps)
        line("M8", "eval_check (SYN-A, compiled kernel)" if circ.has_compiled_kernel() else "eval_check (interpreter)",
             f"{sum(d.group_sizes)} cols x 2^{args.po2 + 2} points", dt, 4 * sum(d.group_sizes) * dom + 16 * dom)
        dt = timed(hal, lambda: circ.eval_check(check, groups, globals_, mix, args.po2, use_interpreter=True), 1)
        line("M8i", "eval_check (SYN-A, step-list interpreter)", f"{sum(d.group_sizes)} cols x 2^{args.po2 + 2} points",
             dt, 4 * sum(d.group_sizes) * dom + 16 * dom)
        # the same evaluated groups under the heavy constraint system (same widths): VALU-bound generated kernels
        from zeth_amd.circuits import syn_heavy
        hdesc = syn_heavy.syn_heavy()
        hd = Desc.parse(hdesc)
        hcirc = hal.load_circuit(hdesc)
        dt = timed(hal, lambda: hcirc.eval_check(check, groups, globals_, mix, args.po2), args.reps)
        line("M8h", f"eval_check (SYN-HEAVY: {len(hd.steps)} steps, {len(hd.taps)} taps, {hcirc.compiled_parts()} generated kernels)",
             f"{sum(hd.group_sizes)} cols x 2^{args.po2 + 2} points", dt, 4 * sum(hd.group_sizes) * dom + 16 * dom)
        dt = timed(hal, lambda: hcirc.eval_check(check, groups, globals_, mix, args.po2, use_interpreter=True), 1)
        line("M8hi", "eval_check (SYN-HEAVY, step-list interpreter)", f"{sum(hd.group_sizes)} cols x 2^{args.po2 + 2} points",
             dt, 4 * sum(hd.group_sizes) * dom + 16 * dom)
    if want("M9"):
        # row f1: the trace-driven witness — host preflight (sequential, one core), then upload + row fill + scan + scatter on the GPU
        from zeth_amd import hal as H
        desc = syn_air.syn_a()
        d = Desc.parse(desc)
        circ = hal.load_circuit(desc)
        wa, wc, wd = d.group_sizes
        A = n - 1994
        pinned = hal.host_alloc(4 * A)
        t0 = time.perf_counter()
        _, ram, cpu_s = H.syn_preflight(0x5EED0000, args.po2, records=pinned)
        line("M9p", "syn_preflight (host, one core: the sequential producer)", f"2^{args.po2} cycles -> {16 * A / 1e6:.1f} MB of records",
             time.perf_counter() - t0, 16 * A)
        drec, data = hal.alloc("records", 4 * A), hal.alloc_elem("data", wd * n)

        def fill():
            hal.write_async(drec, pinned)
            hal.syn_witgen_trace(circ, args.po2, 1994, 0x2E80, drec, ram, None, data)
        dt = timed(hal, fill, args.reps)
        line("M9", "upload of the compact trace + k_syn_rowfill + running-sum scan + preload scatter", f"{wd} x 2^{args.po2} data group from 16-byte records",
             dt, 16 * A + 4 * wd * n)
        hal.sync()
        hal.host_free(pinned)
    hal.close()


if __name__ == "__main__":
    main()
