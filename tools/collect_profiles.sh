#!/bin/bash
# Collect everything profiles/ holds, on the GPU box, from the repo root:
#     bash tools/collect_profiles.sh gpurun_out/profN
# Every step runs under its own `timeout` (a counter pass that wedges must not eat the GPU budget).
# Counter passes run on their own (--pmc never together with sys/hip/hsa tracing), one counter set per pass.
# ONLY="heavy" (or "syn_a", "configs", "micro", "cpp"; several allowed) restricts the run to those sections.
set -u
want() { [ -z "${ONLY:-}" ] || [[ " $ONLY " == *" $1 "* ]]; }
OUT=$(realpath -m "${1:-gpurun_out/prof}"); mkdir -p "$OUT"
ROOT=$PWD; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 12 --warmup 2 --no-live-traffic"
SHORT="python $ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-prof --no-heavy --no-resident --no-block --no-certify --no-live-traffic"
python -c "from zeth_amd import build; build.ensure_built(); build.build_examples()" > /dev/null 2>&1 || echo "collect_profiles: build failed"   # the box gets source only
cd /tmp
# ---- BASELINE config 2 (SYN-A): default (3 seals in flight), serial, PCIe-inclusive ----
want syn_a && timeout 300 $BENCH --ingress host > "$OUT/bench.json" 2> "$OUT/bench.err"
want syn_a && timeout 300 $BENCH --inflight 1 --no-cpu-baseline --no-heavy --no-resident > "$OUT/bench_serial.json" 2>> "$OUT/bench.err"
# ---- the heavy constraint system, same shape ----
want heavy && timeout 300 $BENCH --circuit syn_heavy --no-cpu-baseline > "$OUT/bench_heavy.json" 2>> "$OUT/bench.err"
want heavy && timeout 300 $BENCH --circuit syn_heavy --inflight 1 --no-cpu-baseline > "$OUT/bench_heavy_serial.json" 2>> "$OUT/bench.err"
# ---- BASELINE configs 3 and 5 restated: S = 256 distinct segments; S = 1024 + join tree ----
want configs && timeout 600 python $ROOT/bench.py --config block --segments 256 --no-cpu-baseline > "$OUT/bench_block.json" 2>> "$OUT/bench.err"
want configs && timeout 900 python $ROOT/bench.py --config succinct --segments 1024 --no-cpu-baseline > "$OUT/bench_succinct.json" 2>> "$OUT/bench.err"
# ---- rocprofv3: kernel stats (default + serial + heavy serial), then counters, each on its own ----
want syn_a && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --no-cpu-baseline --no-heavy --no-resident --no-block > /dev/null 2>&1
want syn_a && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -o bench -- $BENCH --inflight 1 --no-cpu-baseline --no-heavy --no-resident --no-block > /dev/null 2>&1
want heavy && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_heavy_serial" -o bench -- $BENCH --circuit syn_heavy --inflight 1 --no-cpu-baseline --no-resident --no-block > /dev/null 2>&1
want syn_a && timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o bench -- $SHORT > /dev/null 2>&1
want syn_a && timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o bench -- $SHORT > /dev/null 2>&1
want syn_a && timeout 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace \
    --output-format csv -d "$OUT/sq" -o bench -- $SHORT > /dev/null 2>&1
want heavy && timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_heavy" -o bench -- $SHORT --circuit syn_heavy > /dev/null 2>&1
want heavy && timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_heavy" -o bench -- $SHORT --circuit syn_heavy > /dev/null 2>&1
want heavy && timeout 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace \
    --output-format csv -d "$OUT/sq_heavy" -o bench -- $SHORT --circuit syn_heavy > /dev/null 2>&1
cd "$ROOT"
want syn_a && python tools/pmc_summary.py "$OUT/fetch" "$OUT/write" "$OUT/traffic.json" > "$OUT/pmc_summary.txt" 2>&1
want syn_a && python tools/sq_summary.py "$OUT/sq" > "$OUT/sq_counters.txt" 2>&1
want heavy && python tools/pmc_summary.py "$OUT/fetch_heavy" "$OUT/write_heavy" "$OUT/traffic_heavy.json" > "$OUT/pmc_summary_heavy.txt" 2>&1
want heavy && python tools/sq_summary.py "$OUT/sq_heavy" > "$OUT/sq_counters_heavy.txt" 2>&1
for d in stats stats_serial stats_heavy_serial; do f=$(ls "$OUT/$d"/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_$d.csv"; done
want micro && timeout 300 python tools/microbench.py > "$OUT/microbench.jsonl" 2> "$OUT/microbench.err"
# ---- the g++-only host driver over the C ABI ----
want cpp && python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
want cpp && LD_LIBRARY_PATH=$ROOT/zeth_amd timeout 300 examples/seal_segments --desc /tmp/syn_a.desc --po2 20 --segments 48 --inflight 3 > "$OUT/cpp_driver.json" 2> "$OUT/cpp_driver.err"
# raw rocprof directories are large: keep the summaries only
rm -rf "$OUT"/stats "$OUT"/stats_serial "$OUT"/stats_heavy_serial "$OUT"/fetch "$OUT"/write "$OUT"/sq "$OUT"/fetch_heavy "$OUT"/write_heavy "$OUT"/sq_heavy
ls -la "$OUT"
