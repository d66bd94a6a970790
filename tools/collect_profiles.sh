#!/bin/bash
# Collect everything profiles/ holds, on the GPU box, from the repo root:
#     bash tools/collect_profiles.sh gpurun_out/profN
# Every step runs under its own `timeout` (a counter pass that wedges must not eat the GPU budget).
# Counter passes run on their own (--pmc never together with sys/hip/hsa tracing), one counter set per pass.
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}"); mkdir -p "$OUT"
ROOT=$PWD; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 12 --warmup 2"
SHORT="python $ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-prof"
cd /tmp
timeout 300 $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 $BENCH --inflight 1 --no-cpu-baseline > "$OUT/bench_serial.json" 2>> "$OUT/bench.err"
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --no-cpu-baseline > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -o bench -- $BENCH --inflight 1 --no-cpu-baseline > /dev/null 2>&1
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o bench -- $SHORT > /dev/null 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o bench -- $SHORT > /dev/null 2>&1
timeout 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace \
    --output-format csv -d "$OUT/sq" -o bench -- $SHORT > /dev/null 2>&1
cd "$ROOT"
python tools/pmc_summary.py "$OUT/fetch" "$OUT/write" "$OUT/traffic.json" > "$OUT/pmc_summary.txt" 2>&1
python tools/sq_summary.py "$OUT/sq" > "$OUT/sq_counters.txt" 2>&1
timeout 300 python tools/microbench.py > "$OUT/microbench.jsonl" 2> "$OUT/microbench.err"
ls -la "$OUT"
