# One GPU-box pass: the whole -m gpu suite, the per-op micro-benchmarks, the default bench line.
#     gpurun --timeout 1800 -- 'bash tools/gpu_check.sh <name>'   -> gpurun_out/<name>/
set -u
O=gpurun_out/${1:-check}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/microbench.py > $O/microbench.jsonl 2> $O/microbench.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log; cut -c1-150 $O/microbench.jsonl; head -c 300 $O/bench_default.json
