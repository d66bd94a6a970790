set -u
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/microbench.py --only M8 > $O/microbench_m8.jsonl 2> $O/microbench.err
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log; cut -c1-170 $O/microbench_m8.jsonl
