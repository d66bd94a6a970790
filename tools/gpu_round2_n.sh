set -u
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -x -k "hash or merkle or poseidon or seal or fold" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/microbench.py --only M3,M4 > $O/microbench.jsonl 2> $O/microbench.err
timeout 400 python bench.py --no-heavy > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log; cat $O/microbench.jsonl | cut -c1-200; head -c 300 $O/bench_default.json
