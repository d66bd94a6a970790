# Round-end evidence: the whole -m gpu suite, the driver's bench command, then everything profiles/ holds.
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
tail -3 $O/pytest.log; head -c 300 $O/bench_default.json
