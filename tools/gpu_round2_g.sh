set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py --steps 12 --no-cpu-baseline --inflight 1 > $O/bench_serial.json 2> $O/bench_serial.err
timeout 400 python bench.py --steps 30 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log
