set -u
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
