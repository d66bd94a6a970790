# Round-3: the new trait-only prover tests (+ anything else passed as pytest args)
set -u
O=gpurun_out/${1:-r3b}; mkdir -p $O; shift
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_round3_gpu.py -m gpu -q "$@" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
