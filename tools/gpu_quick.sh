# A filtered -m gpu run:  gpurun -- 'bash tools/gpu_quick.sh <name> -k "expr"'
set -u
O=gpurun_out/${1:-q}; mkdir -p $O; shift
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x "$@" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
