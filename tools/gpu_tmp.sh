set -u
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -x -k "random_circuits or syn_heavy or eval_check or golden or baseline" ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 200 python tools/microbench.py --only M8,M8h > $O/mb.jsonl 2> $O/mb.err
tail -3 $O/pytest.log; cut -c1-150 $O/mb.jsonl
