set -u
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
for i in 1 2; do
timeout 200 python tools/microbench.py --only M3,M4 > $O/mb_base_$i.jsonl 2>> $O/mb.err
ZKH_LIBRARY=$PWD/.variants/hash_sched.so timeout 200 python tools/microbench.py --only M3,M4 > $O/mb_sched_$i.jsonl 2>> $O/mb.err
done
tail -2 $O/smoke.log; for f in $O/mb_*.jsonl; do echo $f; cut -c1-110 $f; done
