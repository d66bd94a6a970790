set -u
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/exp_codegen.py syn_heavy REGS=96,PREFETCH=4 REGS=96,PREFETCH=8 REGS=96,PREFETCH=16 REGS=128,PREFETCH=8 REGS=64,PREFETCH=4 REGS=96,PREFETCH=4,EPOCH=24 > $O/exp_prefetch_heavy.jsonl 2> $O/exp_prefetch_heavy.err
timeout 300 python tools/exp_codegen.py syn_a REGS=96,PREFETCH=4 REGS=96,PREFETCH=8 REGS=96,PREFETCH=16 REGS=160,PREFETCH=16 > $O/exp_prefetch_syn_a.jsonl 2> $O/exp_prefetch_syn_a.err
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log; cat $O/exp_prefetch_heavy.jsonl $O/exp_prefetch_syn_a.jsonl; head -c 300 $O/bench_default.json
