set -u
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/microbench.py --only M7 > $O/microbench_m7.jsonl 2> $O/microbench.err || timeout 300 python tools/microbench.py > $O/microbench_m7.jsonl 2>> $O/microbench.err
timeout 600 python tools/exp_codegen.py syn_heavy REGS=96 REGS=96,PREFETCH=1 REGS=96,PREFETCH=2 REGS=96,PREFETCH=4 REGS=128,PREFETCH=2 REGS=72,PREFETCH=2 > $O/exp_prefetch_heavy.jsonl 2> $O/exp_prefetch_heavy.err
timeout 300 python tools/exp_codegen.py syn_a REGS=96 REGS=96,PREFETCH=1 REGS=96,PREFETCH=2 REGS=96,PREFETCH=4 > $O/exp_prefetch_syn_a.jsonl 2> $O/exp_prefetch_syn_a.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --circuit syn_heavy --steps 12 --no-cpu-baseline > $O/bench_heavy.json 2> $O/bench_heavy.err
ZKH_SHARE_GPUS=1 timeout 300 python bench.py --gpus 2 --steps 8 --warmup 1 --no-cpu-baseline --no-heavy > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -3 $O/pytest.log; cat $O/microbench_m7.jsonl | cut -c1-160; cat $O/exp_prefetch_heavy.jsonl $O/exp_prefetch_syn_a.jsonl
