# Round-3 first pass: default bench line (certified + block leg), 8 ranks on ONE GPU dry run, microbench.
set -u
O=gpurun_out/${1:-r3a}; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" >> $O/bench_default.err
ZKH_SHARE_GPUS=1 timeout 600 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy > $O/bench_8rank.json 2> $O/bench_8rank.err
echo "8rank rc=$?" >> $O/bench_8rank.err
timeout 300 python tools/microbench.py > $O/microbench.jsonl 2> $O/microbench.err
head -c 1500 $O/bench_default.json; echo; tail -3 $O/bench_default.err; head -c 1200 $O/bench_8rank.json; echo; tail -5 $O/bench_8rank.err
