set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -m zeth_amd.prover > $O/control_roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/
timeout 600 python tools/exp_codegen.py syn_heavy 48,72,96,128 48 > $O/exp_codegen_heavy.jsonl 2> $O/exp_codegen_heavy.err
timeout 300 python tools/exp_codegen.py syn_a 48,72,128,256 48,1000000 > $O/exp_codegen_syn_a.jsonl 2> $O/exp_codegen_syn_a.err
timeout 300 python tools/microbench.py > $O/microbench.jsonl 2> $O/microbench.err
ZKH_NTT_WIDE=1 timeout 300 python tools/microbench.py > $O/microbench_wide.jsonl 2> $O/microbench_wide.err
timeout 300 python bench.py --steps 30 --warmup 2 > $O/bench_syn_a.json 2> $O/bench_syn_a.err
ZKH_NTT_WIDE=1 timeout 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $O/bench_syn_a_wide.json 2> $O/bench_syn_a_wide.err
timeout 300 python bench.py --circuit syn_heavy --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_heavy.json 2> $O/bench_heavy.err
tail -4 $O/pytest.log; cat $O/exp_codegen_heavy.jsonl; grep -h '"M2"\|"M7"\|"M8"' $O/microbench.jsonl $O/microbench_wide.jsonl
