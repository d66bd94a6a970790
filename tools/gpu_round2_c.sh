set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=6 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -m zeth_amd.prover > $O/control_roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/
timeout 600 python tools/exp_codegen.py syn_heavy REGS=96 REGS=96,SOP=0 REGS=72 REGS=128 REGS=96,EPOCH=24 REGS=96,EPOCH=96 REGS=96,PART=6400 > $O/exp_codegen_heavy.jsonl 2> $O/exp_codegen_heavy.err
timeout 300 python tools/exp_codegen.py syn_a REGS=96 REGS=96,SOP=0 REGS=256,EPOCH=24 REGS=256,EPOCH=96 REGS=256,EPOCH=16 > $O/exp_codegen_syn_a.jsonl 2> $O/exp_codegen_syn_a.err
timeout 300 python bench.py --circuit syn_heavy --steps 12 --warmup 2 --inflight 4 --no-cpu-baseline > $O/bench_heavy_if4.json 2> $O/bench_heavy_if4.err
timeout 300 python bench.py --steps 32 --warmup 2 --inflight 4 --no-cpu-baseline > $O/bench_syn_a_if4.json 2> $O/bench_syn_a_if4.err
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
tail -4 $O/pytest.log; cat $O/exp_codegen_heavy.jsonl $O/exp_codegen_syn_a.jsonl
