set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
timeout 120 python -m zeth_amd.prover > gpurun_out/r2a/control_roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json gpurun_out/r2a/
timeout 300 python bench.py --steps 30 --warmup 2 --ingress host > gpurun_out/r2a/bench_syn_a.json 2> gpurun_out/r2a/bench_syn_a.err
timeout 300 python bench.py --circuit syn_heavy --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_heavy.json 2> gpurun_out/r2a/bench_heavy.err
timeout 300 python bench.py --config block --segments 48 --no-cpu-baseline > gpurun_out/r2a/bench_block.json 2> gpurun_out/r2a/bench_block.err
timeout 300 python bench.py --config succinct --segments 16 --no-cpu-baseline > gpurun_out/r2a/bench_succinct.json 2> gpurun_out/r2a/bench_succinct.err
ZKH_SHARE_GPUS=1 timeout 300 python bench.py --gpus 2 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_2rank.json 2> gpurun_out/r2a/bench_2rank.err
timeout 300 python tools/microbench.py > gpurun_out/r2a/microbench.jsonl 2> gpurun_out/r2a/microbench.err
tail -5 gpurun_out/r2a/pytest.log
head -c 600 gpurun_out/r2a/bench_syn_a.json
