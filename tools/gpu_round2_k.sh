set -u
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
ZKH_SHARE_GPUS=1 timeout 400 python bench.py --gpus 2 --config succinct --segments 24 --join-po2 16 --po2 18 --no-cpu-baseline > $O/bench_succinct_2rank.json 2> $O/bench_succinct_2rank.err
ZKH_SHARE_GPUS=1 timeout 400 python bench.py --gpus 2 --config block --segments 32 --po2 18 --no-cpu-baseline > $O/bench_block_2rank.json 2> $O/bench_block_2rank.err
ZKH_SHARE_GPUS=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_torchrun_2rank.json 2> $O/bench_torchrun_2rank.err
tail -3 $O/pytest.log; for f in $O/bench_*2rank.json; do echo $f; head -c 500 $f; echo; done; tail -3 $O/bench_succinct_2rank.err
