# Round-3: P2-JOIN GPU tests + control roots regeneration + succinct bench (S = 128)
set -u
O=gpurun_out/${1:-r3d}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m zeth_amd.prover > $O/roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/control_roots.json
( time timeout 1200 python -m pytest tests/test_p2_join.py tests/test_round3_gpu.py tests/test_keccak_circuit.py -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --config succinct --segments 128 > $O/bench_succinct128.json 2> $O/bench_succinct128.err
tail -12 $O/pytest.log; tail -2 $O/roots.log; head -c 1800 $O/bench_succinct128.json; tail -3 $O/bench_succinct128.err
