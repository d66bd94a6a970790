# Round-3: keccak circuit GPU tests + control roots regeneration
set -u
O=gpurun_out/${1:-r3c}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_keccak_circuit.py -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python -m zeth_amd.prover > $O/roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/control_roots.json
tail -30 $O/pytest.log; tail -3 $O/roots.log
