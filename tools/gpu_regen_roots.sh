# After a change of the hash tables or of a shipped circuit: regenerate zeth_amd/circuits/control_roots.json on the GPU
# box (the code-group commitments are computed by the library itself), bring it back through gpurun_out/, run the suite.
#     gpurun -- 'bash tools/gpu_regen_roots.sh <name>' && cp gpurun_out/<name>/control_roots.json zeth_amd/circuits/
set -u
O=gpurun_out/${1:-roots}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m zeth_amd.prover > $O/regen.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/control_roots.json
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/regen.log; tail -4 $O/pytest.log
