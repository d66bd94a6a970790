# the native host (g++ only) running BASELINE config 5 with in-circuit verification, from program files
set -u
O=gpurun_out/rec; mkdir -p $O
export TMPDIR=/tmp
D=/tmp/zkr; mkdir -p $D
python -m zeth_amd.circuits.rec_verify $D > $O/programs.txt 2>&1
python -m zeth_amd.circuits.recursion $D/recursion.desc >> $O/programs.txt 2>&1
python -m zeth_amd.circuits.syn_air syn_a $D/syn_a.desc >> $O/programs.txt 2>&1
export LD_LIBRARY_PATH=$PWD/zeth_amd:${LD_LIBRARY_PATH:-}
timeout 900 examples/prove_session --desc $D/syn_a.desc --recursion-dir $D --segments ${1:-1024} > $O/prove_session_recursion.json 2> $O/prove_session_recursion.err
echo "rc=$?"; cat $O/prove_session_recursion.json; tail -n 3 $O/prove_session_recursion.err
