# rocprofv3 kernel stats of a 64-segment block + its recursive fold on one lane
set -u
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
D=/tmp/zkr; mkdir -p $D; python -m zeth_amd.circuits.rec_verify $D > /dev/null; python -m zeth_amd.circuits.recursion $D/recursion.desc > /dev/null; python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
( cd /tmp && LD_LIBRARY_PATH=$OLDPWD/zeth_amd timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_fold -o fold -- $OLDPWD/examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 64 --inflight 1 > /dev/null 2>&1; echo "rocprof rc=$?" )
find $O/prof_fold -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fold_kernel_stats.csv; rm -rf $O/prof_fold
head -12 $O/fold_kernel_stats.csv | cut -c1-150
