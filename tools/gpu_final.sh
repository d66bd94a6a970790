# Round-end evidence: the whole -m gpu suite, the driver's bench command (+ its multi-rank form on one GPU), then everything profiles/ holds.
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default rc=$?" ) > $O/bench_default.time 2>&1
ZKH_SHARE_GPUS=1 timeout 600 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy > $O/bench_8rank.json 2> $O/bench_8rank.err
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null; python -m zeth_amd.circuits.p2_join /tmp/p2.desc > /dev/null
timeout 300 examples/prove_session --desc /tmp/syn_a.desc --join-desc /tmp/p2.desc --segments 256 > $O/prove_session_256.json 2> $O/prove_session.err
D=/tmp/zkr; mkdir -p $D; python -m zeth_amd.circuits.rec_verify $D > /dev/null; python -m zeth_amd.circuits.recursion $D/recursion.desc > /dev/null
LD_LIBRARY_PATH=$PWD/zeth_amd timeout 600 examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 1024 > $O/prove_session_recursion_1024.json 2>> $O/prove_session.err
timeout 900 python bench.py --config succinct > $O/bench_succinct_recursion.json 2> $O/bench_succinct_recursion.err; echo "bench_succinct rc=$?" >> $O/bench_default.time
timeout 900 python bench.py --config succinct --no-fused-lift > $O/bench_succinct_recursion_unfused.json 2>> $O/bench_succinct_recursion.err
timeout 900 python bench.py --config succinct --join-circuit p2_join > $O/bench_succinct_p2join.json 2>> $O/bench_succinct_recursion.err
( cd /tmp && LD_LIBRARY_PATH=$OLDPWD/zeth_amd timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_fold -o fold -- $OLDPWD/examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 64 --inflight 1 > /dev/null 2>&1 )
find $O/prof_fold -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fold_kernel_stats.csv; rm -rf $O/prof_fold
bash tools/gpu_big.sh > $O/big.txt 2>&1; cp gpurun_out/big/bench_po2_21.json $O/bench_po2_21.json; cp gpurun_out/big/bench_po2_22.json $O/bench_po2_22.json
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/bench_default.time; tail -1 $O/smoke.log > $O/smoke.txt
tail -3 $O/pytest.log; cat $O/bench_default.time; grep -v amdgpu $O/big.txt; cat $O/smoke.txt; cat $O/prove_session_recursion_1024.json; head -8 $O/fold_kernel_stats.csv | cut -c1-160; head -c 300 $O/bench_default.json; echo; head -c 200 $O/bench_8rank.json; echo; cat $O/prove_session_256.json
