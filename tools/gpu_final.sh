# Round-end evidence: the whole -m gpu suite, the driver's bench command (+ its multi-rank form on one GPU), then everything profiles/ holds.
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
ZKH_SHARE_GPUS=1 timeout 600 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy > $O/bench_8rank.json 2> $O/bench_8rank.err
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null; python -m zeth_amd.circuits.p2_join /tmp/p2.desc > /dev/null
timeout 300 examples/prove_session --desc /tmp/syn_a.desc --join-desc /tmp/p2.desc --segments 256 > $O/prove_session_256.json 2> $O/prove_session.err
tail -3 $O/pytest.log; head -c 300 $O/bench_default.json; echo; head -c 200 $O/bench_8rank.json; echo; cat $O/prove_session_256.json
