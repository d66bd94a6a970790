set -u
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --durations=6 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -m zeth_amd.prover > $O/control_roots.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python tools/exp_codegen.py syn_a REGS=96 REGS=96,EPOCH=8 REGS=96,EPOCH=12 REGS=96,EPOCH=32 REGS=40 > $O/exp_codegen_syn_a.jsonl 2> $O/exp_codegen_syn_a.err
timeout 300 python tools/exp_codegen.py syn_heavy REGS=96 REGS=128 REGS=160 > $O/exp_codegen_heavy.jsonl 2> $O/exp_codegen_heavy.err
mkdir -p /tmp/rc && timeout 120 examples/seal_segments --desc <(python -c "import sys; from zeth_amd.circuits import syn_air; import numpy as np; sys.stdout.buffer.write(np.asarray(syn_air.syn_small(),dtype='<u4').tobytes())") --po2 14 --segments 6 --receipts-dir /tmp/rc > $O/cpp_receipts.json 2> $O/cpp_receipts.err; ls -la /tmp/rc >> $O/cpp_receipts.json
bash tools/collect_profiles.sh $O/prof > $O/collect.log 2>&1
tail -4 $O/pytest.log; cat $O/exp_codegen_heavy.jsonl $O/exp_codegen_syn_a.jsonl; head -c 400 $O/bench_default.json
