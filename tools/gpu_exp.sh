# Generator knob sweep on the GPU box:  gpurun -- 'bash tools/gpu_exp.sh <name>'  -> gpurun_out/<name>/exp_*.jsonl
set -u
O=gpurun_out/${1:-exp}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/exp_codegen.py syn_heavy REGS=96 LAZY=0 REGS=128 REGS=80 PART=4800 PART=2400 PREFETCH=2 PREFETCH=8 EPOCH=96 > $O/exp_heavy.jsonl 2> $O/exp.err
timeout 200 python tools/exp_codegen.py syn_a REGS=96 LAZY=0 PREFETCH=8 > $O/exp_syn_a.jsonl 2>> $O/exp.err
cat $O/exp_heavy.jsonl $O/exp_syn_a.jsonl; tail -3 $O/exp.err
