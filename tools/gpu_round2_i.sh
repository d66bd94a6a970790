set -u
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/exp_codegen.py syn_heavy REGS=96 REGS=96,FLAGS=-DZKH_REDUCE_MIN > $O/exp_min_heavy.jsonl 2> $O/exp_min_heavy.err
timeout 300 python tools/exp_codegen.py syn_a REGS=96 REGS=96,FLAGS=-DZKH_REDUCE_MIN > $O/exp_min_syn_a.jsonl 2> $O/exp_min_syn_a.err
cat $O/exp_min_heavy.jsonl $O/exp_min_syn_a.jsonl; tail -3 $O/exp_min_heavy.err
