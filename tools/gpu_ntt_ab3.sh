# Forward-NTT A/B (round 3, third pass): persistent, software-pipelined strided pass (ZKH_NTT_PERSIST=<workgroups>)
set -u
O=gpurun_out/${1:-ntt_ab3}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
timeout 120 python tools/exp_ntt.py --po2 20 --tag shipped >> $L 2>> $O/err.txt
for g in 256 512 208; do
  ZKH_NTT_PERSIST=$g timeout 120 python tools/exp_ntt.py --po2 20 --tag persist$g >> $L 2>> $O/err.txt
done
timeout 120 python tools/exp_ntt.py --po2 20 --tag shipped >> $L 2>> $O/err.txt
ZKH_NTT_PERSIST=256 timeout 120 python tools/exp_ntt.py --po2 20 --width 16 --tag persist256-w16 >> $L 2>> $O/err.txt
cut -c1-330 $L; tail -3 $O/err.txt
