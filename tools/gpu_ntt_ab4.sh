# Forward-NTT A/B (round 3, fourth pass): 2^22 split 14 + 8 (k_ntt_low14 + the 256-row strided kernel) vs the shipped 12 + 10
set -u
O=gpurun_out/${1:-ntt_ab4}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
for i in 1 2; do
  timeout 120 python tools/exp_ntt.py --po2 20 --tag shipped-12+10 >> $L 2>> $O/err.txt
  ZKH_NTT_SPLIT148=1 timeout 120 python tools/exp_ntt.py --po2 20 --tag split-14+8 >> $L 2>> $O/err.txt
done
ZKH_NTT_SPLIT148=1 timeout 120 python tools/exp_ntt.py --po2 20 --width 16 --tag split-14+8-w16 >> $L 2>> $O/err.txt
cut -c1-330 $L; tail -3 $O/err.txt
