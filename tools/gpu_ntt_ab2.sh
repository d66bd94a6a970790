# Forward-NTT A/B (round 3, second pass): 32-byte runs / 512-lane workgroups (four per CU) vs the shipped 64-byte / 1024-lane tile
set -u
O=gpurun_out/${1:-ntt_ab2}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
for i in 1 2; do
  timeout 120 python tools/exp_ntt.py --po2 20 --tag shipped >> $L 2>> $O/err.txt
  ZKH_NTT_NARROW=1 timeout 120 python tools/exp_ntt.py --po2 20 --tag narrow32B >> $L 2>> $O/err.txt
done
ZKH_NTT_NARROW=1 timeout 120 python tools/exp_ntt.py --po2 20 --width 16 --tag narrow32B-w16 >> $L 2>> $O/err.txt
timeout 120 python tools/exp_ntt.py --po2 20 --width 16 --tag shipped-w16 >> $L 2>> $O/err.txt
cut -c1-330 $L; tail -3 $O/err.txt
