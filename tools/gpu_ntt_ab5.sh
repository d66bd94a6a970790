# Forward-NTT A/B (round 3, fifth pass): sign extension of the lazy butterflies' addend as one v_mad_i64_i32 (build -DZKH_SEXT_MAD)
set -u
O=gpurun_out/${1:-ntt_ab5}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
for i in 1 2 3; do
  timeout 120 python tools/exp_ntt.py --po2 20 --tag shipped >> $L 2>> $O/err.txt
  ZKH_LIBRARY=$PWD/.variants/libzkhal_sextmad.so timeout 120 python tools/exp_ntt.py --po2 20 --tag sext-mad >> $L 2>> $O/err.txt
done
cut -c1-330 $L; tail -3 $O/err.txt
