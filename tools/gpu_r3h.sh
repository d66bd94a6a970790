# Round-3: eval_check generator A/B with real (random) data: mix-power epochs x locality ordering
set -u
O=gpurun_out/${1:-r3h}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
for lib in default l0e16 l1e16 l1e32 noloc; do
  if [ $lib = default ]; then unset ZKH_LIBRARY; else export ZKH_LIBRARY=$PWD/.variants/libzkhal_$lib.so; fi
  echo "{\"lib\": \"$lib\"}" >> $L
  timeout 300 python tools/microbench.py --only M8h,M8 >> $L 2>> $O/err.txt
  timeout 200 python tools/exp_eval_check.py keccak_f 14 >> $L 2>> $O/err.txt
done
unset ZKH_LIBRARY
cut -c1-260 $L; tail -3 $O/err.txt
