# Forward-NTT A/B (round 3): twiddle matrix on/off, column-fast block order on/off, sign extension by mad.
set -u
O=gpurun_out/${1:-ntt_ab}; mkdir -p $O
export TMPDIR=/tmp
L=$O/ab.jsonl; : > $L
for po2 in 20 18; do
  timeout 120 python tools/exp_ntt.py --po2 $po2 --tag matrix+colfast >> $L 2>> $O/err.txt
  ZKH_NTT_NO_COLFAST=1 timeout 120 python tools/exp_ntt.py --po2 $po2 --tag matrix >> $L 2>> $O/err.txt
  ZKH_NTT_NO_MATRIX=1 timeout 120 python tools/exp_ntt.py --po2 $po2 --tag round2 >> $L 2>> $O/err.txt
  ZKH_LIBRARY=$PWD/.variants/libzkhal_sextmad.so timeout 120 python tools/exp_ntt.py --po2 $po2 --tag matrix+colfast+sextmad >> $L 2>> $O/err.txt
done
( time timeout 600 python -m pytest tests -m gpu -q -x -k "ntt or expand or baseline_shape or seal or golden" ) > $O/pytest.log 2>&1
cut -c1-420 $L; tail -4 $O/pytest.log; tail -3 $O/err.txt
