# Round-3: eval_check locality ordering A/B + exactness tests
set -u
O=gpurun_out/${1:-r3e}; mkdir -p $O
export TMPDIR=/tmp
L=$O/eval_check_ab.jsonl; : > $L
for c in "syn_heavy 20" "keccak_f 14" "syn_a 20"; do
  timeout 200 python tools/exp_eval_check.py $c >> $L 2>> $O/err.txt
  ZKH_LIBRARY=$PWD/.variants/libzkhal_noloc.so timeout 200 python tools/exp_eval_check.py $c >> $L 2>> $O/err.txt
done
( time timeout 1500 python -m pytest tests/test_fuzz_gpu.py tests/test_prove_gpu.py tests/test_keccak_circuit.py tests/test_round2_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cat $L; tail -8 $O/pytest.log; tail -3 $O/err.txt
