set -u
O=gpurun_out/r2w; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -x -k "resident or prove_begin or seal" ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
ZKH_SHARE_GPUS=1 timeout 300 python bench.py --gpus 2 --steps 12 --warmup 2 --no-cpu-baseline --no-heavy > $O/two_ranks_one_gpu.json 2> $O/two.err
tail -4 $O/pytest.log; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); print(d["value"], d["syn_heavy"]["segments_per_s"], d["code_group_resident"])
d=json.loads(open("$O/two_ranks_one_gpu.json").read().strip().splitlines()[-1]); print("2 ranks:", d["value"], d["n_gpus"], d.get("code_group_resident",{}).get("segments_per_s"))
PY
tail -3 $O/two.err
