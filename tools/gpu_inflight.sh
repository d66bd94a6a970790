# Seals in flight per GPU vs throughput:  gpurun -- 'bash tools/gpu_inflight.sh <name>'
set -u
O=gpurun_out/${1:-inflight}; mkdir -p $O
export TMPDIR=/tmp
for m in 3 1 2 4 5 6 3; do
  timeout 200 python bench.py --steps 24 --warmup 2 --inflight $m --no-cpu-baseline --no-heavy --no-resident --no-prof > $O/inflight_$m.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads(open("$O/inflight_$m.json").read().strip().splitlines()[-1]); print("inflight",$m, round(d["value"],2), round(d["ms_per_step"],2))
PY
done
ZKH_SHARE_GPUS=1 timeout 300 python bench.py --gpus 2 --steps 12 --warmup 2 --no-cpu-baseline --no-heavy --no-resident > $O/two_ranks_one_gpu.json 2>> $O/err.log; head -c 400 $O/two_ranks_one_gpu.json; echo
timeout 100 python tools/microbench.py --only M1,M2 | cut -c1-120
