# larger segments on one GPU (288 GB HBM): po2 21 and 22 through the same bench (seals verified after the clock)
set -u
O=gpurun_out/big; mkdir -p $O
for p in 21 22; do
  timeout 900 python bench.py --po2 $p --steps 9 --warmup 1 --no-cpu-baseline --no-live-traffic --no-heavy --no-resident --no-block > $O/bench_po2_$p.json 2> $O/bench_po2_$p.err
  echo "po2 $p rc=$?"; python - <<P
import json
try:
    l=json.loads(open("$O/bench_po2_$p.json").read().strip().splitlines()[-1])
    print(round(l["value"],3), "segments/s", round(l["ms_per_step"],1), "ms/step; verified", l.get("timed_seals_verified"), "cycles/s", round(l["value"]*(1<<$p)/1e6,1), "M")
except Exception as e: print("no line", e)
P
  tail -n 2 $O/bench_po2_$p.err | grep -v amdgpu.ids
done
