# Run-to-run spread of the headline on one box:  gpurun -- 'bash tools/gpu_repeat.sh <name>'
set -u
O=gpurun_out/${1:-repeat}; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-heavy --no-resident --no-prof > $O/run_$i.json 2>> $O/err.log
done
timeout 200 python bench.py --steps 200 --warmup 6 --no-cpu-baseline --no-heavy --no-resident --no-prof > $O/run_200.json 2>> $O/err.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/run_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["steps"], round(d["value"],2), round(d["ms_per_step"],2))
PY
rocm-smi --showpower --showclocks 2>/dev/null | head -30
