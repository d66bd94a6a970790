# RECURSION on the GPU box: tests, the default bench line (block leg incl. the lift / join fold), config 5 with in-circuit joins
set -u
O=gpurun_out/rec; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_recursion_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
timeout 900 python bench.py --no-live-traffic --no-heavy --no-resident > $O/bench_default.json 2> $O/bench_default.err
timeout 1200 python bench.py --config succinct --segments ${1:-256} > $O/bench_succinct.json 2> $O/bench_succinct.err
cat $O/tests.txt; tail -3 $O/bench_default.err; python - <<'P'
import json
for f in ("bench_default", "bench_succinct"):
    try:
        l = json.loads(open(f"gpurun_out/rec/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    r = (l.get("block") or {}).get("recursive") or l.get("recursion")
    print(f, "value", round(l["value"], 2), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r or {}).items() if k not in ("programs", "note")})
P
tail -3 $O/bench_succinct.err
