# config 5 in full (1024 segments -> lift + join to one receipt) and the multi-rank shape of it dry-run on one GPU
set -u
O=gpurun_out/rec; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py --config succinct > $O/bench_succinct_1024.json 2> $O/bench_succinct_1024.err
ZKH_SHARE_GPUS=1 timeout 900 python bench.py --gpus 4 --config succinct --segments 32 --inflight 1 > $O/bench_succinct_4rank.json 2> $O/bench_succinct_4rank.err
python - <<'P'
import json
for f in ("bench_succinct_1024", "bench_succinct_4rank"):
    try:
        l = json.loads(open(f"gpurun_out/rec/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    r = l.get("recursion") or {}
    print(f, "value", round(l["value"], 2), "wall", round(l["block_wall_clock_s"], 2), "leaf", round(l["leaf_phase_s"], 2), "follows", l["succinct_root_follows_from_leaf_claims"],
          {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k not in ("programs", "note")})
P
tail -n 3 $O/bench_succinct_1024.err; tail -n 3 $O/bench_succinct_4rank.err
