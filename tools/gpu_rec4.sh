set -u
O=gpurun_out/rec; mkdir -p $O
for k in 3 6; do
  timeout 900 python bench.py --config succinct --segments 256 --fold-inflight $k > $O/bench_succinct_fi$k.json 2> $O/bench_succinct_fi$k.err
done
python - <<'P'
import json
for k in (3, 6):
    try:
        l = json.loads(open(f"gpurun_out/rec/bench_succinct_fi{k}.json").read().strip().splitlines()[-1])
        r = l["recursion"]; print(k, "value", round(l["value"], 2), "leaf", round(l["leaf_phase_s"], 2), "bottom_ms_per_seg", round(r["bottom_ms_per_segment"], 2), "proofs", r["proofs"], "join_ms", round(r["join_ms_each"], 2), "fold", round(r["fold_s"], 2), "load", round(r["program_load_s_all_lanes"], 2))
    except Exception as e:
        print(k, "failed", e)
P
