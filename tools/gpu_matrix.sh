# crash-safety matrix of bench invocations (small sizes)
set -u
O=gpurun_out/matrix; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 400 $O/$name.json | head -c 0)$(python - "$O/$name.json" <<'P'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l.get("recursion") or {}
    print("value", round(l["value"],2), "proofs", r.get("proofs"), "follows", l.get("succinct_root_follows_from_leaf_claims"))
except Exception as e: print("NO LINE", e)
P
)"; }
run heavy_succinct --config succinct --segments 8 --circuit syn_heavy --no-cpu-baseline
run odd9 --config succinct --segments 9 --no-cpu-baseline
run one --config succinct --segments 1 --no-cpu-baseline
run block16 --config block --segments 16 --no-cpu-baseline
run po2_18 --po2 18 --steps 8 --no-cpu-baseline --no-live-traffic --no-heavy --no-resident --block-segments 6
run p2join5 --config succinct --segments 5 --join-circuit p2_join --no-cpu-baseline
run unfused6 --config succinct --segments 6 --no-fused-lift --no-cpu-baseline
tail -n 2 $O/*.err | grep -v amdgpu.ids | grep -v "^$" | head -20
