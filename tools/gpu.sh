#!/bin/bash
# ONE runner for everything measured on the GPU box.  From the repo root:
#     gpurun --timeout 900 -- 'bash tools/gpu.sh <recipe> [name]'        -> gpurun_out/<name>/...
# Recipes (each step under its own `timeout`; counter passes never share a run with tracing domains):
#   tests        the whole -m gpu suite + smoke()
#   new [FILE]   one test file (default tests/test_session_gpu.py), fail fast
#   bench        the driver's command (python bench.py) + its 8-rank form on ONE GPU (ZKH_SHARE_GPUS=1)
#   foldlanes    config 5 (g++ host, streamed) at 4 / 5 / 6 / 8 fold lanes per GPU
#   inflight     the headline at 2 .. 6 seals in flight per GPU
#   torchrun2    the driver's N > 1 launch shape (python -m torch.distributed.run ... bench.py --gpus 2) on ONE GPU
#   succinct4    config 5's N-rank shape as 4 ranks on ONE GPU (native executor per rank + top joins on rank 0)
#   chained      bench.py --config block --chained (S = 256 SYN-C segments, pre == prev.post checked after the clock)
#   config5      BASELINE config 5 (S = 1024 -> one succinct receipt): streamed pipeline, two phases, and the g++ host
#   ab VAR       A/B of one env switch of the library (e.g. ZKH_REC_GRAPH): bench with and without VAR=1, 3 repeats each
#   repro        the DRIVER'S exact command (python bench.py --gpus 1 --steps 20 --warmup 5) five times on this lease: spread of `value` and of
#                every secondary figure (syn_heavy, resident code, block, preflight) -> repro_summary.json  (round-5 verdict, item 1)
#   devices      N > 1 readiness on ONE GPU: 8 ranks sharing it (--allow-shared-gpu; config.devices), the same without the flag (must
#                refuse), the in-process launcher (--launcher session, 8 devices x 1 lane and 1 device x 3 lanes), the RCCL probe at world 1
#   huge         SYN-HUGE (> 250 k steps, > 2 k taps, ~15 k constraints) loaded as data: generator / hipcc / code-object figures, eval_check and
#                seal ms at po2 20, three evaluators + extreme vectors (tests/soak/syn_huge_report.py); the same report for SYN-HEAVY beside it
#   fuzzsoak     tests/test_fuzz_gpu.py over FRESH seeds (ZKH_FUZZ_SEED_OFFSET = 10000, 20000, ...; args: name, rounds): spare GPU minutes spent
#                on shapes the suite never saw; one summary line per round
#   profiles     everything profiles/ holds (tools/collect_profiles.sh)
#   big          po2 21 / 22 segments
#   soak         1000 distinct segments through the g++ driver
#   roots        regenerate zeth_amd/circuits/control_roots.json
#   power        rocm-smi clock / power while one kernel family loops
# Summaries to keep are copied by hand from gpurun_out/<name>/ into profiles/ (tracked).
set -u
R=${1:?recipe}; shift
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keys = ("value", "ms_per_step", "n_gpus", "steps", "timed_seals_verified", "block_wall_clock_s", "block_segments_per_s", "leaf_phase_s")
    print(sys.argv[1].split("/")[-1], {k: (round(l[k], 3) if isinstance(l[k], float) else l[k]) for k in keys if k in l},
          "roofline.frac", round(l.get("roofline", {}).get("frac", 0), 4), "cpu", round(l.get("cpu_baseline", {}).get("value", 0), 4))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
# the box receives SOURCE only (.gpurunignore): the library and the g++ example hosts are built here, once
python -c "from zeth_amd import build; build.ensure_built(); build.build_examples()" > /dev/null 2>&1 || echo "gpu.sh: build failed"
case $R in
tests)
  O=gpurun_out/${1:-tests}; mkdir -p $O
  ( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
  tail -5 $O/pytest.log; tail -2 $O/smoke.log ;;
new)
  O=gpurun_out/new; mkdir -p $O
  ( time timeout 900 python -m pytest ${1:-tests/test_session_gpu.py} -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -30 $O/pytest.log ;;
knobs)
  O=gpurun_out/${1:-knobs}; mkdir -p $O          # generator knobs on SYN-HEAVY around the gathered + locality default (every variant compiled on the box, bit-exact vs built-in)
  timeout 1200 python tools/exp_codegen.py syn_heavy ${KNOB_VARIANTS:-GATHER=0,LOCALITY=0 PART=4800 PART=6400 PART=2400 REGS=80 REGS=112 REGS=128 PREFETCH=2 PREFETCH=8 EPOCH=32 EPOCH=96 LOCWIN=32 LOCWIN=96 REGS=112,PART=4800} > $O/exp_codegen.jsonl 2> $O/err.txt; cat $O/exp_codegen.jsonl ;;
bench)
  O=gpurun_out/${1:-bench}; mkdir -p $O
  ( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; line $O/bench_default.json
  ( time ZKH_SHARE_GPUS=1 timeout 900 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy > $O/bench_8rank_one_gpu.json 2> $O/bench_8rank.err ) 2> $O/bench_8rank.time; line $O/bench_8rank_one_gpu.json
  grep real $O/*.time; tail -3 $O/bench_default.err | grep -v amdgpu.ids ;;
repro)
  O=gpurun_out/${1:-repro}; mkdir -p $O
  for i in 1 2 3 4 5; do
    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err ) 2> $O/driver_cmd_$i.time; line $O/driver_cmd_$i.json
  done
  python - $O <<'PY'
import json, sys
O = sys.argv[1]
ls = [json.loads(open(f"{O}/driver_cmd_{i}.json").read().strip().splitlines()[-1]) for i in range(1, 6)]
def col(f):
    v = [f(l) for l in ls]
    m = sum(v) / len(v)
    return {"runs": [round(x, 3) for x in v], "mean": round(m, 3), "spread_pct": round(100 * (max(v) - min(v)) / m, 2)}
out = {"command": "python bench.py --gpus 1 --steps 20 --warmup 5 (five times, one lease)",
       "value": col(lambda l: l["value"]), "syn_heavy": col(lambda l: l["syn_heavy"]["segments_per_s"]),
       "syn_heavy_min": col(lambda l: l["syn_heavy"]["min"]), "syn_heavy_max": col(lambda l: l["syn_heavy"]["max"]),
       "resident_code": col(lambda l: l["code_group_resident"]["segments_per_s"]), "block": col(lambda l: l["block"]["segments_per_s"]),
       "preflight": col(lambda l: l["block"]["host_preflight_pipeline"]["segments_per_s"]),
       "seal_unloaded_ms": col(lambda l: 1e3 * l["seal_wall_clock_unloaded_s"]), "command_wall_s": col(lambda l: l["command_wall_s"]),
       "unstable_legs": sorted({k for l in ls for k in ("syn_heavy", "code_group_resident") if "unstable" in l.get(k, {})})}
json.dump(out, open(f"{O}/repro_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
  ;;
fuzzsoak)
  O=gpurun_out/${1:-fuzzsoak}; mkdir -p $O
  for k in $(seq 1 ${2:-6}); do
    off=$((k * 10000))
    ZKH_FUZZ_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider > $O/offset_$off.log 2>&1
    echo "offset $off rc=$? $(tail -1 $O/offset_$off.log)"
  done | tee $O/summary.txt ;;
devices)
  O=gpurun_out/${1:-devices}; mkdir -p $O
  ( time timeout 900 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy --allow-shared-gpu > $O/bench_8rank_one_gpu.json 2> $O/bench_8rank.err ) 2> $O/bench_8rank.time; line $O/bench_8rank_one_gpu.json
  timeout 300 python bench.py --gpus 8 --steps 6 --warmup 1 --no-heavy > $O/bench_8rank_refused.out 2> $O/bench_8rank_refused.err; echo "8 ranks on one GPU without --allow-shared-gpu: rc=$? (must be non-zero)"; grep -h "distinct GPUs" $O/bench_8rank_refused.err | head -1 | cut -c1-240
  ( time timeout 600 python bench.py --gpus 8 --launcher session --inflight 1 --steps 6 --warmup 1 --allow-shared-gpu > $O/bench_session_8dev_one_gpu.json 2> $O/bench_session8.err ) 2> $O/bench_session8.time; line $O/bench_session_8dev_one_gpu.json
  ( time timeout 600 python bench.py --gpus 1 --launcher session > $O/bench_session_1gpu.json 2> $O/bench_session1.err ) 2> $O/bench_session1.time; line $O/bench_session_1gpu.json
  ZKH_DIST_BACKEND=nccl timeout 600 python bench.py --gpus 2 --steps 6 --warmup 1 --no-heavy --no-block --no-resident --no-cpu-baseline --allow-shared-gpu > $O/bench_2rank_forced_rccl.json 2> $O/bench_2rank_forced_rccl.err; line $O/bench_2rank_forced_rccl.json
  timeout 300 python tools/rccl_probe_check.py > $O/rccl_probe_world1.json 2> $O/rccl_probe_world1.err; cat $O/rccl_probe_world1.json
  python - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_8rank_one_gpu", "bench_session_8dev_one_gpu", "bench_session_1gpu", "bench_2rank_forced_rccl"):
    try:
        l = json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1]); c = l["config"]
        print(f, "value", round(l["value"], 2), "launcher", c.get("launcher"), "devices", len(c.get("devices") or []), "distinct", c.get("devices_distinct"),
              "rccl", c.get("rccl_probe"), [d.get("pci_bus_id") for d in (c.get("devices") or [])][:2])
    except Exception as e:
        print(f, "no line:", e)
PY
  grep real $O/*.time; for f in $O/*.err; do grep -v amdgpu.ids $f | tail -2; done ;;
huge)
  O=gpurun_out/${1:-huge}; mkdir -p $O
  ( time timeout 1500 python tests/soak/syn_huge_report.py --circuit syn_huge > $O/syn_huge_report.json 2> $O/syn_huge.err ) 2> $O/syn_huge.time
  ( time timeout 900 python tests/soak/syn_huge_report.py --circuit syn_heavy > $O/syn_heavy_report.json 2> $O/syn_heavy.err ) 2> $O/syn_heavy.time
  python - $O <<'PY'
import json, sys
for n in ("syn_huge", "syn_heavy"):
    try:
        r = json.load(open(f"{sys.argv[1]}/{n}_report.json"))
        g = r.get("gpu", {})
        print(n, "steps", r["steps"], "taps", r["taps"], "constraints", r["constraints"], "kernels", r["kernels"], "generator_s", r["generator_s"], "hipcc", r["hipcc"],
              "static", r.get("static"), "bounds", r["bounds"]["violations"])
        print("   gpu:", {k: v for k, v in g.items() if k not in ("equals_oracle_po2_6", "kernels_ms")}, g.get("kernels_ms"))
        print("   agree:", g.get("equals_oracle_po2_6"))
    except Exception as e:
        print(n, "no report:", e)
PY
  grep real $O/*.time; tail -3 $O/syn_huge.err | grep -v amdgpu.ids || true ;;
chained)
  O=gpurun_out/${1:-chained}; mkdir -p $O        # a chained block (claim continuity) and the same block with the host-preflight witness, full size
  timeout 600 python bench.py --config block --chained --no-cpu-baseline > $O/bench_block_chained.json 2> $O/err.txt; line $O/bench_block_chained.json
  timeout 900 python bench.py --config succinct --witness preflight --no-cpu-baseline > $O/bench_succinct_preflight.json 2>> $O/err.txt; line $O/bench_succinct_preflight.json
  grep -v amdgpu.ids $O/err.txt | tail -3 ;;
config5)
  O=gpurun_out/${1:-config5}; mkdir -p $O
  timeout 900 python bench.py --config succinct --no-cpu-baseline > $O/bench_succinct_streamed.json 2> $O/err.txt; line $O/bench_succinct_streamed.json
  timeout 900 python bench.py --config succinct --fold phased --no-cpu-baseline > $O/bench_succinct_phased.json 2>> $O/err.txt; line $O/bench_succinct_phased.json
  timeout 900 python bench.py --config succinct --fold phased --recompute-code --no-cpu-baseline > $O/bench_succinct_phased_recompute.json 2>> $O/err.txt; line $O/bench_succinct_phased_recompute.json
  D=/tmp/zkr; mkdir -p $D; python -m zeth_amd.circuits.rec_verify $D > /dev/null; python -m zeth_amd.circuits.recursion $D/recursion.desc > /dev/null
  python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
  LD_LIBRARY_PATH=$PWD/zeth_amd timeout 600 examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 1024 --noise-seed 11904 > $O/prove_session_1024_streamed.json 2>> $O/err.txt
  LD_LIBRARY_PATH=$PWD/zeth_amd timeout 600 examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 1024 --noise-seed 11904 --two-phase > $O/prove_session_1024_two_phase.json 2>> $O/err.txt
  cat $O/prove_session_1024_*.json | cut -c1-700; grep -v amdgpu.ids $O/err.txt | tail -5 ;;
ab)
  V=${1:?env variable}; O=gpurun_out/ab_$V; mkdir -p $O
  for i in 1 2 3; do
    timeout 300 python bench.py --no-cpu-baseline --no-heavy --no-resident --no-block --no-live-traffic --no-certify > $O/off_$i.json 2>> $O/err.txt; line $O/off_$i.json
    env $V=1 timeout 300 python bench.py --no-cpu-baseline --no-heavy --no-resident --no-block --no-live-traffic --no-certify > $O/on_$i.json 2>> $O/err.txt; line $O/on_$i.json
  done
  timeout 300 python bench.py --inflight 1 --steps 20 --no-cpu-baseline --no-heavy --no-resident --no-block --no-live-traffic --no-certify > $O/off_serial.json 2>> $O/err.txt
  env $V=1 timeout 300 python bench.py --inflight 1 --steps 20 --no-cpu-baseline --no-heavy --no-resident --no-block --no-live-traffic --no-certify > $O/on_serial.json 2>> $O/err.txt
  python - $O <<'PY'
import json, sys
for tag in ("off", "on"):
    l = json.loads(open(f"{sys.argv[1]}/{tag}_serial.json").read().strip().splitlines()[-1])
    print(tag, "serial", round(l["value"], 2), {o["op"]: round(o["ms_per_seal"], 3) for o in l["ops"][:6]})
PY
  ;;
foldlanes)
  O=gpurun_out/${1:-foldlanes}; mkdir -p $O      # config 5 through the g++ host: fold lanes per GPU (the sealing lanes + fold-only contexts) with the streamed fold
  D=/tmp/zkr; mkdir -p $D; python -m zeth_amd.circuits.rec_verify $D > /dev/null; python -m zeth_amd.circuits.recursion $D/recursion.desc > /dev/null
  python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
  for k in 6 4 5 8; do
    LD_LIBRARY_PATH=$PWD/zeth_amd ZKH_FOLD_LANES=$k timeout 600 examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 1024 --noise-seed 11904 > $O/fold_lanes_$k.json 2>> $O/err.txt
    echo -n "fold lanes $k: "; cut -c150-420 $O/fold_lanes_$k.json
  done ;;
overlap)
  O=gpurun_out/${1:-overlap}; mkdir -p $O        # can a VALU-bound and an HBM-bound kernel be paired deliberately (stream priorities, occupancy share)?
  [ -x tools/ubench_overlap ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap
  timeout 120 tools/ubench_overlap | tee $O/ubench_overlap.jsonl ;;
join3)
  O=gpurun_out/${1:-join3}; mkdir -p $O          # config 5 with three children per proof above the bottom level (join3) against the same tree proven with joins only
  D=/tmp/zkr; mkdir -p $D; python -m zeth_amd.circuits.rec_verify $D > /dev/null; python -m zeth_amd.circuits.recursion $D/recursion.desc > /dev/null
  python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
  for rep in 1 2; do for tag in join3 no-join3; do
    f=""; [ $tag = no-join3 ] && f="--no-join3"
    LD_LIBRARY_PATH=$PWD/zeth_amd timeout 600 examples/prove_session --desc /tmp/syn_a.desc --recursion-dir $D --segments 1024 --noise-seed 11904 $f > $O/prove_session_1024_${tag}_$rep.json 2>> $O/err.txt
    echo -n "$tag $rep: "; cut -c1-600 $O/prove_session_1024_${tag}_$rep.json
  done; done
  timeout 900 python bench.py --config succinct --no-cpu-baseline > $O/bench_succinct.json 2>> $O/err.txt; line $O/bench_succinct.json ;;
inflight)
  O=gpurun_out/${1:-inflight}; mkdir -p $O       # seals in flight per GPU: the headline at 2 / 3 / 4 / 5 / 6 lanes, twice each, interleaved
  for rep in 1 2; do for k in 3 2 4 5 6; do
    timeout 300 python bench.py --inflight $k --no-cpu-baseline --no-heavy --no-resident --no-block --no-live-traffic --no-certify --no-prof > $O/k${k}_$rep.json 2>> $O/err.txt
    echo -n "inflight $k: "; line $O/k${k}_$rep.json
  done; done ;;
torchrun2)
  O=gpurun_out/${1:-torchrun2}; mkdir -p $O      # the driver's launch shape for N > 1 (torch.distributed.run), 2 ranks sharing the one GPU
  ( time ZKH_SHARE_GPUS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_torchrun2.out 2> $O/err.txt; echo "rc=$?" ) 2> $O/time.txt
  grep -c '^{"metric"' $O/bench_torchrun2.out; grep '^{"metric"' $O/bench_torchrun2.out | tail -1 > $O/bench_torchrun2.json; line $O/bench_torchrun2.json; grep real $O/time.txt ;;
succinct4)
  O=gpurun_out/${1:-succinct4}; mkdir -p $O      # the N-rank shape of config 5 on ONE GPU: every rank seals + folds its aligned range natively, rank 0 joins the local roots
  ZKH_SHARE_GPUS=1 timeout 600 python bench.py --gpus 4 --config succinct --segments 32 --no-cpu-baseline > $O/bench_succinct_4rank_one_gpu.json 2> $O/err.txt; line $O/bench_succinct_4rank_one_gpu.json
  grep -v amdgpu.ids $O/err.txt | tail -5 ;;
profiles) bash tools/collect_profiles.sh gpurun_out/${1:-prof} ;;
big)
  O=gpurun_out/${1:-big}; mkdir -p $O
  for p in 21 22; do
    timeout 900 python bench.py --po2 $p --steps 9 --warmup 1 --no-cpu-baseline --no-live-traffic --no-heavy --no-resident --no-block > $O/bench_po2_$p.json 2> $O/bench_po2_$p.err
    line $O/bench_po2_$p.json
  done ;;
soak)
  O=gpurun_out/${1:-soak}; mkdir -p $O
  python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
  ( time timeout 900 examples/seal_segments --desc /tmp/syn_a.desc --po2 20 --segments 1000 --inflight 3 ) > $O/soak.json 2> $O/soak.err
  cut -c1-400 $O/soak.json; tail -4 $O/soak.err ;;
roots)
  O=gpurun_out/${1:-roots}; mkdir -p $O
  timeout 600 python -m zeth_amd.prover > $O/regen.log 2>&1 && cp zeth_amd/circuits/control_roots.json $O/control_roots.json
  tail -2 $O/regen.log ;;
power)
  O=gpurun_out/${1:-power}; mkdir -p $O
  sample() { tag=$1; shift; ( timeout 240 "$@" > $O/run_$tag.txt 2>> $O/err.txt ) & pid=$!; : > $O/smi_$tag.txt
    while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk|mclk" | tr '\n' ' ' >> $O/smi_$tag.txt; echo >> $O/smi_$tag.txt; sleep 0.4; done; }
  for m in M1 M2 M3 M4 M8; do sample $m python tools/microbench.py --only $m --reps 4000; done
  sample bench python bench.py --steps 600 --warmup 2 --no-block --no-live-traffic --no-certify --no-cpu-baseline
  for tag in M1 M2 M3 M4 M8 bench; do
    echo "== $tag"; sed -E 's/.*sclk[^(]*\(([0-9]+)Mhz\).*Power \(W\): ([0-9.]+).*/\1 \2/' $O/smi_$tag.txt | awk '$2>600{n++; c[n]=$1; w[n]=$2} END{ if(!n){print "  none"; exit} asort(c); asort(w); printf "  n=%d sclk median %d MHz  power median %d W\n", n, c[int((n+1)/2)], w[int((n+1)/2)]}'
  done ;;
*) echo "unknown recipe $R"; exit 2 ;;
esac
