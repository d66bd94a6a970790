# Clock / power while one kernel family (or the whole bench) runs in a loop
# (round-2 verdict item 4: "if the clock still sags, report rocm-smi power during the pass").
#   gpurun -- 'bash tools/gpu_power.sh power'     -> gpurun_out/power/{smi_*.txt, mb_*.jsonl, summary.txt}
set -u
O=gpurun_out/${1:-power}; mkdir -p $O
export TMPDIR=/tmp
sample() {   # $1 = tag, rest = command
  tag=$1; shift
  ( timeout 240 "$@" > $O/run_$tag.txt 2>> $O/err.txt ) &
  pid=$!
  : > $O/smi_$tag.txt
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk|mclk" | tr '\n' ' ' >> $O/smi_$tag.txt
    echo >> $O/smi_$tag.txt
    sleep 0.4
  done
}
sample M1 python tools/microbench.py --only M1 --reps 8000
sample M2 python tools/microbench.py --only M2 --reps 5000
sample M3 python tools/microbench.py --only M3 --reps 2000
sample M4 python tools/microbench.py --only M4 --reps 6000
sample M8 python tools/microbench.py --only M8 --reps 1200
sample bench python bench.py --steps 600 --warmup 2 --no-block --no-live-traffic --no-certify
sample heavy python bench.py --circuit syn_heavy --steps 400 --warmup 2 --no-block --no-live-traffic --no-certify
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" > $O/smi_maxpower.txt
{
for tag in M1 M2 M3 M4 M8 bench heavy; do
  echo "== $tag: samples above 600 W (count, MHz median, W median)"
  sed -E 's/.*sclk[^(]*\(([0-9]+)Mhz\).*Power \(W\): ([0-9.]+).*/\1 \2/' $O/smi_$tag.txt | awk '$2>600{n++; c[n]=$1; w[n]=$2} END{ if(!n){print "  none"; exit} asort(c); asort(w); printf "  n=%d sclk median %d MHz (min %d max %d)  power median %d W (min %d max %d)\n", n, c[int((n+1)/2)], c[1], c[n], w[int((n+1)/2)], w[1], w[n]}'
  tail -1 $O/run_$tag.txt | cut -c1-220
done
cat $O/smi_maxpower.txt
} | tee $O/summary.txt
