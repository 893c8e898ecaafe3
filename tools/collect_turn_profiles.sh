set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_${1:-r03b}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export QE_USE_GRAPH=0
for wl in c3 c5 c4shard; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${wl}_stats" -- python3 $ROOT/bench.py --workload $wl --steps 4000 --warmup 4000 --no-cpu-baseline > "$OUT/${wl}_bench_under_rocprof.json" 2> "$OUT/${wl}_stats.log" || exit 1
  f=$(find "$OUT/${wl}_stats" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$OUT/${wl}_kernel_stats.csv"
  head -2 "$OUT/${wl}_kernel_stats.csv" | cut -c1-200
  rm -rf "$OUT/${wl}_stats"
done
