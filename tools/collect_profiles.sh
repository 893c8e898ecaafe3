#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics and the separate PMC passes that
# bench.py's roofline block and profiles/ are built from.  Usage: tools/collect_profiles.sh <tag>
# Output: gpurun_out/prof_<tag>/...; summarise locally with tools/summarise_profiles.py <tag>.
set -uo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20000 --warmup 2000 --no-cpu-baseline"
if [ "${SKIP_HEADLINE:-0}" != 1 ]; then
# 1. kernel trace + stats, headline (the program itself after --, no wrapper)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/headline_stats" -- $BENCH > "$OUT/headline_stats.log" 2>&1 || exit 1
# 2. PMC passes, one counter group per run, kernel-trace only
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$grp" | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/headline_pmc_$name" -- $BENCH > "$OUT/headline_pmc_$name.log" 2>&1 || exit 1
done
fi
[ "${ONLY_HEADLINE:-0}" = 1 ] && { echo "headline profiles collected under $OUT"; exit 0; }
# 3. ONE attempt at the graph-replay path under the profiler (rocprofv3 7.2 has been seen to segfault when a
#    captured HIP graph is replayed under --kernel-trace): whatever happens is kept as evidence
( cd /tmp; QE_USE_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3_graph_stats" -- python3 $ROOT/bench.py --workload c3 --steps 1000 --warmup 1000 --no-cpu-baseline > "$OUT/c3_graph_under_rocprofv3.log" 2>&1; echo "exit code $?" >> "$OUT/c3_graph_under_rocprofv3.log" )
# 4. kernel stats of the step-wise / wide shapes with eager launches (QE_USE_GRAPH=0: the profiles describe the
#    kernels, whose durations do not depend on how they are launched; bench.py's own numbers use graph replay)
export QE_USE_GRAPH=0
for wl in c2 c3 c4shard c5 wide; do
    steps=4000; [ "$wl" = wide ] && steps=400
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${wl}_stats" -- python3 $ROOT/bench.py --workload $wl --steps $steps --warmup 4000 --no-cpu-baseline > "$OUT/${wl}_stats.log" 2>&1 || exit 1
done
echo "profiles collected under $OUT"
