#!/usr/bin/env bash
# Runs ON THE GPU BOX: the three PMC passes (one counter group per run, --kernel-trace only) over the c3 workload
# (turnstile path, eager launches), for profiles/<tag>_traffic.json's "c3" entry.  Usage: tools/collect_traffic_c3.sh <tag>
set -uo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export QE_USE_GRAPH=0
BENCH="python3 $ROOT/bench.py --workload c3 --steps 2000 --warmup 1000 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$grp" | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/c3_pmc_$name" -- $BENCH > "$OUT/c3_pmc_$name.log" 2>&1 || exit 1
done
echo "c3 counters collected under $OUT"
