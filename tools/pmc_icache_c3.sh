#!/usr/bin/env bash
# Runs ON THE GPU BOX: instruction-cache counters of the turnstile launches (c3, eager launches), one pass.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export QE_USE_GRAPH=0 PYTHONPATH=$ROOT
WL=${1:-c3}
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES --output-format csv -d "$OUT/$WL" -- python3 $ROOT/bench.py --workload $WL --steps 1000 --warmup 500 --no-cpu-baseline > "$OUT/$WL.log" 2>&1 || { tail -5 "$OUT/$WL.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$WL/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    print(k, "launches", n[k], {c: round(v / max(1, n[k]), 1) for c, v in acc[k].items()})
PY
