#!/usr/bin/env bash
# Runs ON THE GPU BOX: the bench lines DESIGN.md's table is filled from (clean runs, graph replay on); every line
# carries its cpu_baseline (NumPy port + compiled C, one core of the same box).  Copy gpurun_out/final/*.json to
# profiles/<tag>_final_<workload>.json afterwards.
set -uo pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/final
mkdir -p "$OUT"
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/driver_flags.json" || exit 1
python bench.py 2>/dev/null | tail -1 > "$OUT/headline.json" || exit 1
for wl in c2 c3 c4shard c5 tictactoe; do
    timeout -k 10 300 python bench.py --workload $wl --steps 4000 --warmup 2000 2>/dev/null | tail -1 > "$OUT/$wl.json" || exit 1
done
timeout -k 10 300 python bench.py --workload wide --steps 400 --warmup 2000 2>/dev/null | tail -1 > "$OUT/wide.json" || exit 1
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    r = d["roofline"]
    print(os.path.basename(f)[:-5], f"{d['value']/1e6:.1f} M/s", f"{d['ms_per_step']*1e3:.2f} us/step wall", r["kernel"], f"launch {r['avg_launch_us']:.2f} us", f"frac {r['frac']:.5f}",
          "device_region_ms", round(d.get("device_region_ms", 0), 2), "steps", d["steps"])
PY
