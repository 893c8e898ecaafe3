"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the files kept under profiles/."""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = Path("gpurun_out") / f"prof_{tag}"
dst = Path("profiles")


def one(pattern):
    hits = glob.glob(str(src / pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return max(hits, key=lambda h: Path(h).stat().st_mtime)  # gpurun_out/ accumulates earlier collections


def bench_line(log):
    for line in open(log):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit(f"no bench line in {log}")


for wl in ("headline", "c3", "c5", "c4shard", "wide"):
    if wl == "c4shard" and not glob.glob(str(src / "c4shard_stats.log")):
        continue
    shutil.copy(one(f"{wl}_stats/**/*kernel_stats.csv"), dst / f"{tag}_{wl}_kernel_stats.csv")
    json.dump(bench_line(src / f"{wl}_stats.log"), open(dst / f"{tag}_{wl}_bench_under_rocprof.json", "w"), indent=1)


def pmc_sum(group, counter, kernel_substr, wl="headline"):
    total, launches = 0.0, 0
    for row in csv.DictReader(open(one(f"{wl}_pmc_{group}/**/*counter_collection.csv"))):
        if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
            total += float(row["Counter_Value"])
            launches += 1
    return total, launches


kern = "k_rollout_lane"
fetch, n1 = pmc_sum("FETCH_SIZE", "FETCH_SIZE", kern)
write, _ = pmc_sum("WRITE_SIZE", "WRITE_SIZE", kern)
hit, _ = pmc_sum("TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum", kern)
miss, _ = pmc_sum("TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum", kern)
line = bench_line(src / "headline_stats.log")
# all launches of the process: warm-up, the timed call and bench.py's 8 event-timed sample calls
env_steps = (line["steps"] * (1 + line["roofline"]["launches_sampled"] // max(1, line["kernel_launches"])) + line["warmup"]) * line["config"]["agents_per_gpu"]
traffic = 2.0 * fetch * 1024 + write * 1024  # KB counters; gfx950 FETCH_SIZE halves 16-B-per-lane loads
out = {
    "headline": {
        "command": "rocprofv3 --kernel-trace --pmc <counter group> --output-format csv -- python3 bench.py --steps 20000 "
                   "--warmup 2000 --no-cpu-baseline (one pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)",
        "kernel": "k_rollout_lane<float, HashEnv, 4, 128, false, 1, true, true, true> (the build without the general ordered path; 2 of the 94 launches took the build with it)",
        "launches": n1,
        "env_steps_all_launches": env_steps,
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
        "l2_hit_rate": hit / max(1.0, hit + miss),
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B-per-lane loads "
                      "(MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE taken as is",
        "traffic_bytes_all_launches": traffic,
        "traffic_bytes_per_env_step": traffic / env_steps,
        "algorithmic_bytes_per_env_step": line["roofline"]["alg_bytes_per_env_step"],
    }
}
if glob.glob(str(src / "c3_pmc_FETCH_SIZE.log")):  # tools/collect_traffic_c3.sh: the turnstile kernel at 4096 agents
    kern = "k_step_turn"
    fetch, n1 = pmc_sum("FETCH_SIZE", "FETCH_SIZE", kern, "c3")
    write, _ = pmc_sum("WRITE_SIZE", "WRITE_SIZE", kern, "c3")
    hit, _ = pmc_sum("TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum", kern, "c3")
    miss, _ = pmc_sum("TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum", kern, "c3")
    line = bench_line(src / "c3_pmc_FETCH_SIZE.log")
    agents = line["config"]["agents_per_gpu"]
    env_steps = (n1 - 0) * agents  # one launch = one vector step (a few select-only launches included: < 0.1 %)
    traffic = 2.0 * fetch * 1024 + write * 1024
    out["c3"] = {
        "command": "QE_USE_GRAPH=0 rocprofv3 --kernel-trace --pmc <counter group> --output-format csv -- python3 bench.py --workload c3 "
                   "--steps 2000 --warmup 1000 --no-cpu-baseline (one pass per group, tools/collect_traffic_c3.sh)",
        "kernel": "k_step_turn<float, HashEnv, 4>",
        "launches": n1,
        "env_steps_all_launches": env_steps,
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
        "l2_hit_rate": hit / max(1.0, hit + miss),
        "correction": "upper bound: FETCH_SIZE doubled as for the headline kernel although only the row gather uses 16-B-per-lane "
                      "loads (list heads, links and agent state are 4/8-B loads, counted in full); raw FETCH_SIZE + WRITE_SIZE = "
                      f"{(fetch + write) * 1024 / env_steps:.0f} B per env-step",
        "note": "six scattered 8-byte accesses per env-step (two list heads read + exchanged, the cell store, the row) each move a "
                "whole line: the kernel is bound by the latency of these dependent accesses, not by their bytes",
        "traffic_bytes_all_launches": traffic,
        "traffic_bytes_per_env_step": traffic / env_steps,
        "algorithmic_bytes_per_env_step": line["roofline"]["alg_bytes_per_env_step"],
    }
json.dump(out, open(dst / f"{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
