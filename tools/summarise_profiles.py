"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the files kept under profiles/."""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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


for wl in ("headline", "c2", "c3", "c5", "c4shard", "wide"):
    if not glob.glob(str(src / f"{wl}_stats.log")):
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


import subprocess

commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
if subprocess.run(["git", "status", "--porcelain", "--", "dist_classicrl_amd/csrc"], capture_output=True, text=True).stdout.strip():
    commit += "+uncommitted kernel changes"


def dominant(wl, prefix):
    """Symbol (as bench.py's roofline.kernel_symbol spells it) and full name of the kernel with the largest total time
    whose name starts with one of `prefix`."""
    best = None
    for row in csv.DictReader(open(one(f"{wl}_stats/**/*kernel_stats.csv"))):
        name = row["Name"]
        if any(("qe::" + p) in name for p in prefix) and (best is None or float(row["TotalDurationNs"]) > best[1]):
            best = (name, float(row["TotalDurationNs"]), int(row["Calls"]), float(row["AverageNs"]))
    name = best[0]
    sym = name[name.index("qe::k_") + 4:name.index(">(") + 1]
    return sym, best


kern_sym, kern_row = dominant("headline", ("k_rollout_lane", "k_rollout_df"))
kern = kern_sym.split("<")[0]
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
        "kernel_symbol": kern_sym,
        "commit": commit,
        "kernel_stats": {"calls": kern_row[2], "average_ns": kern_row[3], "total_ns": kern_row[1]},
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
        "kernel_symbol": "k_step_turn<float, qe::HashEnv, 4, false>",
        "commit": commit,
        "launches": n1,
        "env_steps_all_launches": env_steps,
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
        "l2_hit_rate": hit / max(1.0, hit + miss),
        "correction": "upper bound: FETCH_SIZE doubled as for the headline kernel although only the row gather and the row records "
                      "use 16-B-per-lane loads (agent state: 4/8-B loads, counted in full); raw FETCH_SIZE + WRITE_SIZE = "
                      f"{(fetch + write) * 1024 / env_steps:.0f} B per env-step",
        "note": "per env-step the kernel reads two 64-B row records and the 64-B row, performs four memory-side atomics on the "
                "records (max + add per touched row) and stores two record entries, the cell, seven 4-B words of agent state and an "
                "8-B log record; every scattered access moves at least a 32-B sector, an atomic one each way.  The kernel is bound "
                "by the latency of these dependent accesses, not by their bytes",
        "traffic_bytes_all_launches": traffic,
        "traffic_bytes_per_env_step": traffic / env_steps,
        "algorithmic_bytes_per_env_step": line["roofline"]["alg_bytes_per_env_step"],
    }
json.dump(out, open(dst / f"{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
