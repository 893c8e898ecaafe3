"""Per-step SQ counter summary of the rollout kernel from tools/pmc_rollout.sh output.   python tools/pmc_summary.py DIR [steps_per_dispatch]"""
import csv, collections, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
for f in [f"{d}/pmc1/p1_counter_collection.csv", f"{d}/pmc2/p2_counter_collection.csv"]:
    rows = [r for r in csv.DictReader(open(f)) if "rollout" in r["Kernel_Name"]]
    # keep the dispatches with the most work (the full-size chunks)
    by = collections.defaultdict(dict)
    for r in rows: by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    key = "SQ_WAVE_CYCLES" if "pmc1" in f else "SQ_ACTIVE_INST_VALU"
    top = max(v.get(key, 0) for v in by.values())
    sel = [v for v in by.values() if v.get(key, 0) > 0.9 * top]
    print(f"{f}: {len(sel)} full-size dispatches")
    for c in sorted(sel[0]):
        print(f"   {c:22s} {sum(v[c] for v in sel) / len(sel) / steps:10.1f} per step (all waves)")
