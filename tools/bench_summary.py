"""One line per bench.py JSON file: value, wall and device time per vector step."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    dev = r["avg_launch_us"] / max(1.0, r["units_per_launch"]) * d["config"]["agents_per_gpu"]
    print(f"{f}: {d['value'] / 1e6:8.2f} M env-steps/s  {d['ms_per_step'] * 1e3:7.3f} us/step wall  {dev:7.3f} us/step device  "
          f"frac {r['frac']:.5f}  contested {d['contested_agent_steps']}")
