"""Diagnostic: idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(sys.argv[1]))))
rows = rows[len(rows) // 3:]  # skip warm-up
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [(rows[k + 1][0] - rows[k][1], rows[k][2], rows[k + 1][2]) for k in range(len(rows) - 1)]
print(f"kernels {len(rows)}, busy {busy/1e6:.2f} ms of {span/1e6:.2f} ms ({100*busy/span:.1f} %)")
by = {}
for g, a, b in gaps:
    k = (a, b)
    by.setdefault(k, []).append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {k[0]} -> {k[1]}: n={len(v)} mean gap {sum(v)/len(v)/1e3:.1f} us, total {sum(v)/1e6:.2f} ms")
names = {}
for s, e, n in rows:
    names.setdefault(n, []).append(e - s)
for n, v in names.items():
    print(f"  {n}: n={len(v)} mean {sum(v)/len(v)/1e3:.1f} us")
