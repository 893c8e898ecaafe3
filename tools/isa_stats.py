"""Resource usage and instruction mix of ONE kernel in a hipcc -S listing.

    python tools/isa_stats.py qe.s <mangled-name-substring> [--dump out.s]
"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if pat in l.split(":")[0] and ":" in l and not l.startswith(("\t", ".", ";", " ")))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".size") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
meta_end = next(i for i in range(end, len(lines)) if "; Occupancy" in lines[i])
print(lines[start][:160])
for l in lines[end:meta_end + 1]:
    if re.search(r"; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|LDSByteSize|Occupancy|SGPRBlocks)", l) or "spill" in l.lower():
        print("  ", l.strip())
ops = collections.Counter(l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";")))
keys = ["flat_load", "scratch_", "v_readlane", "v_writelane", "s_load", "ds_", "global_load", "global_store", "global_atomic",
        "v_mul_lo", "v_mul_hi", "v_mad_u64", "s_barrier", "s_waitcnt", "v_cndmask", "v_cmp", "s_cbranch", "v_"]
for k in keys:
    print(f"  {k:14s} {sum(v for o, v in ops.items() if o.startswith(k))}")
print("  total instr   ", sum(ops.values()))
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
