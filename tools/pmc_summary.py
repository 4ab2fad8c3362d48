"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv): python tools/pmc_summary.py <csv> [name-filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and flt not in n:
        continue
    key = n + " grid=" + r.get("Grid_Size", "?")
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(56), "n".rjust(4), " ".join(c[-18:].rjust(18) for c in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(cnt[k])
    print(k[:56].ljust(56), str(n).rjust(4), " ".join(f"{v.get(c, 0) / n:18.0f}" for c in names))
