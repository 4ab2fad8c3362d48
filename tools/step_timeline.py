"""Timeline of one bench step from a rocprofv3 kernel trace (--in-flight 1): every launch with its start offset, duration and
the idle gap before it, then totals.   python tools/step_timeline.py <kernel_trace.csv> [step_from_end=2] [--all]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]  # noqa: E731
# a step starts at the first pyramid kernel after a non-pyramid kernel
starts = [i for i, r in enumerate(rows) if "k_pyramid" in r["Kernel_Name"] and (i == 0 or "k_pyramid" not in rows[i - 1]["Kernel_Name"])]
a, b = starts[-back], starts[-back + 1] if back > 1 else len(rows)
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0.0
agg = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end) / 1e3
    d = (e - s) / 1e3
    busy += d; gaps += gap
    g = agg.setdefault(name(r), [0, 0.0, 0.0]); g[0] += 1; g[1] += d; g[2] += gap
    if "--all" in sys.argv or d > 100 or gap > 20:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {d:8.1f}  gap {gap:6.1f}  {name(r):46s} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev_end = max(prev_end, e)
print(f"\nstep span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy:.1f} us, idle gaps {gaps:.1f} us over {len(step)} launches")
nxt = int(rows[b]["Start_Timestamp"]) if b < len(rows) else None
if nxt:
    print(f"gap to the next step's first kernel: {(nxt - prev_end) / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:46s} n={v[0]:4d} {v[1] / 1e3:8.3f} ms   gaps before {v[2]:7.1f} us")
