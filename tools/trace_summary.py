"""Summarise the last bench step of a rocprofv3 kernel trace: python tools/trace_summary.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_pyramid" in r["Kernel_Name"]]
step = rows[starts[-1]:]
agg = {}
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += d
    if d >= min_us:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {d:8.1f}us {name:44s} grid={r["Grid_Size_X"]}x{r["Grid_Size_Y"]} vgpr={r["VGPR_Count"]} lds={r["LDS_Block_Size"]}')
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:44s} n={v[0]:4d} {v[1] / 1e3:8.3f} ms")
print(f"kernel time of the step: {tot / 1e3:.3f} ms over {len(step)} launches")
