#!/bin/bash
# per-kernel averages of a short bench run under rocprofv3:  bash tools/kstats.sh [bench flags]   -> gpurun_out/kstats.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kstats && mkdir -p gpurun_out/kstats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kstats -o s -f csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/kstats/run.log 2>&1
python - > gpurun_out/kstats.txt <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 8 + 2 + 4   # timed + warm-up + the cross-check calls after the region
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("ms per step, kernels above 0.05 ms/step; total %.3f" % (tot / 1e6 / steps))
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    if ms >= 0.05:
        print("%8.3f  %5s x %8.4f  %s" % (ms, r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:110]))
line = [l for l in open("gpurun_out/kstats/run.log") if l.startswith("{")]
if line:
    d = json.loads(line[-1]); print("bench:", d["value"], "frames/s", d["ms_per_step"], "ms/step; pnet", d["roofline"]["kernel_ms_per_step"], "pyramid", d["roofline"]["pyramid_ms_per_step"])
PY
cat gpurun_out/kstats.txt
