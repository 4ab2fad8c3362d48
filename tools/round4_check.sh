#!/bin/bash
# Round-4 evidence in one GPU call: GPU tests, the driver's bench line, rocprofv3 kernel stats of the SAME command, the RCCL branch
# with one rank, two gloo ranks on one card.   gpurun --timeout 1100 -- 'bash tools/round4_check.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4
rm -rf $O && mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/r4/stats/**/*kernel_stats.csv", recursive=True)[0]
line = json.loads([l for l in open("gpurun_out/r4/stats.log") if l.startswith("{")][-1])
for r in csv.DictReader(open(f)):
    if "k_pnet_fused" in r["Name"]:
        print("rocprofv3 k_pnet_fused: calls", r["Calls"], "avg ms", round(float(r["AverageNs"]) / 1e6, 4), "| bench line under rocprofv3: kernel_ms_per_step", line["roofline"]["kernel_ms_per_step"], "frac", line["roofline"]["frac"])
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 2 > $O/bench_inflight2.json 2> $O/bench_inflight2.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --force-dist > $O/bench_force_dist.json 2> $O/bench_force_dist.err; cut -c1-200 $O/bench_force_dist.json
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 6 --no-cpu-baseline > $O/bench_gloo2_sharded.json 2> $O/bench_gloo2_sharded.err; cut -c1-200 $O/bench_gloo2_sharded.json
timeout -k 10 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $O/bench_sustained.json 2> $O/bench_sustained.err; cut -c1-200 $O/bench_sustained.json
