#!/bin/bash
# Two contexts in flight with the fused PNet launch ordered behind the END of the device's previous call (the default: the launch
# runs alone, the pyramid kernels overlap the other call's tail) vs left to the hardware queues (--pnet-gate 0) vs one batch in
# flight.  Interleaved A/B on one box + rocprofv3 kernel stats.      gpurun -- 'bash tools/gate_ab.sh'  -> gpurun_out/gate_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gate_ab.txt
: > $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'pnet_ms(events)', r['kernel_ms_per_step'], 'frac', r['frac'], 'alone', r['kernel_ms_alone'], 'pyr', r['pyramid_ms_per_step'], 'crc', d['config']['emb_crc32'])"; }
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 2 2>/dev/null | line "inflight=2 gate=1" >> $O
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 2 --pnet-gate 0 2>/dev/null | line "inflight=2 gate=0" >> $O
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 2>/dev/null | line "inflight=1       " >> $O
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 3 2>/dev/null | line "inflight=3 gate=1" >> $O
  echo "round $i done"
done
rm -rf gpurun_out/gate_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/gate_stats -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 2 > gpurun_out/gate_stats.log 2>&1
python - >> $O <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gate_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_pnet_fused" in r["Name"]:
        print("rocprofv3 (--in-flight 2, gate on) k_pnet_fused: calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6)
PY
grep "^{" gpurun_out/gate_stats.log | tail -1 | line "under rocprofv3  " >> $O
cat $O
