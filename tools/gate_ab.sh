#!/bin/bash
# The fused PNet launches of two contexts in flight: ordered explicitly (the shipped default) vs left to the hardware queues
# (TRL_PNET_GATE=0, tuning build), vs one batch in flight.  Interleaved A/B on one box + rocprofv3 kernel stats of the default.
#   gpurun -- 'bash tools/gate_ab.sh'  -> gpurun_out/gate_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TUNE="$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/libtruely_hip_tuning.so"
O=gpurun_out/gate_ab.txt
: > $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'pnet_ms(events)', r['kernel_ms_per_step'], 'frac', r['frac'], 'alone', r['kernel_ms_alone'], 'crc', d['config']['emb_crc32'])"; }
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line "gate=1 inflight=2" >> $O
  TRUELY_HIP_LIB=$TUNE TRL_PNET_GATE=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line "gate=0 inflight=2" >> $O
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 2>/dev/null | line "        inflight=1" >> $O
  echo "round $i done"
done
rm -rf gpurun_out/gate_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/gate_stats -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/gate_stats.log 2>&1
python - >> $O <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gate_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_pnet_fused" in r["Name"]:
        print("rocprofv3 (driver's command) k_pnet_fused: calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6)
PY
tail -1 gpurun_out/gate_stats.log | line "under rocprofv3  " >> $O
cat $O
