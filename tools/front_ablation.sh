#!/bin/bash
# Durations of the R-/O-Net front kernels with their crop or their conv+pool phase disabled (TRL_FRONT_SKIP bits: 1 = R-Net crop,
# 2 = R-Net conv, 4 = O-Net crop, 8 = O-Net conv): timing-only ablations under rocprofv3, run on the GPU box:
#   gpurun -- 'bash tools/front_ablation.sh'    -> gpurun_out/front_ablation.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TRUELY_HIP_LIB="$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/libtruely_hip_tuning.so"   # the TRL_* switches exist in the tuning build only (make -C .../csrc TUNING=1)
O=gpurun_out/front_ablation
rm -rf $O && mkdir -p $O
: > gpurun_out/front_ablation.txt
for skip in 0 1 2 4 8; do
  export TRL_FRONT_SKIP=$skip
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s$skip -o s -f csv -- python3 bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $O/s$skip.log 2>&1 || { echo "run $skip failed"; tail -5 $O/s$skip.log; exit 1; }
  f=$(find $O/s$skip -name "*kernel_stats.csv" | head -1)
  echo "TRL_FRONT_SKIP=$skip" >> gpurun_out/front_ablation.txt
  python3 - "$f" >> gpurun_out/front_ablation.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_mtcnn_front" in r["Name"]:
        nm = r["Name"].split("k_mtcnn_front")[1].split("(")[0]
        print(f"   k_mtcnn_front{nm:18s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  echo "skip $skip done"
done
rm -rf $O
cat gpurun_out/front_ablation.txt
