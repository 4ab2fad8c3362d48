#!/bin/bash
# VALU / MFMA instruction counts of k_pnet_fused with one compute phase disabled at a time (TRL_PNET_SKIP bits: 2 = conv1+pool,
# 4 = conv2, 8 = conv3+heads+candidates, 32 = emit no candidates -- set in every run so that garbage maps cannot overflow the lists;
# timing-only ablations).  gpurun -- 'bash tools/pnet_phase_pmc.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TRUELY_HIP_LIB="$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/libtruely_hip_tuning.so"   # the TRL_* switches exist in the tuning build only (make -C .../csrc TUNING=1)
O=gpurun_out/pnet_phase
rm -rf $O && mkdir -p $O
: > gpurun_out/pnet_phase_pmc.txt
for k in 32 34 36 40 46; do
  export TRL_PNET_SKIP=$k
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/s$k -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline > $O/s$k.log 2>&1 || { tail -5 $O/s$k.log; exit 1; }
  echo "TRL_PNET_SKIP=$k" >> gpurun_out/pnet_phase_pmc.txt
  python3 tools/pmc_summary.py $(find $O/s$k -name "*counter_collection.csv" | head -1) k_pnet_fused >> gpurun_out/pnet_phase_pmc.txt
  echo "skip $k done"
done
rm -rf $O
cat gpurun_out/pnet_phase_pmc.txt
