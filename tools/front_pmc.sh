#!/bin/bash
# SQ instruction-mix counters of the front kernels (separate --pmc passes, kernel trace only).  gpurun -- 'bash tools/front_pmc.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/front_pmc
rm -rf $O && mkdir -p $O
B="--steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/sq1 -o p -f csv -- python3 bench.py $B > $O/sq1.log 2>&1 || { tail -5 $O/sq1.log; exit 1; }
echo "sq1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/sq2 -o p -f csv -- python3 bench.py $B > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
echo "sq2 done"
for p in sq1 sq2; do python3 tools/pmc_summary.py $(find $O/$p -name "*counter_collection.csv" | head -1) > gpurun_out/front_pmc_$p.txt; grep -E "^kernel|mtcnn_front<(24|48)" gpurun_out/front_pmc_$p.txt | head -4; done
rm -rf $O
