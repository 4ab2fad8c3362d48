#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_front
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline --in-flight 1 --embed-group 1 --steps 8 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/base -o s -f csv -- python3 bench.py $B > $O/base.log 2>&1 || exit 1
export TRUELY_HIP_LIB=$GRAFT_REPO_ROOT/alt_front_w4.so
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/w4 -o s -f csv -- python3 bench.py $B > $O/w4.log 2>&1 || exit 1
for d in base w4; do echo "== $d"; grep -E "k_mtcnn_front|k_pnet_fused|k_pyramid_stream" $O/$d/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160; tail -1 $O/$d.log | cut -c60-130; done
