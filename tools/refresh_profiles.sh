#!/bin/bash
# Regenerates the evidence under gpurun_out/refresh/ that profiles/round4_* is built from.  Run ON the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh'
# then, back in the container:  python tools/collect_profiles.py round4
# PMC passes are separate runs with --kernel-trace only (never combined with --stats / sys-trace).  The per-step kernel breakdown,
# the timeline and the PMC passes use one batch in flight (--in-flight 1) so that a step's launches are not interleaved with
# another context's -- since round 4 that is the bench's default; `--in-flight 2` is a variant.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -C "$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/csrc" TUNING=1 -j8 -s   # the tuning build (TRL_* switches), if it is not there yet
O=gpurun_out/refresh
TUNE_LIB="$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/libtruely_hip_tuning.so"   # TRL_* switches: tuning build only
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench done"; cut -c1-300 $O/bench.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --in-flight 2 $B > $O/bench_inflight2.json 2> $O/bench_inflight2.err
timeout -k 10 300 python bench.py --steps 400 --warmup 5 $B > $O/bench_sustained.json 2> $O/bench_sustained.err; echo "sustained done"
timeout -k 10 300 python bench.py --config 2 --steps 10 --warmup 2 $B > $O/bench_config2.json 2> $O/bench_config2.err
timeout -k 10 300 python bench.py --config 4 --steps 6 --warmup 2 $B > $O/bench_config4.json 2> $O/bench_config4.err
timeout -k 10 300 python bench.py --prelu general --steps 20 --warmup 5 $B > $O/bench_prelu_general.json 2> $O/bench_prelu_general.err
timeout -k 10 300 python bench.py --config 0 --steps 20 --warmup 5 > $O/bench_config0.json 2> $O/bench_config0.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 1 > $O/bench_embed_group1.json 2> $O/bench_embed_group1.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --mode streams --ingest nv12 --steps 6 $B > $O/bench_gloo2_streams_nv12.json 2> $O/bench_gloo2_streams_nv12.err
timeout -k 10 300 python bench.py --ingest nv12 --steps 10 $B > $O/bench_ingest_nv12.json 2> $O/bench_ingest_nv12.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 6 $B > $O/bench_gloo2_sharded.json 2> $O/bench_gloo2_sharded.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --mode streams --steps 6 $B > $O/bench_gloo2_streams.json 2> $O/bench_gloo2_streams.err
echo "bench variants done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats2 -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight 2 > $O/stats2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c2 -o s -f csv -- python3 bench.py --config 2 --steps 4 --warmup 1 $B > $O/stats_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c4 -o s -f csv -- python3 bench.py --config 4 --steps 3 --warmup 1 $B > $O/stats_c4.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -f csv -- python3 bench.py --steps 2 --warmup 1 $B > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -f csv -- python3 bench.py --steps 2 --warmup 1 $B > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/sq1 -o p -f csv -- python3 bench.py --steps 2 --warmup 1 $B > $O/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/sq2 -o p -f csv -- python3 bench.py --steps 2 --warmup 1 $B > $O/sq2.log 2>&1
echo "pmc done"
timeout -k 10 200 python tools/time_facenet.py 20 256 > $O/facenet_ms.txt 2>&1
timeout -k 10 200 python tools/time_facenet.py 10 768 >> $O/facenet_ms.txt 2>&1
timeout -k 10 200 python tools/time_facenet.py 8 1024 >> $O/facenet_ms.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_fn2048 -o s -f csv -- python3 tools/time_facenet.py 4 2048 > $O/stats_fn2048.log 2>&1
timeout -k 10 200 python tools/time_facenet.py 6 2048 >> $O/facenet_ms.txt 2>&1
timeout -k 10 200 python tools/fn_stamps.py 2 5 16 58 60 61 2>&1 | grep -E "launch|fn stamps" > $O/facenet_stamps.txt
echo "facenet done"
# where the waves of the fused PNet kernel spend their time (DBG instantiation: shader clocks per phase and barrier), its phase
# ablation with SQ counters, and the batch sweeps
TRUELY_HIP_LIB=$TUNE_LIB TRL_PNET_CLOCK=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 $B --embed-group 1 > $O/bench_pnet_clock.json 2> $O/pnet_clock.err; grep TRL_PNET_CLOCK $O/pnet_clock.err > $O/pnet_phase_clocks.txt
bash tools/pnet_phase_pmc.sh > /dev/null 2>&1; cp gpurun_out/pnet_phase_pmc.txt $O/pnet_phase_pmc.txt
bash tools/front_ablation.sh > /dev/null 2>&1; cp gpurun_out/front_ablation.txt $O/front_ablation.txt
echo "phase evidence done"
timeout -k 10 300 python tools/crowded_timing.py --pathological > $O/crowded_timing.jsonl 2> $O/crowded_timing.err
timeout -k 10 600 python tools/run_wall_time.py $O/run_wall_time.json --long > $O/run_wall_time.log 2>&1
echo "crowded content + run() wall time done"
