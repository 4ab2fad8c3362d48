cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TRUELY_HIP_LIB="$GRAFT_REPO_ROOT/truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/libtruely_hip_tuning.so"   # the TRL_* switches exist in the tuning build only (make -C .../csrc TUNING=1)
mkdir -p gpurun_out/tune
for t in 32x32 32x64 64x32 64x64 64x96 128x32 128x64 32x128 16x64; do TRL_FN_FORCE1=$t rocprofv3 --kernel-trace -d gpurun_out/tune/s_$t -o t -f csv -- python3 tools/time_facenet.py 3 > gpurun_out/tune/s_$t.log 2>&1; echo $t $(tail -1 gpurun_out/tune/s_$t.log | grep -o "facenet.*"); done
for t in 16x32 16x64 32x32 32x64 48x32 48x64; do TRL_FN_FORCE4=$t rocprofv3 --kernel-trace -d gpurun_out/tune/q_$t -o t -f csv -- python3 tools/time_facenet.py 3 > gpurun_out/tune/q_$t.log 2>&1; echo split $t $(grep -o "facenet 256.*" gpurun_out/tune/q_$t.log | tail -1); done
