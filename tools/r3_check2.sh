#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_check2
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "overlapped or begin_end or large_embedder or grouped or in_flight" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/calibrate_general_prelu.py 64 > $O/calib.txt 2>&1; tail -8 $O/calib.txt
for g in 1 2 3 4; do
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group $g > $O/bench_g$g.json 2> $O/bench_g$g.err && cut -c60-130 $O/bench_g$g.json || exit 1
done
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 $B --embed-group 3 > $O/bench_g3_100.json 2> $O/bench_g3_100.err && cut -c60-130 $O/bench_g3_100.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 $B --embed-group 1 > $O/bench_g1_100.json 2> $O/bench_g1_100.err && cut -c60-130 $O/bench_g1_100.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 3 --in-flight 3 > $O/bench_g3_f3.json 2> $O/bench_g3_f3.err && cut -c60-130 $O/bench_g3_f3.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 3 --in-flight 1 > $O/bench_g3_f1.json 2> $O/bench_g3_f1.err && cut -c60-130 $O/bench_g3_f1.json &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -f csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/stats.log 2>&1
ls $O/stats
