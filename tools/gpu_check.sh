#!/bin/bash
# Working check on the GPU box: the whole GPU test suite, then the bench lines a kernel change moves.
#   gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'   -> gpurun_out/gpu_check/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gpu_check
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/bench.json 2> $O/bench.err && cut -c1-400 $O/bench.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --driver threads --embed-group 1 > $O/bench_threads.json 2> $O/bench_threads.err && cut -c1-200 $O/bench_threads.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight 1 > $O/bench_if1.json 2> $O/bench_if1.err && cut -c1-200 $O/bench_if1.json &&
timeout -k 10 300 python bench.py --prelu general --steps 20 --warmup 5 $B > $O/bench_general.json 2> $O/bench_general.err && cut -c1-200 $O/bench_general.json &&
timeout -k 10 300 python bench.py --prelu general --steps 20 --warmup 5 $B --in-flight 1 > $O/bench_general_if1.json 2> $O/bench_general_if1.err && cut -c1-200 $O/bench_general_if1.json
