#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_check3
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; rc=$?; tail -16 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/bench.json 2> $O/bench.err && cut -c60-130 $O/bench.json &&
timeout -k 10 300 python bench.py --prelu general --steps 20 --warmup 5 $B > $O/bench_general.json 2> $O/bench_general.err && cut -c60-130 $O/bench_general.json &&
timeout -k 10 300 python bench.py --prelu general --steps 20 --warmup 5 $B --in-flight 1 > $O/bench_general_if1.json 2> $O/bench_general_if1.err && cut -c60-130 $O/bench_general_if1.json &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight 1 > $O/bench_if1.json 2> $O/bench_if1.err && cut -c60-130 $O/bench_if1.json &&
timeout -k 10 300 python bench.py --config 0 --steps 20 --warmup 5 > $O/bench_config0.json 2> $O/bench_config0.err && cut -c60-130 $O/bench_config0.json &&
timeout -k 10 300 python tools/run_wall_time.py $O/run_config0.json > $O/run_wall.log 2>&1; tail -3 $O/run_wall.log
