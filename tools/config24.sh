cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c24; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline"
python bench.py --config 2 --steps 10 --warmup 2 $B > $O/bench_config2.json 2>/dev/null
python bench.py --config 4 --steps 6 --warmup 2 $B > $O/bench_config4.json 2>/dev/null
python bench.py --config 2 --steps 10 --warmup 2 $B --in-flight 2 > $O/bench_config2_inflight2.json 2>/dev/null
python bench.py --config 4 --steps 6 --warmup 2 $B --in-flight 2 > $O/bench_config4_inflight2.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c2 -o s -f csv -- python3 bench.py --config 2 --steps 4 --warmup 1 $B > $O/stats_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c4 -o s -f csv -- python3 bench.py --config 4 --steps 3 --warmup 1 $B > $O/stats_c4.log 2>&1
for f in $O/bench_config*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['pyramid_ms_per_step'])"; done
