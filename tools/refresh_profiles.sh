#!/bin/bash
# Regenerates the evidence under gpurun_out/ that profiles/ is built from.  Run ON the GPU box:
#   gpurun --timeout 900 -- 'bash tools/refresh_profiles.sh'
# then, back in the container:  python tools/collect_profiles.py
# PMC passes are separate runs with --kernel-trace only (never combined with --stats / sys-trace).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"; cat $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -f csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
echo "write done"
