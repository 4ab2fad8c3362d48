#!/bin/bash
# Regenerates the evidence under gpurun_out/ that profiles/ is built from.  Run ON the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
# then, back in the container:  python tools/collect_profiles.py
# PMC passes are separate runs with --kernel-trace only (never combined with --stats / sys-trace).
# bench.py keeps two batches in flight by default; the per-step kernel breakdown and the PMC traffic are taken with one
# batch in flight (--in-flight 1) so that a step's launches are not interleaved with another context's.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"; cat $O/bench.json
timeout -k 10 300 python bench.py --in-flight 1 --no-cpu-baseline > $O/bench_inflight1.json 2> $O/bench_inflight1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats2 -o s -f csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/stats2.log 2>&1
echo "stats (default command) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -f csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 > $O/stats.log 2>&1
echo "stats (one batch in flight) done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $O/fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $O/write.log 2>&1
echo "write done"
