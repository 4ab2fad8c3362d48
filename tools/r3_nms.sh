#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_nms
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
B="--no-cpu-baseline --in-flight 1 --embed-group 1 --steps 8 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/new -o s -f csv -- python3 bench.py $B > $O/new.log 2>&1 || exit 1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3_nms/new/s_kernel_stats.csv')))
for r in rows:
    n=r['Name'].replace('(anonymous namespace)::','')
    if any(k in n for k in ('k_nms','k_pnet_fused')):
        print(n[:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us min', round(float(r['MinNs'])/1e3,1))
PY
