#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_pnet3
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], 'pnet', r['kernel_ms_per_step'], 'frac', r['frac'], 'crc', d['config']['emb_crc32'])
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "overlapped" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for g in 1 2 3; do for rep in 1 2; do
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group $g > $O/d_g${g}_$rep.json 2> $O/d_g${g}_$rep.err || exit 1
show $O/d_g${g}_$rep.json "F2 G$g"
done; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 3 --in-flight 3 > $O/d_f3.json 2> $O/d_f3.err && show $O/d_f3.json "F3 G3"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 3 --in-flight 1 > $O/d_f1.json 2> $O/d_f1.err && show $O/d_f1.json "F1 G3"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group 1 --driver threads > $O/d_thr.json 2> $O/d_thr.err && show $O/d_thr.json "F2 G1 threads"
