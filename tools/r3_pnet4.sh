#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_pnet4
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], 'pnet', r['kernel_ms_per_step'], 'frac', r['frac'], 'crc', d['config']['emb_crc32'])
PY
}
for g in 2 4 6 2 4; do
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --embed-group $g > $O/d_g${g}.json 2> $O/d_g${g}.err || exit 1
show $O/d_g${g}.json "F2 G$g steps20"
done
for g in 2 4; do
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 $B --embed-group $g > $O/l_g${g}.json 2> $O/l_g${g}.err || exit 1
show $O/l_g${g}.json "F2 G$g steps100"
done
