#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_pnet2
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], 'pnet', r['kernel_ms_per_step'], 'frac', r['frac'], 'crc', d['config']['emb_crc32'])
PY
}
for r in 1 3 4 6 8 12; do
TRL_PNET_RUN=$r timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight 1 --embed-group 1 > $O/b_run$r.json 2> $O/b_run$r.err || exit 1
show $O/b_run$r.json "if1 g1 run $r"
done
for r in 4 6 8; do
TRL_PNET_RUN=$r timeout -k 10 300 python bench.py --gpus 1 --steps 40 --warmup 5 $B > $O/d_run$r.json 2> $O/d_run$r.err || exit 1
show $O/d_run$r.json "default run $r"
done
