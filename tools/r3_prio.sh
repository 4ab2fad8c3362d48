#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_prio
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], 'pnet', r['kernel_ms_per_step'], 'frac', r['frac'], 'crc', d['config']['emb_crc32'])
PY
}
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for mode in producer own own_low; do for f in 1 2; do
TRUELY_EMBED_STREAM=$mode timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight $f > $O/b_${mode}_$f.json 2> $O/b_${mode}_$f.err || exit 1
show $O/b_${mode}_$f.json "$mode F$f"
done; done
