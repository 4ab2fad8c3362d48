#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_pnet
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "carry or prelu_variants or maps or fused_pnet" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for r in 8 1 4 16; do
TRL_PNET_RUN=$r timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --in-flight 1 > $O/bench_run$r.json 2> $O/bench_run$r.err || exit 1
python - $O/bench_run$r.json $r <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('run',sys.argv[2], d['value'], d['ms_per_step'], 'pnet', r['kernel_ms_per_step'], 'frac', r['frac'], 'crc', d['config']['emb_crc32'])
PY
done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/bench.json 2> $O/bench.err && cut -c60-130 $O/bench.json
