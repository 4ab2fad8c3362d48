#!/bin/bash
# A/B of two builds of the library on one box, interleaved: bash tools/ab.sh <alt.so> [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ALT=$1; shift
FLAGS=${@:---steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --embed-group 1}
for i in 1 2 3 4; do
  for l in "" "$ALT"; do
    if [ -n "$l" ]; then export TRUELY_HIP_LIB=$GRAFT_REPO_ROOT/$l; else unset TRUELY_HIP_LIB; fi
    python bench.py $FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${l:-HEAD}', d['value'], d['roofline']['kernel_ms_per_step'], d['config']['emb_crc32'])"
  done
done
