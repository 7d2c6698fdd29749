#!/bin/bash
# round 6: tuning switches re-measured with two batches in flight (A/B in one box)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6kn; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 3 "$@" 2> $O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$O/$tag.json').read())
    print('$tag', d['ms_per_step'], d['value'], d['step_ms']['median'], round(d['kernels']['conv_split_bf16x3']['ms'],2))
except Exception as e:
    print('$tag ERR', e)
PY
}
run base_a
run presplit --tune USE_PRESPLIT=1
run nosplitk --tune USE_SPLITK=0
run base_b
run nors --tune CONV_NO_RS=1
run nows --tune CONV_NO_WS=1
run nohalo --tune CONV_NO_HALO=1
run tile128x64 --tune CONV_TILE=128x64
run base_c
run presplit_nosplitk --tune USE_PRESPLIT=1 --tune USE_SPLITK=0
