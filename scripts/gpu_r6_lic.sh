#!/bin/bash
# round 6: the lift inside the consumer -- parity tests, then A/B of the C2 step in one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "lift or mlp2_pool or without_feature_volume" 2>&1 | tail -15
for t in "LIFT_IN_CONSUMER=0" "LIFT_IN_CONSUMER=1" "MLP_GATHER_XCD_GROUP=0" "MLP_GATHER_XCD_GROUP=32" "LIFT_IN_CONSUMER=0" "LIFT_IN_CONSUMER=1"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --tune $t 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 $t', d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'],3) for n,v in d['kernels'].items() if v['ms']>0.25})"
done
