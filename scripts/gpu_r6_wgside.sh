#!/bin/bash
# round 6: kernel gradients on the side stream (ops.Tuning.WGRAD_SIDE_STREAM): tests + C3 A/B in one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6wg; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -5
for t in 1 0 1 0; do for p in bf16 fp16; do
  timeout 300 python bench.py --mode train --workload c3 --precision $p --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs --tune WGRAD_SIDE_STREAM=$t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$t $p', d['ms_per_step'], d['step_ms'], d['train_logs'].get('loss'), d['train_logs'].get('l2_grads'))"
done; done
