#!/bin/bash
# quick GPU check of the newest tests + the C2 line (no extra legs)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -x -k "projection or whole_model or interleave or fused_adam or fp16" -s 2>&1 | grep -E "passed|failed|Error|error|whole-model|assert" | tail -15
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'],3) for n,v in d['kernels'].items() if v['ms']>0.1})"; done
