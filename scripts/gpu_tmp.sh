#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q -x -k "plane_fuse or train_step or gradient" 2>&1 | tail -3
timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d.get('step_ms'))"
