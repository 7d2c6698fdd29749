#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "lift_pool_bwd or mlp2_pool_max" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q --timeout=800 -k "c3_full" 2>&1 | tail -12
timeout 600 python bench.py --mode train --workload c3 --precision bf16 --steps 6 --warmup 2 --dump gpurun_out/r03/launches_train_bf16.json > gpurun_out/r03/bench_train_bf16.log 2>&1
tail -1 gpurun_out/r03/bench_train_bf16.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:14]: print(k,v)
"
