#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_presplit.py tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout=600 -k "presplit or conv_ps or template or voting or rotate or resnet_unit or gn_norm" 2>&1 | tail -6
SNAP_BENCH_DUMP=gpurun_out/r03/launches_c4.json timeout 600 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/r03/bench_c4.log 2>&1
tail -1 gpurun_out/r03/bench_c4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['roofline'])
for k,v in d['kernels'].items(): print(k,v)
"
