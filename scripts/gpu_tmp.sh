#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q -x -k "lift or observ or train_step or gradient" 2>&1 | tail -4
rm -rf gpurun_out/c3prof; mkdir -p gpurun_out/c3prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/c3prof" -o c3 -- python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs) > gpurun_out/c3prof.log 2>&1
cp gpurun_out/c3prof/c3_kernel_stats.csv gpurun_out/c3_kernel_stats.csv
rm -rf gpurun_out/c3prof
grep -i "lift_pool\|rocprim" gpurun_out/c3_kernel_stats.csv | cut -c1-60,100-260
timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d.get('step_ms'), {n: round(v['ms'],3) for n,v in d.get('kernels',{}).items() if v['ms']>0.5})"
