#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q -x -k "lift or observ or train_step or gradient" 2>&1 | tail -2
rm -rf gpurun_out/c3prof; mkdir -p gpurun_out/c3prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/c3prof" -o c3 -- python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs) > gpurun_out/c3prof.log 2>&1
cp gpurun_out/c3prof/c3_kernel_stats.csv gpurun_out/c3_kernel_stats.csv
rm -rf gpurun_out/c3prof
grep -i "lift_pool_bwd" gpurun_out/c3_kernel_stats.csv | awk -F, '{print substr($1,1,70), $(NF-6), $(NF-5), $(NF-4), "min", $(NF-2), "max", $(NF-1)}'
