#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_rs.py -m gpu -q -x -k "split_k or splitk or stat" 2>&1 | tail -3
rm -rf gpurun_out/c2prof; mkdir -p gpurun_out/c2prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/c2prof" -o c2 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs) > gpurun_out/c2prof.log 2>&1
grep -i "splitk\|gn_finalize" gpurun_out/c2prof/c2_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/c2prof
