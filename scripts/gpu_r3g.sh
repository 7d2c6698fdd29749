#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "mlp2_pool_max" 2>&1 | tail -3
rm -rf gpurun_out/r03/proftrain
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03/proftrain" -o snap -- \
  python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 2 --warmup 1) > gpurun_out/r03/proftrain.log 2>&1
f=$(find gpurun_out/r03/proftrain -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -32 "$f" | cut -c1-200
