#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/trace" -o snap -- \
  python "$OLDPWD/bench.py" --mode train --workload c3 --precision bf16 --steps 4 --warmup 2 --no-extra-legs) > gpurun_out/trace.log 2>&1
f=$(find gpurun_out/trace -name '*kernel_trace.csv' | head -1)
tail -1 gpurun_out/trace.log | cut -c1-200
[ -n "$f" ] && python tools/trace_gaps.py "$f" ${1:-300} && rm -rf gpurun_out/trace
