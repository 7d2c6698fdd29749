#!/bin/bash
# round 3, call A: pre-split engine tests, unit A/B, bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_presplit.py -m gpu -q -x --timeout=600 2>&1 | tail -40 > gpurun_out/r03/presplit_tests.log
echo "== presplit tests =="; tail -25 gpurun_out/r03/presplit_tests.log
timeout 600 python tools/unit_bench.py --iters 10 --json gpurun_out/r03/unit_bench.json > gpurun_out/r03/unit_bench.log 2>&1
echo "== unit bench =="; cat gpurun_out/r03/unit_bench.log | cut -c1-400
SNAP_BENCH_DUMP=gpurun_out/r03/launches_ps.json timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03/bench_ps.log 2>&1
echo "== bench =="; tail -2 gpurun_out/r03/bench_ps.log | cut -c1-1500
