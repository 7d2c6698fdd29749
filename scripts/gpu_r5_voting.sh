#!/bin/bash
# One gpurun call: the frequency-domain voting -- parity tests, the C4 bench line, kernel stats.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05v
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "voting or template or overlap" -s 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
SNAP_BENCH_DUMP=$O/r05_c4_launches.json timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 2>$O/bench_c4.err | tail -1 > $O/r05_c4_bench.json
head -c 1500 $O/r05_c4_bench.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c4" -o snap -- \
  python "$R/bench.py" --workload c4 --steps 2 --warmup 1) > $O/prof_c4.log 2>&1
cp $O/prof_c4/snap_kernel_stats.csv $O/r05_c4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_c4
head -12 $O/r05_c4_kernel_stats.csv | cut -c1-200
