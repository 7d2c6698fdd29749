#!/bin/bash
# HBM traffic of one bench step: two SEPARATE PMC passes (counters only, no sys/hip trace).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_$C" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline) > gpurun_out/pmc_$C.log 2>&1
  tail -1 gpurun_out/pmc_$C.log | cut -c1-200
done
python tools/make_hbm_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
  gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/hbm_traffic.json 2
