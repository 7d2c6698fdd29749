#!/bin/bash
# FETCH_SIZE of the lift kernels with and without the BEV-tiled traversal (counters only).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for mode in 1 0; do
  rm -rf gpurun_out/pmc_lift_$mode
  (cd /tmp && SNAP_LIFT_BEV_TILES=$mode timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv \
     -d "$R/gpurun_out/pmc_lift_$mode" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline) > gpurun_out/pmc_lift_$mode.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_lift_$mode/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'lift_pool' in k or 'mlp2_pool_kernel' in k:
        k = k.split('(')[0][-40:]
        agg[k] += float(r['Counter_Value']); cnt[k] += 1
for k, v in agg.items():
    print('tiles=$mode', k, cnt[k], 'launches, FETCH_SIZE KiB', round(v), '-> GB/step (x2 correction, 2 steps):', round(v * 1024 * 2 / 2 / 1e9, 2))
PY
done
