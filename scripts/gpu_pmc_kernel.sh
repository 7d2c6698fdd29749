#!/bin/bash
# PMC passes (counters only) over one micro-benchmark: bash scripts/gpu_pmc_kernel.sh <bench> <kernel-substr> "<counters pass 1>" ["<pass 2>" ...]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
B=$1; K=$2; shift 2
i=0
for C in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmck$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmck$i" -o pmc -- \
    python "$R/tools/kernel_bench.py" $B 3) > gpurun_out/pmck$i.log 2>&1
  tail -2 gpurun_out/pmck$i.log
  K="$K" python - "$i" <<'PY'
import csv, collections, glob, os, sys
f = glob.glob(f'gpurun_out/pmck{sys.argv[1]}/*counter_collection.csv')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if os.environ['K'] not in k:
            continue
        k = k.split('(')[0][-50:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        print(k, {a: f'{b / cnt[(k, a)]:.4g}' for a, b in v.items()}, 'dispatches', max(cnt[(k, a)] for a in v))
PY
done
