#!/bin/bash
# PMC pass (counters only; no sys/hip trace): per-dispatch SQ counters for one bench step.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc
R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d "$R/gpurun_out/pmc" -o pmc -- \
  python "$R/bench.py" ${BENCH_ARGS:---steps 1 --warmup 1 --no-cpu-baseline}) > gpurun_out/pmc.log 2>&1
tail -3 gpurun_out/pmc.log
ls gpurun_out/pmc
python - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/pmc/*counter_collection.csv')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get('SQ_BUSY_CYCLES', 0)))[:14]:
        print(k, {a: f'{b:.3g}' for a, b in v.items()})
PY
