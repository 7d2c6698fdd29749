#!/bin/bash
# round 6: kernel stats of the C5 (ViT-B/16 encoder) step
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r6c5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | cut -c1-600
python bench.py --workload c5 --steps 10 --warmup 3 --in-flight 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o snap -- \
    python "$R/bench.py" --workload c5 --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extra-legs) > $O/prof.log 2>&1
cp $O/prof/snap_kernel_stats.csv $O/c5_kernel_stats.csv; rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/c5_kernel_stats.csv')))
for r in rows[:32]:
    print('%-100s %5.1f x %8.1f us = %8.1f us/step'%(r['Name'][:100], int(r['Calls'])/4, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/4e3))
print(sum(float(r['TotalDurationNs']) for r in rows)/4e6)
PY
