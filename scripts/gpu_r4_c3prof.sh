#!/bin/bash
# rocprofv3 kernel stats of the C3 train step (bf16): 1 warm-up + 3 steps -> gpurun_out/r04p/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c3" -o snap -- \
  python "$R/bench.py" --mode train --workload c3 --precision ${PRECISION:-bf16} --steps 3 --warmup 1 --no-extra-legs) > $O/prof_c3.log 2>&1
cp $O/prof_c3/snap_kernel_stats.csv $O/r04_c3_train_${PRECISION:-bf16}_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_c3
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/r04_c3_train_${PRECISION:-bf16}_kernel_stats.csv')))
for r in rows[:32]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/4:7.1f} {float(r['TotalDurationNs'])/1e6/4:8.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f}us")
print('sum', sum(float(r['TotalDurationNs']) for r in rows)/1e6/4)
PY
