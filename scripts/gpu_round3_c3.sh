#!/bin/bash
# Section 6 of scripts/gpu_round3_profiles.sh alone: the C3 training-step artefacts.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 300 python bench.py --mode train --workload c3 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r03_c3_train_bench.json
SNAP_BENCH_DUMP=$O/r03_c3_train_bf16_launches.json timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r03_c3_train_bf16_bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c3" -o snap -- \
  python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 3 --warmup 1) > $O/prof_c3.log 2>&1
cp $O/prof_c3/snap_kernel_stats.csv $O/r03_c3_train_bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_c3
python -c "
import json
for f in ('r03_c3_train_bench','r03_c3_train_bf16_bench'):
  j=json.load(open('$O/'+f+'.json')); print(f, j['value'], j['ms_per_step'])"
