#!/bin/bash
# round 6: is the C3 train step host-bound?  kernel trace (idle time between kernels) + host enqueue time
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6c3g; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof" -o c3 -- \
  python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs) > $O/prof.log 2>&1
python tools/trace_gaps.py $O/prof/c3_kernel_trace.csv 120 | tee $O/gaps.txt
python - <<PY
import csv
rows=[]
for r in csv.DictReader(open('$O/prof/c3_kernel_trace.csv')):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:70]))
rows.sort()
# write a compact version of the last 130 ms for offline analysis
t_end=rows[-1][1]
with open('$O/c3_tail.csv','w') as f:
    for s,e,n in rows:
        if s>=t_end-130e6: f.write(f"{s-t_end},{e-t_end},{n}\n")
PY
rm -rf $O/prof
python - <<PY
import time, torch, sys
sys.argv=['bench.py']
import bench
PY
timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train_c3 unprofiled', d['ms_per_step'], d['step_ms'])"
