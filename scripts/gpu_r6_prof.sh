#!/bin/bash
# round 6: quick kernel-level profile of the C2 step (rocprofv3 kernel stats, 1 warm-up + 3 steps)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6q; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "inside_the_consumer or without_feature_volume" 2>&1 | tail -5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o snap -- \
  python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs "$@") > $O/prof.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/prof/snap_kernel_stats.csv')))
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs'])/1e6/4:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
print('total ms/step', sum(float(r['TotalDurationNs']) for r in rows)/1e6/4)
PY
