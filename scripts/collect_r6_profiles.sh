#!/bin/bash
# Copies what scripts/gpu_r6_final.sh left under gpurun_out/r06p and gpurun_out/r06f into profiles/ (the
# files profiles/README.md lists for round 6).
set -eu
cd "$(dirname "$0")/.."
cp gpurun_out/r06p/r06_*.json gpurun_out/r06p/r06_*.csv gpurun_out/r06p/r06_*.txt gpurun_out/r06p/r06_*.log profiles/
cp gpurun_out/r06f/r06_*.log profiles/
python - <<'PY'
import json
d = json.loads(open('profiles/r06_c2_bench.json').read().strip().splitlines()[-1])
print('C2', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], ' traffic_stale', d['roofline'].get('traffic_stale'),
      ' conv', d['kernels']['conv_split_bf16x3']['ms'])
for leg in ('train_c3', 'train_c3_fp16', 'c4', 'c5'):
  print(leg, d[leg]['ms_per_step'])
PY
tail -2 profiles/r06_gpu_suite.log
