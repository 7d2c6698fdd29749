#!/bin/bash
# Copies what scripts/gpu_r5_final.sh left under gpurun_out/r05p and gpurun_out/r05f into profiles/ (the
# files profiles/README.md lists for round 5).
set -eu
cd "$(dirname "$0")/.."
cp gpurun_out/r05p/r05_*.json gpurun_out/r05p/r05_*.csv gpurun_out/r05p/r05_*.txt gpurun_out/r05p/r05_*.log profiles/
cp gpurun_out/r05f/r05_*.log profiles/
python - <<'PY'
import json
d = json.loads(open('profiles/r05_c2_bench.json').read().strip().splitlines()[-1])
print('C2', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], ' traffic_stale', d['roofline'].get('traffic_stale'),
      ' conv', d['kernels']['conv_split_bf16x3']['ms'])
for leg in ('train_c3', 'train_c3_fp16', 'c4', 'c5'):
  print(leg, d[leg]['ms_per_step'])
PY
tail -2 profiles/r05_gpu_suite.log
