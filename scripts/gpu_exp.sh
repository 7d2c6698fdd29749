#!/bin/bash
# Tuning experiments: bench under env-knob variants; prints conv time per variant.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  env $v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/exp.log 2>&1
  python - "$v" <<'PY'
import json, sys
line = [l for l in open('gpurun_out/exp.log') if l.startswith('{')]
if not line:
    print(sys.argv[1], 'FAILED'); print(open('gpurun_out/exp.log').read()[-800:])
else:
    r = json.loads(line[-1])
    k = r['kernels']
    print(sys.argv[1], 'scenes/s', r['value'], 'ms', r['ms_per_step'], '| conv', k['conv_igemm']['ms'], 'TF', k['conv_igemm']['tflops'],
          '| lift', k['lift_pool']['ms'], 'gn', k['group_norm_stats']['ms'], 'pose', k['pose_score']['ms'], 'sim', k['sim_softmax']['ms'], 'ransac', k['ransac_sample']['ms'])
PY
done
