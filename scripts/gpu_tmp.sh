#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q -x -k "vertical_pool or train_step or gradient or volume" 2>&1 | tail -4
timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs --dump gpurun_out/c3_launches.json 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d.get('step_ms'), {n: round(v['ms'],3) for n,v in d.get('kernels',{}).items() if v['ms']>0.5})"
python - <<'PY'
import json
d=json.load(open('gpurun_out/c3_launches.json'))
for k in d:
  if 'vertical' in k or 'lift' in k: print(k, [(round(x[1],3)) for x in d[k]])
PY
