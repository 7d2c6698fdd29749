#!/bin/bash
# One gpurun call: the whole -m gpu suite + the driver's bench line + smoke.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05c
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/tests.log
cat $O/tests.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench_c2.log 2>$O/bench_c2.err
tail -1 $O/bench_c2.log > $O/r05_c2_bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05c/r05_c2_bench.json'))
print('C2', d['ms_per_step'], d['value'], 'roofline', d['roofline'].get('frac'), d['roofline'].get('ms'))
for k in ('f32_exact', 'volume_materialized', 'train_c3', 'train_c3_fp16', 'c4', 'c5'):
  print(k, {a: b for a, b in d.get(k, {}).items() if a in ('ms_per_step', 'error', 'is_finite', 'loss_scale', 'leg_wall_s')})
print({n: round(v['ms'], 3) for n, v in d['kernels'].items() if v['ms'] > 0.2})
PY
