#!/bin/bash
# round 6: batches in flight (snap_amd.pipeline, bench.py --in-flight N): test + default bench line
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6if; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "in_flight or evaluator" 2>&1 | tail -4
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['config']['batches_in_flight'], d['step_ms'])
for k in ('one_batch_at_a_time','f32_exact','volume_materialized','train_c3','train_c3_fp16','c4','c5'):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('error'))
print(d['roofline']['frac'], d['roofline']['traffic_stale'])
PY
