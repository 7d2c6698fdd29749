#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_conv_rs.py -m gpu -q -x -k "raw_row" 2>&1 | tail -6
timeout 300 python tools/conv_raw_bench.py 2>&1 | tail -12
for v in 0 1; do
python - <<PY
import json, subprocess, sys
from snap_amd import ops
PY
SNAP_NO_RAW=$v timeout 300 python - <<'PY'
import os, json, sys
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-extra-legs']
from snap_amd import ops
ops.CONV_NO_RAW = os.environ['SNAP_NO_RAW'] == '1'
import bench
d = bench.main(sys.argv[1:], emit=False)
print('NO_RAW=' + os.environ['SNAP_NO_RAW'], d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'], 3) for n, v in d['kernels'].items() if v['ms'] > 0.2})
PY
done
