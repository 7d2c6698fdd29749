#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
run() { timeout 300 python - "$@" <<'PY'
import sys, json, io, contextlib
from snap_amd import ops
mode = sys.argv[1]
if mode == 'overlap': ops.OVERLAP_AERIAL_TRAIN = True
import bench
sys.argv = ['bench.py', '--mode', 'train', '--workload', 'c3', '--precision', 'bf16', '--steps', '8', '--warmup', '3', '--no-cpu-baseline', '--no-extra-legs']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
  bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(mode, d['ms_per_step'], d['step_ms'], 'loss', d.get('loss'), d.get('grad_norm'))
PY
}
run default; run overlap; run default; run overlap
