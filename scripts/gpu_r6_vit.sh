#!/bin/bash
# round 6: the one-part pre-split engine under the ViT encoder -- tests, C5 A/B (process-default ops.BF16_PS), kernel stats
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r6vit; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py tests/test_gpu_kernels.py -m gpu -q -k "vit or attention or layer_norm or bf16_ring or bf16_outputs or dense" 2>&1 | tail -4 | tee $O/tests.log
for rep in 1 2; do for ps in 1 0; do
  echo "== BF16_PS=$ps"
  python -c "
import sys
from snap_amd import ops
ops.BF16_PS = bool($ps)
import bench
bench.main(['--workload', 'c5', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-extra-legs'])" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {n:round(v['ms'],3) for n,v in d.get('kernels',{}).items() if v['ms']>0.2})"
done; done 2>&1 | tee $O/ab.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o snap -- \
    python "$R/bench.py" --workload c5 --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extra-legs) > $O/prof.log 2>&1
cp $O/prof/snap_kernel_stats.csv $O/c5_kernel_stats.csv; rm -rf $O/prof
head -8 $O/c5_kernel_stats.csv | cut -c1-150
