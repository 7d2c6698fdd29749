#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for lib in "" snap_amd/lib/alt_vf4/libsnap_hip.so "" snap_amd/lib/alt_vf4/libsnap_hip.so; do
  SNAP_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --workload c4 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=${lib:-default}', d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'],3) for n,v in d['kernels'].items()})"
done
SNAP_HIP_LIB=$PWD/snap_amd/lib/alt_vf4/libsnap_hip.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "voting_fft or rotate_templates" 2>&1 | tail -2
