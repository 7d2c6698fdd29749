#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python -m pytest tests/ -m gpu -q -x -k "mlp2 or without_feature_volume or whole_scene" 2>&1 | tail -3
b() { SNAP_HIP_LIB=${1:+$PWD/$1} timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 ${1:-default}', d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'],3) for n,v in d['kernels'].items() if v['ms']>0.4})"; }
b ""; b snap_amd/lib/alt_mlpold/libsnap_hip.so; b ""; b snap_amd/lib/alt_mlpold/libsnap_hip.so
timeout 300 python tools/mlp_pool_bench.py 2>/dev/null | tail -6
SNAP_HIP_LIB=$PWD/snap_amd/lib/alt_mlpold/libsnap_hip.so timeout 300 python tools/mlp_pool_bench.py 2>/dev/null | tail -6
