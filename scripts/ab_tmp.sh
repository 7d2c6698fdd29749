#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for lib in "" snap_amd/lib/alt_fastpro/libsnap_hip.so "" snap_amd/lib/alt_fastpro/libsnap_hip.so; do
  echo "== lib=${lib:-default}"
  SNAP_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python tools/conv_raw_bench.py 2>/dev/null | awk '{print $1,$2,$3,$4,"tiled",$10}' | head -6
  SNAP_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'], round(d['kernels']['conv_split_bf16x3']['ms'],3))"
done
