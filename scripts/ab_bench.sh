#!/bin/bash
# A/B of two builds of libsnap_hip.so in ONE box (box-to-box spread is ~1.5 %):
#   make -C snap_amd/csrc OUT=../lib/alt/libsnap_hip.so OBJDIR=../lib/alt/obj [EXTRA=...]   (the "alt" build)
#   bash scripts/ab_bench.sh            # alternates default / alt, inference + train step
cd "$(dirname "$0")/.."
for lib in "" "snap_amd/lib/alt/libsnap_hip.so" "" "snap_amd/lib/alt/libsnap_hip.so"; do echo "== lib=${lib:-default}"; for mode in infer train; do
 if [ $mode = infer ]; then A="--steps 10 --warmup 3 --no-cpu-baseline"; else A="--mode train --workload c3 --steps 6 --warmup 2"; fi
 SNAP_HIP_LIB=${lib:+$PWD/$lib} python bench.py $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$mode', d['ms_per_step'], {n:round(v['ms'],2) for n,v in k.items() if v['ms'] > 1.5}, d['roofline']['achieved'])"; done; done
