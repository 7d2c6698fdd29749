#!/bin/bash
# A/B of builds of libsnap_hip.so inside ONE box: bash scripts/ab_lib.sh "" snap_amd/lib/alt/libsnap_hip.so ...
# (make -C snap_amd/csrc OUT=../lib/alt/libsnap_hip.so OBJDIR=../lib/alt/obj [EXTRA=...] builds one)
cd "$(dirname "$0")/.."
for lib in "$@"; do echo "== lib=${lib:-default}";
 SNAP_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(d['ms_per_step'], {n:round(v['ms'],2) for n,v in k.items() if v['ms'] > 0.3})"; done
