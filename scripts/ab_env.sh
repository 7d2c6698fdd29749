#!/bin/bash
# A/B of env-knob variants of ONE build inside one box: bash scripts/ab_env.sh "VAR=0" "VAR=1" ...
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in "$@"; do echo "== $v";
 env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(d['ms_per_step'], {n:round(v['ms'],2) for n,v in k.items() if v['ms'] > 0.3})"; done; done
