#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6if2; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --digest "$@" > $O/$tag.json 2> $O/$tag.err; }
run c2_if2_a   --steps 30 --warmup 3 --in-flight 2
run c2_if2_noaer --steps 30 --warmup 3 --in-flight 2 --tune OVERLAP_AERIAL=0
run c2_if1_noaer --steps 30 --warmup 3 --in-flight 1 --tune OVERLAP_AERIAL=0
run c2_if2_b   --steps 30 --warmup 3 --in-flight 2
run c4_if1 --workload c4 --steps 10 --warmup 2 --in-flight 1
run c4_if2 --workload c4 --steps 10 --warmup 2 --in-flight 2
run c5_if1 --workload c5 --steps 10 --warmup 2 --in-flight 1
run c5_if2 --workload c5 --steps 10 --warmup 2 --in-flight 2
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    dg=d.get('step_digests',[])
    print(f.split('/')[-1], d['config'].get('batches_in_flight'), d['ms_per_step'], d['value'], 'digest', sum(x[0] for x in dg), sum(x[1] for x in dg))
PY
# host enqueue time of one step (GPU idle before, no sync inside)
python - <<PY
import time, torch, bench
PY
python - <<PY
import time, torch, sys
sys.argv=['bench.py']
import bench
from snap_amd import ops
dev=torch.device('cuda',0)
loc,cfg,meta,variables,batch=bench.build('c2',dev,0,materialize_volume=False)
loc.engine='bf16x3'
for i in range(3):
    p=loc.apply(variables,batch,train=False,rngs={'sampling':i})
torch.cuda.synchronize()
ts=[]
for i in range(5):
    t=time.perf_counter(); p=loc.apply(variables,batch,train=False,rngs={'sampling':10+i}); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    ts.append((round((t1-t)*1e3,2), round((t2-t)*1e3,2)))
print('host enqueue ms / total ms per step (GPU idle at start):', ts)
PY
