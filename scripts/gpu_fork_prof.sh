#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=. TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/fork
for F in True False; do
cat > /tmp/run_$F.py <<PY
import sys
sys.path.insert(0, '$R')
from snap_amd.models import resnet
resnet.FORK_SHORTCUT = $F
import bench
bench.main(['--mode', 'train', '--workload', 'c3', '--precision', 'bf16', '--steps', '3', '--warmup', '1'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fork/p_$F -o snap -- python /tmp/run_$F.py) > gpurun_out/fork/log_$F.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/fork/p_$F/snap_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)/4e6
add=[(r['Calls'], int(r['TotalDurationNs'])/4e6) for r in rows if 'CUDAFunctor_add<float>' in r['Name']]
gn=[(r['Calls'], round(int(r['TotalDurationNs'])/4e6,3)) for r in rows if 'gn_bwd_apply' in r['Name']]
print('fork=$F total/step', round(tot,2), 'adds', add, 'gn_bwd_apply', gn)
PY
done
rm -rf gpurun_out/fork/p_True gpurun_out/fork/p_False
