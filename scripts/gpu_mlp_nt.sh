#!/bin/bash
# mlp2_pool at 256 threads / 128 rows (default) against 512 threads / 256 rows
# (alt build first: scripts/build_alt.sh mlpnt512 mlp_pool.hip -DSNAP_MLP_POOL_NT=512): tests + bench
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "mlp2_pool or fused or plane" --timeout=300 2>&1 | tail -5
for lib in default mlpnt512; do
  if [ "$lib" = default ]; then unset SNAP_HIP_LIB; else export SNAP_HIP_LIB=snap_amd/lib/alt_$lib/libsnap_hip.so; fi
  timeout 300 python bench.py --no-extra-legs --steps 30 --dump gpurun_out/r03/launches_mlp_$lib.json > gpurun_out/r03/bench_mlp_$lib.log 2>&1
  python - <<PY
import json
line=open('gpurun_out/r03/bench_mlp_$lib.log').read().strip().splitlines()[-1]
d=json.loads(line); print('$lib', d['ms_per_step'], d['step_ms'])
l=json.load(open('gpurun_out/r03/launches_mlp_$lib.json'))
print({k: round(sum(x[1] for x in v),3) for k,v in l.items()})
PY
done
