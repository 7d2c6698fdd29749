#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_presplit.py -m gpu -q --timeout=600 2>&1 | tail -8
echo "== default =="; timeout 300 python tools/ps_gemm_bench.py 2>&1 | grep -v amdgpu.ids
for a in ${ABL:-256 512}; do
  echo "== ablate $a =="; SNAP_HIP_LIB=snap_amd/lib/alt_psabl$a/libsnap_hip.so timeout 300 python tools/ps_gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
