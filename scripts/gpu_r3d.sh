#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_presplit.py -m gpu -q --timeout=600 2>&1 | tail -3
for i in 1 2; do
echo "== default =="; timeout 300 python tools/ps_gemm_bench.py 2>&1 | grep -v amdgpu.ids
echo "== no stagger =="; SNAP_HIP_LIB=snap_amd/lib/alt_psnostag/libsnap_hip.so timeout 300 python tools/ps_gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
