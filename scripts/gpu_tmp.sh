#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
for l in "" snap_amd/lib/alt_mlpw2/libsnap_hip.so snap_amd/lib/alt_mlpnoscan/libsnap_hip.so ""; do
echo "lib=${l:-default}"; SNAP_HIP_LIB=${l:+$PWD/$l} timeout 300 python tools/mlp_pool_bench.py 2>/dev/null | tail -1; done
