#!/bin/bash
# PMC passes (counters only) over one conv layer: bash scripts/gpu_pmc_conv.sh <tag> <layer> <math> [ENV=VAL ...]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=$1; LAYER=$2; MATH=$3; shift 3
for kv in "$@"; do export "$kv"; done
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
         "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcc_${TAG}_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmcc_${TAG}_$i" -o pmc -- \
    python "$R/tools/conv_one.py" $LAYER $MATH 3) > gpurun_out/pmcc_${TAG}_$i.log 2>&1
  K="conv_" python - "gpurun_out/pmcc_${TAG}_$i" "$TAG" <<'PY'
import csv, collections, glob, os, sys
f = glob.glob(f'{sys.argv[1]}/*counter_collection.csv')
if not f:
    print(sys.argv[2], 'no counter csv'); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_' not in k or 'pack' in k:
        continue
    k = k.split('(')[0][-60:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    print(sys.argv[2], k, {a: f'{b / cnt[(k, a)]:.4g}' for a, b in v.items()})
PY
done
