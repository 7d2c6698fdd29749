#!/bin/bash
# PMC passes over the pre-split GEMM micro-benchmark
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 -L > $R/gpurun_out/r03/counters_list.txt 2>&1)
grep -c . gpurun_out/r03/counters_list.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/r03/pmcg$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/r03/pmcg$i" -o pmc -- \
    python "$R/tools/ps_gemm_bench.py" --iters 2 --only 131072) > gpurun_out/r03/pmcg$i.log 2>&1
  tail -1 gpurun_out/r03/pmcg$i.log | cut -c1-200
  python - "$i" <<'PY'
import csv, collections, glob, os, sys
f = glob.glob(f'gpurun_out/r03/pmcg{sys.argv[1]}/**/*counter_collection.csv', recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'conv_ps' not in k:
            continue
        k = k.split('(')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        print(k, {a: f'{b / cnt[(k, a)]:.4g}' for a, b in v.items()}, 'dispatches', max(cnt[(k, a)] for a in v))
else:
    print('no csv')
PY
done
