#!/bin/bash
# round 6: wave-state counters of single conv launches (where does a wave's time go: parked / issue-stalled / active)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6pmc; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/$O/sq_counters.txt)
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $R/$O/mem_counters.txt)
wc -l $O/*.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
         "SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES" \
         "SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
         "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for L in 9 6 10; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/p${i}_L$L" -o pmc -- \
      python "$R/tools/conv_one_time.py" $L bf16x3) > $O/p${i}_L$L.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for f in sorted(glob.glob('$O/p*_L*/pmc_counter_collection.csv')):
    L=f.split('_L')[1].split('/')[0]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'conv_split' in k or 'halo' in k or 'conv1x1' in k:
            a=agg[(L,k[:70])][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for (L,k),v in sorted(agg.items()):
    print('LAYER',L,k)
    wc=v.get('SQ_WAVE_CYCLES',[1,1]); wcv=wc[0]/wc[1]
    for c,(s,n) in sorted(v.items()):
        print(f'   {c:36s} {s/n:14.0f}  /wave_cycles {s/n/wcv:7.3f}')
PY
rm -rf $O/p*_L*/ 2>/dev/null
