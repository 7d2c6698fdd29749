#!/bin/bash
# round 6: wave-state / LDS counters of the voting_fft kernels inside the C4 bench (counters only: no other trace domain)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6vfpmc; rm -rf $O; mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES" \
         "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/p$i" -o pmc -- \
     python "$R/bench.py" --workload c4 --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline) > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-200
done
python - <<PY | tee $O/summary.txt
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for f in sorted(glob.glob('$O/p*/pmc_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'vf_' in k or 'pose_score' in k or 'sim_split' in k:
            # the big launches only (grid size separates the dot launch from the small ones)
            key=(k.split('(')[0][-40:], r.get('Grid_Size','?'))
            a=agg[key][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for key,v in sorted(agg.items()):
    wc=v.get('SQ_WAVE_CYCLES',[1,1]); wcv=wc[0]/max(wc[1],1)
    if wcv < 1e6: continue
    print('KERNEL',key)
    for c,(s,n) in sorted(v.items()):
        print(f'   {c:30s} {s/n:16.0f}  /wave_cycles {s/n/wcv:7.3f}')
PY
rm -rf $O/p*/ 2>/dev/null
