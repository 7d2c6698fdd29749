#!/bin/bash
# The counter passes of scripts/gpu_round3_profiles.sh alone (HBM traffic + utilisation summary).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmc_$C" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs) > $O/pmc_$C.log 2>&1
done
python tools/make_hbm_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv $O/r03_c2_hbm_traffic.json 2 > $O/traffic.log 2>&1
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmcs$i" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs) > $O/pmcs$i.log 2>&1
done
python tools/make_pmc_summary.py $O/pmc_a.json $O/pmcs1/pmc_counter_collection.csv > /dev/null 2>&1
python tools/make_pmc_summary.py $O/pmc_b.json $O/pmcs2/pmc_counter_collection.csv > /dev/null 2>&1
python - <<PY
import json
a = json.load(open('$O/pmc_a.json')); b = json.load(open('$O/pmc_b.json'))
for fam, e in b['families'].items():
  a['families'].setdefault(fam, {}).update(e)
json.dump(a, open('$O/r03_c2_pmc_summary.json', 'w'), indent=1)
print(a['families'].get('conv_split'))
PY
rm -rf $O/pmcs1 $O/pmcs2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/traffic.log
