#!/bin/bash
# Utilisation counters of one bench step (two PMC passes, counters only) -> gpurun_out/pmc_summary.json
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs$i
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmcs$i" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline) > gpurun_out/pmcs$i.log 2>&1
done
# the two passes have their own SQ_BUSY_CYCLES: summarise them separately, then merge
python tools/make_pmc_summary.py gpurun_out/pmc_summary_a.json gpurun_out/pmcs1/pmc_counter_collection.csv > /dev/null
python tools/make_pmc_summary.py gpurun_out/pmc_summary_b.json gpurun_out/pmcs2/pmc_counter_collection.csv > /dev/null
python - <<'PY'
import json
a = json.load(open('gpurun_out/pmc_summary_a.json')); b = json.load(open('gpurun_out/pmc_summary_b.json'))
for fam, e in b['families'].items():
    a['families'].setdefault(fam, {}).update(e)
json.dump(a, open('gpurun_out/pmc_summary.json', 'w'), indent=1)
for fam, e in sorted(a['families'].items()):
    print(fam, e)
PY
