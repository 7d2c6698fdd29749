#!/bin/bash
# One gpurun call: every artefact profiles/README.md lists for round 2 (written under gpurun_out/r02/).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
# 1. bench lines (the default engine with the CPU baseline; the other two engines without)
SNAP_BENCH_DUMP=$O/launches_bf16x3.json timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_bf16x3.log 2>&1
tail -1 $O/bench_bf16x3.log > $O/r02_c2_bench_bf16x3.json
for m in bf16x6 f32; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --math $m 2>/dev/null | tail -1 > $O/r02_c2_bench_$m.json
done
# (the default engine with the dense feature volume written: the unfused MLP -> fill -> pool chain)
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --materialize-volume 2>/dev/null | tail -1 > $O/r02_c2_bench_bf16x3_volume.json
# 2. rocprofv3 kernel stats of the same command
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o snap -- \
  python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline) > $O/prof.log 2>&1
cp $O/prof/snap_kernel_stats.csv $O/r02_c2_kernel_stats.csv 2>/dev/null
# 3. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate counter-only passes
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmc_$C" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline) > $O/pmc_$C.log 2>&1
done
python tools/make_hbm_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv $O/r02_c2_hbm_traffic.json 2 > $O/traffic.log 2>&1
# 4. utilisation counters
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmcs$i" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline) > $O/pmcs$i.log 2>&1
done
python tools/make_pmc_summary.py $O/pmc_a.json $O/pmcs1/pmc_counter_collection.csv > /dev/null 2>&1
python tools/make_pmc_summary.py $O/pmc_b.json $O/pmcs2/pmc_counter_collection.csv > /dev/null 2>&1
python - <<PY
import json
a = json.load(open('$O/pmc_a.json')); b = json.load(open('$O/pmc_b.json'))
for fam, e in b['families'].items():
    a['families'].setdefault(fam, {}).update(e)
json.dump(a, open('$O/r02_c2_pmc_summary.json', 'w'), indent=1)
PY
# 5. C4 (eval path) on the f32 and the default engine + kernel stats
for m in f32 bf16x3; do
  timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 --math $m 2>/dev/null | tail -1 > $O/r02_c4_bench_$m.json
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c4" -o snap -- \
  python "$R/bench.py" --workload c4 --steps 2 --warmup 1) > $O/prof_c4.log 2>&1
cp $O/prof_c4/snap_kernel_stats.csv $O/r02_c4_kernel_stats.csv 2>/dev/null
# 6. whole-scene parity of every engine (one oracle run) + the eval variant on the default engine
timeout 400 python tools/fullsize_parity.py --math f32,bf16x6,bf16x3,bf16x3+plane --out $O/r02_c2_fullsize_parity.json > $O/parity.log 2>&1
timeout 400 python tools/fullsize_parity.py --math bf16x3+plane --eval --out $O/r02_c2_fullsize_parity_eval.json > $O/parity_eval.log 2>&1
# 7. training step (C3) and the ViT workload (C5)
timeout 300 python bench.py --mode train --workload c3 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r02_c3_train_bench.json
timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r02_c3_train_bf16_bench.json
timeout 300 python bench.py --mode train --workload c3 --precision bf16x3 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r02_c3_train_bf16x3_bench.json
timeout 300 python bench.py --workload c5 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/r02_c5_vit_bench.json
rm -rf $O/prof $O/prof_c4 $O/pmcs1 $O/pmcs2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
head -c 600 $O/r02_c2_bench_bf16x3.json; echo
cat $O/traffic.log
