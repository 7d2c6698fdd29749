#!/bin/bash
# One gpurun call: every artefact profiles/README.md lists for round 6 (written under gpurun_out/r06p/).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
# 2. rocprofv3 kernel stats of the same workload (1 warm-up + 3 timed steps, no extra legs), ONE batch at a
#    time: with two batches in flight (the default) two launches share the GPU and both durations are inflated;
#    the line's roofline times its last step alone for the same reason.  (_inflight2: the default, for the record.)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o snap -- \
  python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --in-flight 1) > $O/prof.log 2>&1
cp $O/prof/snap_kernel_stats.csv $O/r06_c2_kernel_stats.csv 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof2" -o snap -- \
  python "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs) > $O/prof2.log 2>&1
cp $O/prof2/snap_kernel_stats.csv $O/r06_c2_kernel_stats_inflight2.csv 2>/dev/null
# 3. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate counter-only passes
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmc_$C" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --in-flight 1) > $O/pmc_$C.log 2>&1
done
SNAP_GIT_HEAD=${SNAP_GIT_HEAD:-unknown} python tools/make_hbm_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv $O/r06_c2_hbm_traffic.json 2 > $O/traffic.log 2>&1
# 1. the headline line exactly as the driver runs it (default flags: CPU baseline + the extra legs), AFTER the
#    traffic record of these sources exists (bench.py reads profiles/r06_c2_hbm_traffic.json: traffic_stale false)
cp $O/r06_c2_hbm_traffic.json profiles/r06_c2_hbm_traffic.json 2>/dev/null
SNAP_BENCH_DUMP=$O/r06_c2_launches.json timeout 900 python bench.py > $O/bench_c2.log 2>&1
tail -1 $O/bench_c2.log > $O/r06_c2_bench.json
# 4. utilisation counters
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$O/pmcs$i" -o pmc -- \
    python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --in-flight 1) > $O/pmcs$i.log 2>&1
done
python tools/make_pmc_summary.py $O/pmc_a.json $O/pmcs1/pmc_counter_collection.csv > /dev/null 2>&1
python tools/make_pmc_summary.py $O/pmc_b.json $O/pmcs2/pmc_counter_collection.csv > /dev/null 2>&1
python - <<PY
import json
try:
  a = json.load(open('$O/pmc_a.json')); b = json.load(open('$O/pmc_b.json'))
  for fam, e in b['families'].items():
    a['families'].setdefault(fam, {}).update(e)
  json.dump(a, open('$O/r06_c2_pmc_summary.json', 'w'), indent=1)
except Exception as e:
  print('pmc summary failed', e)
PY
# 5. C4 (eval path): the pre-split GEMM engine behind the exhaustive voting + kernel stats
SNAP_BENCH_DUMP=$O/r06_c4_launches.json timeout 300 python bench.py --workload c4 --steps 12 --warmup 2 2>/dev/null | tail -1 > $O/r06_c4_bench.json
timeout 300 python bench.py --workload c4 --steps 12 --warmup 2 --in-flight 1 2>/dev/null | tail -1 > $O/r06_c4_bench_one_at_a_time.json
timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 --voting direct 2>/dev/null | tail -1 > $O/r06_c4_bench_direct.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c4" -o snap -- \
  python "$R/bench.py" --workload c4 --steps 2 --warmup 1 --in-flight 1) > $O/prof_c4.log 2>&1
cp $O/prof_c4/snap_kernel_stats.csv $O/r06_c4_kernel_stats.csv 2>/dev/null
# 6. training step (C3): f32 / bf16 lines, kernel stats of the bf16 step
timeout 300 python bench.py --mode train --workload c3 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r06_c3_train_bench.json
SNAP_BENCH_DUMP=$O/r06_c3_train_bf16_launches.json timeout 300 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r06_c3_train_bf16_bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_c3" -o snap -- \
  python "$R/bench.py" --mode train --workload c3 --precision bf16 --steps 3 --warmup 1) > $O/prof_c3.log 2>&1
cp $O/prof_c3/snap_kernel_stats.csv $O/r06_c3_train_bf16_kernel_stats.csv 2>/dev/null
timeout 300 python bench.py --workload c5 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/r06_c5_vit_bench.json
timeout 300 python bench.py --mode train --workload c3 --precision fp16 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r06_c3_train_fp16_bench.json
python tools/family_table.py $O/r06_c3_train_bf16_launches.json > $O/r06_c3_train_bf16_layers.txt 2>&1
timeout 200 python tools/wgrad_bench.py 2>/dev/null | tail -1 > $O/r06_wgrad_bench.json
# 7. the operand path's own ceiling (global -> LDS by LDS-DMA, no compute): tools/lds_dma_probe.hip
[ -x tools/lds_dma_probe ] && timeout 120 ./tools/lds_dma_probe > $O/r06_lds_dma_probe.log 2>&1
# 8. whole-scene parity on the bench's configuration (one oracle run) + the eval variant
timeout 500 python tools/fullsize_parity.py --math f32,bf16x3,bf16x3+plane --out $O/r06_c2_fullsize_parity.json > $O/parity.log 2>&1
timeout 300 python -m pytest tests/test_gpu_distributed.py -m gpu -q -s 2>&1 | grep -E 'RCCL1_OK|DIST_GPU_OK|passed|failed' > $O/r06_rccl_single_rank.log
rm -rf $O/prof $O/prof2 $O/prof_c4 $O/prof_c3 $O/pmcs1 $O/pmcs2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
head -c 700 $O/r06_c2_bench.json; echo
cat $O/traffic.log
