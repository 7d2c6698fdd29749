#!/bin/bash
# One gpurun call: parity tests, smoke, bench line, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [stage...]
# Stages: tests smoke bench prof   (default: all)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${*:-tests smoke bench prof}"
for s in $STAGES; do
  case $s in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -60 > gpurun_out/tests.log
      echo "== tests =="; tail -40 gpurun_out/tests.log ;;
    testsall)
      # one process per kernel family: a GPU memory fault aborts only its group.
      : > gpurun_out/tests.log
      for k in "conv or dense or weight_standardize" "gn_stats" "lift or project" \
               "vertical_pool or plane_fuse or matching_zero" "sim_softmax or ransac or poses_from_corr" \
               "pose_score or refine_lattice" "rotate_templates or exhaustive"; do
        echo "##### -k '$k'" >> gpurun_out/tests.log
        timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "$k" 2>&1 | tail -45 >> gpurun_out/tests.log
      done
      echo "##### model" >> gpurun_out/tests.log
      timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout=600 2>&1 | tail -60 >> gpurun_out/tests.log
      echo "== tests (grouped) =="; grep -E "#####|passed|failed|error|Error|assert|beyond|mismatch|Fault|fault|Abort" gpurun_out/tests.log | head -120 ;;
    extra)
      timeout 1500 python -m pytest tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -q --timeout=900 2>&1 | tail -60 > gpurun_out/tests_extra.log
      echo "== golden + full-size =="; tail -40 gpurun_out/tests_extra.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
      echo "== smoke =="; tail -5 gpurun_out/smoke.log ;;
    bench)
      SNAP_BENCH_DUMP=gpurun_out/launches.json timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
      echo "== bench =="; tail -5 gpurun_out/bench.log ;;
    train)
      timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -m gpu -q --timeout=600 2>&1 | tail -40 > gpurun_out/tests_train.log
      echo "== train tests =="; tail -5 gpurun_out/tests_train.log ;;
    benchtrain)
      SNAP_BENCH_DUMP=gpurun_out/launches_train.json timeout 1200 python bench.py --mode train --workload c3 --steps 8 --warmup 2 > gpurun_out/bench_train.log 2>&1
      echo "== bench train =="; tail -5 gpurun_out/bench_train.log | cut -c1-3000 ;;
    benchtrainbf16)
      SNAP_BENCH_DUMP=gpurun_out/launches_train_bf16.json timeout 1200 python bench.py --mode train --workload c3 --precision bf16 --steps 8 --warmup 2 > gpurun_out/bench_train_bf16.log 2>&1
      echo "== bench train bf16 =="; tail -5 gpurun_out/bench_train_bf16.log | cut -c1-3000 ;;
    proftrain)
      rm -rf gpurun_out/proftrain
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/proftrain" -o snap -- \
        python "$OLDPWD/bench.py" --mode train --workload c3 --steps 2 --warmup 1 ${TRAIN_ARGS:-}) > gpurun_out/proftrain.log 2>&1
      echo "== prof train =="; tail -2 gpurun_out/proftrain.log | cut -c1-300
      f=$(find gpurun_out/proftrain -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && head -40 "$f" | cut -c1-220 ;;
    prof)
      rm -rf gpurun_out/prof
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o snap -- \
        python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline) > gpurun_out/prof.log 2>&1
      echo "== prof =="; tail -3 gpurun_out/prof.log
      f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && head -25 "$f" ;;
  esac
done
