#!/bin/bash
# Round-4 check: the whole -m gpu suite, smoke, the driver's bench command (with the C3/C4/C5 legs).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${*:-tests smoke bench}"
for s in $STAGES; do
  case $s in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -40 > gpurun_out/tests.log
      echo "== tests =="; tail -15 gpurun_out/tests.log ;;
    new)
      timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_kernels.py tests/test_gpu_backward.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout=600 -k "rccl or nan or pose_score_bwd or mlp2_pool or vertical_pool or matching_dim_64 or c4_exhaustive" -s 2>&1 | tail -60 > gpurun_out/tests_new.log
      echo "== new tests =="; grep -E "passed|failed|RCCL1_OK|\[dist\]|Error|assert" gpurun_out/tests_new.log | head -30 ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
      echo "== smoke =="; tail -3 gpurun_out/smoke.log ;;
    bench)
      T0=$SECONDS
      SNAP_BENCH_DUMP=gpurun_out/launches.json timeout 1200 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err
      echo "== bench == wall $((SECONDS - T0)) s"; tail -c 6000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err ;;
    trace)
      # kernel trace of 3 steps: GPU busy / idle time per step (tools/trace_gaps.py)
      rm -rf gpurun_out/trace
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/trace" -o snap -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs) > gpurun_out/trace.log 2>&1
      f=$(find gpurun_out/trace -name '*kernel_trace.csv' | head -1)
      echo "== trace =="; tail -2 gpurun_out/trace.log | cut -c1-400
      [ -n "$f" ] && python tools/trace_gaps.py "$f" > gpurun_out/trace_gaps.txt && cat gpurun_out/trace_gaps.txt && rm -rf gpurun_out/trace ;;
  esac
done
