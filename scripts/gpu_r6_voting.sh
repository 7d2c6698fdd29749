#!/bin/bash
# One gpurun call: packed-math voting_fft -- parity tests, the C4 bench line A/B (default vs the scalar-math alt build), kernel stats.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r6vf
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "voting or template or overlap or fft" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
for rep in 1 2; do
for lib in "" ${ALTS:-snap_amd/lib/alt_vfold/libsnap_hip.so}; do
  echo "== lib=${lib:-default}"
  SNAP_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --workload c4 --steps 12 --warmup 3 --digest 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('step_digests') or [None])[0], {n:round(v['ms'],3) for n,v in d.get('kernels',{}).items()})"
done; done 2>&1 | tee $O/ab.log
for tag in packed scalar; do
  lib=""; [ $tag = scalar ] && lib=$R/snap_amd/lib/alt_vfold/libsnap_hip.so
  (cd /tmp && SNAP_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$tag" -o snap -- \
    python "$R/bench.py" --workload c4 --steps 3 --warmup 1 --in-flight 1) > $O/prof_$tag.log 2>&1
  cp $O/prof_$tag/snap_kernel_stats.csv $O/c4_kernel_stats_$tag.csv 2>/dev/null
  rm -rf $O/prof_$tag
  echo "-- $tag"; head -8 $O/c4_kernel_stats_$tag.csv | cut -c1-160
done
