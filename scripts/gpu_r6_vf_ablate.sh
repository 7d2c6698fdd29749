#!/bin/bash
# round 6: timing ablations of the voting dot launch (alt builds, WRONG results): which phase the pass waits for
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6vfab; rm -rf $O; mkdir -p $O
for tag in default "$@"; do
  lib=""; [ $tag != default ] && lib=$R/snap_amd/lib/alt_$tag/libsnap_hip.so
  (cd /tmp && SNAP_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$tag" -o snap -- \
    python "$R/bench.py" --workload c4 --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline) > $O/prof_$tag.log 2>&1
  python - $O/prof_$tag/snap_kernel_stats.csv $tag <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'vf_' in r['Name']:
        print(sys.argv[2], r['Name'].split('::')[-1][:28], r['Calls'], 'max_us', round(float(r['MaxNs'])/1e3,1), 'total_ms', round(float(r['TotalDurationNs'])/1e6,3))
PY
  rm -rf $O/prof_$tag
done 2>&1 | tee $O/ablate.log
