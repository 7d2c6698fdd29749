#!/bin/bash
# Final round-6 GPU pass: the whole -m gpu suite (log kept), smoke, the profile artefacts, the two tool logs.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
mkdir -p gpurun_out/r06f
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06f/r06_gpu_suite.log
tail -4 gpurun_out/r06f/r06_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r06f/smoke.log
timeout 300 python tools/torch_ops_in_step.py > gpurun_out/r06f/r06_torch_ops_c2.log 2>&1
timeout 300 python tools/torch_ops_in_step.py --mode train --workload c3 --precision bf16 > gpurun_out/r06f/r06_torch_ops_c3.log 2>&1
tail -1 gpurun_out/r06f/r06_torch_ops_c2.log
bash scripts/gpu_round6_profiles.sh > gpurun_out/r06f/profiles.log 2>&1
tail -3 gpurun_out/r06f/profiles.log
