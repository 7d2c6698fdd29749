#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_baseline_configs.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -3
