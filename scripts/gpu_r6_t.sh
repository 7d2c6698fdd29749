cd /root/repo; export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "inside_the_consumer" 2>&1 | tail -40
