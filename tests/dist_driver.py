"""Launched by test_distributed.py under torch.distributed.run (gloo, CPU).

Exercises bench.py's multi-rank path (rank/world parsing, per-rank scene shard,
barrier + max-over-ranks timing, rank-0 JSON) with the TEST-ONLY oracle backend
standing in for the HIP kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import oracle_ops  # noqa: E402
from snap_amd import ops  # noqa: E402

for name in oracle_ops.ALL_OPS:
  setattr(ops, name, getattr(oracle_ops, name))

import bench  # noqa: E402

if __name__ == '__main__':
  bench.main(sys.argv[1:])
