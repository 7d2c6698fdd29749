"""Developer aid, NOT part of the test suite: `python -m pytest -p tools.dryrun_plugin -m gpu ...`
runs the TEST CODE of the `-m gpu` files on a machine without a GPU by pointing
`snap_amd.ops` at the numpy oracle, to catch typos / shape errors before spending GPU
minutes.  Results obtained this way prove nothing about the kernels (oracle vs oracle);
nothing in tests/, bench.py or the driver ever loads this plugin."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  import helpers
  import oracle_ops
  helpers.DEVICE = 'cpu'
  from snap_amd import ops
  for name in oracle_ops.ALL_OPS:
    setattr(ops, name, getattr(oracle_ops, name))
