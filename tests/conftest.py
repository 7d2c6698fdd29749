"""pytest configuration: the `gpu` marker and the oracle-backend fixture."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a ROCm GPU (run on the MI355X box with `-m gpu`)'
  )


@pytest.fixture
def oracle_backend(monkeypatch):
  """Swap every ``snap_amd.ops`` kernel wrapper for its numpy-oracle twin.

  TEST-ONLY: lets the host-side module code (pytrees, configs, shapes, call
  flags) run on a machine without a GPU.  The product has no such switch.
  """
  import oracle_ops
  from snap_amd import ops

  for name in oracle_ops.ALL_OPS:
    monkeypatch.setattr(ops, name, getattr(oracle_ops, name))
  return oracle_ops


def pytest_terminal_summary(terminalreporter):
  """How often helpers.assert_same_argmax took its fp32 near-tie escape in this session."""
  try:
    import helpers
  except ImportError:
    return
  n = helpers.NEAR_TIE
  if n['rows']:
    terminalreporter.write_line(
        f"[argmax parity] {n['rows']} rows compared with the oracle's argmax, "
        f"{n['escapes']} near-tie escape(s)")
