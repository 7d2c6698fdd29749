"""snap_amd.pipeline on a host without a GPU: the ring degrades to the caller's (only) queue."""
import pytest
import torch

from snap_amd import pipeline


def test_batches_in_flight_is_a_no_op_without_a_gpu():
  ring = pipeline.BatchesInFlight(3, torch.device('cpu'))
  assert ring.n == 1 and ring.streams == [None]
  seen = []
  for i in range(4):
    with ring.slot(i) as s:
      assert s is None
      seen.append(i)
  ring.synchronize()
  ring.join()
  assert seen == [0, 1, 2, 3]
  with pytest.raises(ValueError):
    pipeline.BatchesInFlight(0)
