"""Batches in flight: consecutive batches of an inference / evaluation loop on alternating HIP streams.

One forward pass of the localiser is ~400 dependent kernel launches; between two dependent launches
the GPU drains the last wave of workgroups of the first and dispatches the second (a few
microseconds each, and a partly empty chip during every tail).  The batches of an evaluation loop
do not depend on each other (``snap/evaluator.py:205-238`` iterates the dataset; under JAX the
dispatch is asynchronous but the device still executes one batch after the other), so batch i + 1
can be enqueued on a SECOND stream while batch i runs: its kernels fill the tails and dispatch gaps
of the other stream's.  Every batch does all of its work and produces the same bits whatever the
number of streams (``tests/test_gpu_model.py::test_batches_in_flight_same_bits``).

Measured at C2 (8 scenes per batch, MI355X, one box, ``bench.py --in-flight N``): N = 1 21.2-21.3 ms
per batch, N = 2 20.0-20.3 ms, N = 3 21.0 ms; C4 (launch-bound) 5.4 -> 4.2 ms; C5 18.6 -> 17.4 ms.
Ordering one stage across the streams (the encoders of batch i + 1 behind the encoders of batch i,
so that heads meet tails) measured SLOWER (20.9 ms): what pays is any second queue of ready
kernels, not which ones.

The caching allocator pools blocks per stream, so N batches in flight hold N sets of activations
(~25 GB each at C2 of the 288 GB).  Tensors a batch produces are consumed on its own stream; a
result handed to the host (``.cpu()``) synchronises that stream only.
"""
import contextlib

import torch


class BatchesInFlight:
  """``n`` streams taken in turn (``n`` = 1, or a CPU device: torch's current stream, no-op).

    ring = BatchesInFlight(2, device)
    for i, batch in enumerate(batches):
      with ring.slot(i):
        out[i] = model.apply(variables, batch, ...)     # enqueued on stream i % 2
    ring.join()                                          # the current stream waits for both
  """

  def __init__(self, n=2, device=None):
    n = int(n)
    if n < 1:
      raise ValueError(f'BatchesInFlight: n must be >= 1, got {n}')
    dev = torch.device(device) if device is not None else None
    self.cuda = torch.cuda.is_available() and (dev is None or dev.type == 'cuda')
    self.n = n if self.cuda else 1
    self.streams = [None] * self.n
    self._first = None                # event behind the first batch through the ring
    self._seen_first = set()
    if self.cuda and self.n > 1:
      self.device = dev if (dev is not None and dev.index is not None) else torch.device('cuda', torch.cuda.current_device())
      self.home = torch.cuda.current_stream(self.device)
      self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
      for s in self.streams:
        s.wait_stream(self.home)      # inputs / parameters were produced on the current stream

  def stream(self, i):
    return self.streams[i % self.n]

  @contextlib.contextmanager
  def slot(self, i):
    """Work enqueued inside runs on stream ``i % n``, behind whatever the caller's stream holds now
    (the batch's inputs).  Tensors allocated OUTSIDE the slot and read inside it must stay alive until
    the slot is synchronised / joined (the allocator knows only the stream that allocated them)."""
    s = self.streams[i % self.n]
    if s is None:
      yield None
      return
    s.wait_stream(torch.cuda.current_stream(self.device))
    k = i % self.n
    if self._first is not None and k not in self._seen_first:
      # the other streams start behind the FIRST batch: whatever it built lazily on its stream (device
      # constants, packed tables) is complete before another stream reads it
      s.wait_event(self._first)
      self._seen_first.add(k)
    with torch.cuda.stream(s):
      yield s
    if self._first is None:
      self._first = s.record_event()
      self._seen_first.add(k)

  def synchronize(self, i=None):
    """Host waits for slot ``i`` (default: all of them)."""
    for k, s in enumerate(self.streams):
      if s is not None and (i is None or k == i % self.n):
        s.synchronize()

  def join(self):
    """The stream that was current at construction waits for every slot (no host wait)."""
    for s in self.streams:
      if s is not None:
        self.home.wait_stream(s)
