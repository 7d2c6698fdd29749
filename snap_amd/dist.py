"""Data-parallel exchange step of the training loop (the only collective of the path).

The reference shards scenes over devices with ``jax.pmap(axis_name='batch')`` and has
exactly three cross-device operations:
  * ``jax.lax.pmean(grad, 'batch')``            snap/trainer.py:225-234
  * the finite flag of ``DynamicScale`` / the non-finite step skip   :223-229,:260-277
  * ``psum`` of (sum(metric), count) pairs      snap/trainer.py:57-67
Here: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" on CPU for tests).  Gradients are packed into a few large flat fp32
buckets -- xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring
all-reduce is per-link bound and wants few, large messages: the ~48 M-parameter
model is ~193 MB = 3 buckets of 64 MiB.  Inference needs no collective at all.
"""
from typing import Dict, Iterable, List, Tuple

import torch
import torch.distributed as dist


# Testing aid: with a process group of ONE rank every path below would
# short-circuit and never touch the collective library.  ``FORCE_COLLECTIVES = True`` makes an
# initialised group of any size issue its all-reduces, so that a single-GPU box can prove that
# RCCL loads next to libsnap_hip.so, that the bucket buffers are device tensors RCCL accepts and
# that hooks issued from autograd threads order correctly with RCCL's stream.
FORCE_COLLECTIVES = False


def _world(group=None):
  if not (dist.is_available() and dist.is_initialized()):
    return 1
  return dist.get_world_size(group)


def _exchanges(group=None) -> bool:
  """True when the exchange step must really run its collectives."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def flatten_tree(tree, prefix=''):
  """Nested dict of tensors -> list of (name, tensor) in a deterministic order."""
  out = []
  if isinstance(tree, dict):
    for k in sorted(tree):
      out.extend(flatten_tree(tree[k], f'{prefix}/{k}' if prefix else k))
  elif tree is not None:
    out.append((prefix, tree))
  return out


def allreduce_mean_(tensors: Iterable[torch.Tensor], group=None,
                    bucket_bytes: int = 64 << 20) -> int:
  """In-place mean over ranks of every tensor (``pmean``), bucketed.

  Returns the number of all-reduce calls issued.  Tensors of one call must share a
  device; fp32 accumulation (lower-precision grads are up-cast in the bucket).
  """
  tensors = [t for t in tensors if t is not None]
  world = _world(group)
  if not _exchanges(group) or not tensors:
    return 0
  calls = 0
  bucket: List[torch.Tensor] = []
  size = 0

  def flush():
    nonlocal bucket, size, calls
    if not bucket:
      return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in bucket])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    off = 0
    for t in bucket:
      n = t.numel()
      t.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
      off += n
    calls += 1
    bucket, size = [], 0

  for t in tensors:
    nbytes = t.numel() * 4
    if bucket and size + nbytes > bucket_bytes:
      flush()
    bucket.append(t)
    size += nbytes
  flush()
  return calls


class OverlappedGradReducer:
  """``pmean`` of the gradients, overlapped with the backward pass.

  The leaves are bucketed in REVERSE order (the order in which the backward pass tends to
  finish them); a post-accumulate hook on every leaf copies its ``.grad`` into the bucket's
  flat buffer and, when the last gradient of a bucket has arrived, issues that bucket's
  all-reduce asynchronously (RCCL runs it on its own stream while the remaining backward
  kernels keep the compute stream busy).  ``finish()`` waits for the outstanding buckets
  and returns the averaged gradients in leaf order.  Leaves that receive no gradient count
  as zeros (checked in ``finish``).  Same values as ``allreduce_mean_`` (same bucket sums).
  """

  def __init__(self, leaves: List[torch.Tensor], group=None, bucket_bytes: int = 64 << 20):
    self.leaves = list(leaves)
    self.group = group
    self.world = _world(group)
    self.exchanges = _exchanges(group)
    self.buckets: List[List[int]] = []
    cur: List[int] = []
    size = 0
    for i in reversed(range(len(self.leaves))):
      nbytes = self.leaves[i].numel() * 4
      if cur and size + nbytes > bucket_bytes:
        self.buckets.append(cur)
        cur, size = [], 0
      cur.append(i)
      size += nbytes
    if cur:
      self.buckets.append(cur)
    self.where = {}
    for bi, idxs in enumerate(self.buckets):
      off = 0
      for i in idxs:
        self.where[i] = (bi, off)
        off += self.leaves[i].numel()
    self.flat = [None] * len(self.buckets)
    self.remaining = [len(b) for b in self.buckets]
    self.handles = [None] * len(self.buckets)
    self.hooks = []
    self.calls = 0
    self.calls_in_backward = 0     # buckets whose all-reduce was issued from a gradient hook

  def _buffer(self, bi):
    if self.flat[bi] is None:
      n = sum(self.leaves[i].numel() for i in self.buckets[bi])
      self.flat[bi] = torch.zeros(n, dtype=torch.float32, device=self.leaves[self.buckets[bi][0]].device)
    return self.flat[bi]

  def _arrived(self, i):
    bi, off = self.where[i]
    g = self.leaves[i].grad
    self._buffer(bi)[off:off + g.numel()].copy_(g.reshape(-1))
    self.leaves[i].grad = None
    self.remaining[bi] -= 1
    if self.remaining[bi] == 0:
      before = self.calls
      self._launch(bi)
      self.calls_in_backward += self.calls - before

  def _launch(self, bi):
    if self.handles[bi] is None and self.exchanges:
      self.handles[bi] = dist.all_reduce(self._buffer(bi), op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True)
      self.calls += 1

  def attach(self):
    for i, t in enumerate(self.leaves):
      self.hooks.append(t.register_post_accumulate_grad_hook(lambda _t, i=i: self._arrived(i)))
    return self

  def finish(self) -> List[torch.Tensor]:
    for h in self.hooks:
      h.remove()
    self.hooks = []
    for bi in range(len(self.buckets)):     # leaves without a gradient: their slice stays zero
      if self.remaining[bi] > 0:
        self.remaining[bi] = 0
        self._launch(bi)
    out = [None] * len(self.leaves)
    for bi, idxs in enumerate(self.buckets):
      if self.handles[bi] is not None:
        self.handles[bi].wait()
      flat = self._buffer(bi)
      if self.world > 1:
        flat.mul_(1.0 / self.world)
      for i in idxs:
        _, off = self.where[i]
        out[i] = flat[off:off + self.leaves[i].numel()].reshape(self.leaves[i].shape)
    return out


def allreduce_mean_tree_(grads: Dict, group=None, bucket_bytes: int = 64 << 20) -> int:
  """``pmean`` of a nested gradient dict (the Flax param tree layout)."""
  return allreduce_mean_([t for _, t in flatten_tree(grads)], group, bucket_bytes)


def all_finite(tensors: Iterable[torch.Tensor], group=None) -> bool:
  """True iff every element on every rank is finite (trainer.py:260-266)."""
  return bool(all_finite_tensor(tensors, group).item() > 0)


def all_finite_tensor(tensors: Iterable[torch.Tensor], group=None) -> torch.Tensor:
  """``all_finite`` as a 0-d f32 tensor (1 = finite) left on the device: the caller reads it back
  together with its other step scalars in ONE transfer instead of one host sync per scalar."""
  tensors = [t for t in tensors if t is not None]
  if tensors:
    # max|.| per tensor in one multi-tensor launch; inf / nan propagate into the max.
    peak = torch.stack(torch._foreach_norm(tensors, float('inf')))
    ok = torch.isfinite(peak).all().to(torch.float32)
  else:
    ok = torch.ones((), dtype=torch.float32)
  if _exchanges(group):
    if dist.get_backend(group) == 'nccl' and not ok.is_cuda:     # (no tensors: RCCL needs a device buffer)
      ok = ok.to(torch.device('cuda', torch.cuda.current_device()))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
  return ok


def psum_metric_normalizer(metrics: Dict[str, Tuple[torch.Tensor, torch.Tensor]], group=None):
  """Sum (sum(metric), count) pairs over ranks (trainer.py:57-67); one collective."""
  if not metrics:
    return {}
  keys = sorted(metrics)
  flat = torch.stack([
      torch.stack([metrics[k][0].to(torch.float64).sum(), metrics[k][1].to(torch.float64).sum()])
      for k in keys
  ])
  if _exchanges(group):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
  return {k: (flat[i, 0], flat[i, 1]) for i, k in enumerate(keys)}


def reduce_batch_metrics(metrics: Dict[str, torch.Tensor], batch_mask: torch.Tensor, group=None):
  """Per-example metric vectors [B] -> global masked means (trainer.py:57-67,258)."""
  # metric_mask = batch_mask * isfinite(v): a non-finite value (e.g. on a padding example)
  # leaves both the sum and the count (trainer.py:57-67), instead of NaN * 0 = NaN.
  keys, vals = reduce_batch_metrics_tensor(metrics, batch_mask, group)
  return dict(zip(keys, vals.cpu().tolist()))


def reduce_batch_metrics_tensor(metrics: Dict[str, torch.Tensor], batch_mask: torch.Tensor, group=None):
  """``reduce_batch_metrics`` without the host round trips: (sorted keys, f64 tensor of the global
  masked means on the device) -- ONE stacked collective, no per-metric ``float()``."""
  if not metrics:
    return [], torch.zeros(0, dtype=torch.float64, device=batch_mask.device)
  # one [K, B] tensor instead of ~8 launches per metric (17 metrics at C3: ~150 of the step's ~520 torch
  # launches): the metrics are stacked per dtype (the key order that comes back says which row is which; it is
  # the same on every rank -- sorted keys, grouped by dtype in order of first appearance)
  groups = {}
  for k in sorted(metrics):
    groups.setdefault(metrics[k].dtype, []).append(k)
  keys = [k for ks in groups.values() for k in ks]
  B = batch_mask.numel()
  V = torch.cat([torch.stack([metrics[k].reshape(-1).expand(B) for k in ks]).to(torch.float64)
                 for ks in groups.values()])
  keep = batch_mask.to(torch.bool).reshape(1, -1) & torch.isfinite(V)
  flat = torch.stack([torch.where(keep, V, 0.0).sum(1), keep.to(torch.float64).sum(1)], 1)
  if _exchanges(group):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
  return keys, flat[:, 0] / torch.clamp(flat[:, 1], min=1.0)
