"""Launched by test_distributed.py (gloo, 2 ranks): checks snap_amd.dist."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snap_amd import dist as sdist  # noqa: E402


def main():
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()
  g = torch.Generator().manual_seed(0)
  shapes = [(3, 3, 8, 16), (16,), (257, 64), (1,), (5, 7)]
  base = [torch.randn(s, generator=g) for s in shapes]
  grads = {'b': {'k': base[0] * (rank + 1), 'z': base[1] * (rank + 1)},
           'a': base[2] * (rank + 1), 't': base[3] * (rank + 1), 'h': (base[4] * (rank + 1)).half()}
  calls = sdist.allreduce_mean_tree_(grads, bucket_bytes=4096)
  scale = sum(r + 1 for r in range(world)) / world
  flat = dict(sdist.flatten_tree(grads))
  assert torch.allclose(flat['b/k'], base[0] * scale, atol=1e-6)
  assert torch.allclose(flat['a'], base[2] * scale, atol=1e-6)
  assert flat['h'].dtype == torch.float16 and torch.allclose(flat['h'].float(), base[4] * scale, atol=1e-2)
  assert calls >= 2, calls                       # 4 KiB buckets -> several messages
  assert sdist.all_finite(flat.values())
  bad = [torch.ones(3), torch.tensor([float('inf') if rank == 1 else 1.0])]
  assert not sdist.all_finite(bad)               # a non-finite value on ANY rank
  metrics = {'err': torch.tensor([1.0, 3.0]) * (rank + 1), 'hit': torch.tensor([1.0, 0.0])}
  mask = torch.tensor([True, rank == 0])
  red = sdist.reduce_batch_metrics(metrics, mask)
  # rank 0: err 1, 3 (both valid); rank r > 0: err r + 1 (its second entry masked); world 2 -> (1+3+2)/3
  want_err = (4.0 + sum(r + 1 for r in range(1, world))) / (world + 1)
  assert abs(red['err'] - want_err) < 1e-9 and abs(red['hit'] - world / (world + 1.0)) < 1e-9
  # overlapped reducer: hooks fire during backward(), buckets are reduced asynchronously
  leaves = [(b.clone() * 0 + 1.0).requires_grad_(True) for b in base[:4]]
  unused = torch.ones(3, requires_grad=True)                 # never reaches the loss
  red = sdist.OverlappedGradReducer(leaves + [unused], bucket_bytes=4096).attach()
  loss = sum(((rank + 1) * b * l).sum() for b, l in zip(base[:4], leaves))
  loss.backward()
  avg = red.finish()
  for b, gavg in zip(base[:4], avg):
    assert torch.allclose(gavg, b * scale, atol=1e-6)
  assert float(avg[4].abs().max()) == 0.0 and all(l.grad is None for l in leaves)
  assert red.calls == len(red.buckets) >= 2, (red.calls, len(red.buckets))
  # every rank derives the SAME bucket plan (a rank with a different plan would pair its buckets
  # with other ranks' buckets of another size: a hang or silent corruption on RCCL)
  import hashlib
  h = int(hashlib.sha256(repr(red.buckets).encode()).hexdigest()[:12], 16)
  plans = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
  dist.all_gather(plans, torch.tensor([h], dtype=torch.int64))
  assert len({int(p) for p in plans}) == 1, plans
  if rank == 0:
    print('BUCKET_PLAN_EQUAL', world)
  if rank == 0:
    print('DIST_SYNC_OK', calls)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
