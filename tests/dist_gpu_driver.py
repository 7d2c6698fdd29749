"""Launched by test_gpu_distributed.py under torch.distributed.run: TWO ranks on ONE GPU.

Checks, on device tensors and through the real HIP kernels, the exchange step of the training
loop (snap/trainer.py:225-234,260-277,57-67): the all-reduced gradients of a 2-rank
``train_step`` equal the single-process mean of the two ranks' gradients, the overlapped
reducer issues bucket all-reduces while the backward pass is still running, the finite flag and
the metric psum agree across ranks.  Backend: argv[1] ('nccl' == RCCL, or 'gloo')."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
  backend = sys.argv[1]
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  dist.init_process_group(backend)
  rank, world = dist.get_rank(), dist.get_world_size()
  assert world == 2
  probe = torch.full((1024,), float(rank + 1), device=dev)
  dist.all_reduce(probe)                       # RCCL refuses two ranks on one device here, if it does
  torch.cuda.synchronize()
  assert float(probe[0]) == 3.0
  own = [dist.new_group([r]) for r in range(world)][rank]      # a 1-rank group = "single process"

  import helpers
  from snap_amd import dist as sdist
  from snap_amd import models, trainer
  from snap_amd.data import synthetic
  cfg = helpers.tiny_localizer_config(num_pose_samples=32, retries=2)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  params = helpers.params_to_device(model.flax_model.init(0, device='cpu')['params'], dev)
  batches = [helpers.batch_to_device(synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=40 + r), dev)
             for r in range(world)]
  for b in batches:
    b['batch_mask'] = torch.ones(2, dtype=torch.bool, device=dev)
  state = trainer.TrainState.create(params, rng=3)
  leaves = [t for _, t in trainer.flatten_params(state.params)]
  for t in leaves:
    t.requires_grad_(True)
  rngs = [1000 + r for r in range(world)]

  def grads_of(r, group, overlap):
    out = trainer._forward_backward(state, batches[r], model, leaves, rngs[r], group, False, overlap)
    for t in leaves:
      t.grad = None
    return [g.detach().clone() for g in out[0]], float(out[1].detach())

  # distributed: each rank its own batch, gradients all-reduced (overlapped and plain reducer)
  g_ovl, loss_r = grads_of(rank, None, True)
  g_plain, _ = grads_of(rank, None, False)
  # single process: both batches locally (1-rank group: no collective), then the mean
  singles = [grads_of(r, own, False) for r in range(world)]
  mean = [sum(gs) / world for gs in zip(*[s[0] for s in singles])]
  worst = 0.0
  for a, b, c in zip(g_ovl, g_plain, mean):
    scale = float(c.abs().max()) + 1e-12
    worst = max(worst, float((a - c).abs().max()) / scale, float((b - c).abs().max()) / scale)
  # (the lift's scatter-add backward uses float atomics: run-to-run noise ~1e-6)
  assert worst < 2e-4, worst
  assert abs(loss_r - singles[rank][1]) <= 1e-5 * abs(loss_r)

  # overlap: with small buckets most all-reduces are issued from gradient hooks, i.e. while the
  # autograd engine is still running the rest of the backward pass
  with torch.enable_grad():
    pred = model.flax_model.apply({'params': state.params}, batches[rank], train=True,
                                  rngs={'sampling': rngs[rank]}, mutable=False)
    losses, _ = model.loss_metrics_function(pred, batches[rank], state.params)
    red = sdist.OverlappedGradReducer(leaves, None, bucket_bytes=64 << 10).attach()
    losses['total'].mean().backward()
    avg = red.finish()
  for t in leaves:
    t.grad = None
  assert len(red.buckets) >= 4 and red.calls == len(red.buckets)
  assert red.calls_in_backward >= len(red.buckets) - 1, (red.calls_in_backward, len(red.buckets))
  assert all(torch.isfinite(a).all() for a in avg)

  # finite flag (MIN all-reduce) and metric psum on device tensors
  assert sdist.all_finite(g_ovl)
  bad = [torch.ones(3, device=dev), torch.tensor([float('nan') if rank == 1 else 1.0], device=dev)]
  assert not sdist.all_finite(bad)
  metrics = {'err': torch.tensor([1.0, 3.0], device=dev) * (rank + 1)}
  red_m = sdist.reduce_batch_metrics(metrics, torch.tensor([True, rank == 0], device=dev))
  assert abs(red_m['err'] - 2.0) < 1e-9

  # a whole train_step: same parameters on both ranks afterwards
  for t in leaves:
    t.requires_grad_(False)
  st2, _, logs = trainer.train_step(state, batches[rank], model=model, lr_fn=lambda s: 1e-3)
  digest = torch.stack([t.double().sum() for _, t in trainer.flatten_params(st2.params)]).sum().reshape(1)
  both = [torch.zeros_like(digest) for _ in range(world)]
  dist.all_gather(both, digest)
  assert float((both[0] - both[1]).abs()) <= 1e-9 * float(both[0].abs() + 1), both
  assert logs['is_finite']
  if rank == 0:
    print(f'DIST_GPU_OK backend={backend} worst_rel_grad_err={worst:.2e} '
          f'buckets={len(red.buckets)} issued_during_backward={red.calls_in_backward}')
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
