"""Launched by test_gpu_distributed.py: ONE rank, backend 'nccl' (== RCCL on ROCm), on the one GPU
of the test box, with ``snap_amd.dist.FORCE_COLLECTIVES`` so that every exchange of the training
step (snap/trainer.py:225-234 pmean of the gradients, :260-277 finite flag, :57-67 metric psum)
really goes through RCCL instead of short-circuiting at world size 1.  What this proves on a
single-GPU box: librccl loads next to libsnap_hip.so in one process, the bucket buffers are device
tensors RCCL accepts, all-reduces issued from autograd hook context order correctly with the
compute stream (the reduced gradients equal the collective-free ones), and a whole ``train_step``
runs on the 'nccl' backend."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  dist.init_process_group('nccl', rank=0, world_size=1)
  assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
  probe = torch.full((1 << 20,), 2.5, device=dev)
  dist.all_reduce(probe)
  torch.cuda.synchronize()
  assert float(probe[0]) == 2.5 and float(probe[-1]) == 2.5

  import helpers
  from snap_amd import dist as sdist
  from snap_amd import models, trainer
  from snap_amd.data import synthetic
  cfg = helpers.tiny_localizer_config(num_pose_samples=32, retries=2)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  params = helpers.params_to_device(model.flax_model.init(0, device='cpu')['params'], dev)
  batch = helpers.batch_to_device(synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=40), dev)
  batch['batch_mask'] = torch.ones(2, dtype=torch.bool, device=dev)
  state = trainer.TrainState.create(params, rng=3)
  leaves = [t for _, t in trainer.flatten_params(state.params)]
  for t in leaves:
    t.requires_grad_(True)

  def grads_of(overlap):
    out = trainer._forward_backward(state, batch, model, leaves, 1000, None, False, overlap)
    for t in leaves:
      t.grad = None
    return [g.detach().clone() for g in out[0]]

  # world size 1 short-circuits without the switch: the collective-free gradients
  assert not sdist._exchanges(None)
  g_local = grads_of(False)
  sdist.FORCE_COLLECTIVES = True
  assert sdist._exchanges(None)
  g_plain = grads_of(False)       # bucketed all-reduce after the backward pass
  g_ovl = grads_of(True)          # all-reduces issued from gradient hooks during the backward pass
  worst = 0.0
  for a, b, c in zip(g_ovl, g_plain, g_local):
    scale = float(c.abs().max()) + 1e-12
    worst = max(worst, float((a - c).abs().max()) / scale, float((b - c).abs().max()) / scale)
  assert worst < 2e-4, worst

  # small buckets: most all-reduces go out from hook context while the backward pass still runs
  with torch.enable_grad():
    pred = model.flax_model.apply({'params': state.params}, batch, train=True,
                                  rngs={'sampling': 1000}, mutable=False)
    losses, _ = model.loss_metrics_function(pred, batch, state.params)
    red = sdist.OverlappedGradReducer(leaves, None, bucket_bytes=64 << 10).attach()
    losses['total'].mean().backward()
    avg = red.finish()
  for t in leaves:
    t.grad = None
  assert all(f.is_cuda for f in red.flat if f is not None)
  assert len(red.buckets) >= 4 and red.calls == len(red.buckets)
  assert red.calls_in_backward >= len(red.buckets) - 1, (red.calls_in_backward, len(red.buckets))
  worst_small = 0.0
  for a, c in zip(avg, g_local):
    worst_small = max(worst_small, float((a - c).abs().max()) / (float(c.abs().max()) + 1e-12))
  assert worst_small < 2e-4, worst_small
  n_plain = sdist.allreduce_mean_([g.clone() for g in g_local], None, bucket_bytes=64 << 10)
  assert n_plain >= 4

  assert sdist.all_finite(g_ovl)
  assert sdist.all_finite([])
  assert not sdist.all_finite([torch.ones(3, device=dev), torch.tensor([float('nan')], device=dev)])
  red_m = sdist.reduce_batch_metrics({'err': torch.tensor([1.0, 3.0], device=dev)},
                                     torch.tensor([True, True], device=dev))
  assert abs(red_m['err'] - 2.0) < 1e-9

  for t in leaves:
    t.requires_grad_(False)
  before = torch.stack([t.double().sum() for t in leaves]).sum()
  st2, _, logs = trainer.train_step(state, batch, model=model, lr_fn=lambda s: 1e-3)
  after = torch.stack([t.double().sum() for _, t in trainer.flatten_params(st2.params)]).sum()
  assert logs['is_finite'] and float((after - before).abs()) > 0

  maps = open('/proc/self/maps').read()
  rccl = sorted({l.split()[-1] for l in maps.splitlines() if 'librccl' in l or 'libnccl' in l})
  snap = sorted({l.split()[-1] for l in maps.splitlines() if 'libsnap_hip' in l})
  assert rccl and snap, (rccl, snap)
  ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
  print(f'RCCL1_OK backend={dist.get_backend()} rccl={ver} lib={os.path.basename(rccl[0])} '
        f'next_to={os.path.basename(snap[0])} buckets={len(red.buckets)} '
        f'issued_during_backward={red.calls_in_backward} worst_rel_grad_err={max(worst, worst_small):.2e}')
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
