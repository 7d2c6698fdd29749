"""Which torch (aten) ops launch GPU work inside one C2 inference step, and from which source line:
the host-glue launches around the HIP kernels (~75 elementwise / cat / index / copy
launches per step).  Prints per (source line, op): calls per step and device time.

  python tools/torch_ops_in_step.py [--workload c2]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from snap_amd import ops  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='c2')
  ap.add_argument('--mode', default='infer', choices=['infer', 'train'])
  ap.add_argument('--precision', default='bf16')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  train = args.mode == 'train'
  loc, cfg, meta, variables, batch = bench.build(args.workload, dev, 0, materialize_volume=train)
  if train:
    from snap_amd import models, trainer
    from snap_amd.configs import train_localization
    model = models.get_model('bev_localizer')(cfg, meta)
    state = trainer.TrainState.create(variables['params'], rng=0)

    def step(i):
      trainer.train_step(state, batch, model=model, lr_fn=lambda s: 1e-4, precision=args.precision)
  else:
    loc.engine = 'bf16x3'

    def step(i):
      loc.apply(variables, batch, train=False, rngs={'sampling': i})
  for i in range(3):
    step(i)
  torch.cuda.synchronize()
  from torch.profiler import ProfilerActivity, profile
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(7)
    torch.cuda.synchronize()
  agg = collections.defaultdict(lambda: [0, 0.0])
  for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or not ev.kernels:
      continue
    where = '?'
    for fr in (ev.stack or []):
      if 'snap_amd' in fr and 'site-packages' not in fr:
        where = fr.split('snap_amd/')[-1]
        break
    if where == '?':           # no Python stack on this build: the input shapes identify the call site
      where = str([tuple(sh) for sh in (ev.input_shapes or []) if sh])[:90]
    k = (where, ev.name)
    agg[k][0] += 1
    agg[k][1] += sum(kk.duration for kk in ev.kernels)
  tot_n = tot_t = 0
  for (where, name), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:4d} {t:9.1f} us  {name:28s} {where}')
    tot_n += n; tot_t += t
  print(f'total: {tot_n} kernel-launching aten ops, {tot_t / 1e3:.3f} ms of device time')


if __name__ == '__main__':
  main()
