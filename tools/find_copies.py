"""Where do the torch-side copies / elementwise launches of one inference step come from?
(torch.profiler with python stacks; prints the aten ops that launch device kernels or memcpys,
grouped by the innermost snap_amd source line.)   python tools/find_copies.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from snap_amd import ops  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  ops.MATMUL_PRECISION = 'bf16x3'
  loc, cfg, meta, variables, batch = bench.build('c2', dev, 0, materialize_volume=False)
  for i in range(2):
    loc.apply(variables, batch, train=False, rngs={'sampling': i})
  torch.cuda.synchronize()
  from torch.profiler import ProfilerActivity, profile
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    loc.apply(variables, batch, train=False, rngs={'sampling': 7})
    torch.cuda.synchronize()
  agg = collections.defaultdict(lambda: [0, 0.0, set()])
  for ev in prof.events():
    if not ev.name.startswith('aten::'):
      continue
    dt = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
      continue                     # top-level aten ops only
    if dt <= 0:
      continue
    where = '?'
    for fr in (ev.stack or []):
      if 'snap_amd' in fr or 'bench.py' in fr:
        where = fr.split('/root/repo/')[-1] if '/root/repo/' in fr else fr
        break
    a = agg[(where, ev.name)]
    a[0] += 1
    a[1] += dt
    a[2].add(str(ev.input_shapes)[:80])
  tot = 0.0
  for (where, name), (n, t, shp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print(f'{t:8.1f} us  x{n:<3d} {name:28s} {where[:90]}  {sorted(shp)[:2]}')
  print('total', round(tot, 1), 'us of device time in top-level aten ops')


if __name__ == '__main__':
  main()
