"""Isolated timing of the fused fusion-MLP / vertical max-pool kernel at the C2 map size (8 scenes x
128 x 128 columns x 60 levels, 87 % of the voxels observed, 72 % of those by one view): the 128-row
kernel with the two-stage GEMM0 loop (ops.MLP_POOL_NO_RING), with the three-stage ring (the default) and the opt-in
256-row kernel (ops.MLP_POOL_WIDE), on the same pre-split rows."""
import argparse
import json

import torch

from snap_amd import ops


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--cols', type=int, default=8 * 128 * 128)
  args = ap.parse_args()
  dev = torch.device('cuda')
  Z, cin, H, D = 60, 257, 256, 128
  ks = (cin + 15) // 16
  M = args.cols * Z
  g = torch.Generator(device='cuda').manual_seed(1)
  # pre-split rows: any bf16 pairs do (timing only); keep them finite
  xs = (torch.randn((M, ks * 16), generator=g, device=dev) * 0.5).bfloat16().float().view(M, ks * 16)
  u = torch.rand(M, generator=g, device=dev)
  cls = ((u < 0.87).to(torch.uint8) * (1 + (torch.rand(M, generator=g, device=dev) > 0.72).to(torch.uint8)))
  w0 = torch.randn((cin, H), generator=g, device=dev) / cin ** 0.5
  b0 = torch.zeros(H, device=dev)
  w1 = torch.randn((H, D), generator=g, device=dev) / H ** 0.5
  b1 = torch.zeros(D, device=dev)
  out = {}
  planes = {}
  for name, narrow, ring3 in (('rows128', True, False), ('rows128_ring3', True, True), ('rows256', False, False),
                              ('rows128_b', True, False), ('rows128_ring3_b', True, True)):
    ops.MLP_POOL_WIDE = not narrow
    ops.MLP_POOL_NO_RING = not ring3
    kw = dict(cin=cin, Z=Z, x_split=True, zero_slabs=(8, 8))
    for _ in range(2):
      p, v = ops.mlp2_pool_max(xs, cls, w0, b0, w1, b1, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
      p, v = ops.mlp2_pool_max(xs, cls, w0, b0, w1, b1, **kw)
    e1.record()
    torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / args.reps, 4)
    planes[name] = p
  ops.MLP_POOL_WIDE = False
  ops.MLP_POOL_NO_RING = False
  out['equal_ring3'] = bool(torch.equal(planes['rows128'], planes['rows128_ring3']))
  out['equal'] = bool(torch.equal(planes['rows128'], planes['rows256']))
  print(json.dumps(out))


if __name__ == '__main__':
  main()
