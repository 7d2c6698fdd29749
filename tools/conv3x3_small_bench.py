"""3x3 layers with few rows (aerial encoder, StreetView stage 4): the engine's own tile choice
against forced 128-row tiles (halo body + split-K by channel tiles)."""
import json

import numpy as np
import torch

from snap_amd import ops

SHAPES = [(8, 34, 34, 256, 256), (8, 17, 17, 512, 512), (40, 17, 17, 512, 512), (8, 68, 68, 128, 128),
          (40, 34, 34, 256, 256)]


def main():
  ops.MATMUL_PRECISION = 'bf16x3'
  dev = torch.device('cuda')
  for i, (N, H, W, Cin, Cout) in enumerate(SHAPES):
    g = torch.Generator().manual_seed(i)
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / np.sqrt(9 * Cin)).to(dev)
    mu, sc = ops.group_norm_stats(x, torch.ones(Cin, device=dev))
    beta = torch.zeros(Cin, device=dev)
    row = {'shape': [N, H, W, Cin, Cout]}
    ref = None
    for name, tile, halo in (('auto', None, False), ('128x128', '128x128', False), ('128x64', '128x64', False),
                             ('128x128 im2col', '128x128', True)):
      ops.CONV_TILE = tile
      ops.CONV_NO_HALO = halo
      kw = dict(padding=((1, 1), (1, 1)), prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta))
      for _ in range(3):
        y = ops.conv2d(x, w, **kw)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20):
        y = ops.conv2d(x, w, **kw)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / 20
      if ref is None:
        ref = y
      row[name] = {'ms': round(ms, 4), 'TF': round(2.0 * N * H * W * 9 * Cin * Cout / ms / 1e9, 1),
                   'maxdiff': float((y - ref).abs().max())}
    ops.CONV_TILE = None
    ops.CONV_NO_HALO = False
    print(json.dumps(row), flush=True)


if __name__ == '__main__':
  main()
