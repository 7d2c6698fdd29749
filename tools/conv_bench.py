"""Times representative conv / dense layers of the C2 step on both matrix-core engines.

  python tools/conv_bench.py [--math f32,bf16] [--iters 10]
"""
import argparse
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from snap_amd import ops  # noqa: E402

# name, (N, H, W, Cs), (KH, KW, Cin, Cout), stride, pad, prologue
LAYERS = [
    ('fusion_mlp_L0  M=2M K=257 N=256', (1, 1, 2_000_000, 260), (1, 1, 257, 256), 1, 0, ops.PRO_NONE),
    ('fusion_mlp_L1  M=2M K=256 N=128', (1, 1, 2_000_000, 256), (1, 1, 256, 128), 1, 0, ops.PRO_NONE),
    ('proj_mlp       M=590k K=128 N=160', (1, 1, 36 * 128 * 128, 128), (1, 1, 128, 160), 1, 0, ops.PRO_RELU),
    ('stage1 3x3 64  36x136x136', (36, 136, 136, 64), (3, 3, 64, 64), 1, 1, ops.PRO_GN_RELU),
    ('stage1 1x1 64->256', (36, 136, 136, 64), (1, 1, 64, 256), 1, 0, ops.PRO_GN_RELU),
    ('stage2 3x3 128 36x68x68', (36, 68, 68, 128), (3, 3, 128, 128), 1, 1, ops.PRO_GN_RELU),
    ('stage3 3x3 256 36x34x34', (36, 34, 34, 256), (3, 3, 256, 256), 1, 1, ops.PRO_GN_RELU),
    ('stage4 3x3 512 36x17x17', (36, 17, 17, 512), (3, 3, 512, 512), 1, 1, ops.PRO_GN_RELU),
    ('aerial s1 3x3 64 8x128x128', (8, 128, 128, 64), (3, 3, 64, 64), 1, 1, ops.PRO_GN_RELU),
    ('stage3 1x1 1024->256 40x34x34', (40, 34, 34, 1024), (1, 1, 1024, 256), 1, 0, ops.PRO_GN_RELU),
    ('stage3 1x1 256->1024 40x34x34', (40, 34, 34, 256), (1, 1, 256, 1024), 1, 0, ops.PRO_GN_RELU),
    ('stage2 1x1 128->512 40x68x68', (40, 68, 68, 128), (1, 1, 128, 512), 1, 0, ops.PRO_GN_RELU),
]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--math', default='f32,bf16x6,bf16x3,bf16')
  ap.add_argument('--iters', type=int, default=10)
  args = ap.parse_args()
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  for name, xs, ws, stride, pad, pro in LAYERS:
    x = torch.randn(xs, device=dev, generator=g)
    w = torch.randn(ws, device=dev, generator=g) / (ws[0] * ws[1] * ws[2]) ** 0.5
    gn = None
    if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
      gn = (torch.zeros(xs[0], ws[2], device=dev), torch.ones(xs[0], ws[2], device=dev),
            torch.zeros(ws[2], device=dev))
    kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), cin=ws[2], prologue=pro, gn=gn)
    Ho = (xs[1] + 2 * pad - ws[0]) // stride + 1
    Wo = (xs[2] + 2 * pad - ws[1]) // stride + 1
    flops = 2.0 * xs[0] * Ho * Wo * ws[0] * ws[1] * ws[2] * ws[3]
    line = f'{name:38s}'
    for math in args.math.split(','):
      # (timed: the conv launch only -- the weight image is prepared once, outside the loop)
      if math == 'bf16':
        w._snap_packed = {math: ops.pack_weights_bf16(w)}
      elif math in ops.SPLIT_PARTS:
        w._snap_packed = {math: ops.pack_weights_split_bf16(w, ops.SPLIT_PARTS[math])}
      for _ in range(2):
        ops.conv2d(x, w, math=math, **kw)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.iters):
        ops.conv2d(x, w, math=math, **kw)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.iters
      line += f'  {math}: {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF'
    print(line, flush=True)


if __name__ == '__main__':
  main()
