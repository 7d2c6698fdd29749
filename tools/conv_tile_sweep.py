"""Times the C2 layers of tools/conv_bench.LAYERS (+ the top-gap layers of the launch table) on the split
engine for every forced output tile: is the engine's automatic tile choice the fastest?

  python tools/conv_tile_sweep.py
"""
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from snap_amd import ops  # noqa: E402
import conv_bench  # noqa: E402

EXTRA = [
    ('stage3 1x1 256->1024 +res 40x34x34', (40, 34, 34, 256), (1, 1, 256, 1024), 1, 0, ops.PRO_GN_RELU),
    ('stage4 1x1 512->2048 40x17x17', (40, 17, 17, 512), (1, 1, 512, 2048), 1, 0, ops.PRO_GN_RELU),
    ('stage4 1x1 2048->512 40x17x17', (40, 17, 17, 2048), (1, 1, 2048, 512), 1, 0, ops.PRO_GN_RELU),
    ('stage2 1x1 512->128 40x68x68', (40, 68, 68, 512), (1, 1, 512, 128), 1, 0, ops.PRO_GN_RELU),
    ('aerial s3 3x3 256 8x34x34', (8, 34, 34, 256), (3, 3, 256, 256), 1, 1, ops.PRO_GN_RELU),
    ('aerial s3 1x1 1024->256 8x34x34', (8, 34, 34, 1024), (1, 1, 1024, 256), 1, 0, ops.PRO_GN_RELU),
    # the aerial encoder's other small-M layers
    ('aerial s3 1x1 256->1024 8x34x34', (8, 34, 34, 256), (1, 1, 256, 1024), 1, 0, ops.PRO_GN_RELU),
    ('aerial s3 1x1 1024->512 8x34x34', (8, 34, 34, 1024), (1, 1, 1024, 512), 1, 0, ops.PRO_GN_RELU),
    ('aerial s2 1x1 128->512 8x68x68', (8, 68, 68, 128), (1, 1, 128, 512), 1, 0, ops.PRO_GN_RELU),
    ('aerial s2 3x3 128 8x68x68', (8, 68, 68, 128), (3, 3, 128, 128), 1, 1, ops.PRO_GN_RELU),
    ('aerial s2 1x1 512->128 8x68x68', (8, 68, 68, 512), (1, 1, 512, 128), 1, 0, ops.PRO_GN_RELU),
    ('aerial s4 3x3 512 8x17x17', (8, 17, 17, 512), (3, 3, 512, 512), 1, 1, ops.PRO_GN_RELU),
    ('aerial s4 1x1 512->2048 8x17x17', (8, 17, 17, 512), (1, 1, 512, 2048), 1, 0, ops.PRO_GN_RELU),
    ('aerial s4 1x1 2048->512 8x17x17', (8, 17, 17, 2048), (1, 1, 2048, 512), 1, 0, ops.PRO_GN_RELU),
    ('aerial s1 1x1 256->64 8x136x136', (8, 136, 136, 256), (1, 1, 256, 64), 1, 0, ops.PRO_GN_RELU),
]


def main():
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  math = 'bf16x3'
  for name, xs, ws, stride, pad, pro in conv_bench.LAYERS[3:] + EXTRA:
    x = torch.randn(xs, device=dev, generator=g)
    w = torch.randn(ws, device=dev, generator=g) / (ws[0] * ws[1] * ws[2]) ** 0.5
    gn = (torch.zeros(xs[0], ws[2], device=dev), torch.ones(xs[0], ws[2], device=dev), torch.zeros(ws[2], device=dev))
    w._snap_packed = {math: ops.pack_weights_split_bf16(w, 2)}
    kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), cin=ws[2], prologue=pro, gn=gn)
    line = f'{name:40s}'
    for tile in (None, '128x128', '128x64', '64x128', '64x64'):
      for no_rs in (False, True):
        if tile is not None and not no_rs:
          continue
        ops.CONV_TILE, ops.CONV_NO_RS = tile, no_rs
        try:
          for _ in range(3):
            ops.conv2d(x, w, math=math, **kw)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(20):
            ops.conv2d(x, w, math=math, **kw)
          e1.record()
          torch.cuda.synchronize()
          line += f'  {tile or "auto"}{"/tiled" if no_rs else ""}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f}'
        except Exception as e:
          line += f'  {tile}: ERR'
        finally:
          ops.CONV_TILE, ops.CONV_NO_RS = None, False
    print(line, flush=True)


if __name__ == '__main__':
  main()
