"""The K >= 256 1 x 1 layers of the C2 encoders on the split engine: raw-row LDS-ring body (conv_raw.hip)
vs the tiled plain body (ops.CONV_RAW_RING off), isolated launches with a GroupNorm prologue.

  python tools/conv_raw_bench.py
"""
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from snap_amd import ops  # noqa: E402

LAYERS = [
    ('stage3 1x1 1024->256 40x34x34', (40, 34, 34, 1024), 256),
    ('stage3 1x1 256->1024 40x34x34', (40, 34, 34, 256), 1024),
    ('stage2 1x1 512->128 40x68x68', (40, 68, 68, 512), 128),
    ('stage2 1x1 256->128 40x136x136', (40, 136, 136, 256), 128),
    ('stage2 1x1 256->64 40x136x136', (40, 136, 136, 256), 64),
    ('stage4 1x1 2048->512 40x17x17', (40, 17, 17, 2048), 512),
    ('stage4 1x1 512->2048 40x17x17', (40, 17, 17, 512), 2048),
    ('stage3 1x1 1024->512 40x34x34', (40, 34, 34, 1024), 512),
    ('aerial 1x1 1024->256 8x34x34', (8, 34, 34, 1024), 256),
    ('aerial 1x1 256->1024 8x34x34', (8, 34, 34, 256), 1024),
]


def main():
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  for name, xs, cout in LAYERS:
    cin = xs[-1]
    x = torch.randn(xs, device=dev, generator=g)
    w = torch.randn((1, 1, cin, cout), device=dev, generator=g) / cin ** 0.5
    gn = (torch.zeros(xs[0], cin, device=dev), torch.ones(xs[0], cin, device=dev), torch.zeros(cin, device=dev))
    w._snap_packed = {'bf16x3': ops.pack_weights_split_bf16(w, 2)}
    line = f'{name:34s}'
    for label, no_raw, no_rs in (('raw', False, True), ('tiled', True, True), ('auto', False, False)):
      ops.CONV_RAW_RING, ops.CONV_NO_RS = not no_raw, no_rs
      try:
        for _ in range(3):
          ops.conv2d(x, w, math='bf16x3', prologue=ops.PRO_GN_RELU, gn=gn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
          ops.conv2d(x, w, math='bf16x3', prologue=ops.PRO_GN_RELU, gn=gn)
        e1.record()
        torch.cuda.synchronize()
        line += f'  {label}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us'
      finally:
        ops.CONV_RAW_RING, ops.CONV_NO_RS = False, False
    print(line, flush=True)


if __name__ == '__main__':
  main()
