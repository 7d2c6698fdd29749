"""Times the masked MLP's kernel gradients at C3 size (flat GEMMs over a row list) on both plans.

  python tools/wgrad_bench.py [--rows 3932160] [--density 0.4]
"""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snap_amd import _lib, ops, ops_bwd


def timeit(fn, n=5):
  fn(); torch.cuda.synchronize()
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  ev[0].record()
  for _ in range(n):
    fn()
  ev[1].record(); torch.cuda.synchronize()
  return ev[0].elapsed_time(ev[1]) / n


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rows', type=int, default=3932160)
  ap.add_argument('--density', type=float, default=0.4)
  ap.add_argument('--math', default='bf16')
  a = ap.parse_args()
  dev = 'cuda'
  lib = _lib.load()
  M = a.rows
  g = torch.Generator(device=dev).manual_seed(1)
  mask = torch.rand(M, device=dev, generator=g) < a.density
  index, count = ops.compact_rows(mask)
  n = int(count.item())
  hd = ops_bwd.HALF_DTYPE[a.math]
  x = torch.randn(M, 260, device=dev, generator=g)
  gout = torch.randn(M, 128, device=dev, generator=g)
  h0 = torch.randn(M, 256, device=dev, generator=g).to(hd)       # compact rows (first n used)
  g1 = torch.randn(M, 256, device=dev, generator=g).to(hd)
  gbig = torch.randn(M, 256, device=dev, generator=g)
  out = {'rows': M, 'observed': n, 'math': a.math}
  cases = {
      'dW0 x(f32,list)[256] x g1(half)[256]': lambda plans=3: ops_bwd.conv2d_wgrad(
          x.reshape(1, 1, M, 260), g1.reshape(1, 1, M, 256), (1, 1, 256, 256), rows_z=index, row_count=count, math=a.math, plans=plans),
      'dW1 h0(half)[256] x g(f32,list)[128]': lambda plans=3: ops_bwd.conv2d_wgrad(
          h0.reshape(1, 1, M, 256), gout.reshape(1, 1, M, 128), (1, 1, 256, 128), rows_dy=index, row_count=count, math=a.math, plans=plans),
      'dW f32 x f32 lists [256]x[256]': lambda plans=3: ops_bwd.conv2d_wgrad(
          x.reshape(1, 1, M, 260), gbig.reshape(1, 1, M, 256), (1, 1, 256, 256), rows_z=index, rows_dy=index, row_count=count, math=a.math, plans=plans),
  }
  for name, fn in cases.items():
    r = {}
    for wide in (0, 1):
      r['wide' if wide else 'tiles128'] = round(timeit(lambda: fn(2 | wide)), 4)
    out[name] = r
  print(json.dumps(out))


if __name__ == '__main__':
  main()
