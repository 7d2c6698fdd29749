"""Big-GEMM timing of the pre-split engine (conv_ps.hip) next to conv_split, per row tile.

  [SNAP_HIP_LIB=snap_amd/lib/alt_<x>/libsnap_hip.so] python tools/ps_gemm_bench.py [--shapes ...]
"""
import argparse
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from snap_amd import ops  # noqa: E402

SHAPES = [
    # name, M, K (taps x Cin), N, k, H, W
    ('gemm 65536x1024x1024', 65536, 1024, 1024),
    ('gemm 65536x256x1024', 65536, 256, 1024),
    ('gemm 131072x4096x256', 131072, 4096, 256),
    ('gemm 16384x32768x576', 16384, 32768, 576),
]


def timeit(fn, iters):
  for _ in range(2):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--split', action='store_true', help='also time conv_split on the f32 input')
  ap.add_argument('--only', default=None)
  args = ap.parse_args()
  ops.MATMUL_PRECISION = 'bf16x3'
  dev = 'cuda'
  for name, M, K, N in SHAPES:
    if args.only and args.only not in name:
      continue
    x = torch.randn((1, 1, M, K), device=dev)
    w = torch.randn((1, 1, K, N), device=dev) / K ** 0.5
    w._snap_packed = {'bf16x3': ops.pack_weights_split_bf16(w, 2)}
    xs = ops.presplit(x)
    flops = 2.0 * M * K * N
    line = f'{name:28s}'
    for tile in (1, 2) + ((3,) if N % 192 == 0 else ()):
      ms = timeit(lambda: ops.conv2d(xs, w, ps_tile=tile), args.iters)
      line += f'  ps t{tile}: {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF'
    if args.split:
      ms = timeit(lambda: ops.conv2d(x, w), args.iters)
      line += f'  split: {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF'
    print(line, flush=True)


if __name__ == '__main__':
  main()
