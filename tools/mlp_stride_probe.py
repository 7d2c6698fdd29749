"""Does the row stride of `pooled` (257 live channels) matter to the masked MLP's first-layer GEMMs?  Times the
forward Dense (row list in, half out) and the kernel gradient for strides 260 / 272 / 288 floats.

  python tools/mlp_stride_probe.py [--rows 3932160] [--density 0.5]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snap_amd import ops, ops_bwd


def timeit(fn, n=5):
  fn(); torch.cuda.synchronize()
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  ev[0].record()
  for _ in range(n):
    fn()
  ev[1].record(); torch.cuda.synchronize()
  return round(ev[0].elapsed_time(ev[1]) / n, 4)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rows', type=int, default=3932160)
  ap.add_argument('--density', type=float, default=0.5)
  a = ap.parse_args()
  dev = 'cuda'
  M = a.rows
  g = torch.Generator(device=dev).manual_seed(1)
  mask = torch.rand(M, device=dev, generator=g) < a.density
  index, count = ops.compact_rows(mask)
  W0 = torch.randn(257, 256, device=dev, generator=g) / 16
  b0 = torch.randn(256, device=dev, generator=g)
  g1 = torch.randn(M, 256, device=dev, generator=g).to(torch.bfloat16)
  out = {'rows': M, 'observed': int(count.item())}
  x = torch.randn(M, 260, device=dev, generator=g)
  gfull = torch.randn(M, 128, device=dev, generator=g)
  W1t = torch.randn(128, 256, device=dev, generator=g) / 11
  for tile in (None, '128x128', '128x64', '64x128', '64x64'):
    ops.CONV_TILE = tile
    fwd = lambda: ops.dense(x, W0, b0, cin=257, relu=True, rows_in=index, row_count=count, out_half=True, math='bf16')
    dg1 = lambda: ops.dense(gfull, W1t, None, cin=128, rows_in=index, row_count=count, out_half=True, math='bf16')
    out[f'tile_{tile}'] = {'fwd_L0': timeit(fwd), 'dgrad_L1': timeit(dg1)}
  ops.CONV_TILE = None
  del x, gfull
  for cs in (260, 288):
    x = torch.randn(M, cs, device=dev, generator=g)
    x[:, 257:] = 0
    fwd = lambda: ops.dense(x, W0, b0, cin=257, relu=True, rows_in=index, row_count=count, out_half=True, math='bf16')
    wg = lambda: ops_bwd.conv2d_wgrad(x.reshape(1, 1, M, cs), g1.reshape(1, 1, M, 256), (1, 1, 256, 256), rows_z=index,
                                      row_count=count, math='bf16')
    Wt = W0[:256].t().contiguous()
    gi = torch.empty(M, cs, device=dev)
    dg = lambda: ops.dense(g1, Wt, None, cin=256, rows_out=index, row_count=count, out=gi, out_stride=cs, math='bf16')
    out[f'stride{cs}'] = {'fwd_L0': timeit(fwd), 'dW0': timeit(wg), 'dx': timeit(dg)}
    del x, gi
  print(json.dumps(out))


if __name__ == '__main__':
  main()
