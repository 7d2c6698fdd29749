"""Isolated timings of the bottleneck units' closing 1x1 convolutions (C2 shapes): the tiled split
engine (ops.CONV_NO_RS = True) against the row-stationary kernel (conv_rs.hip).  HIP events over
`reps` back-to-back launches on an otherwise idle GPU."""
import argparse
import json

import numpy as np
import torch

from snap_amd import ops

SHAPES = [  # N, H, W, Cin, Cout
    (40, 136, 136, 64, 256), (40, 68, 68, 128, 512), (40, 34, 34, 256, 1024),
    (8, 136, 136, 64, 256), (8, 68, 68, 128, 512), (8, 34, 34, 256, 1024),
]
REDUCTIONS = [  # the first stage's 256 -> 64 / 128 (run with --reductions --no-res)
    (40, 136, 136, 256, 64), (40, 136, 136, 256, 128), (8, 136, 136, 256, 64), (8, 136, 136, 256, 128),
]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--only', type=int, default=None)
  ap.add_argument('--stats', default='raw')
  ap.add_argument('--no-res', action='store_true')
  ap.add_argument('--nsplit', type=int, default=0)
  ap.add_argument('--reductions', action='store_true')
  args = ap.parse_args()
  ops.MATMUL_PRECISION = 'bf16x3'
  ops.CONV_RS_NSPLIT = args.nsplit
  ops.CONV_RS_FORCE = True
  dev = torch.device('cuda')
  out = []
  for i, (N, H, W, Cin, Cout) in enumerate(REDUCTIONS if args.reductions else SHAPES):
    if args.only is not None and i != args.only:
      continue
    g = torch.Generator().manual_seed(i)
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    w = (torch.randn((1, 1, Cin, Cout), generator=g) / np.sqrt(Cin)).to(dev)
    res = None if (args.no_res or args.reductions) else torch.randn((N, H, W, Cout), generator=g).to(dev)
    gamma = torch.ones(Cin, device=dev)
    beta = torch.zeros(Cin, device=dev)
    mu, sc = ops.group_norm_stats(x, gamma)
    row = {'shape': [N, H, W, Cin, Cout]}
    for name, no_rs, no_ws in (('tiled', True, False), ('rs', False, True), ('ws', False, False)):
      ops.CONV_NO_RS = no_rs
      ops.CONV_NO_WS = no_ws
      kw = dict(prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta), residual=res,
                emit_gn_stats=None if args.stats == 'none' else args.stats)
      for _ in range(3):
        y = ops.conv2d(x, w, **kw)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.reps):
        y = ops.conv2d(x, w, **kw)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.reps
      M = N * H * W
      byt = 4.0 * M * (Cin + Cout * (1 if res is None else 2))
      row[name] = {'ms': round(ms, 4), 'TF': round(2.0 * M * Cin * Cout / ms / 1e9, 1),
                   'TBs': round(byt / ms / 1e9, 2)}
    ops.CONV_NO_RS = False
    ops.CONV_NO_WS = False
    print(json.dumps(row), flush=True)
    out.append(row)


if __name__ == '__main__':
  main()
