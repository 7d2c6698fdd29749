"""A/B of one ResNet bottleneck unit (snap_amd/models/resnet.py::residual_unit) at the C2 shapes:
the pre-split path (gn_norm_split + conv_ps for the 3x3 / closing 1x1) against the fused-prologue
path (conv_split for every conv), with the per-launch breakdown of each.

  python tools/unit_bench.py [--iters 10] [--json out.json]
"""
import argparse
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from snap_amd import ops  # noqa: E402
from snap_amd.models import base, resnet  # noqa: E402

# name, N, H, W, Cin, nmid, stride
UNITS = [
    ('sv stage1 unit2  40x136x136 256/64', 40, 136, 136, 256, 64, 1),
    ('sv stage2 unit2  40x68x68  512/128', 40, 68, 68, 512, 128, 1),
    ('sv stage3 unit2  40x34x34 1024/256', 40, 34, 34, 1024, 256, 1),
    ('sv stage4 unit2  40x17x17 2048/512', 40, 17, 17, 2048, 512, 1),
    ('sv stage2 unit1  40x136x136 256/128 s2', 40, 136, 136, 256, 128, 2),
    ('aerial stage1 u2  8x136x136 256/64', 8, 136, 136, 256, 64, 1),
    ('aerial stage3 u2  8x34x34 1024/256', 8, 34, 34, 1024, 256, 1),
    ('aerial stage4 u2  8x17x17 2048/512', 8, 17, 17, 2048, 512, 1),
]

CONFIGS = [
    # label, USE_PRESPLIT, PS_RES_INIT, PS_TILE
    ('fused', False, False, 0),
    ('ps', True, True, 0),
    ('ps noinit', True, False, 0),
    ('ps t128', True, True, 1),
    ('ps t256', True, True, 2),
]


def run_unit(p, x, stride, nmid):
  ctx = base.ForwardContext()
  ks = resnet._kernels(p, [])
  ctx.standardize_all(ks, ops.weight_standardize_multi)
  ops.pack_weights_split_multi([ctx._lookup(k) for k in ks], ops.MATMUL_PRECISION)
  return ctx


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--json', default=None)
  ap.add_argument('--only', default=None)
  args = ap.parse_args()
  dev = 'cuda'
  ops.MATMUL_PRECISION = 'bf16x3'
  out = {}
  for name, N, H, W, Cin, nmid, stride in UNITS:
    if args.only and args.only not in name:
      continue
    gen = torch.Generator().manual_seed(0)
    p = resnet._init_unit(gen, 'cpu', Cin, nmid, stride)
    p = {k: {kk: vv.to(dev) for kk, vv in v.items()} for k, v in p.items()}
    x = torch.randn((N, H, W, Cin), device=dev)
    # x as a conv output with fused statistics (what the previous unit hands over)
    line = {}
    for label, use_ps, res_init, tile in CONFIGS:
      ops.USE_PRESPLIT, ops.PS_RES_INIT, ops.PS_TILE = use_ps, res_init, tile
      ctx = run_unit(p, x, stride, nmid)
      for _ in range(2):
        resnet.residual_unit(ctx, p, x, stride, nmid)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.iters):
        resnet.residual_unit(ctx, p, x, stride, nmid)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.iters
      prof = ops.KernelProfiler()
      ops.set_profiler(prof)
      resnet.residual_unit(ctx, p, x, stride, nmid)
      ops.set_profiler(None)
      torch.cuda.synchronize()
      parts = []
      for fam in prof.records:
        for tag, t, fl, by in prof.launches(fam):
          parts.append((fam if tag is None else tag, round(t, 4)))
      line[label] = dict(ms=round(ms, 4), launches=parts)
      del ctx
    ops.USE_PRESPLIT, ops.PS_RES_INIT, ops.PS_TILE = True, True, 0
    out[name] = line
    print(name)
    for label, rec in line.items():
      print(f'  {label:10s} {rec["ms"]:7.3f} ms   ' +
            '  '.join(f'{t.split("_K")[-1] if "_K" in t else t}:{v:.3f}' for t, v in rec['launches']))
    sys.stdout.flush()
  if args.json:
    with open(args.json, 'w') as f:
      json.dump(out, f, indent=1)


if __name__ == '__main__':
  main()
