"""Single-kernel micro-benchmarks at the C2 shapes (HIP-event timed; used for tuning and
as the target of rocprofv3 --pmc passes).   python tools/kernel_bench.py pose_score [iters]
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snap_amd import ops  # noqa: E402


def timeit(fn, iters):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True)
  e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def pose_score(iters):
  B, Nq, X, Y, P = 8, 4652, 128, 128, 10001
  g = torch.Generator(device='cuda').manual_seed(0)
  sim = torch.rand((B, Nq, X, Y), device='cuda', generator=g)
  ang = (torch.rand((B, P), device='cuda', generator=g) - 0.5) * 2 * math.pi
  t = torch.rand((B, P, 2), device='cuda', generator=g) * 25.6
  poses = torch.cat([ang[..., None], t], -1).contiguous()
  q_xy = (torch.rand((B, Nq, 2), device='cuda', generator=g) - 0.5) * 16.0
  valid = torch.ones((B, Nq), dtype=torch.bool, device='cuda')
  ms = timeit(lambda: ops.pose_score(sim, poses, q_xy, valid, None, 0.2), iters)
  by = 4.0 * sim.numel()
  print(f'pose_score: {ms:.4f} ms  {by / ms / 1e6:.1f} GB/s  ({by / ms / 1e6 / 8000:.3f} of 8 TB/s)')


def pose_score_c4(iters):
  """Eval-path shapes (BASELINE configs[3] / SURVEY C4): 256x256 map, 20 001 sampled poses,
  then the 41^3 refinement lattice -- the band-tiled kernel (a 256 KB plane does not fit LDS)."""
  B, Nq, X, Y = 1, 4652, 256, 256
  g = torch.Generator(device='cuda').manual_seed(0)
  sim = torch.rand((B, Nq, X, Y), device='cuda', generator=g)
  q_xy = (torch.rand((B, Nq, 2), device='cuda', generator=g) - 0.5) * 16.0
  valid = torch.ones((B, Nq), dtype=torch.bool, device='cuda')
  for P in (20001, 68921):
    ang = (torch.rand((B, P), device='cuda', generator=g) - 0.5) * 2 * math.pi
    t = torch.rand((B, P, 2), device='cuda', generator=g) * 51.2
    poses = torch.cat([ang[..., None], t], -1).contiguous()
    ms = timeit(lambda: ops.pose_score(sim, poses, q_xy, valid, None, 0.2), iters)
    by = 4.0 * sim.numel()
    print(f'pose_score C4 P={P}: {ms:.3f} ms  plane bytes {by / 1e9:.2f} GB -> {by / ms / 1e6:.0f} GB/s; '
          f'gather volume 16 B x P x Nq = {16.0 * P * Nq / 1e9:.2f} GB -> {16.0 * P * Nq / ms / 1e6:.0f} GB/s')


def voting_c4(iters):
  """Exhaustive (x, y, theta) correlation at H = W = 256, R = 36, Dm = 32 (k14 + k15)."""
  from snap_amd.models import pose_exhaustive_voting as pev
  from snap_amd.models import types
  from snap_amd.utils import grids
  H = W = 256
  R, Dm = 36, 32
  g = torch.Generator(device='cuda').manual_seed(0)
  fq = torch.nn.functional.normalize(torch.randn((H, W, Dm), device='cuda', generator=g), dim=-1)
  fm = torch.nn.functional.normalize(torch.randn((H, W, Dm), device='cuda', generator=g), dim=-1)
  vq = torch.rand((H, W), device='cuda', generator=g) < 0.9
  vm = torch.ones((H, W), dtype=torch.bool, device='cuda')
  grid = grids.Grid2D((H, W), 0.2)
  pq = types.FeaturePlane(features=fq, valid=vq)
  pm = types.FeaturePlane(features=fm, valid=vm)
  ms = timeit(lambda: pev.exhaustive_pose_voting(pq, pm, R, grid), max(2, iters // 5))
  flop = 2.0 * R * (2 * H - 1) * (2 * W - 1) * H * W * Dm
  by = 4.0 * (R * H * W * Dm + H * W * Dm + R * (2 * H - 1) * (2 * W - 1))
  print(f'voting C4: {ms:.1f} ms  direct-form {flop / 1e12:.1f} TFLOP -> {flop / ms / 1e9:.1f} TFLOP/s '
        f'({flop / ms / 1e9 / 157.3:.3f} of the f32 MFMA peak); algorithmic bytes {by / 1e6:.0f} MB')


if __name__ == '__main__':
  which = sys.argv[1] if len(sys.argv) > 1 else 'pose_score'
  iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
  {'pose_score': pose_score, 'pose_score_c4': pose_score_c4, 'voting_c4': voting_c4}[which](iters)
