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


def lift_pool(iters):
  raise SystemExit('not wired yet')


if __name__ == '__main__':
  which = sys.argv[1] if len(sys.argv) > 1 else 'pose_score'
  iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
  {'pose_score': pose_score, 'lift_pool': lift_pool}[which](iters)
