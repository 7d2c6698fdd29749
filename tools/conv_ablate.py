"""Tuning aid: times the conv engine on a few shapes under the ablation bits of an ALT build
(-DSNAP_CONV_SPLIT_ABLATE=1 reads the environment variable SNAP_ALT_ABLATE; timing only, wrong results)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import ops

def bench(fn, n=5):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n

shapes = [('mlp0', 1, 1, 4000000, 256, 1, 256), ('c3x3', 40, 34, 34, 256, 3, 256), ('c1x1', 40, 68, 68, 512, 1, 128)]
for name, N, H, W, Cin, k, Cout in shapes:
  x = torch.randn(N, H, W, Cin, device='cuda')
  w = torch.randn(k, k, Cin, Cout, device='cuda') * 0.05
  mu = torch.zeros(N, Cin, device='cuda'); sc = torch.ones(N, Cin, device='cuda'); beta = torch.zeros(Cin, device='cuda')
  for pro in (0, 2):
    for ab in (0, 1, 2, 8, 9, 3):
      os.environ['SNAP_ALT_ABLATE'] = str(ab)     # read by the ALT build only (-DSNAP_CONV_SPLIT_ABLATE=1)
      pad = ((k // 2, k // 2), (k // 2, k // 2))
      kw = dict(padding=pad, prologue=pro, gn=(mu, sc, beta) if pro else None)
      ms = bench(lambda: ops.conv2d(x, w, **kw))
      fl = 2.0 * N * H * W * k * k * Cin * Cout
      print(f'{name} pro={pro} ablate={ab}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF', flush=True)
os.environ['SNAP_ALT_ABLATE'] = '0'
