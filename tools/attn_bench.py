"""Tool (not product): the attention kernel on C5's shape (20 images x 1024 tokens x 12 heads x 64)."""
import sys, torch
sys.path.insert(0, '.')
from snap_amd import ops
torch.manual_seed(0)
B, N, H = 20, 1024, 12
qkv = torch.randn(B, N, 3, H, 64, device='cuda')
qh = qkv.to(torch.bfloat16)
def timeit(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
fl = 4.0 * B * H * N * N * 64 / 1e9
for name, fn in (('f32 qkv', lambda: ops.attention(qkv)), ('bf16 qkv', lambda: ops.attention(qh, out_half=True))):
  t = timeit(fn)
  print(f'{name}: {t*1e3:.1f} us  {fl/t:.0f} TFLOP/s')
