"""Tool (not product): the one-part pre-split engine vs the training-precision engine on the ViT-B/16 dense shapes."""
import sys, torch
sys.path.insert(0, '.')
from snap_amd import ops
dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
torch.manual_seed(0)
def timeit(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for (K, N) in [(768, 2304), (768, 768), (768, 3072), (3072, 768), (768, 128)]:
  x = torch.randn(M, K, device=dev)
  w = torch.randn(K, N, device=dev) / K ** 0.5
  b = torch.randn(N, device=dev)
  xb = x.to(torch.bfloat16)
  ref = (xb.float() @ w.to(torch.bfloat16).float()) + b
  y0 = ops.dense(x, w, b, math='bf16')
  y1 = ops.dense(xb, w, b, math='bf16', bf16_ring=True)
  y2 = ops.dense(xb, w, b, math='bf16')
  yh = ops.dense(xb, w, b, math='bf16', gelu=True, out_half=True, bf16_ring=True)
  refh = torch.nn.functional.gelu(ref, approximate='tanh')
  e = lambda a: float((a.float() - ref).abs().max())
  print(f'K={K} N={N}: err f32-in {e(y0):.2e} ps1 {e(y1):.2e} xh {e(y2):.2e} ps1==xh {bool(torch.equal(y1, y2))} gelu-half err {float((yh.float() - refh).abs().max()):.2e}')
  t0 = timeit(lambda: ops.dense(x, w, b, math='bf16'))
  t1 = timeit(lambda: ops.dense(xb, w, b, math='bf16', bf16_ring=True))
  t2 = timeit(lambda: ops.dense(xb, w, b, math='bf16'))
  t3 = timeit(lambda: ops.dense(xb, w, b, math='bf16', residual=y0, bf16_ring=True))
  t4 = timeit(lambda: ops.dense(xb, w, b, math='bf16', gelu=True, out_half=True, bf16_ring=True))
  fl = 2.0 * M * K * N / 1e9
  print(f'   f32-in {t0*1e3:7.1f} us {fl/t0:6.0f} TF | ps1 {t1*1e3:7.1f} us {fl/t1:6.0f} TF | xh {t2*1e3:7.1f} us {fl/t2:6.0f} TF | ps1+res {t3*1e3:7.1f} | ps1 gelu half-out {t4*1e3:7.1f}')
