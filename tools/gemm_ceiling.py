"""Ceiling probe (tool, not product): library bf16 GEMM rates on the ViT-B/16 shapes of C5 (20 480 tokens)."""
import torch, time
dev = 'cuda'
M = 20480
for (K, N) in [(768, 2304), (768, 768), (768, 3072), (3072, 768)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'M={M} K={K} N={N}: {ms*1e3:.1f} us  {2*M*K*N/ms/1e9:.0f} TFLOP/s')
