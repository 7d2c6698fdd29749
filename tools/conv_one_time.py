"""Average launch time of one layer of tools/conv_bench.LAYERS on one engine (helper of
conv_ablate_split.py)."""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from snap_amd import ops  # noqa: E402
import conv_bench  # noqa: E402

name, xs, ws, stride, pad, pro = conv_bench.LAYERS[int(sys.argv[1])]
math = sys.argv[2]
os.environ['SNAP_ALT_ABLATE'] = sys.argv[3] if len(sys.argv) > 3 else '0'   # timing-only ablation bits (alt build only)
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(xs, device=dev, generator=g)
w = torch.randn(ws, device=dev, generator=g) / (ws[0] * ws[1] * ws[2]) ** 0.5
gn = None
if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
  gn = (torch.zeros(xs[0], ws[2], device=dev), torch.ones(xs[0], ws[2], device=dev), torch.zeros(ws[2], device=dev))
if math in ops.SPLIT_PARTS:
  w._snap_packed = {math: ops.pack_weights_split_bf16(w, ops.SPLIT_PARTS[math])}
kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), cin=ws[2], prologue=pro, gn=gn)
for _ in range(3):
  ops.conv2d(x, w, math=math, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
  ops.conv2d(x, w, math=math, **kw)
e1.record()
torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  {name}')
