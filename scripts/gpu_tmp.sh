#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 300 python - <<'PY'
import torch
from snap_amd import ops
dev='cuda'
M,D,Dm=131072,128,32
g=torch.Generator().manual_seed(0)
P=[torch.randn(M,D,generator=g).to(dev) for _ in range(3)]
V=[(torch.rand(M,generator=g)>0.3).to(dev) for _ in range(3)]
Wm=torch.randn(D,Dm,generator=g).to(dev)*0.1; bm=torch.zeros(Dm).to(dev)
def t(fn,n=30):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b)/n*1e3
for npl in (1,2,3):
  for wf in (True, False):
    for vs in ('mask','none'):
      vv=[V[i] if vs=='mask' else None for i in range(npl)]
      print(npl, 'fused' if wf else 'nofused', vs, round(t(lambda: ops.plane_fuse_match(P[:npl], vv, 'max', Wm, bm, want_fused=wf)),1), 'us')
PY
