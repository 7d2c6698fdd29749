#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
timeout 300 python - <<'PY'
import collections, traceback, torch, sys
sys.path.insert(0, '.')
import bench
from snap_amd import ops
dev = torch.device('cuda', 0)
loc, cfg, meta, variables, batch = bench.build('c2', dev, 0, materialize_volume=False)
loc.engine = 'bf16x3'
def step(i): loc.apply(variables, batch, train=False, rngs={'sampling': i})
for i in range(3): step(i)
torch.cuda.synchronize()
counts = collections.Counter()
def site():
  for fr in reversed(traceback.extract_stack()[:-2]):
    if 'snap_amd' in fr.filename:
      return f"{fr.filename.split('snap_amd/')[-1]}:{fr.lineno}"
  return '?'
orig_to = torch.Tensor.to
def to(self, *a, **k):
  out = orig_to(self, *a, **k)
  if not self.is_cuda and out.is_cuda: counts[('to', site())] += 1
  return out
torch.Tensor.to = to
orig_tensor = torch.tensor
def tensor(*a, **k):
  out = orig_tensor(*a, **k)
  if out.is_cuda: counts[('tensor', site())] += 1
  return out
torch.tensor = tensor
orig_as = torch.as_tensor
def as_tensor(*a, **k):
  out = orig_as(*a, **k)
  if out.is_cuda and not (a and isinstance(a[0], torch.Tensor) and a[0].is_cuda): counts[('as_tensor', site())] += 1
  return out
torch.as_tensor = as_tensor
orig_copy = torch.Tensor.copy_
def copy_(self, src, *a, **k):
  if self.is_cuda and isinstance(src, torch.Tensor) and not src.is_cuda: counts[('copy_ h2d', site())] += 1
  elif self.is_cuda and isinstance(src, torch.Tensor) and src.is_cuda and self.is_contiguous() and src.is_contiguous(): counts[('copy_ d2d contiguous', site())] += 1
  return orig_copy(self, src, *a, **k)
torch.Tensor.copy_ = copy_
orig_clone = torch.Tensor.clone
def clone(self, *a, **k):
  if self.is_cuda: counts[('clone', site())] += 1
  return orig_clone(self, *a, **k)
torch.Tensor.clone = clone
orig_contig = torch.Tensor.contiguous
def contiguous(self, *a, **k):
  if self.is_cuda and not self.is_contiguous(): counts[('contiguous(copy)', site())] += 1
  return orig_contig(self, *a, **k)
torch.Tensor.contiguous = contiguous
step(7)
torch.cuda.synchronize()
for k, v in counts.most_common(60): print(v, k)
PY
