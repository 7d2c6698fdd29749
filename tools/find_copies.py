"""Where do the small copies of one C2 inference step come from?  Wraps the torch entry points
that end in a hipMemcpy / blit kernel (`copy_`, `clone`, `to`, `contiguous`, `cat`, `stack`,
indexed assignment, `item`) for ONE step and groups the calls by the snap_amd source line."""
import collections
import sys
import traceback

import torch

import bench
from snap_amd import ops

COUNTS = collections.Counter()


def _site():
  for fr in reversed(traceback.extract_stack()[:-2]):
    if 'snap_amd' in fr.filename or fr.filename.endswith('bench.py'):
      return f'{fr.filename.split("repo/")[-1]}:{fr.lineno} {fr.line.strip()[:90]}'
  return '?'


def _wrap(owner, name):
  orig = getattr(owner, name)

  def f(*a, **k):
    t = a[0] if a and isinstance(a[0], torch.Tensor) else None
    note = ''
    if t is not None and name in ('contiguous',) and t.is_contiguous():
      return orig(*a, **k)
    if t is not None and name == 'to':
      r = orig(*a, **k)
      if r is t:
        return r
      note = f' {t.device.type}->{r.device.type}'
      COUNTS[(name + note, _site())] += 1
      return r
    COUNTS[(name + note, _site())] += 1
    return orig(*a, **k)
  setattr(owner, name, f)
  return orig


def main():
  device = torch.device('cuda')
  ops.MATMUL_PRECISION = 'bf16x3'
  loc, cfg, meta, variables, batch = bench.build('c2', device, 0, materialize_volume=False)
  step = lambda i: loc.apply(variables, batch, train=False, rngs={'sampling': i})
  for i in range(3):
    step(i)
  torch.cuda.synchronize()
  saved = [(torch.Tensor, n, _wrap(torch.Tensor, n)) for n in
           ('copy_', 'clone', 'to', 'contiguous', '__setitem__', 'item', 'cpu', 'tolist', 'fill_', 'zero_')]
  saved += [(torch, n, _wrap(torch, n)) for n in ('cat', 'stack', 'tensor', 'as_tensor', 'where', 'zeros', 'ones', 'full')]
  step(3)
  torch.cuda.synchronize()
  for o, n, f in saved:
    setattr(o, n, f)
  tot = collections.Counter()
  for (name, site), n in COUNTS.items():
    tot[name] += n
  print(dict(tot))
  for (name, site), n in COUNTS.most_common(80):
    print(f'{n:4d} {name:22s} {site}')


if __name__ == '__main__':
  sys.exit(main())
