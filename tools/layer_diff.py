"""Per-layer conv timings of two `bench.py --dump` files side by side (A/B of a kernel change)."""
import collections
import json
import sys


def agg(path):
  d = json.load(open(path))
  a = collections.OrderedDict()
  for fam, rows in d.items():
    if not fam.startswith('conv'):
      continue
    for tag, ms, fl, by in rows:
      t = tag.replace('RS_', '').replace('PS_', '').replace('WS_', '')
      x = a.setdefault(t, [0, 0.0, 0.0, 0.0, tag[:3] if tag[:3] in ('RS_', 'PS_', 'WS_') else ''])
      x[0] += 1; x[1] += ms; x[2] += fl; x[3] += by
  return a, {k: sum(x[1] for x in v) for k, v in d.items()}


def main():
  a0, t0 = agg(sys.argv[1])
  a1, t1 = agg(sys.argv[2])
  only = sys.argv[3] if len(sys.argv) > 3 else None
  print({k: (round(t0.get(k, 0), 3), round(t1.get(k, 0), 3)) for k in sorted(set(t0) | set(t1))})
  for k, v in sorted(a1.items(), key=lambda kv: -kv[1][1]):
    if only and v[4] != only:
      continue
    b = a0.get(k)
    print(f'{v[4]:3s}{k:40s} n={v[0]} {b[1] if b else float("nan"):.3f} -> {v[1]:.3f} ms  '
          f'{v[2] / v[1] / 1e9:6.1f} TF {v[3] / v[1] / 1e9:5.2f} TB/s')


if __name__ == '__main__':
  main()
