"""Per-layer table of one bench step's conv launches (bench.py --dump / SNAP_BENCH_DUMP file):
launch count, time, TFLOP/s, GB/s on algorithmic bytes, and the layer's floor
max(3 * flops / 2.5 PF, bytes / 6.3 TB/s) with the gap to it.

  python tools/launch_table.py profiles/r03_c2_launches.json [family]
"""
import collections
import json
import sys


def main():
  d = json.load(open(sys.argv[1]))
  fam = sys.argv[2] if len(sys.argv) > 2 else 'conv_split_bf16x3'
  agg = collections.OrderedDict()
  for tag, ms, flops, nbytes in d[fam]:
    a = agg.setdefault(tag, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += flops; a[3] += nbytes
  tot = tot_floor = 0.0
  print(f'{"layer":50s} {"n":>3s} {"ms":>7s} {"us/launch":>9s} {"TF":>6s} {"GB/s":>6s} {"floor":>6s} {"gap":>6s}')
  for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    floor = max(3 * a[2] / 2.5e15, a[3] / 6.3e12) * 1e3
    tot += a[1]; tot_floor += floor
    print(f'{k:50s} {a[0]:3d} {a[1]:7.3f} {a[1] / a[0] * 1e3:9.1f} {a[2] / a[1] / 1e9:6.1f} {a[3] / a[1] / 1e6:6.0f} '
          f'{floor:6.3f} {a[1] - floor:6.3f}')
  print(f'total {tot:.3f} ms, floor {tot_floor:.3f} ms, launches {sum(a[0] for a in agg.values())}')


if __name__ == '__main__':
  main()
