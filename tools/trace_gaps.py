"""GPU busy / idle time from a rocprofv3 --kernel-trace CSV: the union of the kernel intervals over
the last `window` ms of the trace (the timed steps), the idle time between them, the histogram of
the gaps and which kernels they precede.  Answers "how much of a step is launch gaps".

  python tools/trace_gaps.py <kernel_trace.csv> [window_ms]
"""
import collections
import csv
import sys


def main():
  path = sys.argv[1]
  rows = []
  for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
  rows.sort()
  t_end = max(e for _, e, _ in rows)
  window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
  rows = [r for r in rows if r[0] >= t_end - window]
  t0 = rows[0][0]
  busy = 0
  cur_s, cur_e = rows[0][0], rows[0][1]
  gaps = []
  for s, e, name in rows[1:]:
    if s > cur_e:
      busy += cur_e - cur_s
      gaps.append((s - cur_e, name))
      cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  busy += cur_e - cur_s
  span = cur_e - t0
  ksum = sum(e - s for s, e, _ in rows)
  print(f'window {span / 1e6:.3f} ms, {len(rows)} kernels, busy(union) {busy / 1e6:.3f} ms '
        f'({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.3f} ms, kernel-duration sum {ksum / 1e6:.3f} ms')
  hist = collections.Counter()
  for g, _ in gaps:
    b = 1 if g < 1000 else 2 if g < 3000 else 5 if g < 5000 else 10 if g < 10000 else 30 if g < 30000 else 100 if g < 100000 else 1000
    hist[b] += g
  for b in sorted(hist):
    n = sum(1 for g, _ in gaps if (1 if g < 1000 else 2 if g < 3000 else 5 if g < 5000 else 10 if g < 10000 else 30 if g < 30000 else 100 if g < 100000 else 1000) == b)
    print(f'  gaps < {b:5d} us: {n:5d} gaps, {hist[b] / 1e6:7.3f} ms')
  by = collections.Counter()
  for g, name in gaps:
    by[name] += g
  print('idle time in front of:')
  for name, g in by.most_common(15):
    print(f'  {g / 1e6:7.3f} ms  {name}')


if __name__ == '__main__':
  main()
