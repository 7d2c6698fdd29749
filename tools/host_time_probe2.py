"""Tool (not product): host time to ENQUEUE one inference step on an idle GPU (device synchronised first, nothing waited for
inside): the pure launch-side cost of a step, free of the back-pressure of a full queue."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
rec = []
orig_main = bench.main
import snap_amd.pipeline as pipeline
orig_slot = pipeline.BatchesInFlight.slot
import contextlib
@contextlib.contextmanager
def slot(self, i):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  with orig_slot(self, i) as s:
    yield s
  rec.append(1e3 * (time.perf_counter() - t0))
pipeline.BatchesInFlight.slot = slot
for w in sys.argv[1:] or ['c2', 'c5', 'c4']:
  rec.clear()
  bench.main(['--workload', w, '--steps', '10', '--warmup', '4', '--no-cpu-baseline', '--no-extra-legs'], emit=False)
  r = sorted(rec[6:-1])
  print(w, 'host ms to enqueue one step on an idle GPU (median):', round(r[len(r) // 2], 3), ' min', round(r[0], 3))
