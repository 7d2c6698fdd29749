"""Tool (not product): how long the HOST needs to enqueue one C3 training step, not counting the step's closing
device -> host read (trainer.py: the finite flag / gradient norm / metrics leave through one .cpu() per step)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
stamp = {}
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
  if 'first' not in stamp:
    stamp['first'] = time.perf_counter()
  return orig_cpu(self, *a, **k)
torch.Tensor.cpu = cpu
from snap_amd import trainer
orig = trainer.train_step
rec = []
def wrapped(*a, **k):
  stamp.clear()
  t0 = time.perf_counter()
  out = orig(*a, **k)
  t1 = time.perf_counter()
  rec.append((1e3 * (stamp.get('first', t1) - t0), 1e3 * (t1 - t0)))
  return out
trainer.train_step = wrapped
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
bench.main(['--mode', 'train', '--workload', 'c3', '--precision', prec, '--steps', '8', '--warmup', '3', '--no-cpu-baseline', '--no-extra-legs'], emit=False)
r = rec[4:-1]
print('C3', prec, 'host ms until the closing read (median):', sorted(x[0] for x in r)[len(r) // 2], ' whole call:', sorted(x[1] for x in r)[len(r) // 2])
