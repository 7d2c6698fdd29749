"""world_size-2 gloo test (CPU) of the multi-GPU path of bench.py.

The hot path shards by scene with no data-path collective; what needs
covering is the launch contract: one process per rank, per-rank scene shard,
barrier-bracketed timing with MAX over ranks, one JSON line on rank 0.
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def test_bench_two_ranks_gloo():
  cmd = [
      sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
      '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
      os.path.join(ROOT, 'tests', 'dist_driver.py'),
      '--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'tiny', '--device', 'cpu',
  ]
  env = dict(os.environ, OMP_NUM_THREADS='2')
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, f'expected exactly one JSON line, got {len(lines)}: {out.stdout[-2000:]}'
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['warmup'] == 1
  assert rec['scaling'] == 'weak' and rec['higher_is_better'] is True
  assert rec['config']['global_batch'] == 4 and rec['config']['scenes_per_gpu'] == 2
  # value = all ranks' scenes / max-over-ranks time
  expect = rec['config']['global_batch'] * rec['steps'] / (rec['ms_per_step'] * rec['steps'] / 1e3)
  assert abs(rec['value'] - expect) / expect < 1e-3
  assert 'cpu_baseline' not in rec    # rank 0 at N=1 only


def test_bench_eight_ranks_gloo():
  """The launch contract at the node's full width (8 ranks, as `bench.py --gpus 8` is launched):
  rank -> shard mapping, distinct per-rank data, one JSON line, whole-job value, and the
  process-group facts a multi-GPU record must carry (backend, world size, per-rank shards)."""
  cmd = [
      sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
      '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
      os.path.join(ROOT, 'tests', 'dist_driver.py'),
      '--gpus', '8', '--steps', '1', '--warmup', '1', '--workload', 'tiny', '--device', 'cpu',
  ]
  env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, f'expected exactly one JSON line, got {len(lines)}: {out.stdout[-2000:]}'
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 8 and rec['config']['global_batch'] == 16 and rec['config']['scenes_per_gpu'] == 2
  d = rec['distributed']
  assert d['world_size'] == 8 and d['backend'] == 'gloo' and d['ranks'] == list(range(8))
  assert len(set(d['shard_digests'])) == 8, d           # every rank generated its OWN scenes
  expect = rec['config']['global_batch'] * rec['steps'] / (rec['ms_per_step'] * rec['steps'] / 1e3)
  assert abs(rec['value'] - expect) / expect < 1e-3


def test_gradient_and_metric_sync_eight_ranks_gloo():
  """The training exchange step (bucketed gradient mean, finite flag, metric psum, overlapped
  reducer) at world size 8; every rank derives the same bucket plan."""
  cmd = [
      sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
      '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
      os.path.join(ROOT, 'tests', 'dist_sync_driver.py'),
  ]
  env = dict(os.environ, OMP_NUM_THREADS='1')
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
  assert out.returncode == 0, out.stderr[-3000:]
  assert 'DIST_SYNC_OK' in out.stdout and 'BUCKET_PLAN_EQUAL 8' in out.stdout


def test_bench_rejects_world_size_mismatch():
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'tiny',
       '--device', 'cpu'],
      capture_output=True, text=True, timeout=300, cwd=ROOT,
      env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')},
  )
  assert out.returncode != 0 and 'WORLD_SIZE' in (out.stderr + out.stdout)


def test_gradient_and_metric_sync_two_ranks_gloo():
  """snap_amd.dist: bucketed pmean of a gradient tree, finite flag, metric psum
  (the exchange step of snap/trainer.py:57-67,225-234,260-277)."""
  cmd = [
      sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
      '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
      os.path.join(ROOT, 'tests', 'dist_sync_driver.py'),
  ]
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-3000:]
  assert 'DIST_SYNC_OK' in out.stdout
