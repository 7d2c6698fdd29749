"""`-m gpu`: the data-parallel exchange step on device tensors -- two ranks sharing the one GPU of
the test box (the only collective of the path is the gradient all-reduce).

RCCL (backend 'nccl') is tried first; a communicator with two ranks on ONE device is something
RCCL may refuse ("duplicate GPU"), in which case the same driver runs over gloo with CUDA
tensors -- the reducer, bucket layout, hooks and overlap logic are backend-independent, so that
still exercises everything but RCCL's own transport.  The test prints which backend ran."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _run(backend, timeout):
  cmd = [
      sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
      '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
      os.path.join(ROOT, 'tests', 'dist_gpu_driver.py'), backend,
  ]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
  try:
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
  except subprocess.TimeoutExpired as e:
    class _R:
      returncode = -9
      stdout = (e.stdout or b'').decode() if isinstance(e.stdout, bytes) else (e.stdout or '')
      stderr = 'timeout'
    return _R()


def test_two_rank_train_step_gradients_on_one_gpu():
  out = _run('nccl', 240)
  ran = 'nccl'
  if out.returncode != 0 or 'DIST_GPU_OK' not in out.stdout:
    why = (out.stderr or '')[-400:].replace('\n', ' | ')
    print(f'[dist] RCCL with two ranks on one device did not run ({why}); falling back to gloo + device tensors')
    out = _run('gloo', 600)
    ran = 'gloo'
  assert out.returncode == 0, out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if 'DIST_GPU_OK' in l]
  assert line, out.stdout[-2000:]
  print(f'[dist] {line[0]} (ran on {ran})')


def test_rccl_single_rank_forced_collectives():
  """Backend 'nccl' (RCCL) at world size 1 with ``dist.FORCE_COLLECTIVES``: the exchange step of a
  real tiny train_step goes through librccl on the GPU (no fallback: RCCL must run here)."""
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4',
             MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1',
             LOCAL_RANK='0')
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dist_rccl1_driver.py')],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
  assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
  line = [l for l in out.stdout.splitlines() if 'RCCL1_OK' in l]
  assert line and 'backend=nccl' in line[0], out.stdout[-2000:]
  print(f'[dist] {line[0]}')
