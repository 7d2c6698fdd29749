"""Where the cycles of the split engine's plain 1 x 1 body go, phase by phase (alt build only:
scripts/build_alt.sh timeline conv_split.hip -DSNAP_CONV_TIMELINE=1; SNAP_HIP_LIB=snap_amd/lib/alt_timeline/libsnap_hip.so).

  SNAP_HIP_LIB=... python tools/conv_timeline.py
"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from snap_amd import _lib, ops  # noqa: E402

LAYERS = [
    ('stage3 1x1 1024->256 40x34x34', (40, 34, 34, 1024), 256),
    ('stage2 1x1 512->128 40x68x68', (40, 68, 68, 512), 128),
    ('stage4 1x1 2048->512 40x17x17', (40, 17, 17, 2048), 512),
]
PHASES = ['issue loads / DMA', 'fragment fetch', 'MFMA issue', 'wait A rows', 'prologue + split + stores', 'barrier', 'loop control']


def main():
  lib = _lib.load()
  lib.snap_debug_conv_timeline.restype = ctypes.c_int
  lib.snap_debug_conv_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  ops.CONV_NO_RS = True
  for name, xs, cout in LAYERS:
    cin = xs[-1]
    x = torch.randn(xs, device=dev, generator=g)
    w = torch.randn((1, 1, cin, cout), device=dev, generator=g) / cin ** 0.5
    gn = (torch.zeros(xs[0], cin, device=dev), torch.ones(xs[0], cin, device=dev), torch.zeros(cin, device=dev))
    w._snap_packed = {'bf16x3': ops.pack_weights_split_bf16(w, 2)}
    for _ in range(3):
      ops.conv2d(x, w, math='bf16x3', prologue=ops.PRO_GN_RELU, gn=gn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, w, math='bf16x3', prologue=ops.PRO_GN_RELU, gn=gn)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 8, np.uint64)
    assert lib.snap_debug_conv_timeline(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(1024, 4, 8).astype(np.float64)
    live = t[..., 7] > 0
    steps = t[..., 7][live]
    per = t[..., :7][live] / steps[:, None]          # cycles per k-step and wave
    mean = per.mean(0)
    print(f'{name}: {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented), {int(live.sum())} waves sampled, '
          f'{steps.mean():.0f} k-steps, {mean.sum():.0f} cycles per k-step and wave')
    for ph, c in zip(PHASES, mean):
      print(f'   {ph:28s} {c:8.0f} cycles  {100 * c / mean.sum():5.1f} %')


if __name__ == '__main__':
  main()
