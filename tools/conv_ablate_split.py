"""Ablation timing of the split-bf16 conv engine on one layer (tuning only; results are wrong
with any bit set): python tools/conv_ablate_split.py <layer index> <math>.
Needs a library built with the hooks compiled in:
  make -C snap_amd/csrc OUT=../lib/alt/libsnap_hip.so OBJDIR=../lib/alt/obj EXTRA=-DSNAP_CONV_SPLIT_ABLATE=1
  SNAP_HIP_LIB=snap_amd/lib/alt/libsnap_hip.so python tools/conv_ablate_split.py 6 bf16x3
Bits (ops.CONV_ABLATE, passed to conv_one_time.py as its third argument): 1 no A loads, 2 no B DMA, 4 no MFMAs, 8 no prologue/split math,
16 no A LDS stores, 32 no fragment fetches."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
layer, math = sys.argv[1], sys.argv[2]
for bits, what in [(0, 'full'), (4, 'no MFMA'), (1, 'no A loads'), (2, 'no B DMA'), (3, 'no loads at all'),
                   (8, 'no convert math'), (24, 'no convert, no A stores'), (32, 'no fragment fetch'),
                   (36, 'no frag fetch, no MFMA'), (63 - 4, 'MFMA only'), (27, 'frag fetch + MFMA only'),
                   (63, 'loop skeleton + barriers')]:
  out = subprocess.run([sys.executable, os.path.join(HERE, 'conv_one_time.py'), layer, math, str(bits)],
                       capture_output=True, text=True)
  print(f'{bits:3d} {what:28s} {out.stdout.strip()}', flush=True)
