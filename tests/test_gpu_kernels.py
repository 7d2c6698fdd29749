"""`-m gpu` parity tests: every HIP kernel (through the C ABI) vs the numpy oracle.

Same seeded inputs on both sides; sizes the oracle finishes in seconds.
Tolerances: bit-exact for masks / indices; fp32 round-off class (<= 1e-4 of the
tensor's magnitude, stated per test) for floating point -- the north-star bound is
1e-3 on feature maps.
"""
import os

import numpy as np
import pytest
import torch

import helpers
import oracle_ops
from snap_amd import ops

pytestmark = pytest.mark.gpu

DEV = helpers.DEVICE


def rnd(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(shape, generator=g) * scale)


def _unaligned(t):
  """The same values at a 4-byte (not 16-byte) aligned address: kernels whose fast variant needs
  aligned operands then take their general variant (the library has no environment switches)."""
  buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
  out = buf[1:].view(t.shape)
  out.copy_(t)
  assert out.data_ptr() % 16 != 0 and out.is_contiguous()
  return out


def both(fn_name, args_cpu, kwargs=None, to_gpu=None):
  """Run ops.<fn> on the GPU and oracle_ops.<fn> on the CPU with the same inputs."""
  kwargs = kwargs or {}
  def mv(x):
    if isinstance(x, torch.Tensor):
      return x.to(DEV).contiguous()
    if isinstance(x, (list, tuple)):
      return type(x)(mv(i) for i in x)
    return x
  got = getattr(ops, fn_name)(*mv(args_cpu), **{k: mv(v) for k, v in kwargs.items()})
  want = getattr(oracle_ops, fn_name)(*args_cpu, **kwargs)
  if DEV == 'cuda':
    torch.cuda.synchronize()
  return got, want


# ----------------------------------------------------------------------------
# conv engine
# ----------------------------------------------------------------------------
CONV_CASES = [
    # name, N,H,W,Cin, KH,KW,Cout, stride, pad
    ('1x1_64_256', 2, 9, 11, 64, 1, 1, 256, 1, 0),
    ('1x1_s2_proj', 2, 10, 12, 128, 1, 1, 64, 2, 0),
    ('3x3_s1', 1, 13, 9, 32, 3, 3, 64, 1, 1),
    ('3x3_s2', 2, 14, 10, 64, 3, 3, 128, 2, 1),
    ('7x7_root_scalar', 1, 24, 20, 3, 7, 7, 32, 2, 3),
    ('3x3_root_scalar', 1, 12, 12, 3, 3, 3, 64, 1, 1),
    ('1x1_cout160', 1, 8, 8, 128, 1, 1, 160, 1, 0),
    ('1x1_cin20_ktail', 1, 6, 7, 20, 1, 1, 36, 1, 0),
    ('big_template', 1, 20, 20, 8, 7, 7, 12, 1, 0),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_plain(case):
  _, N, H, W, Cin, KH, KW, Cout, stride, pad = case
  x = rnd((N, H, W, Cin), 1)
  w = rnd((KH, KW, Cin, Cout), 2, 1.0 / np.sqrt(KH * KW * Cin))
  kw = dict(stride=stride, padding=((pad, pad), (pad, pad)))
  got, want = both('conv2d', (x, w), kw)
  helpers.report('conv ' + case[0], got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('tile', ['128x128', '128x64', '64x128', '64x64'])
@pytest.mark.parametrize('bk', ['16', '32'])
def test_conv_every_tile_variant(tile, bk, monkeypatch):
  """Force each (tile, K-slab depth) instantiation of the engine on shapes with
  M / N / K tails, a GroupNorm prologue and residual + bias + ReLU epilogues."""
  monkeypatch.setattr(ops, 'CONV_TILE', tile)
  monkeypatch.setattr(ops, 'CONV_BK', int(bk))
  N, H, W, Cin, Cout = 2, 15, 13, 96, 200
  x = rnd((N, H, W, Cin), 31) + 0.2
  w = rnd((3, 3, Cin, Cout), 32, 1 / np.sqrt(9 * Cin))
  gamma, beta = rnd((Cin,), 33) + 1, rnd((Cin,), 34) * 0.1
  mu, sc = oracle_ops.group_norm_stats(x, gamma)
  res = rnd((N, H, W, Cout), 35)
  bias = rnd((Cout,), 36)
  kw = dict(padding=((1, 1), (1, 1)), prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta), residual=res,
            bias=bias, relu=True)
  got, want = both('conv2d', (x, w), kw)
  helpers.report(f'conv tile {tile} bk {bk}', got, want, atol=5e-5, rtol=1e-5)
  xs = rnd((1, 1, 700, 260), 37)
  ws = rnd((1, 1, 257, 256), 38, 1 / 16.0)
  got, want = both('conv2d', (xs, ws), dict(cin=257))
  helpers.report(f'dense k257 tile {tile} bk {bk}', got, want, atol=5e-5, rtol=1e-5)
  x3 = torch.rand((1, 20, 18, 3), generator=torch.Generator().manual_seed(39))
  w3 = rnd((7, 7, 3, 64), 40, 0.1)
  got, want = both('conv2d', (x3, w3), dict(stride=2, padding=((3, 3), (3, 3)),
                                            prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0)))
  helpers.report(f'scalar path tile {tile}', got, want, atol=5e-5, rtol=1e-5)


# bf16-operand engine (training precision): compared with the restatement that rounds both
# operands to bf16 after the f32 prologue (oracle/encoder.py:bf16_round); what is left is the
# f32 accumulation order, so the tolerance stays in the fp32 round-off class.
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_bf16_plain(case):
  _, N, H, W, Cin, KH, KW, Cout, stride, pad = case
  x = rnd((N, H, W, Cin), 1)
  w = rnd((KH, KW, Cin, Cout), 2, 1.0 / np.sqrt(KH * KW * Cin))
  kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), math='bf16')
  got, want = both('conv2d', (x, w), kw)
  helpers.report('conv bf16 ' + case[0], got, want, atol=3e-5, rtol=1e-5)
  # ... and it is NOT the f32 result (the engine really ran in bf16) where the shape allows it
  if Cin >= 4 and Cin % 4 == 0 and DEV == 'cuda':
    exact = oracle_ops.conv2d(x, w, stride=stride, padding=((pad, pad), (pad, pad)))
    assert (got.cpu() - exact).abs().max() > 1e-4


# IEEE-half operand engine (math='fp16': the reference's dtype=float16 train config,
# train_localization.py:93): the same kernels on v_mfma_f32_32x32x16_f16, compared with the
# restatement that rounds both operands to binary16 (oracle/encoder.py:fp16_round).
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fp16_plain(case):
  _, N, H, W, Cin, KH, KW, Cout, stride, pad = case
  x = rnd((N, H, W, Cin), 1)
  w = rnd((KH, KW, Cin, Cout), 2, 1.0 / np.sqrt(KH * KW * Cin))
  kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), math='fp16')
  got, want = both('conv2d', (x, w), kw)
  helpers.report('conv fp16 ' + case[0], got, want, atol=3e-5, rtol=1e-5)
  if Cin >= 4 and Cin % 4 == 0 and DEV == 'cuda':
    exact = oracle_ops.conv2d(x, w, stride=stride, padding=((pad, pad), (pad, pad)))
    bf = ops.conv2d(x.to(DEV), w.to(DEV), math='bf16', **{k: v for k, v in kw.items() if k != 'math'})
    e16 = float((got.cpu() - exact).abs().max())
    assert 1e-6 < e16 < float((bf.cpu() - exact).abs().max())       # half really ran: 11 bits beat 8


def test_conv_fp16_prologues_overflow_and_underflow():
  """GroupNorm prologues / residual / bias / ReLU on the half engine; an operand beyond 65504 becomes
  inf (and reaches the output: what DynamicScale watches for), one below 2^-25 flushes to zero."""
  N, H, W, Cin, Cout = 2, 15, 13, 96, 200
  x = rnd((N, H, W, Cin), 31) + 0.2
  w = rnd((3, 3, Cin, Cout), 32, 1 / np.sqrt(9 * Cin))
  gamma, beta = rnd((Cin,), 33) + 1, rnd((Cin,), 34) * 0.1
  res = rnd((N, H, W, Cout), 35)
  bias = rnd((Cout,), 36)
  for pro, relu_first in ((ops.PRO_GN_RELU, False), (ops.PRO_RELU_GN, True)):
    mu, sc = oracle_ops.group_norm_stats(x, gamma, relu_first=relu_first)
    kw = dict(padding=((1, 1), (1, 1)), prologue=pro, gn=(mu, sc, beta), residual=res,
              bias=bias, relu=True, math='fp16')
    got, want = both('conv2d', (x, w), kw)
    helpers.report(f'conv fp16 pro {pro}', got, want, atol=5e-5, rtol=1e-5)
  xs = rnd((1, 1, 700, 260), 37)
  ws = rnd((1, 1, 257, 256), 38, 1 / 16.0)
  got, want = both('conv2d', (xs, ws), dict(cin=257, prologue=ops.PRO_RELU, math='fp16'))
  helpers.report('dense fp16 k257', got, want, atol=5e-5, rtol=1e-5)
  xo = rnd((1, 4, 4, 64), 39)
  wo = rnd((1, 1, 64, 64), 40, 0.1)
  xo[0, 1, 2, 5] = 7.0e4                         # > 65504: inf in half
  xo[0, 2, 2, 7] = 1.0e-9                        # < 2^-25: zero in half
  y = ops.conv2d(xo.to(DEV), wo.to(DEV), math='fp16').cpu()
  assert bool(torch.isinf(y[0, 1, 2]).all()) and bool(torch.isfinite(y[0, 2, 2]).all())
  got, want = both('conv2d', (xo, wo), dict(math='fp16'))
  helpers.report('conv fp16 overflow', got, want, atol=3e-5, rtol=1e-5)
  assert bool(torch.isfinite(ops.conv2d(xo.to(DEV), wo.to(DEV), math='bf16')).all())


@pytest.mark.parametrize('tile', ['128x128', '128x64', '64x128', '64x64'])
def test_conv_bf16_every_tile_variant(tile, monkeypatch):
  monkeypatch.setattr(ops, 'CONV_TILE', tile)
  N, H, W, Cin, Cout = 2, 15, 13, 96, 200
  x = rnd((N, H, W, Cin), 31) + 0.2
  w = rnd((3, 3, Cin, Cout), 32, 1 / np.sqrt(9 * Cin))
  gamma, beta = rnd((Cin,), 33) + 1, rnd((Cin,), 34) * 0.1
  res = rnd((N, H, W, Cout), 35)
  bias = rnd((Cout,), 36)
  for pro, relu_first in ((ops.PRO_GN_RELU, False), (ops.PRO_RELU_GN, True)):
    mu, sc = oracle_ops.group_norm_stats(x, gamma, relu_first=relu_first)
    kw = dict(padding=((1, 1), (1, 1)), prologue=pro, gn=(mu, sc, beta), residual=res,
              bias=bias, relu=True, math='bf16')
    got, want = both('conv2d', (x, w), kw)
    helpers.report(f'conv bf16 tile {tile} pro {pro}', got, want, atol=5e-5, rtol=1e-5)
  xs = rnd((1, 1, 700, 260), 37)
  ws = rnd((1, 1, 257, 256), 38, 1 / 16.0)
  got, want = both('conv2d', (xs, ws), dict(cin=257, prologue=ops.PRO_RELU, math='bf16'))
  helpers.report(f'dense bf16 k257 tile {tile}', got, want, atol=5e-5, rtol=1e-5)
  x2 = rnd((2, 11, 9, 40), 41)
  w2 = rnd((3, 3, 40, 72), 42, 1 / np.sqrt(360))
  prev = rnd((2, 3, 3, 72), 43)
  got, want = both('conv2d', (x2, w2), dict(stride=2, padding=((1, 1), (1, 1)), up_prev=None,
                                            prologue=ops.PRO_AFFINE, in_affine=(0.5, 0.25),
                                            math='bf16'))
  helpers.report(f'conv bf16 s2 affine tile {tile}', got, want, atol=5e-5, rtol=1e-5)
  x3 = rnd((2, 6, 6, 64), 44)
  w3 = rnd((1, 1, 64, 72), 45, 1 / 8.0)
  got, want = both('conv2d', (x3, w3), dict(up_prev=prev, math='bf16'))
  helpers.report(f'conv bf16 upsample-add tile {tile}', got, want, atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,Cin,Cout,k,pro', [(3, 17, 17, 64, 256, 1, ops.PRO_GN_RELU), (2, 15, 13, 128, 200, 3, ops.PRO_RELU_GN),
                                                  (5, 12, 11, 512, 128, 1, ops.PRO_GN_RELU), (2, 34, 34, 256, 64, 3, ops.PRO_GN_RELU)])
def test_conv_bf16_groupnorm_table_keeps_the_bits(N, H, W, Cin, Cout, k, pro, math_, monkeypatch):
  """Training-precision engines, GroupNorm prologue: the statistics of a slab's 32 channels come from an LDS table
  fetched two slabs ahead (row tiles of <= Ho Wo pixels: at most two images) instead of two loads per staged row.
  Same operands, same arithmetic: bit-identical to the per-row loader (``CONV_NO_PLAIN`` pins it), tiles that
  straddle images and split-K shapes included."""
  x = (rnd((N, H, W, Cin), 61) + 0.2).to(DEV)
  w = rnd((k, k, Cin, Cout), 62, 1 / np.sqrt(k * k * Cin)).to(DEV)
  gamma, beta = (rnd((Cin,), 63) + 1).to(DEV), (rnd((Cin,), 64) * 0.1).to(DEV)
  mu, sc = ops.group_norm_stats(x, gamma, relu_first=pro == ops.PRO_RELU_GN)
  pad = (k - 1) // 2
  kw = dict(padding=((pad, pad), (pad, pad)), prologue=pro, gn=(mu, sc, beta), math=math_)
  table = ops.conv2d(x, w, **kw)
  monkeypatch.setattr(ops, 'CONV_NO_PLAIN', True)
  rows = ops.conv2d(x, w, **kw)
  assert torch.equal(table, rows), float((table - rows).abs().max())
  assert float(table.abs().max()) > 0


# split-bf16 engine (f32-grade accuracy on the bf16 matrix cores): compared with the EXACT conv of
# the f32 operands, like the f32 engine.  'bf16x6' (3 parts, 6 products) is held to the f32
# engine's own tolerance; 'bf16x3' (2 parts, 3 products, ~2^-17 per product) to 1e-4.
SPLIT_TOL = {'bf16x6': 2e-5, 'bf16x3': 1e-4}


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_split_plain(case, math):
  _, N, H, W, Cin, KH, KW, Cout, stride, pad = case
  x = rnd((N, H, W, Cin), 1)
  w = rnd((KH, KW, Cin, Cout), 2, 1.0 / np.sqrt(KH * KW * Cin))
  kw = dict(stride=stride, padding=((pad, pad), (pad, pad)), math=math)
  got, want = both('conv2d', (x, w), kw)
  helpers.report(f'conv {math} ' + case[0], got, want, atol=SPLIT_TOL[math], rtol=1e-5)


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
@pytest.mark.parametrize('tile', ['128x128', '128x64', '64x128', '64x64'])
def test_conv_split_every_tile_variant(tile, math, monkeypatch):
  monkeypatch.setattr(ops, 'CONV_TILE', tile)
  tol = 2.5 * SPLIT_TOL[math]
  N, H, W, Cin, Cout = 2, 15, 13, 96, 200
  x = rnd((N, H, W, Cin), 31) + 0.2
  w = rnd((3, 3, Cin, Cout), 32, 1 / np.sqrt(9 * Cin))
  gamma, beta = rnd((Cin,), 33) + 1, rnd((Cin,), 34) * 0.1
  res = rnd((N, H, W, Cout), 35)
  bias = rnd((Cout,), 36)
  for pro, relu_first in ((ops.PRO_GN_RELU, False), (ops.PRO_RELU_GN, True)):
    mu, sc = oracle_ops.group_norm_stats(x, gamma, relu_first=relu_first)
    kw = dict(padding=((1, 1), (1, 1)), prologue=pro, gn=(mu, sc, beta), residual=res,
              bias=bias, relu=True, math=math)
    got, want = both('conv2d', (x, w), kw)
    helpers.report(f'conv {math} tile {tile} pro {pro}', got, want, atol=tol, rtol=1e-5)
  xs = rnd((1, 1, 700, 260), 37)
  ws = rnd((1, 1, 257, 256), 38, 1 / 16.0)
  got, want = both('conv2d', (xs, ws), dict(cin=257, prologue=ops.PRO_RELU, math=math))
  helpers.report(f'dense {math} k257 tile {tile}', got, want, atol=tol, rtol=1e-5)
  x2 = rnd((2, 11, 9, 40), 41)
  w2 = rnd((3, 3, 40, 72), 42, 1 / np.sqrt(360))
  prev = rnd((2, 3, 3, 72), 43)
  got, want = both('conv2d', (x2, w2), dict(stride=2, padding=((1, 1), (1, 1)),
                                            prologue=ops.PRO_AFFINE, in_affine=(0.5, 0.25),
                                            math=math))
  helpers.report(f'conv {math} s2 affine tile {tile}', got, want, atol=tol, rtol=1e-5)
  x3 = rnd((2, 6, 6, 64), 44)
  w3 = rnd((1, 1, 64, 72), 45, 1 / 8.0)
  got, want = both('conv2d', (x3, w3), dict(up_prev=prev, math=math))
  helpers.report(f'conv {math} upsample-add tile {tile}', got, want, atol=tol, rtol=1e-5)


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
def test_conv_split_rows_gnstats_splitk_and_scales(math):
  """Row-indexed launches, GroupNorm partial sums, split-K, degenerate shapes and operand
  magnitudes far from 1 (the split is exponent-agnostic: power-of-two scaling is exact)."""
  tol = 2.5 * SPLIT_TOL[math]
  g = torch.Generator().manual_seed(50)
  M, Cin, Cout = 3000, 260, 256
  x = rnd((M, Cin), 51)
  w = rnd((257, Cout), 52, 1 / 16.0)
  bias = rnd((Cout,), 53)
  mask = torch.rand(M, generator=g) > 0.6
  want = oracle_ops.dense(x, w, bias, cin=257, relu=True)
  index, count = ops.compact_rows(mask.to(DEV))
  out = torch.zeros(M, Cout, device=DEV)
  ops.dense(x.to(DEV), w.to(DEV), bias.to(DEV), cin=257, relu=True, rows_in=index, rows_out=index,
            row_count=count, out=out, math=math)
  helpers.report(f'{math} row-indexed dense', out[mask.to(DEV)], want[mask], atol=tol, rtol=1e-5)
  assert float(out[~mask.to(DEV)].abs().max()) == 0.0
  N, H, W, C1, C2 = 2, 24, 20, 64, 128
  xi = rnd((N, H, W, C1), 54)
  wi = rnd((3, 3, C1, C2), 55, 1 / 24.0)
  gamma = rnd((C2,), 56) + 1
  ops.USE_SPLITK = False
  try:
    y = ops.conv2d(xi.to(DEV), wi.to(DEV), padding=((1, 1), (1, 1)), emit_gn_stats='raw', math=math)
  finally:
    ops.USE_SPLITK = True
  assert hasattr(y, '_snap_gn_partial')
  mu_f, sc_f = ops.group_norm_stats(y, gamma.to(DEV))
  mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), gamma)
  helpers.report(f'{math} fused gn mu', mu_f, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report(f'{math} fused gn sc', sc_f, sc_w, atol=1e-5, rtol=5e-5)
  xs = rnd((2, 8, 8, 512), 57)
  ws = rnd((3, 3, 512, 128), 58, 1 / np.sqrt(9 * 512))
  kw = dict(padding=((1, 1), (1, 1)), math=math)
  got, want = both('conv2d', (xs, ws), kw)
  helpers.report(f'{math} split-K', got, want, atol=tol, rtol=1e-5)
  for (N, H, W, Cin, k, Cout, pad) in [(1, 1, 1, 8, 1, 4, 0), (1, 1, 5, 4, 1, 8, 0), (1, 3, 3, 12, 3, 4, 1)]:
    x = rnd((N, H, W, Cin), 60 + Cin)
    w = rnd((k, k, Cin, Cout), 61, 1 / np.sqrt(k * k * Cin))
    got, want = both('conv2d', (x, w), dict(padding=((pad, pad), (pad, pad)), math=math))
    helpers.report(f'conv {math} tiny {N}x{H}x{W}x{Cin}', got, want, atol=tol, rtol=1e-5)
  # scaling both operands by powers of two scales the result exactly (bit for bit)
  xa = rnd((1, 9, 9, 32), 70)
  wa = rnd((3, 3, 32, 64), 71, 0.1)
  base = ops.conv2d(xa.to(DEV), wa.to(DEV), padding=((1, 1), (1, 1)), math=math)
  big = ops.conv2d((xa * 2.0 ** 40).to(DEV), (wa * 2.0 ** -70).to(DEV), padding=((1, 1), (1, 1)), math=math)
  assert torch.equal(big, base * 2.0 ** -30)


HALO_CASES = [
    # N, H, W, Cin, Cout, GroupNorm prologue, epilogue extras
    (2, 34, 34, 64, 64, True, False),       # stage-3 width, BN = 64
    (3, 20, 40, 32, 200, False, True),      # plain prologue, bias + residual + relu, ragged Cout
    (1, 136, 136, 16, 64, True, False),     # stage-1 width: the largest stage (402 rows)
    (2, 68, 68, 48, 128, True, True),       # stage-2 width
    (1, 16, 143, 16, 32, False, False),     # widest image the stage holds (416 rows)
    (5, 12, 12, 32, 128, True, False),      # image == stage size (130 + 2 W = 154 > 144: im2col body)
    (4, 13, 13, 32, 128, True, False),      # 169 pixels >= 156: several images per row tile
]


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
@pytest.mark.parametrize('N,H,W,Cin,Cout,gn,extras', HALO_CASES)
def test_conv_split_halo_3x3(N, H, W, Cin, Cout, gn, extras, math):
  """3x3 / stride 1 / pad 1 on the halo body (input staged once per channel tile, taps read
  shifted rows of the stage; out-of-image taps masked per lane), incl. tiles that straddle
  images, GroupNorm tables of two images, the fused output statistics."""
  tol = 2.5 * SPLIT_TOL[math]
  x = rnd((N, H, W, Cin), 500 + W) + 0.1
  w = rnd((3, 3, Cin, Cout), 501, 1 / np.sqrt(9 * Cin))
  kw = dict(padding=((1, 1), (1, 1)), math=math)
  if gn:
    gamma, beta = rnd((Cin,), 502) + 1, rnd((Cin,), 503) * 0.1
    mu, sc = oracle_ops.group_norm_stats(x, gamma, groups=min(32, Cin // 2))
    kw.update(prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta))
  if extras:
    kw.update(residual=rnd((N, H, W, Cout), 504), bias=rnd((Cout,), 505), relu=True)
  ops.USE_SPLITK = False
  try:
    got, want = both('conv2d', (x, w), kw)
    helpers.report(f'halo conv {math} {N}x{H}x{W}x{Cin}->{Cout}', got, want, atol=tol, rtol=1e-5)
    if not extras and Cout % 32 == 0:
      gkw = {k: (tuple(t.to(DEV) for t in v) if k == 'gn' else v) for k, v in kw.items()}
      y = ops.conv2d(x.to(DEV), w.to(DEV), emit_gn_stats='raw', **gkw)
      g2 = rnd((Cout,), 506) + 1
      mu_f, sc_f = ops.group_norm_stats(y, g2.to(DEV))
      mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), g2)
      helpers.report('halo fused gn mu', mu_f, mu_w, atol=1e-5, rtol=1e-5)
      helpers.report('halo fused gn sc', sc_f, sc_w, atol=1e-5, rtol=5e-5)
  finally:
    ops.USE_SPLITK = True


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
@pytest.mark.parametrize('N,H,W,Cout,affine', [(2, 64, 64, 64, True), (1, 96, 80, 32, False), (3, 32, 32, 160, True)])
def test_conv_split_rgb_root(N, H, W, Cout, affine, math):
  """The 7 x 7 / stride 2 / pad 3 root convolution of an RGB image stored with 4 floats per pixel
  (cin = 3): on the split engine a K slab is 4 consecutive pixels of one kernel row (root weight
  image, 14 slabs); the 4th float of a pixel and the 8th tap of a row meet zero weights."""
  x = torch.rand((N, H, W, 4), generator=torch.Generator().manual_seed(900 + W))
  x[..., 3] = 0.37                                  # (must not matter)
  w = rnd((7, 7, 3, Cout), 901, 1 / np.sqrt(147))
  kw = dict(stride=2, padding=((3, 3), (3, 3)), cin=3, math=math)
  if affine:
    kw.update(prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0))
  got, want = both('conv2d', (x, w), kw)
  assert got.shape == (N, H // 2, W // 2, Cout)
  helpers.report(f'rgb root {math} {N}x{H}x{W}->{Cout}', got, want, atol=2.5 * SPLIT_TOL[math], rtol=1e-5)
  exact = ops.conv2d(x.to(DEV), w.to(DEV), **{**kw, 'math': 'f32'})
  assert float((got - exact).abs().max()) <= 2.5 * SPLIT_TOL[math] * float(exact.abs().max())


@pytest.mark.parametrize('math', ['f32', 'bf16x3', 'bf16x6'])
def test_conv_emits_both_groupnorm_statistics(math):
  """emit_gn_stats='both' (the closing 1 x 1 conv of a ResNet stage): the statistics of y
  (GroupNorm -> ReLU readers) and of relu(y) (ReLU -> GroupNorm readers, the FPN levels) out of
  ONE epilogue on the split engines (a dedicated kernel variant; row tiles straddling images);
  a request the engine does not honour (f32) silently leaves the second statistic to the
  stand-alone pass."""
  N, H, W, Cin, Cout = 3, 97, 120, 64, 256      # 273 x 2 tiles of 128 x 128 (the engine's choice)
  x = rnd((N, H, W, Cin), 950)
  w = rnd((1, 1, Cin, Cout), 951, 1 / 8.0)
  res = rnd((N, H, W, Cout), 952)
  g_in = rnd((Cin,), 954) + 1
  b_in = rnd((Cin,), 955)
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV))
  kw = dict(prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in.to(DEV)), residual=res.to(DEV), math=math)
  y = ops.conv2d(xd, w.to(DEV), emit_gn_stats='both', **kw)
  y_raw = ops.conv2d(xd, w.to(DEV), emit_gn_stats='raw', **kw)
  assert torch.equal(y, y_raw)
  assert hasattr(y, '_snap_gn_partial')
  assert hasattr(y, '_snap_gn_partial_relu') == (math != 'f32')
  (mu_a, sc_a), (mu_b, sc_b) = (ops.group_norm_stats(t, g_in.new_ones(Cout).to(DEV)) for t in (y, y_raw))
  assert torch.equal(mu_a, mu_b) and torch.equal(sc_a, sc_b)      # (the first statistic: unchanged)
  gamma = rnd((Cout,), 953) + 1
  for relu_first in (False, True):
    mu_f, sc_f = ops.group_norm_stats(y, gamma.to(DEV), relu_first=relu_first)
    mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), gamma, relu_first=relu_first)
    helpers.report(f'both stats mu relu_first={relu_first}', mu_f, mu_w, atol=1e-5, rtol=1e-5)
    helpers.report(f'both stats sc relu_first={relu_first}', sc_f, sc_w, atol=1e-5, rtol=5e-5)


@pytest.mark.parametrize('shape', [(8, 34, 34, 256, 256, 3), (3, 17, 19, 512, 512, 3), (5, 9, 7, 1024, 256, 1),
                                   (2, 12, 12, 1024, 512, 1)])
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('math', ['bf16x3', 'bf16', 'fp16'])
def test_split_k_launches_emit_groupnorm_statistics(shape, relu, math):
  """Deep reductions over few rows run split-K; their reduce pass emits the GroupNorm partial sums
  (per 32-row slab: tiles straddling images, ragged last tile, residual) -- the output is bitwise
  what the launch without statistics writes, the statistics agree with the stand-alone pass.  The
  split engine and the training-precision engines (whose forward convolutions in a C3 step took a
  stand-alone statistics pass after every split-K launch before round 5)."""
  N, H, W, Cin, Cout, k = shape
  x = rnd((N, H, W, Cin), 1200 + Cin)
  w = rnd((k, k, Cin, Cout), 1201 + Cout, 1 / np.sqrt(k * k * Cin))
  res = rnd((N, H, W, Cout), 1202)
  g_in = rnd((Cin,), 1203) * 0.3 + 1
  b_in = rnd((Cin,), 1204) * 0.3
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV))
  pad = ((k // 2, k // 2), (k // 2, k // 2))
  kw = dict(padding=pad, prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in.to(DEV)), residual=res.to(DEV), math=math)
  y = ops.conv2d(xd, w.to(DEV), emit_gn_stats='relu' if relu else 'raw', **kw)
  assert getattr(y, '_snap_gn_partial', (None, 0, None))[1] == 32, 'not a split-K launch with fused statistics'
  ops.SPLITK_STATS = False
  try:
    y0 = ops.conv2d(xd, w.to(DEV), emit_gn_stats='raw', **kw)
  finally:
    ops.SPLITK_STATS = True
  assert not hasattr(y0, '_snap_gn_partial')
  assert torch.equal(y, y0)
  gamma = rnd((Cout,), 1205) * 0.3 + 1
  mu_f, sc_f = ops.group_norm_stats(y, gamma.to(DEV), relu_first=relu)
  mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), gamma, relu_first=relu)
  helpers.report('split-K stats mu', mu_f, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('split-K stats sc', sc_f, sc_w, atol=1e-5, rtol=5e-5)


def test_conv_split_accuracy_class():
  """The split engines against float64 on a deep reduction (K = 4608), next to the exact f32
  engine: 'bf16x6' must sit in the f32 engine's error class (<= 2x its rms error), 'bf16x3'
  within 2^-15 of the product scale."""
  x = rnd((2, 20, 20, 512), 80)
  w = rnd((3, 3, 512, 256), 81, 1 / np.sqrt(9 * 512))
  exact = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(),
                                     padding=1).permute(0, 2, 3, 1)
  errs = {}
  for math in ('f32', 'bf16x6', 'bf16x3'):
    y = ops.conv2d(x.to(DEV), w.to(DEV), padding=((1, 1), (1, 1)), math=math).cpu().double()
    errs[math] = float((y - exact).pow(2).mean().sqrt() / exact.pow(2).mean().sqrt())
  print('relative rms error vs float64:', errs)
  assert errs['bf16x6'] <= 2.0 * errs['f32'] + 1e-8
  assert errs['bf16x3'] <= 2.0 ** -15


def test_conv_bf16_degenerate_shapes():
  """One output pixel, one image row, K smaller than a slab, Cout = 4, an all-masked row list."""
  for (N, H, W, Cin, k, Cout, pad) in [(1, 1, 1, 8, 1, 4, 0), (1, 1, 5, 4, 1, 8, 0), (1, 3, 3, 12, 3, 4, 1)]:
    x = rnd((N, H, W, Cin), 60 + Cin)
    w = rnd((k, k, Cin, Cout), 61, 1 / np.sqrt(k * k * Cin))
    got, want = both('conv2d', (x, w), dict(padding=((pad, pad), (pad, pad)), math='bf16'))
    helpers.report(f'conv bf16 tiny {N}x{H}x{W}x{Cin}', got, want, atol=3e-5, rtol=1e-5)
  M = 70000
  x = rnd((M, 64), 62)
  w = rnd((64, 32), 63, 1 / 8.0)
  mask = torch.zeros(M, dtype=torch.bool)
  index, count = ops.compact_rows(mask.to(DEV))
  out = torch.full((M, 32), 7.0, device=DEV)
  ops.dense(x.to(DEV), w.to(DEV), rows_in=index, rows_out=index, row_count=count, out=out, math='bf16')
  assert int(count.item()) == 0 and float(out.min()) == 7.0      # nothing written


def test_conv_bf16_rows_gnstats_splitk():
  """Row-indexed launches, GroupNorm partial sums and split-K on the bf16 engine."""
  g = torch.Generator().manual_seed(50)
  M, Cin, Cout = 3000, 260, 256
  x = rnd((M, Cin), 51)
  w = rnd((257, Cout), 52, 1 / 16.0)
  bias = rnd((Cout,), 53)
  mask = torch.rand(M, generator=g) > 0.6
  want = oracle_ops.dense(x, w, bias, cin=257, relu=True)
  want_rows = want[mask]
  index, count = ops.compact_rows(mask.to(DEV))
  out = torch.zeros(M, Cout, device=DEV)
  ops.dense(x.to(DEV), w.to(DEV), bias.to(DEV), cin=257, relu=True, rows_in=index, rows_out=index,
            row_count=count, out=out, math='bf16')
  want_b = oracle_ops.conv2d(x.reshape(1, 1, M, Cin), w.reshape(1, 1, 257, Cout), cin=257, bias=bias,
                             relu=True, math='bf16').reshape(M, Cout)
  helpers.report('bf16 row-indexed dense', out[mask.to(DEV)], want_b[mask], atol=5e-5, rtol=1e-5)
  assert float(out[~mask.to(DEV)].abs().max()) == 0.0
  del want_rows
  # statistics out of the epilogue == statistics of the produced tensor
  N, H, W, C1, C2 = 2, 24, 20, 64, 128
  xi = rnd((N, H, W, C1), 54)
  wi = rnd((3, 3, C1, C2), 55, 1 / 24.0)
  gamma = rnd((C2,), 56) + 1
  ops.USE_SPLITK = False     # (a launch that splits K takes its statistics afterwards)
  try:
    y = ops.conv2d(xi.to(DEV), wi.to(DEV), padding=((1, 1), (1, 1)), emit_gn_stats='raw', math='bf16')
  finally:
    ops.USE_SPLITK = True
  assert hasattr(y, '_snap_gn_partial')
  mu_f, sc_f = ops.group_norm_stats(y, gamma.to(DEV))
  mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), gamma)
  helpers.report('bf16 fused gn mu', mu_f, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('bf16 fused gn sc', sc_f, sc_w, atol=1e-5, rtol=5e-5)
  # split-K (deep K, few tiles) against the single-pass launch
  xs = rnd((2, 8, 8, 512), 57)
  ws = rnd((3, 3, 512, 128), 58, 1 / np.sqrt(9 * 512))
  kw = dict(padding=((1, 1), (1, 1)), math='bf16')
  got, want = both('conv2d', (xs, ws), kw)
  helpers.report('bf16 split-K', got, want, atol=5e-5, rtol=1e-5)
  ops.USE_SPLITK = False
  try:
    single = ops.conv2d(xs.to(DEV), ws.to(DEV), **kw)
  finally:
    ops.USE_SPLITK = True
  helpers.report('bf16 split-K vs single pass', got, single.cpu(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('math', ['f32', 'bf16'])
def test_conv_random_shapes_fuzz(math):
  """24 seeded random (shape, stride, padding, prologue, epilogue) combinations per engine vs the
  oracle: odd sizes, Cin / Cout off the tile multiples, asymmetric padding, every fusion."""
  rng = np.random.default_rng(2025 if math == 'f32' else 2026)
  for it in range(24):
    N = int(rng.integers(1, 4))
    k = int(rng.choice([1, 1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 2]))
    H, W = int(rng.integers(k, 23)), int(rng.integers(k, 23))
    Cin = int(rng.choice([4, 8, 12, 20, 32, 36, 64, 100, 132]))
    Cs = Cin + int(rng.choice([0, 0, 4]))                       # padded row stride
    Cout = int(rng.choice([4, 8, 28, 64, 68, 132, 200]))
    pt, pb, pl, pr = (int(v) for v in rng.integers(0, k, 4))
    Ho = (H + pt + pb - k) // stride + 1
    Wo = (W + pl + pr - k) // stride + 1
    if Ho < 1 or Wo < 1:
      continue
    x = torch.zeros((N, H, W, Cs))
    x[..., :Cin] = rnd((N, H, W, Cin), 1000 + it) + 0.1
    w = rnd((k, k, Cin, Cout), 2000 + it, 1.0 / np.sqrt(k * k * Cin))
    pro = int(rng.choice([ops.PRO_NONE, ops.PRO_RELU, ops.PRO_AFFINE, ops.PRO_GN_RELU, ops.PRO_RELU_GN]))
    kw = dict(stride=stride, padding=((pt, pb), (pl, pr)), cin=Cin, prologue=pro, math=math)
    if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
      if Cin % 4:
        continue
      groups = 4 if Cin % 32 else 32
      gamma, beta = rnd((Cin,), 3000 + it) * 0.3 + 1, rnd((Cin,), 4000 + it) * 0.1
      mu, sc = oracle_ops.group_norm_stats(x[..., :Cin].contiguous(), gamma, groups=groups,
                                           relu_first=pro == ops.PRO_RELU_GN)
      kw['gn'] = (mu, sc, beta)
    if pro == ops.PRO_AFFINE:
      kw['in_affine'] = (float(rng.uniform(0.5, 2)), float(rng.uniform(-1, 1)))
    if rng.random() < 0.5:
      kw['bias'] = rnd((Cout,), 5000 + it)
    if rng.random() < 0.4:
      kw['relu'] = True
    if rng.random() < 0.4:
      kw['residual'] = rnd((N, Ho, Wo, Cout), 6000 + it)
    if rng.random() < 0.3:
      kw['row_mask'] = torch.rand((N, Ho, Wo), generator=torch.Generator().manual_seed(7000 + it)) > 0.3
    got, want = both('conv2d', (x, w), kw)
    scale = max(1.0, float(want.abs().max()))
    helpers.report(f'fuzz {math} #{it} N{N} {H}x{W} k{k} s{stride} Cin{Cin}/{Cs} Cout{Cout} pro{pro} '
                   f'{sorted(set(kw) - {"stride", "padding", "cin", "prologue", "math"})}',
                   got, want, atol=4e-5 * scale, rtol=1e-5)


def test_conv_affine_root():
  x = torch.rand((2, 20, 18, 3), generator=torch.Generator().manual_seed(3))
  w = rnd((7, 7, 3, 64), 4, 0.1)
  kw = dict(stride=2, padding=((3, 3), (3, 3)), prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0))
  got, want = both('conv2d', (x, w), kw)
  helpers.report('conv affine', got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('mode', [ops.PRO_GN_RELU, ops.PRO_RELU_GN])
@pytest.mark.parametrize('C', [64, 128, 256])
def test_gn_stats_and_fused_conv(mode, C):
  N, H, W = 3, 7, 9
  x = rnd((N, H, W, C), 5) * 2 + 0.7
  gamma = rnd((C,), 6) * 0.5 + 1
  beta = rnd((C,), 7) * 0.2
  relu_first = mode == ops.PRO_RELU_GN
  (mu_g, sc_g), (mu_w, sc_w) = both('group_norm_stats', (x, gamma), dict(relu_first=relu_first))
  helpers.report('gn mu', mu_g, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('gn sc', sc_g, sc_w, atol=1e-5, rtol=2e-5)
  y_g, y_w = both('group_norm_apply', (x, mu_w, sc_w, beta, mode))
  helpers.report('gn apply', y_g, y_w, atol=1e-5, rtol=1e-5)
  w = rnd((3, 3, C, 64), 8, 1 / np.sqrt(9 * C))
  kw = dict(padding=((1, 1), (1, 1)), prologue=mode, gn=(mu_w, sc_w, beta))
  got, want = both('conv2d', (x, w), kw)
  helpers.report('conv gn-fused', got, want, atol=3e-5, rtol=1e-5)


def test_gn_stats_wide_and_narrow():
  for C, HW in [(32, (5, 5)), (512, (4, 6)), (1024, (3, 3)), (2048, (2, 3))]:
    x = rnd((2, *HW, C), 9) + 0.3
    gamma = torch.ones(C)
    (mu_g, sc_g), (mu_w, sc_w) = both('group_norm_stats', (x, gamma))
    helpers.report(f'gn mu C={C}', mu_g, mu_w, atol=1e-5, rtol=1e-5)
    helpers.report(f'gn sc C={C}', sc_g, sc_w, atol=1e-5, rtol=2e-5)


def test_gn_stats_many_slabs():
  x = rnd((2, 40, 37, 64), 10) + 1.0  # 1480 pixels -> 3 slabs
  gamma = rnd((64,), 11) + 1
  (mu_g, sc_g), (mu_w, sc_w) = both('group_norm_stats', (x, gamma))
  helpers.report('gn mu slabs', mu_g, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('gn sc slabs', sc_g, sc_w, atol=1e-5, rtol=2e-5)


def test_conv_epilogues():
  N, H, W, Cin, Cout = 2, 8, 6, 64, 128
  x = rnd((N, H, W, Cin), 12)
  w = rnd((1, 1, Cin, Cout), 13, 1 / np.sqrt(Cin))
  bias = rnd((Cout,), 14)
  res = rnd((N, H, W, Cout), 15)
  prev = rnd((N, H // 2, W // 2, Cout), 16)
  mask = torch.rand((N, H, W), generator=torch.Generator().manual_seed(17)) > 0.4
  got, want = both('conv2d', (x, w), dict(bias=bias, relu=True))
  helpers.report('bias+relu', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('conv2d', (x, w), dict(residual=res))
  helpers.report('residual', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('conv2d', (x, w), dict(up_prev=prev))
  helpers.report('upsample-add', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('conv2d', (x, w), dict(bias=bias, row_mask=mask))
  helpers.report('rowmask', got, want, atol=2e-5, rtol=1e-5)


def test_dense_padded_k257():
  """Fusion-MLP shape: K = 257 inside a 260-float row stride."""
  M, K, Ks, Co = 300, 257, 260, 256
  x = torch.zeros(M, Ks)
  x[:, :K] = rnd((M, K), 18)
  x[:, K:] = 123.0  # garbage in the pad must be ignored
  w = rnd((K, Co), 19, 1 / np.sqrt(K))
  b = rnd((Co,), 20)
  got, want = both('dense', (x, w, b), dict(cin=K, relu=True))
  helpers.report('dense k257', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('dense', (x[:, :128].contiguous(), w[:128], b), dict(prologue=ops.PRO_RELU))
  helpers.report('dense relu-in', got, want, atol=2e-5, rtol=1e-5)


def test_weight_standardize_and_maxpool():
  for shape in [(7, 7, 3, 64), (3, 3, 64, 64), (1, 1, 256, 128), (1, 1, 64, 40)]:
    w = rnd(shape, 21) * 0.3 + 0.05
    got, want = both('weight_standardize', (w,))
    helpers.report(f'wstd {shape}', got, want, atol=2e-5, rtol=1e-5)
  for shape in [(2, 9, 11, 64), (1, 16, 16, 32)]:
    x = rnd(shape, 22)
    got, want = both('max_pool_3x3s2', (x,))
    helpers.report(f'maxpool {shape}', got, want, atol=0, rtol=0)


# ----------------------------------------------------------------------------
# lift
# ----------------------------------------------------------------------------
def _lift_scene(B, V, h, w, fd, nb, N, seed, k_radial=0.02):
  from snap_amd.data import synthetic
  from snap_amd.utils import grids
  g = grids.Grid3D((16, 16, 8), 0.4)
  batch = synthetic.make_batch(B, g, V, (h * 4, w * 4), seed=seed, with_aerial=False,
                               with_gt=False, k_radial=k_radial)
  scene = batch['map']
  cam = scene['camera'].scale(torch.tensor([0.25, 0.25]))
  rng = np.random.default_rng(seed)
  pts = np.stack([rng.uniform(-1, 7.4, (B, N)), rng.uniform(-1, 7.4, (B, N)),
                  rng.uniform(-1.5, 4.0, (B, N))], -1).astype(np.float32)
  f = rnd((B, V, h, w, fd + nb), seed + 1)
  return f, cam.packed(), scene['T_view2scene'].packed(), torch.tensor(pts)


@pytest.mark.parametrize('K,V', [(0, 1), (0, 3), (2, 3), (4, 6), (4, 20), (8, 9)])
def test_lift_pool(K, V):
  fd, nb = 32, 8
  f, cam, Rt, pts = _lift_scene(2, V, 12, 16, fd, nb, 4000, seed=30 + V)
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0))
  (pg, vg), (pw, vw) = both('lift_pool', (f, cam, Rt, pts), kw)
  assert vw.float().mean() > 0.05, 'test scene has (almost) no visible voxels'
  keep = _lift_validity_equal_up_to_borders(f'lift K{K} V{V}', vg, vw, cam, Rt, pts, kw)
  helpers.report('lift pooled', pg.cpu()[keep], pw[keep], atol=2e-4, rtol=1e-4)


def _lift_validity_equal_up_to_borders(name, vg, vw, cam, Rt, pts, kw):
  """Voxel validity must EQUAL the oracle's; a differing voxel is accepted only if a float64
  re-projection puts it on a visibility boundary (image edge, near plane, FoV limit,
  max_view_distance) of some view -- see helpers.assert_validity_mismatches_on_borders.
  Returns the mask of voxels whose validity agrees."""
  scene = {'camera': oracle_ops.unpack_cameras(cam, True), 'T_view2scene': oracle_ops.unpack_transforms(Rt)}
  helpers.assert_validity_mismatches_on_borders(
      name, vg, vw.numpy() if hasattr(vw, 'numpy') else vw, scene, pts.numpy(), (1.0, 1.0),
      max_view_distance=kw.get('max_view_distance'))
  return ~(vg.cpu() != vw)


@pytest.mark.parametrize('weighted,use_var,minmax', [(False, True, False), (False, True, True),
                                                     (True, False, False), (True, True, True),
                                                     (False, False, True)])
@pytest.mark.parametrize('K,V', [(0, 3), (2, 5)])
def test_lift_pool_fusion_options(K, V, weighted, use_var, minmax):
  """scores = None (do_weighted_fusion=False), fusion_use_variance=False, fusion_add_minmax=True
  (streetview_encoder.py:141-178): mean | var? | max, min? | score_max?."""
  fd, nb = 32, 8
  f, cam, Rt, pts = _lift_scene(2, V, 12, 16, fd, nb if weighted else 0, 3000, seed=70 + V)
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0),
            weighted=weighted, use_variance=use_var, add_minmax=minmax)
  (pg, vg), (pw, vw) = both('lift_pool', (f, cam, Rt, pts), kw)
  assert pg.shape[-1] == ops.pooled_stride(fd, weighted, use_var, minmax) == pw.shape[-1]
  keep = _lift_validity_equal_up_to_borders('lift options', vg, vw, cam, Rt, pts, kw)
  helpers.report(f'lift pooled w{int(weighted)} v{int(use_var)} m{int(minmax)}', pg.cpu()[keep], pw[keep],
                 atol=2e-4, rtol=1e-4)


def test_lift_and_pose_score_random_configs_fuzz():
  """12 seeded random lift configurations (views, top-K, feature / bin widths, map sizes, radial
  distortion, depth ranges, max view distance, N off the 32-voxel batches) and 12 pose-scoring
  configurations (plane sizes on both kernels, P, Nq, masks) vs the oracle."""
  rng = np.random.default_rng(4040)
  for it in range(12):
    V = int(rng.integers(1, 9))
    K = 0 if V == 1 or rng.random() < 0.4 else int(rng.integers(1, min(V, 8)))
    if K >= V:
      K = 0
    fd = int(rng.choice([8, 32, 64, 128]))
    nb = int(rng.choice([4, 8, 32]))
    h, w = int(rng.integers(6, 20)), int(rng.integers(6, 20))
    N = int(rng.integers(33, 3000))
    f, cam, Rt, pts = _lift_scene(int(rng.integers(1, 3)), V, h, w, fd, nb, N, seed=500 + it,
                                  k_radial=float(rng.choice([0.0, 0.02, 0.08])))
    dmin = float(rng.choice([0.5, 1.0]))
    kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(dmin, dmin * float(rng.choice([8, 32]))))
    if K and rng.random() < 0.4:
      kw['max_view_distance'] = float(rng.uniform(3, 8))
    (pg, vg), (pw, vw) = both('lift_pool', (f, cam, Rt, pts), kw)
    keep = _lift_validity_equal_up_to_borders(f'lift fuzz #{it}', vg, vw, cam, Rt, pts, kw)
    helpers.report(f'lift fuzz #{it} V{V} K{K} fd{fd} nb{nb} {h}x{w} N{N}', pg.cpu()[keep], pw[keep],
                   atol=2e-4, rtol=1e-4)
  for it in range(12):
    X, Y = int(rng.integers(8, 200)), int(rng.integers(8, 200))
    B, Nq, P = int(rng.integers(1, 3)), int(rng.integers(1, 90)), int(rng.integers(1, 900))
    cell = float(rng.choice([0.2, 0.25, 0.5]))
    sim = torch.tensor(rng.random((B, Nq, X, Y), dtype=np.float32))
    poses = np.stack([rng.uniform(-np.pi, np.pi, (B, P)),
                      rng.uniform(-0.3 * X * cell, 1.3 * X * cell, (B, P)),
                      rng.uniform(-0.3 * Y * cell, 1.3 * Y * cell, (B, P))], -1).astype(np.float32)
    q_xy = torch.tensor(rng.uniform(-3, 3, (B, Nq, 2)).astype(np.float32))
    valid_q = torch.tensor(rng.random((B, Nq)) > 0.2)
    map_valid = torch.tensor(rng.random((B, X, Y)) > 0.1) if rng.random() < 0.7 else None
    oob = bool(rng.random() < 0.5) and map_valid is not None     # (masking needs the map validity)
    got, want = both('pose_score', (sim, torch.tensor(poses), q_xy, valid_q, map_valid, cell), dict(mask_oob=oob))
    helpers.report(f'pose_score fuzz #{it} {X}x{Y} B{B} Nq{Nq} P{P} oob{oob}', got, want, atol=2e-4, rtol=1e-5)


def test_lift_pool_full_width_and_maxdist():
  fd, nb = 128, 32
  f, cam, Rt, pts = _lift_scene(1, 5, 10, 10, fd, nb, 3000, seed=41)
  kw = dict(K=4, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 32.0),
            max_view_distance=6.0)
  (pg, vg), (pw, vw) = both('lift_pool', (f, cam, Rt, pts), kw)
  mism = (vg.cpu() != vw)
  assert mism.float().mean() < 2e-3
  helpers.report('lift pooled fd128', pg.cpu()[~mism], pw[~mism], atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('X,Y,Z,K,V', [(11, 13, 7, 0, 3), (16, 8, 60, 0, 4), (9, 24, 60, 2, 5), (8, 8, 3, 0, 1),
                                       (301, 1, 60, 0, 1), (70, 3, 7, 0, 3), (33, 2, 9, 2, 4)])
def test_lift_bev_tiled_traversal_is_a_pure_reordering(X, Y, Z, K, V):
  """grid_yz: the 8 x 8-column-block traversal (one block per XCD at a time) writes the same
  rows, bit for bit, as the linear voxel order -- grids that are not multiples of the block,
  blocks whose voxel count is not a multiple of a workgroup's 256."""
  fd, nb = 128, 32
  f, cam, Rt, pts = _lift_scene(2, V, 12, 16, fd, nb, X * Y * Z, seed=300 + X)
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0))
  args = [t.to(DEV) for t in (f, cam, Rt, pts)]
  p0, v0 = ops.lift_pool(*args, **kw)
  p1, v1 = ops.lift_pool(*args, grid_yz=(Y, Z), **kw)
  assert v0.float().mean() > 0.05
  assert torch.equal(v0, v1) and torch.equal(p0, p1)
  # valid_rows_only: the rows of voxels some view sees are the same bits; the others are not written
  p2, v2 = ops.lift_pool(*args, grid_yz=(Y, Z), valid_rows_only=True, **kw)
  assert torch.equal(v2, v0) and torch.equal(p2[v0], p0[v0])
  # out_split: the same rows as [slab][hi | lo][16] bf16 -- hi + lo reproduces the split of the f32 row
  if K == 0 or K <= 4:
    p3, v3 = ops.lift_pool(*args, grid_yz=(Y, Z), out_split=True, **kw)
    ks = (2 * fd + 1 + 15) // 16
    parts = p3.view(torch.int32).view(torch.int16).view(torch.bfloat16).reshape(*p3.shape[:2], ks, 2, 16)
    ref = torch.zeros(*p0.shape[:2], ks * 16, device=p0.device)
    ref[..., :2 * fd + 1] = p0[..., :2 * fd + 1]
    hi = ref.bfloat16()
    lo = (ref - hi.float()).bfloat16()
    assert torch.equal(v3, v0)
    assert torch.equal(parts[..., 0, :].reshape(*ref.shape), hi) and torch.equal(parts[..., 1, :].reshape(*ref.shape), lo)
    # class_rows: valid[] becomes 0 / 1 (one visible observation) / 2 (several); a class-1 row has
    # an all-zero variance and does not write those slabs -- everything else is the same bits
    p4, v4, c4 = ops.lift_pool(*args, grid_yz=(Y, Z), out_split=True, class_rows=True, **kw)
    assert torch.equal(v4, v0) and torch.equal(c4 != 0, v0) and int(c4.max()) <= 2
    one, many = c4 == 1, c4 == 2
    assert int(one.sum()) > 0 and (V == 1 or int(many.sum()) > 0)
    p4s, p3s = (t.reshape(*t.shape[:2], ks, 16) for t in (p4, p3))
    nv = fd // 16
    assert torch.equal(p4s[many], p3s[many])
    assert torch.equal(p4s[one][:, :nv], p3s[one][:, :nv]) and torch.equal(p4s[one][:, 2 * nv:], p3s[one][:, 2 * nv:])
    assert float(p3s[one][:, nv:2 * nv].abs().max()) == 0.0      # (the slabs that are skipped ARE zero)


@pytest.mark.parametrize('shape,ph,pw,pc', [((5, 37, 29, 3), 27, 3, 1), ((2, 3, 16, 16, 3), 16, 16, 1),
                                            ((3, 20, 31, 5), 12, 1, 0), ((1, 8, 8, 4), 0, 0, 0)])
def test_pad_image_is_torch_pad(shape, ph, pw, pc):
  """pad_to_multiple (+ zero channels) in one pass == torch.nn.functional.pad, bit for bit."""
  x = rnd(shape, 1200 + ph).to(DEV)
  want = torch.nn.functional.pad(x, (0, pc, 0, pw, 0, ph))
  got = ops.pad_image(x, ph, pw, pc)
  assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize('batched', [False, True])
def test_voxel_points_is_the_broadcast(batched):
  B, X, Y, Z = 3, 7, 5, 11
  xy = rnd((B, X, Y, 2) if batched else (X, Y, 2), 1300).to(DEV)
  z = rnd((B, Z), 1301).to(DEV)
  want = torch.empty(B, X, Y, Z, 3, device=DEV)
  want[..., :2] = (xy if batched else xy[None].expand(B, X, Y, 2))[:, :, :, None, :]
  want[..., 2] = z[:, None, None, :]
  assert torch.equal(ops.voxel_points(xy, z), want)


def test_project_points():
  f, cam, Rt, pts = _lift_scene(2, 4, 12, 16, 8, 4, 5000, seed=50, k_radial=0.05)
  (p2g, vig, dg), (p2w, viw, dw) = both('project_points', (cam, Rt, pts, True))
  mism = vig.cpu() != viw
  assert mism.float().mean() < 1e-3
  helpers.report('depth', dg, dw, atol=1e-5, rtol=1e-5)
  vis = viw & ~mism
  helpers.report('p2d (visible)', p2g.cpu()[vis], p2w[vis], atol=2e-4, rtol=1e-5)


# ----------------------------------------------------------------------------
# BEV
# ----------------------------------------------------------------------------
@pytest.mark.parametrize('Z,D', [(12, 128), (60, 128), (64, 32), (1, 256), (70, 128)])
@pytest.mark.parametrize('pooling', ['max', 'sum', 'mean'])
def test_vertical_pool(pooling, Z, D):
  """Z <= 64: the wave-per-column kernel (alternate levels per half-wave); Z = 70: the
  half-wave-per-column kernel."""
  vol = rnd((2, 9, 7, Z, D), 60)
  valid = torch.rand((2, 9, 7, Z), generator=torch.Generator().manual_seed(61)) > 0.6
  valid[0, 0] = False  # fully invalid columns
  valid[1, 1] = True   # fully valid columns
  (pg, vg), (pw, vw) = both('vertical_pool', (vol, valid, pooling))
  helpers.report('vpool valid', vg, vw, 0)
  helpers.report('vpool plane', pg, pw, atol=1e-5, rtol=1e-6)


@pytest.mark.parametrize('math', ['f32', 'bf16x3', 'bf16'])
def test_encoder_ops_propagate_nan_like_jnp(math):
  """jnp.maximum / nn.relu / nn.max_pool / the GroupNorm statistics propagate a NaN; so do the fused
  prologues (GroupNorm -> ReLU, ReLU -> GroupNorm, ReLU) and epilogues of every conv engine, the
  statistics kernels and the max-pool: a NaN pixel poisons its image's normalised activations (and
  nothing of the other image), as in the reference -- where it ends in the trainer's non-finite step
  skip (trainer.py:260-277).  Against the numpy oracle, NaN positions included."""
  N, H, W, Cin, Cout = 2, 9, 11, 64, 128
  x = rnd((N, H, W, Cin), 501) + 0.2
  x[1, 4, 5, 7] = float('nan')
  w = rnd((3, 3, Cin, Cout), 502, 1 / np.sqrt(9 * Cin))
  gamma, beta = rnd((Cin,), 503) + 1, rnd((Cin,), 504) * 0.1
  for pro, relu_first in ((ops.PRO_GN_RELU, False), (ops.PRO_RELU_GN, True)):
    (mg, sg), (mw, sw) = both('group_norm_stats', (x, gamma), dict(relu_first=relu_first))
    helpers.report('gn mean (NaN-aware)', mg, mw, atol=1e-5, rtol=1e-5)
    helpers.report('gn scale (NaN-aware)', sg, sw, atol=1e-4, rtol=1e-4)
    assert bool(torch.isnan(mg[1]).any()) and not bool(torch.isnan(mg[0]).any())
    kw = dict(padding=((1, 1), (1, 1)), prologue=pro, gn=(mw, sw, beta), relu=True, math=math)
    got, want = both('conv2d', (x, w), kw)
    helpers.report(f'conv {math} pro {pro} (NaN-aware)', got, want, atol=2e-4 if math == 'bf16' else 5e-5, rtol=1e-4)
    assert bool(torch.isnan(got[1]).all()) and not bool(torch.isnan(got[0]).any())
  got, want = both('conv2d', (x, w), dict(padding=((1, 1), (1, 1)), prologue=ops.PRO_RELU, relu=True, math=math))
  helpers.report(f'conv {math} relu prologue (NaN-aware)', got, want, atol=2e-4 if math == 'bf16' else 5e-5, rtol=1e-4)
  assert int(torch.isnan(got).any(-1).sum()) == 9               # the 3 x 3 footprint of the one NaN pixel
  got, want = both('max_pool_3x3s2', (x,))
  helpers.report('max pool (NaN-aware)', got, want, atol=0)
  assert bool(torch.isnan(got).any())


@pytest.mark.parametrize('Z', [60, 70])
def test_pooling_propagates_nan_of_observed_voxels(Z):
  """bev_mapper.py:63-78: ``jnp.max(features, where=valid_any_or_all, initial=-inf)`` -- a NaN of an
  OBSERVED voxel makes that channel of the column NaN; a NaN of a masked voxel (or anywhere in a
  column without observations) never reaches the plane.  Both pooling paths against oracle/bev.py:
  the stand-alone kernels (wave-per-column Z <= 64, half-wave Z = 70) and the fused MLP + max kernel
  (NaN planted in the MLP's INPUT rows: it has to survive both layers, the hidden ReLU included,
  and the integer atomic max)."""
  D, ncols = 128, 41
  g = torch.Generator().manual_seed(71 + Z)
  vol = torch.randn(ncols, Z, D, generator=g)
  valid = torch.rand(ncols, Z, generator=g) > 0.5
  valid[0] = False
  valid[1] = True
  valid[2] = False; valid[2, 3] = True
  vol[1, 5, 7] = float('nan')                      # observed voxel, one channel
  vol[2, 3, :] = float('nan')                      # the single observed level of its column
  vol[0, 4, 9] = float('nan')                      # column without observations
  zi = int((~valid[5]).nonzero()[0]); vol[5, zi, 11] = float('nan')      # masked voxel of an observed column
  zv = int(valid[6].nonzero()[0]); vol[6, zv, 0] = -float('nan')          # negative-signed NaN, observed
  vol[7, int(valid[7].nonzero()[-1]), 1] = float('inf')
  (pg, vg), (pw, vw) = both('vertical_pool', (vol, valid, 'max'))
  helpers.report('vpool valid', vg, vw, 0)
  helpers.report('vpool plane (NaN-aware)', pg, pw, atol=0)
  pg = pg.cpu()
  assert bool(torch.isnan(pg[1, 7])) and int(torch.isnan(pg[1]).sum()) == 1
  assert bool(torch.isnan(pg[2]).all()) and not bool(torch.isnan(pg[0]).any())
  assert not bool(torch.isnan(pg[5]).any()) and bool(torch.isnan(pg[6, 0])) and float(pg[7, 1]) == float('inf')
  # sum / mean propagate by arithmetic; checked against the oracle too
  for pooling in ('sum', 'mean'):
    (pg2, _), (pw2, _) = both('vertical_pool', (vol, valid, pooling))
    helpers.report(f'vpool {pooling} (NaN-aware)', pg2, pw2, atol=1e-4, rtol=1e-5)
  # modality fusion: the same masked max over stacked planes (bev_mapper.py:225-252)
  pa, pb = torch.randn(50, D, generator=g), torch.randn(50, D, generator=g)
  va, vb = torch.rand(50, generator=g) > 0.3, torch.rand(50, generator=g) > 0.3
  va[0], vb[0] = True, True; pa[0, 3] = float('nan')          # valid modality: propagates
  va[1], vb[1] = False, True; pa[1, 4] = float('nan')         # masked modality: ignored
  got, want = both('plane_fuse_match', ([pa, pb], [va, vb]), dict(pooling='max'))
  helpers.report('fused plane (NaN-aware)', got[0], want[0], atol=0)
  assert bool(torch.isnan(got[0][0, 3])) and not bool(torch.isnan(got[0][1]).any())
  # the fused MLP + max kernel, NaN planted in the input rows
  cin, stride, H = 257, 260, 256
  M = ncols * Z
  x = torch.randn(M, stride, generator=g)
  x[:, cin:] = 0
  mask = valid.reshape(-1).clone()
  w0 = torch.randn(cin, H, generator=g) / cin ** 0.5
  b0 = torch.randn(H, generator=g) * 0.1
  w1 = torch.randn(H, D, generator=g) / H ** 0.5
  b1 = torch.randn(D, generator=g) * 0.1
  r_obs = 1 * Z + 5                                 # observed row (column 1 is fully observed)
  r_msk = 5 * Z + zi                                # masked row of an observed column
  x[r_obs, 100] = float('nan')
  x[r_msk, 3] = float('nan')
  x[0 * Z + 2, 8] = float('nan')                    # column without observations
  (pg, vg), (pw, vw) = both('mlp2_pool_max', (x, mask, w0, b0, w1, b1), dict(cin=cin, Z=Z))
  helpers.report('mlp2_pool valid', vg, vw, 0)
  helpers.report('mlp2_pool plane (NaN-aware)', pg, pw, atol=1e-4, rtol=1e-4)
  pg = pg.cpu()
  assert bool(torch.isnan(pg[1]).all())             # NaN input -> every hidden unit -> every channel
  assert int(torch.isnan(pg).any(-1).sum()) == 1    # ... and no other column


MLP_POOL_CASES = [
    # cin, stride, H, D, Z, columns, relu_in
    (257, 260, 256, 128, 60, 37, False),     # the reference's fusion MLP / 12 m column at 0.2 m
    (257, 260, 256, 128, 60, 300, True),     # several row tiles, columns straddling them
    (257, 260, 256, 128, 60, 301, False),    # ... and the opt-in 256-row kernel of the pre-split rows over 70 tiles
    (65, 68, 64, 32, 12, 50, False),         # the tiny test models (feature_dim 32)
    (129, 132, 128, 64, 7, 91, False),
    (33, 36, 96, 20, 64, 9, True),           # H not a multiple of 64, D < 32
]


@pytest.mark.parametrize('cin,stride,H,D,Z,ncols,relu_in', MLP_POOL_CASES)
def test_mlp2_pool_max(cin, stride, H, D, Z, ncols, relu_in):
  """Fused fusion MLP + vertical max pooling vs the oracle, and BIT-EXACT against the unfused
  chain on the same engine (conv_split bf16x3 x 2 -> fill_masked_rows -> vertical_pool)."""
  M = ncols * Z
  g = torch.Generator().manual_seed(900 + cin + ncols)
  x = torch.randn(M, stride, generator=g)
  x[:, cin:] = 0
  mask = torch.rand(M, generator=g) > 0.4
  mask.view(ncols, Z)[0] = False               # an unobserved column
  mask.view(ncols, Z)[1] = True                # a fully observed one
  mask.view(ncols, Z)[2] = False
  mask.view(ncols, Z)[2, Z // 2] = True        # a single level
  w0 = torch.randn(cin, H, generator=g) / cin ** 0.5
  b0 = torch.randn(H, generator=g) * 0.1
  w1 = torch.randn(H, D, generator=g) / H ** 0.5
  b1 = torch.randn(D, generator=g) * 0.1
  kw = dict(cin=cin, Z=Z, relu_in=relu_in)
  (pg, vg), (pw, vw) = both('mlp2_pool_max', (x, mask, w0, b0, w1, b1), kw)
  helpers.report('mlp2_pool valid', vg, vw, 0)
  helpers.report('mlp2_pool plane', pg, pw, atol=1e-4, rtol=1e-4)   # bf16x3 class (~1e-5 observed)
  assert float(pg[0].abs().max()) == 0.0
  # the unfused chain, same arithmetic
  xd, md = x.to(DEV), mask.to(DEV)
  pro = ops.PRO_RELU if relu_in else ops.PRO_NONE
  index, count = ops.compact_rows(md)          # (what layers.MLP._masked_rows launches)
  hid = ops.dense(xd, w0.to(DEV), b0.to(DEV), cin=cin, prologue=pro, relu=True, math='bf16x3',
                  rows_in=index, row_count=count)
  vol = ops.dense(hid, w1.to(DEV), b1.to(DEV), math='bf16x3', rows_out=index, row_count=count)
  ops.fill_masked_rows_(vol, md)
  plane, pvalid = ops.vertical_pool(vol.reshape(ncols, Z, D), md.reshape(ncols, Z), 'max')
  assert torch.equal(pvalid, vg)
  assert torch.equal(plane, pg), float((plane - pg).abs().max())
  # a non-finite value that reaches the pooling must look the same in both configurations: the
  # maximum over a column with a NaN voxel is NaN (jnp.max semantics, bev_mapper.py:63-78).  A NaN
  # bias of the last layer makes that channel NaN in every observed column, on both paths.
  b1n = b1.clone()
  b1n[D // 2] = float('nan')
  pn, vn = ops.mlp2_pool_max(xd, md, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1n.to(DEV), **kw)
  voln = ops.dense(hid, w1.to(DEV), b1n.to(DEV), math='bf16x3', rows_out=index, row_count=count)
  ops.fill_masked_rows_(voln, md)
  plane_n, _ = ops.vertical_pool(voln.reshape(ncols, Z, D), md.reshape(ncols, Z), 'max')
  assert torch.equal(torch.isnan(pn), torch.isnan(plane_n))
  assert bool(torch.isnan(pn[vg][:, D // 2]).all()) and not bool(torch.isnan(pn[~vg]).any())
  bad = torch.nan_to_num(pn) != torch.nan_to_num(plane_n)
  assert not bool(bad.any()), (int(bad.sum()), bad.nonzero()[:4].tolist(), pn[bad][:4].tolist(), plane_n[bad][:4].tolist())
  assert torch.equal(vn, vg)
  if not relu_in:
    # the rows handed over pre-split ([slab][hi | lo][16] bf16, the lift's out_split format): same bits
    ks = (cin + 15) // 16
    xp = torch.zeros(M, ks * 16)
    xp[:, :cin] = x[:, :cin]
    hi = xp.bfloat16()
    lo = (xp - hi.float()).bfloat16()
    rows = torch.stack([hi.view(M, ks, 16), lo.view(M, ks, 16)], dim=2).contiguous()      # [M, ks, 2, 16]
    xs = rows.view(torch.int16).view(M, ks * 32).view(torch.float32).contiguous()           # f32 container
    assert xs.shape == (M, ks * 16)
    ps, vs = ops.mlp2_pool_max(xs.to(DEV), md, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1.to(DEV),
                               cin=cin, Z=Z, x_split=True)
    assert torch.equal(vs, vg) and torch.equal(ps, pg), float((ps - pg).abs().max())
    # (the line above ran the three-stage ring) the two-stage loop, and 256-row tiles with 64 rows per wave
    # (H = 256; elsewhere the default kernel again)
    for switch in ('MLP_POOL_NO_RING', 'MLP_POOL_WIDE'):
      setattr(ops, switch, True)
      try:
        ps, vs = ops.mlp2_pool_max(xs.to(DEV), md, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1.to(DEV),
                                   cin=cin, Z=Z, x_split=True)
      finally:
        setattr(ops, switch, False)
      assert torch.equal(vs, vg) and torch.equal(ps, pg), (switch, float((ps - pg).abs().max()))
    # two row classes into one plane: class-1 rows are zero over a slab range and hold GARBAGE
    # there (never read, never multiplied); the plane is the one-list plane of the zeroed rows
    zlo, zn = (ks - 1) // 2, (ks - 1) // 2
    cls = md.to(torch.uint8) * (1 + (torch.rand(M, generator=g) > 0.5).to(torch.uint8)).to(DEV)
    xz = xp.clone()
    xz[(cls == 1).cpu(), 16 * zlo:16 * (zlo + zn)] = 0
    want, vwant = ops.mlp2_pool_max(xz.to(DEV), md, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1.to(DEV),
                                    cin=cin, Z=Z)
    xg = xz.clone()
    xg[(cls == 1).cpu(), 16 * zlo:16 * (zlo + zn)] = float('nan')
    got, vgot = ops.mlp2_pool_max(xg.to(DEV), cls, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1.to(DEV),
                                  cin=cin, Z=Z, zero_slabs=(zlo, zn))
    assert torch.equal(vgot, vwant) and torch.equal(got, want), float((got - want).abs().max())
    hi = xg.bfloat16()
    lo = (xg - hi.float()).bfloat16()
    rows = torch.stack([hi.view(M, ks, 16), lo.view(M, ks, 16)], dim=2).contiguous()
    xs = rows.view(torch.int16).view(M, ks * 32).view(torch.float32).contiguous()
    got, vgot = ops.mlp2_pool_max(xs.to(DEV), cls, w0.to(DEV), b0.to(DEV), w1.to(DEV), b1.to(DEV),
                                  cin=cin, Z=Z, x_split=True, zero_slabs=(zlo, zn))
    assert torch.equal(vgot, vwant) and torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize('X,Y,Z,K,V,fd,nb,H', [(16, 8, 60, 0, 4, 128, 32, 256), (9, 24, 60, 2, 5, 128, 32, 256),
                                               (8, 8, 3, 0, 1, 128, 32, 256), (11, 13, 7, 0, 3, 32, 8, 64),
                                               (40, 24, 60, 0, 4, 128, 32, 256), (301, 1, 60, 0, 1, 128, 32, 256)])
def test_lift_inside_the_consumer(X, Y, Z, K, V, fd, nb, H):
  """Tap records (snap_lift_pool_records_f32) + the gather inside the fused MLP / pool kernel
  (snap_mlp2_pool_max_gather_f32): voxels with ONE visible observation never get a `pooled` row, and the
  plane is, bit for bit, the plane of the rows-through-memory path; the records themselves are checked
  against the numpy restatement of the format (tests/oracle_ops.py)."""
  import oracle_ops
  f, cam, Rt, pts = _lift_scene(2, V, 12, 16, fd, nb, X * Y * Z, seed=500 + X)
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0), grid_yz=(Y, Z),
            valid_rows_only=True, out_split=True, class_rows=True)
  args = [t.to(DEV) for t in (f, cam, Rt, pts)]
  p4, v4, c4 = ops.lift_pool(*args, **kw)
  p5, v5, c5, r5 = ops.lift_pool(*args, tap_records=True, **kw)
  one, many = c4 == 1, c4 == 2
  assert torch.equal(c5, c4) and torch.equal(v5, v4) and int(one.sum()) > 0 and (V == 1 or int(many.sum()) > 0)
  assert torch.equal(p5[many], p4[many])
  # the records reproduce the rows the other path wrote (mean | 0 | score), up to the hi + lo split
  ks = (2 * fd + 1 + 15) // 16
  rows = torch.from_numpy(oracle_ops._rows_from_tap_records(f, r5.cpu(), fd, one.cpu().numpy())).reshape(*c4.shape, -1)
  parts = p4.cpu().view(torch.int32).view(torch.int16).view(torch.bfloat16).reshape(*p4.shape[:2], ks, 2, 16).float()
  stored = (parts[..., 0, :] + parts[..., 1, :]).reshape(*p4.shape[:2], ks * 16)
  oc = one.cpu()
  assert float((rows[oc][:, :fd] - stored[oc][:, :fd]).abs().max()) <= 2e-5 * float(stored[oc][:, :fd].abs().max())
  assert float((rows[oc][:, 2 * fd] - stored[oc][:, 2 * fd]).abs().max()) <= 2e-5
  # the plane: records + in-kernel gather == classed rows through memory
  g = torch.Generator().manual_seed(X)
  cin, D = 2 * fd + 1, fd
  w0 = (torch.randn((cin, H), generator=g) / cin ** 0.5).to(DEV)
  b0 = (torch.randn(H, generator=g) * 0.1).to(DEV)
  w1 = (torch.randn((H, D), generator=g) / H ** 0.5).to(DEV)
  b1 = (torch.randn(D, generator=g) * 0.1).to(DEV)
  nv = fd // 16
  flat = lambda t: t.reshape(-1, t.shape[-1])
  want, vwant = ops.mlp2_pool_max(flat(p4), c4.reshape(-1), w0, b0, w1, b1, cin=cin, Z=Z, x_split=True,
                                  zero_slabs=(nv, nv))
  for group in (0, 1, 8):
    ops.MLP_GATHER_XCD_GROUP, keep = group, ops.MLP_GATHER_XCD_GROUP
    try:
      got, vgot = ops.mlp2_pool_max(flat(p5), c5.reshape(-1), w0, b0, w1, b1, cin=cin, Z=Z, x_split=True,
                                    zero_slabs=(nv, nv), gather=(args[0], r5))
    finally:
      ops.MLP_GATHER_XCD_GROUP = keep
    assert torch.equal(vgot, vwant) and torch.equal(got, want), (group, float((got - want).abs().max()))


@pytest.mark.parametrize('nplanes', [1, 2])
def test_plane_fuse_match(nplanes):
  D, Dm = 128, 32
  planes = [rnd((2, 11, 13, D), 70 + i) for i in range(nplanes)]
  valids = [torch.rand((2, 11, 13), generator=torch.Generator().manual_seed(80 + i)) > 0.3
            for i in range(nplanes)]
  if nplanes == 2:
    valids[1] = None
  planes[0] = planes[0] * valids[0][..., None]
  Wm = rnd((D, Dm), 75, 0.1)
  bm = rnd((Dm,), 76, 0.01)
  (fg, vg, mg), (fw, vw, mw) = both('plane_fuse_match', (planes, valids, 'max', Wm, bm))
  helpers.report('fuse valid', vg, vw, 0)
  helpers.report('fused', fg, fw, atol=1e-6)
  helpers.report('matching', mg, mw, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize('pooling,Dm,normalize', [('max', 32, True), ('mean', 16, True), ('sum', 32, False)])
def test_plane_fuse_match_persistent_kernel_for_128_channels(pooling, Dm, normalize):
  """From 8192 cells of 128 channels the fuse + matching-head launch takes the kernel that keeps the head's
  columns in registers (plane_fuse_match_d128_kernel): against the oracle, and bit for bit what the
  per-cell kernel gives on the same cells in launches below the threshold (every pooling mode, a
  plane without validity, cells no plane covers, NaN, a ragged cell count)."""
  G = lambda t: t.to(DEV).contiguous()
  D, M = 128, 8192 + 1037
  planes = [rnd((M, D), 170 + i) for i in range(3)]
  valids = [torch.rand((M,), generator=torch.Generator().manual_seed(180 + i)) > 0.4 for i in range(3)]
  planes[1][5, 7] = float('nan')
  Wm, bm = rnd((D, Dm), 175, 0.1), rnd((Dm,), 176, 0.01)
  args = ([G(p) for p in planes], [G(v) for v in valids], pooling, G(Wm), G(bm))
  fg, vg, mg = ops.plane_fuse_match(*args, normalize=normalize)
  fw, vw, mw = oracle_ops.plane_fuse_match(planes, valids, pooling, Wm, bm) if normalize else (None, None, None)
  if normalize:
    ok = ~torch.isnan(fw).any(-1)
    helpers.report('fuse valid', vg.cpu(), vw, 0)
    helpers.report('fused', fg.cpu()[ok], fw[ok], atol=1e-6)
    helpers.report('matching', mg.cpu()[ok], mw[ok], atol=2e-6, rtol=1e-5)
  for lo in range(0, M, 4096):        # (4096 cells per launch: the per-cell kernel)
    hi = min(lo + 4096, M)
    f0, v0, m0 = ops.plane_fuse_match([G(p[lo:hi]) for p in planes], [G(v[lo:hi]) for v in valids], pooling,
                                      G(Wm), G(bm), normalize=normalize)
    assert torch.equal(f0.view(torch.int32), fg[lo:hi].view(torch.int32)) and torch.equal(v0, vg[lo:hi])
    assert torch.equal(m0.view(torch.int32), mg[lo:hi].view(torch.int32))
  # one plane, no validity
  f1, v1, m1 = ops.plane_fuse_match([G(planes[0])], [None], pooling, G(Wm), G(bm), normalize=normalize)
  f2, v2, m2 = ops.plane_fuse_match([G(planes[0][:4096])], [None], pooling, G(Wm), G(bm), normalize=normalize)
  assert torch.equal(m1[:4096].view(torch.int32), m2.view(torch.int32)) and bool(v1.all())


def test_matching_zero_norm():
  D, Dm = 32, 8
  plane = torch.zeros(1, 4, 4, D)
  valid = torch.ones(1, 4, 4, dtype=torch.bool)
  Wm = rnd((D, Dm), 77)
  bm = torch.zeros(Dm)
  (_, _, mg), (_, _, mw) = both('plane_fuse_match', ([plane], [valid], 'max', Wm, bm))
  helpers.report('zero-norm matching', mg, mw, atol=0)


# ----------------------------------------------------------------------------
# pose
# ----------------------------------------------------------------------------
def _unit(x):
  return x / x.norm(dim=-1, keepdim=True)


@pytest.mark.parametrize('X,Y,Dm', [(16, 16, 32), (13, 21, 8), (40, 24, 16)])
def test_sim_softmax(X, Y, Dm):
  B, Nq = 2, 70
  fq = _unit(rnd((B, Nq, Dm), 90))
  fm = _unit(rnd((B, X, Y, Dm), 91))
  fq[0, 3] = 0  # an invalid (masked) query point
  nv = torch.tensor([69.0, 70.0])
  scale = float(np.exp(2.0))
  got, want = both('sim_softmax', (fq, fm, scale, True, nv), dict(want_prob=True))
  helpers.report('sim', got[0], want[0], atol=1e-6, rtol=1e-5)
  helpers.report('chunk max', got[1][..., 0], want[1][..., 0], atol=1e-5, rtol=1e-5)
  helpers.report('chunk sum', got[1][..., 1], want[1][..., 1], atol=1e-4, rtol=1e-5)
  helpers.report('prob', got[2], want[2], atol=1e-9, rtol=1e-4)
  helpers.report('rowstats', got[3], want[3], atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize('Dm', [8, 32])
def test_sim_softmax_mfma_kernel_keeps_the_valu_kernels_bits(Dm, monkeypatch):
  """The matrix-core similarity kernel runs the same k-ordered fmaf chain as the VALU kernel it
  replaces (v_mfma_f32_32x32x2_f32 is an exact f32 chain): sim must agree bit for bit -- the
  sampler re-evaluates the selected chunk with that chain -- and the chunk statistics up to the
  order of their 64-term sums.  Ragged sizes: Nq and X*Y not multiples of the 64 x 256 tile."""
  B, Nq, X, Y = 2, 150, 27, 23
  fq = _unit(rnd((B, Nq, Dm), 190)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 191)).to(DEV)
  nv = torch.tensor([149.0, 150.0], device=DEV)
  scale = float(np.exp(2.0))
  fq_u, fm_u = _unaligned(fq), _unaligned(fm)      # 4-byte offset views: the VALU kernel's inputs
  for clip in (True, False):
    sim_v, st_v, _, _ = ops.sim_softmax(fq_u, fm_u, scale, clip, nv)
    sim_m, st_m, _, _ = ops.sim_softmax(fq, fm, scale, clip, nv)
    assert torch.equal(sim_m, sim_v)
    assert torch.equal(st_m[..., 0], st_v[..., 0])
    assert torch.allclose(st_m[..., 1], st_v[..., 1], rtol=1e-5, atol=0)


@pytest.mark.parametrize('math,tol', [('bf16x6', 2e-6), ('bf16x3', 6e-5)])
@pytest.mark.parametrize('X,Y,Dm,Nq', [(16, 16, 32, 70), (27, 23, 16, 150), (40, 24, 64, 33)])
def test_sim_softmax_split_bf16(X, Y, Dm, Nq, math, tol):
  """The contraction on the bf16 matrix cores with split operands: vs the oracle and vs the exact
  f32 kernel (bf16x6 keeps 24 significand bits per operand: the difference to the f32 chain is
  rounding-order level), incl. ragged tiles, a zero query row, confidence weights."""
  B = 2
  fq = _unit(rnd((B, Nq, Dm), 290))
  fm = _unit(rnd((B, X, Y, Dm), 291))
  fq[0, 3] = 0
  nv = torch.tensor([float(Nq - 1), float(Nq)])
  scale = float(np.exp(2.0))
  got, want = both('sim_softmax', (fq, fm, scale, True, nv), dict(math=math))
  helpers.report(f'sim {math}', got[0], want[0], atol=tol / 10, rtol=tol)
  helpers.report(f'chunk max {math}', got[1][..., 0], want[1][..., 0], atol=tol * 10, rtol=tol)
  helpers.report(f'chunk sum {math}', got[1][..., 1], want[1][..., 1], atol=1e-3, rtol=10 * tol)
  exact = ops.sim_softmax(fq.to(DEV), fm.to(DEV), scale, True, nv.to(DEV), math='f32')
  assert float((got[0] - exact[0]).abs().max()) <= tol * float(exact[0].abs().max())
  w = torch.rand(B, Nq, generator=torch.Generator().manual_seed(5)) + 0.1
  w = (w / w.sum(-1, keepdim=True)).contiguous()
  gw, ww = both('sim_softmax', (fq, fm, scale, False, nv), dict(math=math, row_weight=w))
  helpers.report(f'weighted sim {math}', gw[0], ww[0], atol=tol / 10, rtol=tol)


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3'])
@pytest.mark.parametrize('X,Y,Dm,Nq,clip', [(16, 16, 32, 70, True), (32, 24, 16, 150, False), (128, 128, 32, 131, True),
                                            (16, 48, 64, 64, True)])
def test_sim_softmax_full_chunk_kernel_keeps_the_general_kernels_bits(X, Y, Dm, Nq, clip, math):
  """X Y % 256 == 0 takes ``sim_split_fast_kernel`` (operand roles swapped in the MFMA, no tail
  masks, statistics combined per workgroup, non-temporal stores): sim and the chunk statistics are
  BIT-IDENTICAL to the general kernel's, with and without confidence weights, ragged row tiles."""
  B = 2
  fq = _unit(rnd((B, Nq, Dm), 390)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 391)).to(DEV)
  fq[0, 3] = 0
  nv = torch.tensor([float(Nq - 1), float(Nq)], device=DEV)
  scale = float(np.exp(2.0))
  w = torch.rand(B, Nq, generator=torch.Generator().manual_seed(6)) + 0.1
  w = (w / w.sum(-1, keepdim=True)).contiguous().to(DEV)
  for rw in (None, w):
    fast = ops.sim_softmax(fq, fm, scale, clip, nv, math=math, row_weight=rw)
    try:
      ops.SIM_GENERAL_KERNEL = True
      gen = ops.sim_softmax(fq, fm, scale, clip, nv, math=math, row_weight=rw)
    finally:
      ops.SIM_GENERAL_KERNEL = False
    assert torch.equal(fast[0], gen[0]), float((fast[0] - gen[0]).abs().max())
    assert torch.equal(fast[1], gen[1]), float((fast[1] - gen[1]).abs().max())
  assert float(fast[0].abs().max()) > 0


def test_ransac_sample_given_uniforms():
  B, Nq, X, Y, Dm, S = 2, 40, 24, 20, 16, 600
  fq = _unit(rnd((B, Nq, Dm), 95))
  fm = _unit(rnd((B, X, Y, Dm), 96))
  nv = torch.tensor([40.0, 40.0])
  scale = float(np.exp(2.5))
  u = torch.rand((B, S, 2), generator=torch.Generator().manual_seed(97))
  _, stats, _, _ = ops.sim_softmax(fq.to(DEV), fm.to(DEV), scale, True, nv.to(DEV))
  got = ops.ransac_sample(fq.to(DEV), fm.to(DEV), stats, scale, True, S, uniforms=u.to(DEV)).cpu()
  want = oracle_ops.ransac_sample(fq, fm, None, scale, True, S, uniforms=u)
  assert (got[..., 0] == want[..., 0]).all(), 'query row selection differs'
  # the cell may differ by CDF round-off: accept if the float64 CDF at the
  # returned cell brackets the target within 1e-5 of the row mass.
  bad = 0
  q64, m64 = fq.double().numpy(), fm.double().numpy()
  for b in range(B):
    for s in range(S):
      if (got[b, s] == want[b, s]).all():
        continue
      n = int(got[b, s, 0])
      x = np.maximum(np.einsum('d,ijd->ij', q64[b, n], m64[b]), 0).reshape(-1) * scale
      e = np.exp(x - x.max())
      cdf = np.cumsum(e) / e.sum()
      cell = int(got[b, s, 1]) * Y + int(got[b, s, 2])
      lo = cdf[cell - 1] if cell > 0 else 0.0
      t = float(u[b, s, 1])
      if not (lo - 1e-5 <= t <= cdf[cell] + 1e-5):
        bad += 1
  assert bad == 0, f'{bad} samples outside their CDF bracket'


def test_ransac_sample_distribution():
  """Philox path: empirical cell histogram matches prob (chi-square-like bound)."""
  B, Nq, X, Y, Dm, S = 1, 6, 8, 8, 8, 60000
  fq = _unit(rnd((B, Nq, Dm), 98))
  fm = _unit(rnd((B, X, Y, Dm), 99))
  nv = torch.tensor([6.0])
  scale = float(np.exp(2.0))
  _, stats, prob, _ = ops.sim_softmax(fq.to(DEV), fm.to(DEV), scale, True, nv.to(DEV), want_prob=True)
  corr = ops.ransac_sample(fq.to(DEV), fm.to(DEV), stats, scale, True, S, seed=1234).cpu().numpy()[0]
  hist = np.zeros((Nq, X, Y))
  np.add.at(hist, (corr[:, 0], corr[:, 1], corr[:, 2]), 1)
  p = prob.cpu().numpy()[0].astype(np.float64)
  p = p / p.sum()
  exp = p * S
  z = (hist - exp) / np.sqrt(exp + 1e-9)
  assert np.abs(z).max() < 6.0, f'max z-score {np.abs(z).max()}'
  corr2 = ops.ransac_sample(fq.to(DEV), fm.to(DEV), stats, scale, True, S, seed=1234).cpu().numpy()[0]
  assert (corr == corr2).all(), 'sampling is not deterministic for a fixed seed'


def test_poses_from_corr():
  B, Nq, P, retries, X, Y = 2, 50, 200, 4, 30, 28
  rng = np.random.default_rng(100)
  corr = np.stack([rng.integers(0, Nq, (B, P * retries * 2)),
                   rng.integers(0, X, (B, P * retries * 2)),
                   rng.integers(0, Y, (B, P * retries * 2))], -1).astype(np.int32)
  corr[0, 0] = corr[0, 1]  # degenerate pair (identical correspondences)
  q_xy = torch.tensor(rng.uniform(-5, 5, (B, Nq, 2)).astype(np.float32))
  got, want = both('poses_from_corr', (torch.tensor(corr), q_xy, P, retries, 0.2))
  g, w = got.cpu().numpy(), want.numpy()
  # degenerate / antipodal cases aside, angle and translation agree.
  dang = np.abs(np.angle(np.exp(1j * (g[..., 0] - w[..., 0]))))
  ok = dang < 1e-3
  assert ok.mean() > 0.98, f'only {ok.mean():.3f} of poses agree'
  helpers.report('pose t', got.cpu()[..., 1:][torch.tensor(ok)], want[..., 1:][torch.tensor(ok)],
                 atol=2e-3)


@pytest.mark.parametrize('mask_oob', [False, True])
@pytest.mark.parametrize('X,Y', [(32, 32), (25, 37), (192, 160), (131, 260)])   # the last two: row bands
def test_pose_score(mask_oob, X, Y):
  B, Nq, P = 2, 45, 700
  rng = np.random.default_rng(110)
  sim = torch.tensor(rng.random((B, Nq, X, Y), dtype=np.float32))
  cell = 0.2
  poses = np.stack([rng.uniform(-np.pi, np.pi, (B, P)),
                    rng.uniform(-0.2 * X * cell, 1.2 * X * cell, (B, P)),
                    rng.uniform(-0.2 * Y * cell, 1.2 * Y * cell, (B, P))], -1).astype(np.float32)
  q_xy = torch.tensor(rng.uniform(-2, 2, (B, Nq, 2)).astype(np.float32))
  valid_q = torch.tensor(rng.random((B, Nq)) > 0.2)
  map_valid = torch.tensor(rng.random((B, X, Y)) > 0.1)
  got, want = both('pose_score', (sim, torch.tensor(poses), q_xy, valid_q, map_valid, cell),
                   dict(mask_oob=mask_oob))
  helpers.report('pose scores', got, want, atol=2e-4, rtol=1e-5)
  ig, iw = both('argmax_rows', (want.clone(), 1))
  helpers.report('argmax', ig, iw, 0)


def test_refine_lattice_and_argmax_ties():
  from snap_amd.models import pose_estimation
  init = torch.tensor([[0.3, 4.0, 5.0], [-2.0, 1.0, 2.5]])
  offs_r, offs_p = pose_estimation.refinement_offsets('cpu')
  assert offs_r.numel() == 41 and offs_p.numel() == 41
  got, want = both('refine_lattice', (init, offs_r, offs_p))
  helpers.report('lattice', got, want, atol=2e-6)
  s = torch.zeros(3, 1000)
  s[0, 500] = 1; s[0, 700] = 1      # tie -> first index
  s[1, 999] = 2
  s[2, :] = -1.0
  ig, iw = both('argmax_rows', (s, 0))
  helpers.report('argmax ties', ig, iw, 0)
  ig, iw = both('argmax_rows', (s, 600))
  helpers.report('argmax start', ig, iw, 0)
  # long rows (the 68 921-pose lattice) take the 1024-thread kernel: ties -> first index, start offsets
  g = torch.Generator().manual_seed(5)
  s = torch.randn(2, 68921, generator=g)
  s[0, 40000] = 9.0; s[0, 66000] = 9.0
  s[1, 68920] = 9.0
  for start in (0, 1, 50000):
    ig, iw = both('argmax_rows', (s, start))
    helpers.report(f'argmax long rows start {start}', ig, iw, 0)


# ----------------------------------------------------------------------------
# exhaustive voting
# ----------------------------------------------------------------------------
@pytest.mark.parametrize('method', ['direct', 'fft'])
@pytest.mark.parametrize('H,R,D', [(16, 8, 8), (24, 36, 32)])
def test_rotate_templates_and_matching(H, R, D, method):
  from oracle import grids as o_grids
  from oracle import voting as o_voting
  from snap_amd.models import pose_exhaustive_voting as pev
  from snap_amd.models import types
  from snap_amd.utils import grids
  cell = 0.25
  rng = np.random.default_rng(120)
  fq = rng.standard_normal((H, H, D)).astype(np.float32)
  vq = rng.random((H, H)) > 0.15
  fq = fq * vq[..., None]
  fm = rng.standard_normal((H, H, D)).astype(np.float32)
  vm = rng.random((H, H)) > 0.1
  g = grids.Grid2D((H, H), cell)
  og = o_grids.Grid2D((H, H), cell)
  t_w, tv_w = o_voting.sample_query_templates(fq, vq, R, og)
  t_g, tv_g = pev.sample_query_templates(torch.tensor(fq).to(DEV), torch.tensor(vq).to(DEV), R, g)
  mism = tv_g.cpu().numpy() != tv_w
  # booleans are compared EXACTLY; a flip is accepted only where the float64 rotated coordinate
  # lies on a decision boundary (grid border / change of the bilinear tap pair), to 1e-4 cell
  helpers.assert_template_validity_mismatches_on_borders(f'templates H={H} R={R}', tv_g, tv_w, cell)
  helpers.report('templates', t_g.cpu().numpy()[~mism], t_w[~mism], atol=2e-5)
  s_w = o_voting.template_matching(t_w, tv_w, fm, vm)
  s_g = pev.template_matching(torch.tensor(t_w).to(DEV), torch.tensor(tv_w).to(DEV),
                              torch.tensor(fm).to(DEV), torch.tensor(vm).to(DEV), method=method)
  helpers.report('template scores', s_g, s_w, atol=1e-4, rtol=1e-5)
  full = pev.exhaustive_pose_voting(
      types.FeaturePlane(torch.tensor(fq).to(DEV), torch.tensor(vq).to(DEV)),
      types.FeaturePlane(torch.tensor(fm).to(DEV), torch.tensor(vm).to(DEV)), R, g, method=method)
  if mism.sum() == 0:
    helpers.report('exhaustive voting', full, s_w, atol=1e-4, rtol=1e-5)
  assert full.shape == (R, 2 * H - 1, 2 * H - 1)


@pytest.mark.parametrize('H,R,D', [(100, 8, 16), (86, 12, 32), (64, 4, 8)])
def test_voting_fft_rotated_entry_equals_the_explicit_templates(H, R, D):
  """The rotated entry of the frequency-domain voting (templates sampled inside the first transform, masks + counts
  from the tiled validity pass: 64 x 64 tiles, here with partial tiles and several tiles per side) against the
  explicit-template entry fed with ``sample_query_templates`` (the stand-alone rotate kernel's templates and masks):
  the same -inf mask and the same scores."""
  from snap_amd.models import pose_exhaustive_voting as pev
  from snap_amd.models import types
  from snap_amd.utils import grids
  rng = np.random.default_rng(150 + H)
  vq = torch.tensor(rng.random((H, H)) > 0.15).to(DEV)
  fq = (torch.tensor(rng.standard_normal((H, H, D)).astype(np.float32)).to(DEV) * vq[..., None]).contiguous()
  fm = torch.tensor(rng.standard_normal((H, H, D)).astype(np.float32)).to(DEV)
  vm = torch.tensor(rng.random((H, H)) > 0.1).to(DEV)
  g = grids.Grid2D((H, H), 0.25)
  t, tv = pev.sample_query_templates(fq, vq, R, g)
  want = pev.template_matching(t, tv, fm, vm, method='fft')
  got = pev.exhaustive_pose_voting(types.FeaturePlane(fq, vq), types.FeaturePlane(fm, vm), R, g, method='fft')
  fw, fg = torch.isfinite(want), torch.isfinite(got)
  assert torch.equal(fw, fg), int((fw != fg).sum())
  assert 0 < int(fw.sum()) < fw.numel()
  helpers.report(f'rotated fft entry H={H}', got[fg], want[fw], atol=1e-6 * float(want[fw].abs().max()), rtol=0)


@pytest.mark.parametrize('H,R,D,S', [(16, 8, 8, 4), (24, 36, 32, 4), (24, 36, 32, 3), (20, 12, 16, 2)])
def test_template_matching_shift_stacked(H, R, D, S, monkeypatch):
  """The shift-stacked form of the correlation GEMM (R S^2 filters, stride S) is the direct form
  re-tiled: compared with the oracle and with the plain path (same products and k order; only
  the split-K partition can differ), incl. output sizes that S does not divide."""
  from oracle import voting as o_voting
  from snap_amd.models import pose_exhaustive_voting as pev
  rng = np.random.default_rng(140 + S)
  t = rng.standard_normal((R, H, H, D)).astype(np.float32)
  tv = rng.random((R, H, H)) > 0.2
  t = t * tv[..., None]
  fm = rng.standard_normal((H, H, D)).astype(np.float32)
  vm = rng.random((H, H)) > 0.1
  args = [torch.tensor(a).to(DEV) for a in (t, tv, fm, vm)]
  monkeypatch.setattr(pev, 'STACK_SHIFT', 1)
  plain = pev.template_matching(*args, method='direct')
  monkeypatch.setattr(pev, 'STACK_SHIFT', S)
  monkeypatch.setattr(pev, 'STACK_MIN_CELLS', 0)
  stacked = pev.template_matching(*args, method='direct')
  want = o_voting.template_matching(t, tv, fm, vm)
  helpers.report(f'stacked S={S} vs oracle', stacked, want, atol=1e-4, rtol=1e-5)
  helpers.report(f'stacked S={S} vs plain', stacked, plain, atol=2e-5, rtol=1e-6)


@pytest.mark.parametrize('H,W,Hm,Wm,R,D', [
    (16, 16, 16, 16, 8, 8),      # N = 48 = 3 * 4 * 4
    (24, 24, 24, 24, 36, 32),    # N = 96 = 3 * 2 * 4 * 4: the radix-2 stage
    (22, 22, 22, 22, 7, 34),     # N = 64 = 4^3; odd R (an unpaired rotation), two channel groups, D % 4 != 0
    (10, 13, 12, 9, 12, 16),     # rectangular template AND a map of another size
    (43, 43, 43, 43, 36, 64),    # N = 128 = 2 * 4^3; two full channel groups
    (86, 86, 86, 86, 4, 8),      # N = 256
])
@pytest.mark.parametrize('overlap', [0.05, None])
def test_voting_fft_vs_oracle(H, W, Hm, Wm, R, D, overlap):
  """``snap_voting_fft_f32`` against oracle/voting.py's direct sliding-window sum: the -inf mask
  EXACT, finite scores to 2e-5 (f32 transforms; the oracle sums in f32 too)."""
  from oracle import voting as o_voting
  from snap_amd.models import pose_exhaustive_voting as pev
  rng = np.random.default_rng(500 + H + R + D)
  t = rng.standard_normal((R, H, W, D)).astype(np.float32)
  tv = rng.random((R, H, W)) > 0.2
  t = t * tv[..., None]
  fm = rng.standard_normal((Hm, Wm, D)).astype(np.float32)
  vm = rng.random((Hm, Wm)) > 0.1
  want = o_voting.template_matching(t, tv, fm, vm, min_overlap=overlap)
  got = pev.template_matching(*[torch.tensor(a).to(DEV) for a in (t, tv, fm, vm)], min_overlap=overlap,
                              method='fft').cpu().numpy()
  assert got.shape == want.shape
  fw, fg = np.isfinite(want), np.isfinite(got)
  assert (fw == fg).all(), int((fw != fg).sum())
  if overlap is not None:
    assert (want[~fw] == got[~fg]).all()      # -inf, not +inf / NaN
  scale = float(np.abs(want[fw]).max())
  helpers.report(f'fft voting H={H} R={R} D={D}', got[fg], want[fw], atol=2e-5 * max(1.0, scale))


def test_voting_fft_rejects_maps_beyond_its_transform_sizes():
  from snap_amd.models import pose_exhaustive_voting as pev
  assert ops.voting_fft_supported(36, 256, 256, 32, 256, 256)
  assert ops.voting_fft_supported(36, 342, 342, 32, 342, 342)       # 3 * 342 - 2 = 1024
  assert not ops.voting_fft_supported(36, 343, 343, 32, 343, 343)
  q = torch.zeros(4, 8, 8, 4, device=DEV)
  with pytest.raises(ValueError):
    pev._use_fft('fft', 4, (343, 343), 32, (343, 343))
  assert pev._use_fft('auto', 4, (343, 343), 32, (343, 343)) is False     # falls back to the direct form
  assert pev._use_fft('auto', 4, (8, 8), 4, (8, 8)) is False              # small maps stay direct
  assert pev._use_fft('auto', 36, (256, 256), 32, (256, 256)) is True
  # a workspace beyond the budget sends 'auto' back to the direct form ('fft' stays forced)
  keep, pev.FFT_WORKSPACE_BUDGET = pev.FFT_WORKSPACE_BUDGET, 1 << 20
  try:
    assert ops.voting_fft_workspace_bytes(36, 256, 256, 32, 256, 256) > 1 << 20
    assert pev._use_fft('auto', 36, (256, 256), 32, (256, 256)) is False
    assert pev._use_fft('fft', 36, (256, 256), 32, (256, 256)) is True
  finally:
    pev.FFT_WORKSPACE_BUDGET = keep
  del q


@pytest.mark.parametrize('H,W,R,D,S', [(24, 24, 36, 32, 4), (10, 13, 12, 16, 3), (9, 9, 8, 20, 2), (17, 16, 36, 64, 4)])
def test_fused_template_pack_equals_stack_then_pack(H, W, R, D, S):
  """``snap_pack_stacked_templates_split_bf16``: templates [R, H, W, D] -> the split engine's two-part
  weight image of the shift-stacked bank, BIT FOR BIT what ``stack_templates`` followed by
  ``pack_weights_split_bf16(., 2)`` writes (D not a multiple of 16, column padding, every shift)."""
  t = rnd((R, H, W, D), 160 + S).to(DEV)
  bank = ops.stack_templates(t, S, 'rhwd')
  want = ops.pack_weights_split_bf16(bank, 2)
  got = ops.pack_stacked_templates_split(t, S)
  assert got.shape == tuple(bank.shape)
  assert got.data.numel() == want.numel()
  assert torch.equal(got.data.view(torch.int16), want.view(torch.int16))


def test_voting_with_the_fused_template_pack_keeps_its_bits(monkeypatch):
  from snap_amd.models import pose_exhaustive_voting as pev
  H, R, D = 24, 36, 32
  rng = np.random.default_rng(171)
  t = rng.standard_normal((R, H, H, D)).astype(np.float32)
  tv = rng.random((R, H, H)) > 0.2
  t = t * tv[..., None]
  fm = rng.standard_normal((H, H, D)).astype(np.float32)
  vm = rng.random((H, H)) > 0.1
  args = [torch.tensor(a).to(DEV) for a in (t, tv, fm, vm)]
  monkeypatch.setattr(pev, 'STACK_MIN_CELLS', 0)
  monkeypatch.setattr(ops, 'MATMUL_PRECISION', 'bf16x3')
  monkeypatch.setattr(ops, 'FUSED_TEMPLATE_PACK', True)
  fused = pev.template_matching(*args, method='direct')
  monkeypatch.setattr(ops, 'FUSED_TEMPLATE_PACK', False)
  plain = pev.template_matching(*args, method='direct')
  assert torch.equal(fused, plain)
  assert bool(torch.isfinite(fused).any())


def test_overlap_count_on_the_bf16_engine_is_exact(monkeypatch):
  """The 0/1 overlap-count correlation folded onto the bf16 engine (W % 32 == 0 maps) returns the
  same integers as the f32 scalar path -- bit for bit."""
  from snap_amd.models import pose_exhaustive_voting as pev
  H = W = 64
  R = 8
  g = torch.Generator().manual_seed(150)
  cw = (torch.rand((H, W, 1, R), generator=g) < 0.8).float().to(DEV)
  mv = (torch.rand((H, W), generator=g) < 0.9)
  mp, mvp = ops.pad_map(torch.zeros((H, W, 4), device=DEV), mv.to(DEV))
  monkeypatch.setattr(pev, 'STACK_MIN_CELLS', 1 << 30)
  plain = pev._overlap_count(mvp, cw, R, (H, W))
  monkeypatch.setattr(pev, 'STACK_MIN_CELLS', 0)
  folded = pev._overlap_count(mvp, cw, R, (H, W))
  assert plain.shape == folded.shape == (2 * H - 1, 2 * W - 1, R)
  assert torch.equal(plain, folded)
  assert float(plain.max()) > 1000          # real counts, not zeros


def test_exhaustive_identity_kat():
  """Known answer: matching a map with itself peaks at (0, H-1, W-1)."""
  from snap_amd.models import pose_exhaustive_voting as pev
  from snap_amd.models import types
  from snap_amd.utils import grids
  H, D, R = 32, 16, 12
  rng = np.random.default_rng(130)
  f = torch.tensor(rng.standard_normal((H, H, D)).astype(np.float32)).to(DEV)
  v = torch.ones(H, H, dtype=torch.bool, device=DEV)
  plane = types.FeaturePlane(f, v)
  s = pev.exhaustive_pose_voting(plane, plane, R, grids.Grid2D((H, H), 0.5))
  idx = np.unravel_index(int(torch.argmax(s)), s.shape)
  assert tuple(int(i) for i in idx) == (0, H - 1, H - 1)


# ---------------------------------------------------------------------------
# masked-row compaction (row-indexed fusion MLP)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('M,p', [(1, 1.0), (17, 0.5), (4096, 0.0), (4097, 1.0), (100003, 0.6),
                                 (1 << 20, 0.03)])
def test_compact_rows_matches_nonzero(M, p):
  g = torch.Generator().manual_seed(M)
  mask = (torch.rand(M, generator=g) < p)
  index, count = ops.compact_rows(mask.cuda())
  want = torch.nonzero(mask).flatten().to(torch.int32)
  assert int(count.item()) == want.numel()
  assert torch.equal(index[: want.numel()].cpu(), want)
  # unaligned view (the 16-byte fast path must not be taken blindly)
  if M > 5:
    sub = mask.cuda()[3:]                                    # data_ptr off by 3 bytes
    index, count = ops.compact_rows(sub)
    want = torch.nonzero(mask[3:]).flatten().to(torch.int32)
    assert int(count.item()) == want.numel()
    assert torch.equal(index[: want.numel()].cpu(), want)


@pytest.mark.parametrize('layers_,in_dim,stride', [((64, 32), 65, 68), ((128,), 257, 260), ((256, 128), 257, 260)])
def test_masked_row_mlp_is_bitwise_the_dense_mlp(layers_, in_dim, stride, monkeypatch):
  from snap_amd.models import layers
  from snap_amd.utils import config_dict
  cfg = config_dict.ConfigDict(dict(layers=layers_, activation='relu', apply_input_activation=False))
  mlp = layers.MLP(cfg, in_dim=in_dim)
  gen = torch.Generator().manual_seed(5)
  params = helpers.params_to_device(mlp.init_params(gen, 'cpu'), 'cuda')
  for i in range(len(layers_)):
    params[f'Dense_{i}']['bias'] = torch.randn(layers_[i], generator=gen).cuda()
  M = 70001
  x = torch.randn(M, stride, generator=gen).cuda()
  mask = (torch.rand(M, generator=gen) < 0.6).cuda()
  dense = mlp(params, x, row_mask=mask)                      # M < COMPACT_MIN_ROWS: dense path
  monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', 0)
  compact = mlp(params, x, row_mask=mask)
  assert torch.equal(dense, compact)
  assert bool((compact[~mask] == 0).all())
  # all rows masked / none masked
  for m in (torch.zeros(M, dtype=torch.bool).cuda(), torch.ones(M, dtype=torch.bool).cuda()):
    monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', 1 << 30)
    d = mlp(params, x, row_mask=m)
    monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', 0)
    assert torch.equal(d, mlp(params, x, row_mask=m))


# ---------------------------------------------------------------------------
# GroupNorm statistics emitted by the producing conv's epilogue
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('N,H,W,Cin,Cout,k,relu', [
    (3, 17, 19, 64, 128, 1, False),     # HW = 323: row tiles straddle images
    (2, 34, 34, 64, 64, 3, False),      # BN = 64 tiles
    (5, 16, 16, 128, 256, 1, True),     # HW = 256 = 2 tiles exactly; FPN order
    (1, 40, 40, 32, 512, 1, False),
    (40, 12, 12, 64, 256, 1, False),    # HW = 144: >= tile (64x.. tiles) many images
    (8, 64, 64, 64, 256, 1, False),     # 128x128 tiles
    (20, 70, 70, 64, 64, 3, True),      # 128x64 tiles, HW = 4900 straddles
])
def test_gn_stats_from_conv_epilogue(N, H, W, Cin, Cout, k, relu):
  g = torch.Generator().manual_seed(N * 1000 + Cout)
  x = torch.randn(N, H, W, Cin, generator=g).cuda()
  w = (torch.randn(k, k, Cin, Cout, generator=g) / np.sqrt(k * k * Cin)).cuda()
  res = torch.randn(N, H, W, Cout, generator=g).cuda() + 0.7
  gamma = (torch.rand(Cout, generator=g) + 0.5).cuda()
  pad = ((k // 2, k // 2), (k // 2, k // 2))
  ops.USE_SPLITK = False      # small test shapes would otherwise take the split-K route
  try:
    y = ops.conv2d(x, w, padding=pad, residual=res, emit_gn_stats='relu' if relu else 'raw')
  finally:
    ops.USE_SPLITK = True
  assert hasattr(y, '_snap_gn_partial'), 'shape should support fused statistics'
  mu_f, sc_f, rs_f = ops.group_norm_stats(y, gamma, relu_first=relu, want_rstd=True)
  ops.USE_FUSED_GN_STATS = False
  try:
    mu_s, sc_s, rs_s = ops.group_norm_stats(y, gamma, relu_first=relu, want_rstd=True)
  finally:
    ops.USE_FUSED_GN_STATS = True
  helpers.report('mu', mu_f, mu_s, atol=2e-6, rtol=2e-6)
  helpers.report('sc', sc_f, sc_s, atol=1e-6, rtol=5e-6)
  helpers.report('rstd', rs_f, rs_s, atol=1e-6, rtol=5e-6)
  # and against the oracle's two-pass definition in fp64
  v = torch.relu(y) if relu else y
  vg = v.double().reshape(N, H * W, 32, Cout // 32)
  mean = vg.mean((1, 3))
  var = ((vg - mean[:, None, :, None]) ** 2).mean((1, 3))
  want_mu = mean[:, :, None].expand(N, 32, Cout // 32).reshape(N, Cout)
  want_rs = (1 / torch.sqrt(var + 1e-5))[:, :, None].expand(N, 32, Cout // 32).reshape(N, Cout)
  helpers.report('mu vs fp64', mu_f, want_mu.float(), atol=2e-6, rtol=2e-6)
  helpers.report('rstd vs fp64', rs_f, want_rs.float(), atol=1e-6, rtol=5e-6)


# ---------------------------------------------------------------------------
# split-K launches of the conv engine (small-M / deep-K layers)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,kind', [
    (1, 17, 17, 512, 512, 3, 1, 'gn_res'),     # the deep ResNet stage shape
    (2, 9, 9, 2048, 512, 1, 1, 'plain'),
    (1, 1, 4652, 16384, 32, 1, 1, 'plain'),    # similarity backward (d fq)
    (2, 20, 20, 256, 64, 3, 2, 'bias_relu'),
    (1, 12, 12, 67, 128, 3, 1, 'plain'),       # scalar (Cin % 4 != 0) loader
    (3, 8, 8, 1024, 256, 1, 1, 'mask'),
])
def test_conv_split_k_matches_single_pass(N, H, W, Cin, Cout, k, stride, kind):
  g = torch.Generator().manual_seed(Cin + Cout)
  x = torch.randn(N, H, W, Cin, generator=g).cuda()
  w = (torch.randn(k, k, Cin, Cout, generator=g) / np.sqrt(k * k * Cin)).cuda()
  pad = ((k // 2, k // 2), (k // 2, k // 2))
  Ho = (H + 2 * (k // 2) - k) // stride + 1
  Wo = (W + 2 * (k // 2) - k) // stride + 1
  kw = dict(stride=stride, padding=pad)
  if kind == 'gn_res':
    gamma = (torch.rand(Cin, generator=g) + 0.5).cuda()
    beta = torch.randn(Cin, generator=g).cuda()
    mu, sc = ops.group_norm_stats(x, gamma)
    kw.update(prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta),
              residual=torch.randn(N, Ho, Wo, Cout, generator=g).cuda())
  elif kind == 'bias_relu':
    kw.update(bias=torch.randn(Cout, generator=g).cuda(), relu=True)
  elif kind == 'mask':
    kw.update(row_mask=(torch.rand(N * Ho * Wo, generator=g) < 0.5).cuda())
  lib = ops._lib.load()
  split = ops.conv2d(x, w, **kw)
  ops.USE_SPLITK = False
  try:
    single = ops.conv2d(x, w, **kw)
  finally:
    ops.USE_SPLITK = True
  helpers.report('split-K vs single pass', split, single, atol=3e-5, rtol=1e-5)


# ---------------------------------------------------------------------------
# VerticalPooling: 'softmax' / 'weighted' / 'mlp' modes
# ---------------------------------------------------------------------------
def _volume(seed, lead=(2, 9, 7), Z=60, D=128):
  g = torch.Generator().manual_seed(seed)
  vol = torch.randn(*lead, Z, D, generator=g)
  valid = torch.rand(*lead, Z, generator=g) < 0.55
  valid[0, 0, 0] = False          # a column without any valid level
  valid[0, 0, 1] = True           # a fully valid column
  valid[0, 1, 0] = False
  valid[0, 1, 0, Z // 3] = True   # exactly one valid level
  return vol, valid, g


@pytest.mark.parametrize('mode', ['softmax', 'weighted'])
@pytest.mark.parametrize('Z,D', [(60, 128), (12, 32)])
def test_vertical_pool_confidence_modes(mode, Z, D):
  from oracle import bev as o_bev
  vol, valid, g = _volume(31, Z=Z, D=D)
  w = torch.randn(D, 1, generator=g) * 0.3
  b = torch.randn(1, generator=g)
  want = o_bev.vertical_pooling({'pooling': mode}, vol.numpy(), valid.numpy(),
                                {'confidence_head': {'kernel': w.numpy(), 'bias': b.numpy()}})
  plane, pvalid, scores, weights = ops.vertical_pool_conf(
      vol.cuda(), valid.cuda(), w.reshape(-1).cuda(), b.cuda(), mode == 'weighted')
  helpers.report('valid', pvalid, want['valid'], 0)
  helpers.report('scores', scores, want['scores'], atol=2e-5, rtol=1e-5)
  helpers.report('weights', weights, want['weights'], atol=2e-6, rtol=2e-5)
  helpers.report('plane', plane, want['features'], atol=2e-5, rtol=2e-5)
  assert float(plane[0, 0, 0].abs().max()) == 0.0
  assert abs(float(weights[0, 1, 0].sum()) - 1.0) < 1e-6 and float(weights[0, 1, 0, Z // 3]) == 1.0


def test_vertical_pooling_module_modes_match_oracle():
  """The module-level API for every pooling mode, incl. 'mlp' (flatten (Z, D) -> MLP)."""
  from oracle import bev as o_bev
  from snap_amd.models import bev_mapper, types
  from snap_amd.utils import config_dict
  vol, valid, g = _volume(32, lead=(1, 6, 5), Z=12, D=32)
  fv = types.FeatureVolume(features=vol.cuda(), valid=valid.cuda())
  mlp_cfg = dict(layers=(64, 32), activation='relu', apply_input_activation=False)
  for mode in ('max', 'sum', 'mean', 'softmax', 'weighted', 'mlp'):
    cfg = config_dict.ConfigDict(dict(pooling=mode, mlp=mlp_cfg))
    vp = bev_mapper.VerticalPooling(cfg, num_levels=12, feature_dim=32)
    params = vp.init_params(torch.Generator().manual_seed(3), 'cpu')
    pred = vp(helpers.params_to_device(params, 'cuda'), fv)
    want = o_bev.vertical_pooling(cfg.to_dict(), vol.numpy(), valid.numpy(), helpers.params_to_numpy(params))
    helpers.report(f'{mode} plane', pred['plane'].features, want['features'], atol=3e-5, rtol=3e-5)
    helpers.report(f'{mode} valid', pred['plane'].valid, want['valid'], 0)
    assert ('weights' in pred) == (mode in ('softmax', 'weighted'))


@pytest.mark.parametrize('X,Y,Nq', [(24, 20, 40), (128, 128, 300), (9, 7, 5)])
def test_ransac_row_table_gives_identical_samples(X, Y, Nq):
  """The per-row prefix table is a pure speed path: same correspondences, bit for bit."""
  B, Dm, S = 2, 32, 4000
  fq = _unit(rnd((B, Nq, Dm), 195)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 196)).to(DEV)
  nv = torch.full((B,), float(Nq)).to(DEV)
  scale = float(np.exp(2.0))
  _, stats, _, _ = ops.sim_softmax(fq, fm, scale, True, nv)
  a = ops.ransac_sample(fq, fm, stats, scale, True, S, seed=77, row_table=True)
  b = ops.ransac_sample(fq, fm, stats, scale, True, S, seed=77, row_table=False)
  assert torch.equal(a, b)
  u = torch.rand((B, S, 2), generator=torch.Generator().manual_seed(5)).to(DEV)
  a = ops.ransac_sample(fq, fm, stats, scale, True, S, uniforms=u, row_table=True)
  b = ops.ransac_sample(fq, fm, stats, scale, True, S, uniforms=u, row_table=False)
  assert torch.equal(a, b)


# ---------------------------------------------------------------------------
# degenerate inputs
# ---------------------------------------------------------------------------
def test_pose_score_without_valid_points_is_zero():
  B, Nq, X, Y, P = 1, 70, 128, 128, 300
  g = torch.Generator().manual_seed(1)
  sim = torch.rand(B, Nq, X, Y, generator=g).to(DEV)
  poses = torch.rand(B, P, 3, generator=g).to(DEV)
  q_xy = torch.rand(B, Nq, 2, generator=g).to(DEV)
  none = torch.zeros(B, Nq, dtype=torch.bool, device=DEV)
  s = ops.pose_score(sim, poses, q_xy, none, None, 0.2)
  assert float(s.abs().max()) == 0.0
  one = none.clone()
  one[0, 37] = True                       # exactly one valid point
  s1 = ops.pose_score(sim, poses, q_xy, one, None, 0.2)
  want = oracle_ops.pose_score(sim.cpu(), poses.cpu(), q_xy.cpu(), one.cpu(), None, 0.2)
  helpers.report('one valid point', s1, want, atol=1e-6, rtol=1e-6)


def test_lift_with_nothing_visible():
  """All voxels far above every (horizontally looking) camera: pooled rows are zero, nothing
  is valid, and the masked-row MLP on top of it returns zeros (empty row list)."""
  fd, nb = 32, 8
  f, cam, Rt, pts = _lift_scene(1, 3, 12, 16, fd, nb, 70000, seed=77)
  pts = pts.clone()
  pts[..., 2] = 500.0                     # straight up: outside every vertical field of view
  pooled, valid = ops.lift_pool(f.to(DEV), cam.to(DEV), Rt.to(DEV), pts.to(DEV), K=2, fisheye=True,
                                feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0))
  assert not bool(valid.any()) and float(pooled.abs().max()) == 0.0
  index, count = ops.compact_rows(valid)
  assert int(count.item()) == 0
  w = torch.randn(pooled.shape[-1], 64, generator=torch.Generator().manual_seed(2)).to(DEV)
  y = ops.dense(pooled.reshape(-1, pooled.shape[-1]), w, None, rows_in=index, rows_out=index, row_count=count)
  ops.fill_masked_rows_(y, valid.reshape(-1))
  assert float(y.abs().max()) == 0.0


# ----------------------------------------------------------------------------
# ViT encoder pieces (BASELINE.json configs[4]; no reference implementation exists, the oracle
# is the published architecture in float64 -- oracle/vit.py)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize('M,C', [(37, 768), (5, 192), (130, 1024), (9, 64), (1, 4)])
def test_layer_norm(M, C):
  x = rnd((M, C), 201) * 1.7 + 0.3
  gamma, beta = rnd((C,), 202) * 0.3 + 1, rnd((C,), 203) * 0.2
  got, want = both('layer_norm', (x, gamma, beta))
  helpers.report(f'layer_norm {M}x{C}', got, want, atol=2e-5, rtol=1e-5)


def test_dense_gelu_and_residual_epilogues():
  x = rnd((300, 192), 204)
  w = rnd((192, 256), 205, 1 / np.sqrt(192))
  b = rnd((256,), 206)
  res = rnd((300, 256), 207)
  got, want = both('dense', (x, w, b), dict(gelu=True))
  helpers.report('dense + gelu', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('dense', (x, w, b), dict(residual=res))
  helpers.report('dense + residual', got, want, atol=2e-5, rtol=1e-5)
  got, want = both('dense', (x, w, b), dict(gelu=True, math='bf16'))
  helpers.report('dense + gelu bf16', got, want, atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize('B,N,H', [(2, 200, 3), (1, 1024, 2), (3, 64, 1), (1, 129, 12), (2, 1, 1), (1, 7, 2)])
def test_attention(B, N, H):
  """softmax(q k^T / 8) v on the bf16 matrix cores vs float64 on the bf16-rounded operands.
  Tolerance: 4e-3 of the value range (the kernel also rounds its probabilities to bf16,
  relative error 2^-9 each, averaged by the softmax)."""
  qkv = rnd((B, N, 3, H, 64), 210 + N)
  qkv[:, :, 0] *= 2.0                                  # sharper softmax than unit variance
  got, want = both('attention', (qkv,))
  scale = float(qkv[:, :, 2].abs().max())
  helpers.report(f'attention B{B} N{N} H{H}', got, want, atol=4e-3 * scale, rtol=0)
  exact = oracle_ops.attention(qkv, bf16_operands=False)
  helpers.report(f'attention vs exact f64 B{B} N{N} H{H}', got, exact, atol=2e-2 * scale, rtol=0)


def _dev(t):
  return t.to(DEV).contiguous()


@pytest.mark.parametrize('M,K,N', [(300, 192, 256), (1000, 768, 100), (2500, 32, 2304), (257, 3072, 768), (20, 16, 4)])
def test_bf16_ring_engine_is_the_half_input_engine_bit_for_bit(M, K, N):
  """``conv2d(bf16_ring=True)`` (conv_ps.hip, one part: a bf16 input, both operands by LDS-DMA through the ring):
  a 1 x 1 launch gives the bits of the x_half engine (without its split-K) -- plain, bias + GELU, residual, bf16-only
  output -- on row / column counts that leave partial tiles; both agree with float64 on the bf16-rounded operands."""
  x = _dev(rnd((M, K), 220 + M))
  w = _dev(rnd((K, N), 221, 1 / np.sqrt(K)))
  b = _dev(rnd((N,), 222))
  res = _dev(rnd((M, N), 223))
  xb = x.to(torch.bfloat16)
  ref = xb.double() @ w.to(torch.bfloat16).double() + b.double()
  for kw in (dict(), dict(gelu=True), dict(residual=res), dict(relu=True)):
    ring = ops.dense(xb, w, b, math='bf16', bf16_ring=True, **kw)
    with ops.tuning_scope(USE_SPLITK=False):                  # (the ring never splits K: the same order of the k-steps)
      half = ops.dense(xb, w, b, math='bf16', **kw)
    assert torch.equal(ring, half), (kw, float((ring - half).abs().max()))
  ring = ops.dense(xb, w, b, math='bf16', bf16_ring=True)
  helpers.report(f'bf16 ring {M}x{K}x{N}', ring, ref.float(), atol=3e-5 * float(ref.abs().max()) + 1e-6, rtol=0)
  # the bf16-only output is the f32 output rounded
  rh = ops.dense(xb, w, b, math='bf16', bf16_ring=True, gelu=True, out_half=True)
  rf = ops.dense(xb, w, b, math='bf16', bf16_ring=True, gelu=True)
  assert rh.dtype == torch.bfloat16 and torch.equal(rh, rf.to(torch.bfloat16))
  with ops.tuning_scope(BF16_PS=False):                       # (the switch: the call falls back to the x_half engine)
    assert torch.equal(ops.dense(xb, w, b, math='bf16', bf16_ring=True), ops.dense(xb, w, b, math='bf16'))


def test_bf16_ring_engine_3x3_conv():
  """The same engine under a padded 3 x 3 convolution (taps outside the image read zeros through the buffer range
  check): equal to the x_half engine to summation order."""
  x = _dev(rnd((3, 13, 11, 64), 230)).to(torch.bfloat16)
  w = _dev(rnd((3, 3, 64, 96), 231, 1 / 24.0))
  pad = ((1, 1), (1, 1))
  ring = ops.conv2d(x, w, padding=pad, math='bf16', bf16_ring=True)
  half = ops.conv2d(x, w, padding=pad, math='bf16')
  helpers.report('bf16 ring 3x3', ring, half, atol=2e-5 * float(half.abs().max()), rtol=0)


def test_bf16_outputs_of_layer_norm_and_attention_are_the_rounded_f32_outputs():
  x = _dev(rnd((130, 768), 240) * 1.7 + 0.3)
  gamma, beta = _dev(rnd((768,), 241) * 0.3 + 1), _dev(rnd((768,), 242) * 0.2)
  assert torch.equal(ops.layer_norm(x, gamma, beta, out_half=True), ops.layer_norm(x, gamma, beta).to(torch.bfloat16))
  qkv = _dev(rnd((2, 200, 3, 3, 64), 243))
  assert torch.equal(ops.attention(qkv, out_half=True), ops.attention(qkv).to(torch.bfloat16))
  # a bf16 qkv: K / V are the values the f32 kernel rounds to; Q is rounded once more (before its scaling)
  qh = qkv.to(torch.bfloat16)
  got = ops.attention(qh, out_half=True)
  want = oracle_ops.attention(qh.float().cpu(), bf16_operands=True)
  helpers.report('attention bf16 qkv', got.float(), want, atol=8e-3 * float(qkv[:, :, 2].abs().max()), rtol=0)


# ----------------------------------------------------------------------------
# generic grid operators (snap/utils/grids.py:116-153)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize('shape,D', [((9,), 3), ((7, 5), 1), ((6, 8), 5), ((4, 5, 3), 2)])
@pytest.mark.parametrize('with_valid', [False, True])
def test_interpolate_nd_matches_the_oracle(shape, D, with_valid):
  from oracle import grids as o_grids
  from snap_amd.utils import grids as g
  n = len(shape)
  rng = np.random.default_rng(7 + n + D)
  arr = rng.standard_normal((*shape, D)).astype(np.float32)
  K = 400
  pts = rng.uniform(-1.5, np.asarray(shape) + 1.5, (K, n)).astype(np.float32)
  # exact grid lines / centres / borders: zero-weight taps and the in-bounds test
  pts[:40] = np.round(pts[:40] * 2) / 2
  pts[40] = 0.0
  pts[41] = np.asarray(shape, np.float32)               # == size: out of bounds
  pts[42] = np.nextafter(np.asarray(shape, np.float32), 0).astype(np.float32)
  va = (rng.uniform(size=shape) > 0.25) if with_valid else None
  want_v, want_ok = o_grids.interpolate_nd(arr, pts, va)
  got_v, got_ok = g.interpolate_nd(torch.from_numpy(arr).to(DEV), torch.from_numpy(pts).to(DEV),
                                   None if va is None else torch.from_numpy(va).to(DEV))
  assert np.array_equal(got_ok.cpu().numpy(), want_ok)
  helpers.report(f'interpolate_nd {shape}x{D}', got_v, want_v, atol=2e-6, rtol=1e-6)
  with pytest.raises(NotImplementedError):
    g.interpolate_nd(torch.from_numpy(arr).to(DEV), torch.from_numpy(pts).to(DEV), order=0)


def test_argmax_nd_and_expectation_nd():
  from snap_amd.utils import grids as g
  rng = np.random.default_rng(11)
  for extent in ((17,), (6, 9), (5, 4, 7)):
    grid = g.GridND(tuple(extent), 0.5)
    sc = rng.standard_normal((3, 2, *extent)).astype(np.float32)
    flat = sc.reshape(3, 2, -1)
    flat[0, 0, 5] = flat[0, 0, 11] = flat.max() + 1.0          # a tie: the FIRST index wins
    want = np.stack(np.unravel_index(np.argmax(flat, -1), extent), -1)
    got = g.argmax_nd(torch.from_numpy(sc).to(DEV), grid)
    assert np.array_equal(got.cpu().numpy(), want)
    pdf = np.exp(sc)
    pdf /= pdf.reshape(3, 2, -1).sum(-1).reshape(3, 2, *([1] * len(extent)))
    idx = np.stack(np.meshgrid(*[np.arange(e) for e in extent], indexing='ij'), -1)
    want_e = (idx * pdf[..., None].astype(np.float64)).sum(tuple(range(2, 2 + len(extent))))
    got_e = g.expectation_nd(torch.from_numpy(pdf.astype(np.float32)).to(DEV), grid)
    assert got_e.shape == (3, 2, len(extent))
    helpers.report(f'expectation_nd {extent}', got_e, want_e.astype(np.float32), atol=1e-4, rtol=1e-5)


# ----------------------------------------------------------------------------
# confidence-weighted matching (bev_mapper.py:154-157,292-295; bev_localizer.py:165-168)
# ----------------------------------------------------------------------------
def test_confidence_head_and_masked_softmax_rows():
  from oracle import bev as o_bev
  rng = np.random.default_rng(21)
  f = rng.standard_normal((2, 9, 7, 128)).astype(np.float32)
  v = rng.uniform(size=(2, 9, 7)) > 0.3
  w = (rng.standard_normal(128) / 11).astype(np.float32)
  b = 0.37
  got = ops.confidence_head(torch.from_numpy(f).to(DEV), torch.from_numpy(v).to(DEV),
                            torch.from_numpy(w).to(DEV), b)
  want = np.where(v, o_bev.log_sigmoid((f @ w + np.float32(b)).astype(np.float32)), 0)
  helpers.report('confidence head', got, want, atol=2e-6, rtol=1e-5)
  x = (rng.standard_normal((3, 1000)) * 3).astype(np.float32)
  m = rng.uniform(size=(3, 1000)) > 0.4
  m[1] = False                                        # nothing valid: behaves as all-true
  wgt, cdf = ops.masked_softmax_rows(torch.from_numpy(x).to(DEV), torch.from_numpy(m).to(DEV))
  want_w = o_bev.layers_masked_softmax(x.astype(np.float64), m, -1)
  helpers.report('masked softmax rows', wgt, want_w.astype(np.float32), atol=1e-8, rtol=2e-5)
  helpers.report('masked softmax cdf', cdf, np.cumsum(want_w, -1).astype(np.float32), atol=2e-6, rtol=2e-5)
  assert float(wgt[0][~torch.from_numpy(m[0]).to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize('mfma', ['1', '0'])
def test_sim_softmax_with_confidence_weights(mfma):
  from oracle import pose as o_pose
  place = (lambda t: t) if mfma == '1' else _unaligned      # '0': the VALU kernel (unaligned inputs)
  B, Nq, X, Y, Dm = 2, 70, 13, 21, 32
  fq = _unit(rnd((B, Nq, Dm), 290))
  fm = _unit(rnd((B, X, Y, Dm), 291))
  g = torch.Generator().manual_seed(292)
  wts = torch.softmax(torch.randn(B, Nq, generator=g), -1)
  wts[0, 5] = 0.0
  nv = torch.tensor([70.0, 70.0])
  scale = float(np.exp(2.0))
  sim, stats, prob, _ = ops.sim_softmax(place(fq.to(DEV)), place(fm.to(DEV)), scale, True, nv.to(DEV),
                                        want_prob=True, row_weight=wts.to(DEV).contiguous())
  want_sim, want_prob = o_pose.similarity(fq.numpy(), fm.numpy(), np.ones((B, Nq), bool), 2.0, True,
                                          wts.numpy()[..., None, None])
  helpers.report('weighted sim', sim, want_sim, atol=1e-7, rtol=1e-5)
  helpers.report('weighted prob', prob, want_prob, atol=1e-10, rtol=1e-4)
  # the chunk statistics describe the UN-weighted row softmax: unchanged by the weights
  _, stats0, _, _ = ops.sim_softmax(place(fq.to(DEV)), place(fm.to(DEV)), scale, True, nv.to(DEV))
  assert torch.equal(stats, stats0)


def test_ransac_sample_rows_follow_the_confidence_cdf():
  B, Nq, X, Y, Dm, S = 2, 50, 12, 10, 16, 4000
  fq = _unit(rnd((B, Nq, Dm), 295)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 296)).to(DEV)
  nv = torch.full((B,), float(Nq), device=DEV)
  scale = float(np.exp(2.0))
  conf = rnd((B, Nq), 297).to(DEV) * 2
  mask = torch.ones(B, Nq, dtype=torch.bool, device=DEV)
  mask[:, ::3] = False
  w, cdf = ops.masked_softmax_rows(conf.contiguous(), mask)
  _, stats, _, _ = ops.sim_softmax(fq, fm, scale, True, nv, row_weight=w)
  u = torch.rand((B, S, 2), generator=torch.Generator().manual_seed(298)).to(DEV)
  corr = ops.ransac_sample(fq, fm, stats, scale, True, S, uniforms=u, row_cdf=cdf)
  # injected uniforms: the selected row is exactly the inverse CDF of u1
  want_rows = torch.searchsorted(cdf, (u[..., 0] * cdf[:, -1:]).contiguous(), right=True).clamp(max=Nq - 1)
  assert torch.equal(corr[..., 0].long(), want_rows)
  assert not bool((~mask)[torch.arange(B, device=DEV)[:, None], corr[..., 0].long()].any())   # never a masked point
  # Philox draws: row frequencies follow the weights (z-score on every valid row)
  S2 = 200000
  c2 = ops.ransac_sample(fq, fm, stats, scale, True, S2, seed=9, row_cdf=cdf)
  for bb in range(B):
    cnt = torch.bincount(c2[bb, :, 0].long(), minlength=Nq).double().cpu().numpy()
    p = w[bb].double().cpu().numpy()
    z = (cnt - S2 * p) / np.sqrt(np.maximum(S2 * p * (1 - p), 1e-9))
    assert np.abs(z[p > 0]).max() < 5.0
    assert cnt[p == 0].sum() == 0


def test_template_matching_without_padding_is_the_zero_extended_correlation():
  """pose_exhaustive_voting.py:82-91 with do_padding=False: jax.scipy.signal.convolve(mode='full')
  of the flipped templates == the cross-correlation over the map extended by zeros; checked
  against scipy.signal.convolve(method='direct') itself."""
  import scipy.signal
  from snap_amd.models import pose_exhaustive_voting as pev
  rng = np.random.default_rng(31)
  R, H, D = 8, 12, 4
  q = rng.standard_normal((R, H, H, D)).astype(np.float32)
  qv = rng.uniform(size=(R, H, H)) > 0.2
  q = q * qv[..., None]
  m = rng.standard_normal((H, H, D)).astype(np.float32)
  mv = np.ones((H, H), bool)
  got = pev.template_matching(torch.from_numpy(q).to(DEV), torch.from_numpy(qv).to(DEV),
                              torch.from_numpy(m).to(DEV), torch.from_numpy(mv).to(DEV),
                              do_padding=False, min_overlap=None)
  want = np.zeros((R, 2 * H - 1, 2 * H - 1))
  for r in range(R):
    for d in range(D):
      want[r] += scipy.signal.convolve(q[r, ::-1, ::-1, d].astype(np.float64), m[..., d].astype(np.float64),
                                       mode='full', method='direct')
    want[r] /= qv[r].sum()
  helpers.report('template matching, no padding', got, want.astype(np.float32), atol=2e-5, rtol=1e-5)
  with pytest.raises(ValueError):
    pev.template_matching(torch.from_numpy(q).to(DEV), torch.from_numpy(qv).to(DEV),
                          torch.from_numpy(m).to(DEV), torch.from_numpy(mv).to(DEV), do_padding=False)


@pytest.mark.parametrize('X,Y,Nq,B,rad', [(64, 64, 300, 2, 9), (40, 72, 77, 3, 30), (256, 256, 500, 1, 39), (33, 36, 50, 1, 4)])
def test_pose_score_window_is_pose_score_bit_for_bit(X, Y, Nq, B, rad):
  """snap_pose_score_window_f32: poses clustered around one centre pose per scene, scored from one window
  of every point's plane -- the bits of snap_pose_score_f32 (same sample arithmetic, same order of sums):
  windows clipped by the plane's borders, windows as large as the plane, invalid points, a scene whose centre
  lies outside the map."""
  cell = 0.2
  g = torch.Generator().manual_seed(X + Nq)
  sim = torch.randn((B, Nq, X, Y), generator=g).to(DEV)
  q_xy = ((torch.rand((B, Nq, 2), generator=g) - 0.5) * 6.0).to(DEV)
  qn = float(q_xy.norm(dim=-1).max())
  centers = torch.stack([torch.rand(B, generator=g) * 6.28, torch.rand(B, generator=g) * X * cell,
                         torch.rand(B, generator=g) * Y * cell], -1)
  centers[0, 1:] = torch.tensor([-1.0, Y * cell + 0.7])          # (a centre outside the map)
  # poses within the promised radius: |dt| + |q| |da| <= (rad - 1) cells
  P = 3000
  budget = (rad - 1) * cell
  da = (torch.rand(B, P, generator=g) - 0.5) * 2 * min(0.3 * budget / max(qn, 1e-3), 0.5)
  room = budget - qn * da.abs().max()
  ang = torch.rand(B, P, generator=g) * 6.28
  rr = torch.rand(B, P, generator=g) * float(room)
  poses = torch.stack([centers[:, None, 0] + da, centers[:, None, 1] + rr * torch.cos(ang),
                       centers[:, None, 2] + rr * torch.sin(ang)], -1).contiguous().to(DEV)
  vq = (torch.rand(B, Nq, generator=g) > 0.2).to(DEV)
  assert ops.pose_score_window_supported(X, Y, rad)
  want = ops.pose_score(sim, poses, q_xy, vq, None, cell)
  got = ops.pose_score_window(sim, poses, centers.to(DEV), rad, q_xy, vq, cell)
  assert torch.equal(got, want), float((got - want).abs().max())
  assert not ops.pose_score_window_supported(2048, 2048, 400)     # (a window beyond the LDS buffers)


def test_ransac_sample_reading_the_chunk_scores_from_sim():
  """The sampler may read the selected chunk's 64 scores from the sim tensor (x = sim * num_valid,
  one rounding away from the re-evaluated dot products) instead of recomputing them: same
  distribution, and with injected uniforms the same correspondences except where a CDF step falls
  inside that rounding."""
  B, Nq, X, Y, Dm, S = 2, 90, 24, 20, 32, 20000
  fq = _unit(rnd((B, Nq, Dm), 395)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 396)).to(DEV)
  nv = torch.tensor([88.0, 90.0], device=DEV)
  scale = float(np.exp(2.0))
  sim, stats, _, _ = ops.sim_softmax(fq, fm, scale, True, nv)
  u = torch.rand((B, S, 2), generator=torch.Generator().manual_seed(397)).to(DEV)
  a = ops.ransac_sample(fq, fm, stats, scale, True, S, uniforms=u)
  unscale = nv[:, None].expand(B, Nq).contiguous()
  b = ops.ransac_sample(fq, fm, stats, scale, True, S, uniforms=u, sim=sim, row_unscale=unscale)
  assert torch.equal(a[..., 0], b[..., 0])
  same = (a == b).all(-1).float().mean()
  assert float(same) > 0.999, float(same)
  # where they differ, they differ by ONE cell within the chunk (a CDF boundary)
  diff = (a != b).any(-1)
  if bool(diff.any()):
    ca = a[diff][:, 1].long() * Y + a[diff][:, 2].long()
    cb = b[diff][:, 1].long() * Y + b[diff][:, 2].long()
    assert int((ca - cb).abs().max()) <= 1


@pytest.mark.parametrize('X,Y,Nq,S', [(24, 20, 90, 20001), (128, 128, 300, 4003), (256, 256, 64, 1000)])
def test_ransac_four_per_wave_kernel_draws_the_same_samples(X, Y, Nq, S, monkeypatch):
  """The sampler kernel that runs four correspondences per wave in lock step (table + sim path)
  is a scheduling change only: bit-identical correspondences, with the generator and with
  injected uniforms, sample counts that are not multiples of the 16 per workgroup."""
  B, Dm = 2, 32
  fq = _unit(rnd((B, Nq, Dm), 495)).to(DEV)
  fm = _unit(rnd((B, X, Y, Dm), 496)).to(DEV)
  nv = torch.tensor([float(Nq - 2), float(Nq)], device=DEV)
  scale = float(np.exp(2.0))
  sim, stats, _, _ = ops.sim_softmax(fq, fm, scale, True, nv)
  unscale = nv[:, None].expand(B, Nq).contiguous()
  u = torch.rand((B, S, 2), generator=torch.Generator().manual_seed(497)).to(DEV)
  for kw in (dict(uniforms=u), dict(seed=1234)):
    # without the per-row table the library takes the one-correspondence-per-wave kernel
    a = ops.ransac_sample(fq, fm, stats, scale, True, S, sim=sim, row_unscale=unscale, row_table=False, **kw)
    b = ops.ransac_sample(fq, fm, stats, scale, True, S, sim=sim, row_unscale=unscale, **kw)
    assert torch.equal(a, b)
