"""`-m gpu` tests of the pre-split conv engine (conv_ps.hip) and its producers (presplit.hip).

The engine multiplies the operands conv_split.hip would have built on the fly (same split, same
k order, same product order per accumulator), so against the im2col body of that engine the
results are compared BIT FOR BIT; against the numpy oracle with the split engines' tolerance.
"""
import numpy as np
import pytest
import torch

import helpers
import oracle_ops
from snap_amd import ops

pytestmark = pytest.mark.gpu

DEV = helpers.DEVICE
TOL = 2.5e-4       # 'bf16x3' (2 parts, 3 products, ~2^-17 per product): as test_gpu_kernels.SPLIT_TOL x 2.5


def rnd(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _engine():
  prev = ops.MATMUL_PRECISION
  ops.MATMUL_PRECISION = 'bf16x3'
  yield
  ops.MATMUL_PRECISION = prev


def decode(ps):
  """PreSplit -> (hi, lo) f32 tensors [N, H, W, C]."""
  N, H, W, C = ps.shape
  d = ps.data.view(N * H * W, C // 16, 2, 16).float()
  return d[:, :, 0].reshape(N, H, W, C), d[:, :, 1].reshape(N, H, W, C)


def split_ref(v):
  hi = v.to(torch.bfloat16).float()
  lo = (v - hi).to(torch.bfloat16).float()
  return hi, lo


def test_presplit_is_the_two_part_split():
  x = (rnd((3, 7, 5, 48), 1) * torch.logspace(-6, 6, 48)).to(DEV)
  ps = ops.presplit(x)
  hi, lo = decode(ps)
  hr, lr = split_ref(x)
  assert torch.equal(hi, hr) and torch.equal(lo, lr)
  # rows that are not a multiple of the workgroup, C = 16
  x2 = rnd((1, 1, 1001, 16), 2).to(DEV)
  h2, l2 = decode(ops.presplit(x2))
  hr, lr = split_ref(x2)
  assert torch.equal(h2, hr) and torch.equal(l2, lr)


def _producer(N, H, W, C, seed):
  """A conv output [N, H, W, C] carrying fused GroupNorm partial sums, and its GroupNorm params."""
  x0 = rnd((N, H, W, 32), seed).to(DEV)
  w0 = rnd((1, 1, 32, C), seed + 1, 1 / np.sqrt(32.0)).to(DEV)
  ops.USE_SPLITK = False
  try:
    y = ops.conv2d(x0, w0, emit_gn_stats='raw')
  finally:
    ops.USE_SPLITK = True
  assert hasattr(y, '_snap_gn_partial')
  gamma = (rnd((C,), seed + 2) * 0.3 + 1).to(DEV)
  beta = (rnd((C,), seed + 3) * 0.2).to(DEV)
  return y, gamma, beta


@pytest.mark.parametrize('N,H,W,C', [(2, 40, 36, 64), (3, 17, 17, 512), (1, 136, 136, 64), (5, 12, 12, 128)])
def test_gn_norm_split_is_the_fused_prologue(N, H, W, C):
  y, gamma, beta = _producer(N, H, W, C, 10 + C)
  ps = ops.gn_norm_split(y, gamma, beta, want_stats=True)
  assert ps is not None and ps.shape == (N, H, W, C)
  mu, sc = ps.stats
  mu_f, sc_f = ops.group_norm_stats(y, gamma)        # the finalize launch it stands in for
  helpers.report('norm_split mu', mu, mu_f, atol=1e-7, rtol=1e-6)
  helpers.report('norm_split sc', sc, sc_f, atol=1e-7, rtol=1e-6)
  mu_w, sc_w = oracle_ops.group_norm_stats(y.cpu(), gamma.cpu())
  helpers.report('norm_split mu vs oracle', mu, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('norm_split sc vs oracle', sc, sc_w, atol=1e-5, rtol=5e-5)
  # the element arithmetic of apply_pro<GN_RELU>: (v - mu) * sc + beta, no contraction
  p = torch.clamp_min((y - mu[:, None, None, :]) * sc[:, None, None, :] + beta, 0.0)
  hi, lo = decode(ps)
  hr, lr = split_ref(p)
  assert torch.equal(hi, hr) and torch.equal(lo, lr)


PS_CASES = [
    # N, H, W, Cin, k, Cout, stride, residual
    (2, 40, 36, 64, 1, 256, 1, True),       # the closing 1x1 of a unit (+ residual)
    (2, 40, 36, 64, 1, 64, 1, False),       # 64-wide column tile
    (1, 20, 96, 32, 3, 64, 1, False),       # 3x3 (W > 79: conv_split takes its im2col body too)
    (2, 20, 96, 32, 3, 128, 1, False),
    (2, 41, 37, 64, 3, 128, 2, False),      # stride 2, ragged sizes
    (3, 23, 19, 128, 1, 512, 1, True),      # ragged M tail
]


@pytest.mark.parametrize('ps_tile', [1, 2])
@pytest.mark.parametrize('N,H,W,Cin,k,Cout,stride,res', PS_CASES)
def test_conv_ps_equals_conv_split_bitwise(N, H, W, Cin, k, Cout, stride, res, ps_tile):
  y, gamma, beta = _producer(N, H, W, Cin, 100 + Cout + k)
  w = rnd((k, k, Cin, Cout), 7, 1 / np.sqrt(k * k * Cin)).to(DEV)
  pad = k // 2
  ps = ops.gn_norm_split(y, gamma, beta, want_stats=True)
  mu, sc = ps.stats
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
  kw = dict(stride=stride, padding=((pad, pad), (pad, pad)))
  if res:
    kw['residual'] = rnd((N, Ho, Wo, Cout), 8).to(DEV)
  ops.USE_SPLITK = False
  try:
    want = ops.conv2d(y, w, prologue=ops.PRO_GN_RELU, gn=(mu, sc, beta), **kw)
    got = ops.conv2d(ps, w, res_init=False, ps_tile=ps_tile, **kw)
    assert torch.equal(got, want)
    if res:
      got_r = ops.conv2d(ps, w, res_init=True, ps_tile=ps_tile, **kw)
      helpers.report('conv_ps residual in the accumulators', got_r, want, atol=2e-6, rtol=2e-6)
  finally:
    ops.USE_SPLITK = True
  # ... and the oracle, on the normalised input
  p = torch.clamp_min((y - mu[:, None, None, :]) * sc[:, None, None, :] + beta, 0.0)
  ref = oracle_ops.conv2d(p.cpu(), w.cpu(), **{kk: (v.cpu() if torch.is_tensor(v) else v) for kk, v in kw.items()})
  helpers.report('conv_ps vs oracle', got, ref, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize('ps_tile', [1, 2])
@pytest.mark.parametrize('emit', ['raw', 'both'])
def test_conv_ps_output_statistics(emit, ps_tile):
  """The statistics of the OUTPUT (and of relu(output): 'both') from the pre-split engine's
  epilogue, row tiles straddling images, feed a second gn_norm_split."""
  N, H, W, Cin, Cout = 3, 31, 29, 64, 256
  y, gamma, beta = _producer(N, H, W, Cin, 300)
  w = rnd((1, 1, Cin, Cout), 301, 1 / 8.0).to(DEV)
  ps = ops.gn_norm_split(y, gamma, beta)
  res = rnd((N, H, W, Cout), 302).to(DEV)
  z = ops.conv2d(ps, w, residual=res, emit_gn_stats=emit, ps_tile=ps_tile)
  assert hasattr(z, '_snap_gn_partial')
  assert hasattr(z, '_snap_gn_partial_relu') == (emit == 'both')
  g2 = (rnd((Cout,), 303) + 1).to(DEV)
  for relu_first in ((False, True) if emit == 'both' else (False,)):
    mu_f, sc_f = ops.group_norm_stats(z, g2, relu_first=relu_first)
    mu_w, sc_w = oracle_ops.group_norm_stats(z.cpu(), g2.cpu(), relu_first=relu_first)
    helpers.report(f'conv_ps stats mu relu_first={relu_first}', mu_f, mu_w, atol=1e-5, rtol=1e-5)
    helpers.report(f'conv_ps stats sc relu_first={relu_first}', sc_f, sc_w, atol=1e-5, rtol=5e-5)
  b2 = (rnd((Cout,), 304) * 0.1).to(DEV)
  ps2 = ops.gn_norm_split(z, g2, b2, want_stats=True)
  mu_w, sc_w = oracle_ops.group_norm_stats(z.cpu(), g2.cpu())
  helpers.report('second norm_split mu', ps2.stats[0], mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('second norm_split sc', ps2.stats[1], sc_w, atol=1e-5, rtol=5e-5)


def test_conv_ps_split_k_and_plain_presplit():
  """Small M / deep K: split-K on the pre-split engine; a plain presplit() input (no GroupNorm)
  with bias + ReLU + row mask epilogues."""
  x = rnd((2, 8, 8, 512), 400).to(DEV)
  w = rnd((3, 3, 512, 128), 401, 1 / np.sqrt(9 * 512)).to(DEV)
  kw = dict(padding=((1, 1), (1, 1)))
  got = ops.conv2d(ops.presplit(x), w, **kw)
  ops.USE_SPLITK = False
  try:
    single = ops.conv2d(ops.presplit(x), w, **kw)
  finally:
    ops.USE_SPLITK = True
  want = oracle_ops.conv2d(x.cpu(), w.cpu(), **kw)
  helpers.report('conv_ps split-K vs oracle', got, want, atol=TOL, rtol=1e-5)
  helpers.report('conv_ps split-K vs single pass', got, single, atol=2e-5, rtol=1e-5)
  x2 = rnd((1, 1, 3000, 256), 402).to(DEV)
  w2 = rnd((1, 1, 256, 128), 403, 1 / 16.0).to(DEV)
  bias = rnd((128,), 404).to(DEV)
  mask = (torch.rand(3000, generator=torch.Generator().manual_seed(5)) > 0.4).to(DEV)
  got = ops.conv2d(ops.presplit(x2), w2, bias=bias, relu=True, row_mask=mask)
  want = ops.conv2d(x2, w2, bias=bias, relu=True, row_mask=mask)
  assert torch.equal(got, want)


@pytest.mark.parametrize('M,K,N', [(5000, 48, 192), (777, 80, 576), (4096, 256, 384)])
def test_conv_ps_256x192_tile(M, K, N):
  """The large-GEMM tile (256 x 192, two k-steps per ring stage, one workgroup per CU: the
  exhaustive voting's): odd numbers of k-steps (a half-empty last stage), ragged M, column tiles
  that straddle the 128-column blocks of the weight image, split-K."""
  x = rnd((1, 1, M, K), 600 + K).to(DEV)
  w = rnd((1, 1, K, N), 601, 1 / np.sqrt(K)).to(DEV)
  ops.USE_SPLITK = False
  try:
    want = ops.conv2d(x, w)                                   # conv_split, im2col body
    got = ops.conv2d(ops.presplit(x), w, ps_tile=3)
  finally:
    ops.USE_SPLITK = True
  assert torch.equal(got, want)
  got_k = ops.conv2d(ops.presplit(x), w, ps_tile=3)           # split-K where the shape asks for it
  helpers.report('256x192 split-K', got_k, want, atol=2e-5, rtol=1e-5)
  # a strided 5x5 correlation (the voting's shape class) against the oracle
  xi = rnd((1, 37, 41, 32), 602).to(DEV)
  wi = rnd((5, 5, 32, 192), 603, 1 / np.sqrt(800.0)).to(DEV)
  kw = dict(stride=4, padding=((0, 3), (0, 2)))
  got = ops.conv2d(ops.presplit(xi), wi, ps_tile=3, **kw)
  want = oracle_ops.conv2d(xi.cpu(), wi.cpu(), **kw)
  helpers.report('256x192 strided 5x5 vs oracle', got, want, atol=TOL, rtol=1e-5)


def test_resnet_unit_takes_the_presplit_path_and_matches_the_fused_one():
  """One bottleneck unit of the model code: pre-split path on vs off (same engine otherwise)."""
  from snap_amd.models import base, resnet
  gen = torch.Generator().manual_seed(3)
  p = resnet._init_unit(gen, 'cpu', 256, 64, 1)
  p = helpers.params_to_device(p, DEV)
  x = rnd((2, 34, 38, 256), 500).to(DEV)
  outs = {}
  for on in (True, False):
    ops.USE_PRESPLIT = on
    try:
      ops.PACK_EPOCH += 1
      ctx = base.ForwardContext()
      outs[on] = resnet.residual_unit(ctx, p, x, 1, 64)
    finally:
      ops.USE_PRESPLIT = True
  d = float((outs[True] - outs[False]).abs().max()) / float(outs[False].abs().max())
  print(f'unit presplit vs fused: {d:.2e}')
  assert d < 2e-5
