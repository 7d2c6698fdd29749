"""`-m gpu` tests of the stationary-operand 1 x 1 kernels (conv_rs.hip): row-stationary
(activation tile in registers) and weights-stationary (weight panel resident in LDS).

Both multiply exactly the operands the tiled split engine (conv_split.hip) builds -- same
GroupNorm arithmetic, same split, same slab order, same product order per accumulator, residual
added after the sum -- so against that engine (``ops.CONV_NO_RS = True``) the OUTPUT is compared
bit for bit; the GroupNorm statistics they emit are summed in another order (the
weights-stationary kernel: per 32-row slab) and are compared with the oracle.
"""
import numpy as np
import pytest
import torch

import helpers
import oracle_ops
from snap_amd import ops

pytestmark = pytest.mark.gpu

DEV = helpers.DEVICE
TOL = 2.5e-4


def rnd(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _engine():
  prev = ops.MATMUL_PRECISION
  ops.MATMUL_PRECISION = 'bf16x3'
  prev_tile = ops.CONV_TILE
  ops.CONV_TILE = '128x128'     # small test shapes: the 128-row tiles the big layers get by themselves
  ops.CONV_RS_FORCE = True      # ... and the kernel below its row-count threshold
  yield
  ops.MATMUL_PRECISION = prev
  ops.CONV_TILE = prev_tile
  ops.CONV_RS_FORCE = False
  ops.CONV_NO_RS = False
  ops.CONV_NO_WS = False
  ops.CONV_RAW_RING = False


def _layer(N, H, W, Cin, Cout, seed, residual):
  x = rnd((N, H, W, Cin), seed)
  w = rnd((1, 1, Cin, Cout), seed + 1, 1 / np.sqrt(Cin))
  res = rnd((N, H, W, Cout), seed + 2) if residual else None
  g_in = rnd((Cin,), seed + 3) * 0.3 + 1
  b_in = rnd((Cin,), seed + 4) * 0.3
  return x, w, res, g_in, b_in


def _kind(N, H, W, Cin, Cout, residual):
  """0 = tiled body, 1 = row-stationary, 2 = weights-stationary (under the current ops.CONV_* switches)."""
  from snap_amd import _lib
  import ctypes
  d = _lib.SnapConvDesc(N=N, H=H, W=W, Cin=Cin, Cin_stride=Cin, KH=1, KW=1, stride=1, pad_t=0, pad_l=0,
                        Ho=H, Wo=W, Cout=Cout, Cout_stride=Cout, prologue=ops.PRO_GN_RELU,
                        epilogue=ops.EPI_RESIDUAL if residual else 0, in_scale=1.0, in_shift=0.0,
                        tile_hint=128128 + 1000000 * ops._stationary_mode())
  return int(_lib.load().snap_conv2d_stationary_kind(ctypes.byref(d), 2))


def _takes_rs(*a):
  return _kind(*a) != 0


def _run(x, w, res, g_in, b_in, emit, no_rs, relu=False):
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV))
  ops.CONV_NO_RS = no_rs
  ops.USE_SPLITK = False      # (the tiled engine would split K = 256 on these small M: another sum order)
  try:
    y = ops.conv2d(xd, w.to(DEV), prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in.to(DEV)),
                   residual=None if res is None else res.to(DEV), relu=relu, emit_gn_stats=emit)
  finally:
    ops.CONV_NO_RS = False
    ops.USE_SPLITK = True
  return y


# (N, H, W, Cin, Cout, residual): row tiles that straddle images, a ragged last tile, one / several
# column tiles per workgroup, every Cin the kernel takes
CASES = [
    (3, 20, 23, 64, 256, True),
    (2, 31, 29, 64, 512, False),
    (7, 9, 11, 64, 256, True),           # 99 pixels per image: 32-row tiles straddle images all the time (ws only)
    (5, 16, 17, 128, 512, True),
    (2, 40, 37, 128, 256, False),
    (3, 23, 17, 256, 1024, True),
    (1, 70, 66, 256, 512, False),
    (40, 34, 34, 256, 1024, True),       # the C2 stage-3 expansion, full size
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,residual', CASES)
@pytest.mark.parametrize('emit', [None, 'raw', 'both'])
@pytest.mark.parametrize('kernel', ['ws', 'rs'])
def test_stationary_kernels_equal_the_tiled_engine(N, H, W, Cin, Cout, residual, emit, kernel):
  if emit is not None and N == 40:
    pytest.skip('one statistics variant at full size is enough')
  ops.CONV_NO_WS = kernel == 'rs'
  want = 1 if (kernel == 'rs' or Cin == 256 or Cout % 256) else 2
  if kernel == 'ws' and want == 1:
    pytest.skip('the weights-stationary kernel does not take this shape (covered by the rs case)')
  if kernel == 'rs' and H * W < 128:
    pytest.skip('a 128-row tile would touch more than two images')
  assert _kind(N, H, W, Cin, Cout, residual) == want
  layer = _layer(N, H, W, Cin, Cout, 1000 + Cin + Cout, residual)
  y_rs = _run(*layer, emit, no_rs=False)
  y_t = _run(*layer, emit, no_rs=True)
  assert torch.equal(y_rs, y_t), float((y_rs - y_t).abs().max())
  if emit is None:
    return
  assert hasattr(y_rs, '_snap_gn_partial')
  if emit == 'both':                    # (every shape: the tiled engine only at 128 x 128 tiles)
    assert hasattr(y_rs, '_snap_gn_partial_relu')
  gamma = rnd((Cout,), 7) * 0.3 + 1
  for relu_first in ((False, True) if emit == 'both' else (False,)):
    mu_f, sc_f = ops.group_norm_stats(y_rs, gamma.to(DEV), relu_first=relu_first)
    mu_w, sc_w = oracle_ops.group_norm_stats(y_rs.cpu(), gamma, relu_first=relu_first)
    helpers.report(f'rs stats mu relu_first={relu_first}', mu_f, mu_w, atol=1e-5, rtol=1e-5)
    helpers.report(f'rs stats sc relu_first={relu_first}', sc_f, sc_w, atol=1e-5, rtol=5e-5)


@pytest.mark.parametrize('kernel', ['ws', 'rs'])
def test_stationary_kernels_against_the_oracle_and_relu_epilogue(kernel):
  ops.CONV_NO_WS = kernel == 'rs'
  x, w, res, g_in, b_in = _layer(2, 19, 21, 128, 512, 77, True)
  y = _run(x, w, res, g_in, b_in, None, no_rs=False, relu=True)
  mu, sc = oracle_ops.group_norm_stats(x, g_in)
  want = oracle_ops.conv2d(x, w, prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in), residual=res, relu=True)
  helpers.report('rs conv vs oracle', y, want, atol=TOL * float(want.abs().max()))
  assert float(y.min()) >= 0


def test_shapes_outside_the_kernel_take_the_tiled_engine():
  """Cin = 512 / a bias / a stride: the dispatcher leaves them alone (same bits with the switch)."""
  assert not _takes_rs(2, 17, 17, 512, 1024, True)
  ops.CONV_RS_FORCE = False
  assert not _takes_rs(8, 34, 34, 256, 1024, True)      # too few rows to pay (the aerial encoder)
  assert _kind(40, 34, 34, 256, 1024, True) == 1        # Cin = 256: the panel does not fit LDS
  assert _kind(40, 68, 68, 128, 512, True) == 2
  ops.CONV_RS_FORCE = True
  ops.CONV_NO_WS = True
  assert not _takes_rs(3, 11, 11, 64, 256, True)        # fewer than 128 pixels per image (a row tile: two images at most)
  ops.CONV_NO_WS = False
  ops.CONV_RAW_RING = False
  assert not _takes_rs(3, 5, 5, 64, 256, True)          # ... fewer than 32 for the weights-stationary kernel
  assert not _takes_rs(3, 20, 20, 64, 128, True)        # a single column tile
  assert not _takes_rs(3, 20, 20, 256, 256, True)       # Cin = 256 pays from Cout = 512
  x, w, res, g_in, b_in = _layer(2, 17, 17, 512, 1024, 88, True)
  y_a = _run(x, w, res, g_in, b_in, None, no_rs=False)
  y_b = _run(x, w, res, g_in, b_in, None, no_rs=True)
  assert torch.equal(y_a, y_b)


# ---- the weights-stationary 3 x 3 of the first stage (64 -> 64 channels, images wider than the halo body takes)
CASES_3X3 = [
    (2, 20, 90),          # three full tiles per row
    (3, 11, 85),          # a 25-pixel last tile per row
    (2, 9, 91),           # a ONE-pixel last tile per row
    (1, 47, 136),
    (40, 136, 136),       # the C2 StreetView layer, full size
]


def _kind3(N, H, W):
  from snap_amd import _lib
  import ctypes
  d = _lib.SnapConvDesc(N=N, H=H, W=W, Cin=64, Cin_stride=64, KH=3, KW=3, stride=1, pad_t=1, pad_l=1,
                        Ho=H, Wo=W, Cout=64, Cout_stride=64, prologue=ops.PRO_GN_RELU, epilogue=0,
                        in_scale=1.0, in_shift=0.0, tile_hint=1000000 * ops._stationary_mode())
  return int(_lib.load().snap_conv2d_stationary_kind(ctypes.byref(d), 2))


@pytest.mark.parametrize('N,H,W', CASES_3X3)
@pytest.mark.parametrize('emit', [None, 'raw', 'relu'])
def test_weights_stationary_3x3_equals_the_im2col_body(N, H, W, emit):
  if emit is not None and N == 40:
    pytest.skip('one statistics variant at full size is enough')
  ops.CONV_TILE = None
  assert _kind3(N, H, W) == 3
  x = rnd((N, H, W, 64), 3000 + W)
  w = rnd((3, 3, 64, 64), 3001, 1 / 24.0)
  g_in = rnd((64,), 3002) * 0.3 + 1
  b_in = rnd((64,), 3003) * 0.3
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV))
  kw = dict(padding=((1, 1), (1, 1)), prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in.to(DEV)), emit_gn_stats=emit)
  y_ws = ops.conv2d(xd, w.to(DEV), **kw)
  ops.CONV_NO_RS = True
  ops.USE_SPLITK = False
  try:
    assert _kind3(N, H, W) == 0
    y_t = ops.conv2d(xd, w.to(DEV), **kw)
  finally:
    ops.CONV_NO_RS = False
    ops.USE_SPLITK = True
  assert torch.equal(y_ws, y_t), float((y_ws - y_t).abs().max())
  if N <= 3:
    want = oracle_ops.conv2d(x, w, padding=((1, 1), (1, 1)), prologue=ops.PRO_GN_RELU,
                             gn=(*oracle_ops.group_norm_stats(x, g_in), b_in))
    helpers.report('ws 3x3 vs oracle', y_ws, want, atol=TOL * float(want.abs().max()))
  if emit is None:
    return
  assert y_ws._snap_gn_partial[1] == -H * ((W + 29) // 30)      # one slab per row-aligned tile of 30 pixels
  gamma = rnd((64,), 3004) * 0.3 + 1
  mu_f, sc_f = ops.group_norm_stats(y_ws, gamma.to(DEV), relu_first=emit == 'relu')
  mu_w, sc_w = oracle_ops.group_norm_stats(y_ws.cpu(), gamma, relu_first=emit == 'relu')
  helpers.report('ws 3x3 stats mu', mu_f, mu_w, atol=1e-5, rtol=1e-5)
  helpers.report('ws 3x3 stats sc', sc_f, sc_w, atol=1e-5, rtol=5e-5)


# ---- the weights-stationary RGB root convolution (7 x 7 / stride 2 / pad 3, 64 output channels)
@pytest.mark.parametrize('N,H,W', [(2, 64, 96), (1, 50, 70), (3, 33, 130), (40, 544, 680)])
@pytest.mark.parametrize('affine', [True, False])
def test_weights_stationary_root_conv_equals_the_tiled_root_kernel(N, H, W, affine):
  if N == 40 and not affine:
    pytest.skip('one variant at full size is enough')
  ops.CONV_TILE = None
  x = torch.rand((N, H, W, 4), generator=torch.Generator().manual_seed(4000 + W))
  x[..., 3] = 0.37                                  # (must not matter: zero weights)
  w = rnd((7, 7, 3, 64), 4001, 1 / np.sqrt(147))
  kw = dict(stride=2, padding=((3, 3), (3, 3)), cin=3)
  if affine:
    kw.update(prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0))
  xd, wd = x.to(DEV), w.to(DEV)
  y_ws = ops.conv2d(xd, wd, **kw)
  ops.CONV_NO_RS = True
  try:
    y_t = ops.conv2d(xd, wd, **kw)
  finally:
    ops.CONV_NO_RS = False
  assert y_ws.shape == (N, (H + 1) // 2, (W + 1) // 2, 64)
  assert torch.equal(y_ws, y_t), float((y_ws - y_t).abs().max())
  if N <= 3:
    want = oracle_ops.conv2d(x, w, **kw)
    helpers.report('ws root vs oracle', y_ws, want, atol=TOL * float(want.abs().max()))


# ------------------------------------------------------------------------------------------
# conv_raw.hip: 1 x 1 layers with K >= 256 -- raw rows through an LDS ring, GroupNorm + split at
# fragment fetch.  Same arithmetic as the tiled (plain) body: bit for bit.
# ------------------------------------------------------------------------------------------
RAW_CASES = [
    # N, H, W, Cin, Cout, prologue, residual, relu, up_prev, tile
    (3, 34, 34, 1024, 256, 'gn_relu', False, False, False, '128x128'),    # stage-3 reduction (two column tiles)
    (2, 17, 17, 2048, 512, 'gn_relu', False, False, False, '128x128'),    # 289 pixels per image: tiles straddle two images
    (2, 17, 17, 512, 2048, 'gn_relu', True, False, False, '128x128'),     # expansion with a residual, K = 512
    (1, 23, 19, 256, 128, 'gn_relu', False, True, False, '128x64'),       # ragged last tile, 64-column tiles, ReLU epilogue
    (2, 16, 24, 256, 128, 'relu_gn', False, False, True, '128x128'),      # FPN skip conv: ReLU -> GroupNorm, upsample-add epilogue
    (40, 34, 34, 1024, 256, 'gn_relu', False, False, False, None),        # the C2 layer at full size, automatic tile
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,pro,residual,relu,up,tile', RAW_CASES)
@pytest.mark.parametrize('emit', [None, 'raw'])
def test_raw_row_ring_body_equals_the_tiled_body(N, H, W, Cin, Cout, pro, residual, relu, up, tile, emit):
  if emit is not None and (N == 40 or up):
    pytest.skip('one statistics variant per epilogue kind is enough')
  ops.CONV_TILE = tile
  ops.CONV_NO_RS = True                 # (the stationary kernels have their own tests above)
  relu_first = pro == 'relu_gn'
  prologue = ops.PRO_RELU_GN if relu_first else ops.PRO_GN_RELU
  x, w, res, g_in, b_in = _layer(N, H, W, Cin, Cout, 3000 + Cin + Cout, residual)
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV), relu_first=relu_first)
  upp = rnd((N, H // 2, W // 2, Cout), 3100).to(DEV) if up else None
  out = {}
  for no_raw in (False, True):
    ops.CONV_RAW_RING = not no_raw
    ops.USE_SPLITK = False
    try:
      out[no_raw] = ops.conv2d(xd, w.to(DEV), prologue=prologue, gn=(mu, sc, b_in.to(DEV)),
                               residual=None if res is None else res.to(DEV), relu=relu, up_prev=upp,
                               emit_gn_stats=emit)
    finally:
      ops.CONV_RAW_RING = False
      ops.USE_SPLITK = True
  y_raw, y_t = out[False], out[True]
  assert torch.equal(y_raw, y_t), float((y_raw - y_t).abs().max())
  assert float(y_raw.abs().max()) > 0
  mu_w, sc_w = oracle_ops.group_norm_stats(x, g_in, relu_first=relu_first)
  want = oracle_ops.conv2d(x, w, prologue=prologue, gn=(mu_w, sc_w, b_in), residual=res, relu=relu,
                           up_prev=None if upp is None else upp.cpu())
  helpers.report('raw-ring conv vs oracle', y_raw, want, atol=TOL * float(want.abs().max()))
  if emit is not None:
    assert hasattr(y_raw, '_snap_gn_partial')
    gamma = rnd((Cout,), 7) * 0.3 + 1
    mu_f, sc_f = ops.group_norm_stats(y_raw, gamma.to(DEV))
    mu_o, sc_o = oracle_ops.group_norm_stats(y_raw.cpu(), gamma)
    helpers.report('raw-ring stats mu', mu_f, mu_o, atol=1e-5, rtol=1e-5)
    helpers.report('raw-ring stats sc', sc_f, sc_o, atol=1e-5, rtol=5e-5)


def test_raw_row_ring_body_with_split_k_keeps_the_bits():
  """A small-M / deep-K launch that the engine splits along K (each split >= 16 k-steps): the ring body inside
  every split + the shared reduce pass = the tiled body's bits."""
  ops.CONV_TILE = '128x128'
  ops.CONV_NO_RS = True
  # 100 output tiles -> 8 splits of 16 k-steps each (conv_common.h: target 768 workgroups, >= 8 slabs per split)
  x, w, res, g_in, b_in = _layer(4, 40, 40, 2048, 256, 3333, False)
  xd = x.to(DEV)
  mu, sc = ops.group_norm_stats(xd, g_in.to(DEV))
  out = {}
  for no_raw in (False, True):
    ops.CONV_RAW_RING = not no_raw
    try:
      out[no_raw] = ops.conv2d(xd, w.to(DEV), prologue=ops.PRO_GN_RELU, gn=(mu, sc, b_in.to(DEV)))
    finally:
      ops.CONV_RAW_RING = False
  assert torch.equal(out[False], out[True]), float((out[False] - out[True]).abs().max())
