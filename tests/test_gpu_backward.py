"""`-m gpu` gradient parity: every backward HIP kernel vs torch autograd (CPU, fp64)
of a plain torch restatement of the same op (the fp reference for floating-point
kernels), on the same seeded inputs and cotangents."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers
import oracle_ops
from snap_amd import ops
from snap_amd import ops_bwd

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(shape, generator=g) * scale


def G(t):
  return t.to(DEV).contiguous()


# -- torch references (float64, differentiable) ---------------------------------------
def ref_gn(x, gamma, beta, groups, relu_first, relu_after):
  N, H, W, C = x.shape
  v = torch.relu(x) if relu_first else x
  g = v.reshape(N, H * W, groups, C // groups)
  mean = g.mean((1, 3), keepdim=True)
  var = ((g - mean) ** 2).mean((1, 3), keepdim=True)
  y = ((g - mean) / torch.sqrt(var + 1e-5)).reshape(N, H, W, C) * gamma + beta
  return torch.relu(y) if relu_after else y


def ref_conv(z, w, stride, pad):
  y = F.conv2d(z.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=stride, padding=pad)
  return y.permute(0, 2, 3, 1)


# -- wgrad / dgrad -----------------------------------------------------------------------
WG_CASES = [
    ('1x1', 2, 9, 7, 64, 1, 128, 1, 0, ops.PRO_NONE),
    ('3x3_gn', 2, 10, 9, 64, 3, 64, 1, 1, ops.PRO_GN_RELU),
    ('3x3_s2_gn', 2, 12, 10, 128, 3, 128, 2, 1, ops.PRO_GN_RELU),
    ('1x1_relu_gn', 1, 8, 8, 256, 1, 32, 1, 0, ops.PRO_RELU_GN),
    ('dense_k257', 1, 1, 500, 257, 1, 256, 1, 0, ops.PRO_NONE),
    ('root_affine', 1, 18, 16, 3, 7, 32, 2, 3, ops.PRO_AFFINE),
    ('1x1_relu', 1, 6, 6, 128, 1, 160, 1, 0, ops.PRO_RELU),
    ('big_m', 4, 40, 40, 64, 1, 256, 1, 0, ops.PRO_GN_RELU),
]


@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_conv_wgrad(case):
  _, N, H, W, Cin, k, Cout, stride, pad, pro = case
  cs = 260 if Cin == 257 else Cin
  x = torch.zeros(N, H, W, cs)
  x[..., :Cin] = rnd((N, H, W, Cin), 1) + 0.1
  w = rnd((k, k, Cin, Cout), 2, 1 / math.sqrt(k * k * Cin))
  gamma, beta = rnd((Cin,), 3) * 0.3 + 1, rnd((Cin,), 4) * 0.2
  Ho = (H + 2 * pad - k) // stride + 1
  Wo = (W + 2 * pad - k) // stride + 1
  dy = rnd((N, Ho, Wo, Cout), 5)
  xd = x[..., :Cin].double()
  if pro == ops.PRO_GN_RELU:
    z = ref_gn(xd, gamma.double(), beta.double(), 32, False, True)
  elif pro == ops.PRO_RELU_GN:
    z = ref_gn(xd, gamma.double(), beta.double(), 32, True, False)
  elif pro == ops.PRO_AFFINE:
    z = xd * 2 - 1
  elif pro == ops.PRO_RELU:
    z = torch.relu(xd)
  else:
    z = xd
  wd = w.double().requires_grad_(True)
  ref_conv(z, wd, stride, pad).backward(dy.double())
  gn = None
  if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
    mu, sc = oracle_ops.group_norm_stats(x[..., :Cin].contiguous(), gamma,
                                         relu_first=pro == ops.PRO_RELU_GN)
    gn = (G(mu), G(sc), G(beta))
  got = ops_bwd.conv2d_wgrad(G(x), G(dy), tuple(w.shape), stride=stride,
                             padding=((pad, pad), (pad, pad)), prologue=pro, gn=gn,
                             in_affine=(2.0, -1.0) if pro == ops.PRO_AFFINE else (1.0, 0.0))
  ref = wd.grad
  helpers.report('wgrad ' + case[0], got, ref.float(), atol=2e-4 * float(ref.abs().max()) + 1e-5)


@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_conv_wgrad_bf16(case):
  """bf16-operand wgrad engine vs the restatement that rounds prologue(x) and dy to bf16
  (oracle/encoder.py:bf16_round) and accumulates exactly; fp32 round-off class tolerance."""
  from oracle import encoder as o_enc
  _, N, H, W, Cin, k, Cout, stride, pad, pro = case
  cs = 260 if Cin == 257 else Cin
  x = torch.zeros(N, H, W, cs)
  x[..., :Cin] = rnd((N, H, W, Cin), 1) + 0.1
  gamma, beta = rnd((Cin,), 3) * 0.3 + 1, rnd((Cin,), 4) * 0.2
  Ho = (H + 2 * pad - k) // stride + 1
  Wo = (W + 2 * pad - k) // stride + 1
  dy = rnd((N, Ho, Wo, Cout), 5)
  gn = gn_cpu = None
  if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
    mu, sc = oracle_ops.group_norm_stats(x[..., :Cin].contiguous(), gamma,
                                         relu_first=pro == ops.PRO_RELU_GN)
    gn = (G(mu), G(sc), G(beta))
    gn_cpu = (mu, sc, beta)
  aff = (2.0, -1.0) if pro == ops.PRO_AFFINE else (1.0, 0.0)
  z32 = oracle_ops._prologue(x.numpy().astype(np.float32), pro, gn_cpu, aff, Cin)
  bf16_path = cs % 4 == 0 and Cin >= 4
  rz = o_enc.bf16_round(z32) if bf16_path else z32
  rdy = o_enc.bf16_round(dy.numpy()) if bf16_path else dy.numpy()
  wd = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
  ref_conv(torch.from_numpy(rz).double(), wd, stride, pad).backward(torch.from_numpy(rdy).double())
  got = ops_bwd.conv2d_wgrad(G(x), G(dy), (k, k, Cin, Cout), stride=stride,
                             padding=((pad, pad), (pad, pad)), prologue=pro, gn=gn,
                             in_affine=aff, math='bf16')
  ref = wd.grad
  helpers.report('wgrad bf16 ' + case[0], got, ref.float(), atol=2e-5 * float(ref.abs().max()) + 1e-5)


def test_conv_wgrad_bf16_rows():
  """Row lists + device-side row count on the bf16 wgrad engine."""
  from oracle import encoder as o_enc
  M, Cin, Cout = 5000, 260, 128
  x = rnd((M, Cin), 71)
  dy = rnd((M, Cout), 72)
  mask = torch.rand(M, generator=torch.Generator().manual_seed(73)) > 0.55
  index, count = ops.compact_rows(G(mask))
  got = ops_bwd.conv2d_wgrad(G(x).reshape(1, 1, M, Cin), G(dy).reshape(1, 1, M, Cout),
                             (1, 1, 257, Cout), rows_z=index, rows_dy=index, row_count=count,
                             math='bf16')
  zr = o_enc.bf16_round(x.numpy()[mask.numpy()][:, :257]).astype(np.float64)
  dr = o_enc.bf16_round(dy.numpy()[mask.numpy()]).astype(np.float64)
  ref = torch.from_numpy(zr.T @ dr).float().reshape(1, 1, 257, Cout)
  helpers.report('wgrad bf16 rows', got, ref, atol=2e-5 * float(ref.abs().max()) + 1e-5)


@pytest.mark.parametrize('Cin,Cs,Cout,relu', [(257, 260, 256, False), (256, 256, 128, True), (64, 64, 256, False)])
def test_conv_wgrad_bf16_tall_products(Cin, Cs, Cout, relu):
  """The fusion MLP's weight gradients at their real height (M > 2^19 rows, flat), with row lists and
  a device-side row count; vs the rounded-operand restatement in float64.  (256-wide output tiles on
  512 threads were tried for these launches in round 3: the byte counts halve, the time does not
  move -- 4.03 -> 4.33 ms on the largest --, so the 128 x 128 tiles stay.)"""
  from oracle import encoder as o_enc
  M = (1 << 19) + 1234
  x = rnd((M, Cs), 171)
  x[:, Cin:] = 0
  dy = rnd((M, Cout), 172)
  mask = torch.rand(M, generator=torch.Generator().manual_seed(173)) > 0.3
  index, count = ops.compact_rows(G(mask))
  got = ops_bwd.conv2d_wgrad(G(x).reshape(1, 1, M, Cs), G(dy).reshape(1, 1, M, Cout),
                             (1, 1, Cin, Cout), rows_z=index, rows_dy=index, row_count=count,
                             prologue=ops.PRO_RELU if relu else ops.PRO_NONE, math='bf16')
  z = x.numpy()[mask.numpy()][:, :Cin]
  if relu:
    z = np.maximum(z, 0)
  zr = o_enc.bf16_round(z.astype(np.float32)).astype(np.float64)
  dr = o_enc.bf16_round(dy.numpy()[mask.numpy()]).astype(np.float64)
  ref = torch.from_numpy(zr.T @ dr).float().reshape(1, 1, Cin, Cout)
  helpers.report('wgrad bf16 tall', got, ref, atol=2e-5 * float(ref.abs().max()) + 1e-4)


@pytest.mark.parametrize('math_', ['f32', 'bf16', 'fp16'])
def test_conv_wgrad_and_dgrad_random_shapes_fuzz(math_):
  """16 seeded random (shape, stride, padding, prologue) combinations per engine: kernel and data
  gradients vs torch fp64 autograd of the plain restatement (operands rounded to bf16 first for
  the bf16 engine, so the tolerance stays in the f32 round-off class)."""
  from oracle import encoder as o_enc
  from snap_amd import autograd as ag
  rng = np.random.default_rng({'f32': 3030, 'bf16': 3031, 'fp16': 3032}[math_])
  for it in range(16):
    N = int(rng.integers(1, 3))
    k = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2]))
    pad = int(rng.integers(0, k))
    H, W = int(rng.integers(k + 1, 15)), int(rng.integers(k + 1, 15))
    Cin = int(rng.choice([4, 8, 36, 64, 100, 132]))
    Cout = int(rng.choice([4, 28, 64, 132]))
    pro = int(rng.choice([ops.PRO_NONE, ops.PRO_RELU, ops.PRO_AFFINE]))
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    x = rnd((N, H, W, Cin), 100 + it) + 0.1
    w = rnd((k, k, Cin, Cout), 200 + it, 1 / math.sqrt(k * k * Cin))
    dy = rnd((N, Ho, Wo, Cout), 300 + it)
    aff = (1.5, -0.25) if pro == ops.PRO_AFFINE else (1.0, 0.0)
    z32 = oracle_ops._prologue(x.numpy(), pro, None, aff, Cin)
    rnd_ = {'bf16': o_enc.bf16_round, 'fp16': o_enc.fp16_round}.get(math_, lambda a: a)
    zd = torch.from_numpy(rnd_(z32)).double()
    wd = torch.from_numpy(rnd_(w.numpy())).double().requires_grad_(True)
    zd.requires_grad_(True)
    ref_conv(zd, wd, stride, pad).backward(torch.from_numpy(rnd_(dy.numpy())).double())
    tag = f'fuzz {math_} #{it} N{N} {H}x{W} k{k} s{stride} p{pad} Cin{Cin} Cout{Cout} pro{pro}'
    dw = ops_bwd.conv2d_wgrad(G(x), G(dy), tuple(w.shape), stride=stride, padding=((pad, pad), (pad, pad)),
                              prologue=pro, in_affine=aff, math=math_)
    want = wd.grad.float()
    helpers.report('wgrad ' + tag, dw, want, atol=3e-5 * float(want.abs().max()) + 1e-6)
    ops.MATMUL_PRECISION = math_
    try:
      dz = ag.conv_dgrad(G(dy), G(w), (N, H, W, Cin), stride, ((pad, pad), (pad, pad)))
    finally:
      ops.MATMUL_PRECISION = 'f32'
    wantz = zd.grad.float()
    helpers.report('dgrad ' + tag, dz[..., :Cin], wantz, atol=3e-5 * float(wantz.abs().max()) + 1e-6)


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,C,Cout_prev,k', [(3, 9, 7, 64, 256, 1), (2, 13, 11, 128, 128, 3), (2, 5, 6, 256, 1024, 1),
                                                (1, 34, 34, 64, 64, 3), (2, 4, 3, 512, 2048, 1)])
def test_dgrad_reads_the_half_twin_of_the_groupnorm_vjp(N, H, W, C, Cout_prev, k, math_):
  """``group_norm_bwd(half=...)`` writes its f32 gradient AND the same values rounded to the training
  engine's element type; the producing layer's data-gradient convolution reads that twin with both
  operands by LDS-DMA (``conv_bf16_xh_kernel``).  The twin equals the rounded f32 tensor, and the
  convolution's result is BIT-IDENTICAL to the launch that rounds the f32 tensor in its loop --
  1 x 1 and 3 x 3, split-K shapes (deep K, few rows), channel counts up to 2048."""
  from snap_amd import autograd as ag
  x = rnd((N, H, W, C), 410) * 1.5 + 0.4
  gamma, beta = rnd((C,), 411) * 0.3 + 1, rnd((C,), 412) * 0.2
  dz = rnd((N, H, W, C), 413)
  add = rnd((N, H, W, C), 414)
  mu, sc, rstd = ops.group_norm_stats(G(x), G(gamma), want_rstd=True)
  plain = ops_bwd.group_norm_bwd(G(x), G(dz), mu, rstd, G(gamma), G(beta), ops.PRO_GN_RELU, add=G(add))
  dx, dgamma, dbeta = ops_bwd.group_norm_bwd(G(x), G(dz), mu, rstd, G(gamma), G(beta), ops.PRO_GN_RELU,
                                             add=G(add), half=math_)
  assert torch.equal(dx, plain[0]) and torch.equal(dgamma, plain[1]) and torch.equal(dbeta, plain[2])
  twin = ops_bwd.half_twin(dx, math_)
  assert twin is not None and ops_bwd.half_twin(dx, 'fp16' if math_ == 'bf16' else 'bf16') is None
  assert torch.equal(twin, dx.to(twin.dtype))                     # RNE, as the engine's in-loop rounding
  # dx is the gradient w.r.t. the output of the PREVIOUS conv (Cprev -> C channels, k x k)
  w = G(rnd((k, k, Cout_prev, C), 415, 1 / math.sqrt(k * k * Cout_prev)))
  pad = (k - 1) // 2
  ops.MATMUL_PRECISION = math_
  try:
    via_twin = ag.conv_dgrad(dx, w, (N, H, W, Cout_prev), 1, ((pad, pad), (pad, pad)))
    ops_bwd._HALF_TWINS.clear()
    via_f32 = ag.conv_dgrad(dx, w, (N, H, W, Cout_prev), 1, ((pad, pad), (pad, pad)))
  finally:
    ops.MATMUL_PRECISION = 'f32'
  assert torch.equal(via_twin, via_f32), float((via_twin - via_f32).abs().max())
  assert float(via_twin.abs().max()) > 0


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,C,Cprev,k,pro', [(3, 9, 7, 64, 256, 1, ops.PRO_GN_RELU), (2, 13, 11, 128, 128, 3, ops.PRO_GN_RELU),
                                                 (2, 20, 18, 64, 64, 3, ops.PRO_RELU_GN), (2, 5, 6, 256, 1024, 1, ops.PRO_NONE),
                                                 (1, 7, 9, 128, 512, 1, ops.PRO_GN_RELU)])
def test_wgrad_reads_the_half_twin_of_the_groupnorm_vjp(N, H, W, C, Cprev, k, pro, math_):
  """The same twin is the dY operand of the producing layer's KERNEL gradient (2-byte loads, byte
  permutes instead of conversions, every prologue on the Z side): same rounded operands, same plan,
  same summation order -- bit-identical to the launch that rounds the f32 gradient in its loader."""
  x = rnd((N, H, W, C), 420) * 1.5 + 0.4
  gamma, beta = rnd((C,), 421) * 0.3 + 1, rnd((C,), 422) * 0.2
  dz = rnd((N, H, W, C), 423)
  mu, sc, rstd = ops.group_norm_stats(G(x), G(gamma), want_rstd=True)
  dx, _, _ = ops_bwd.group_norm_bwd(G(x), G(dz), mu, rstd, G(gamma), G(beta), ops.PRO_GN_RELU, half=math_)
  assert ops_bwd.half_twin(dx, math_) is not None
  # dx = d(output of the previous conv): xp [N,H,W,Cprev] -> (k x k) -> C channels, GroupNorm prologue on xp
  xp = G(rnd((N, H, W, Cprev), 424) + 0.2)
  gn = None
  if pro in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
    gp, bp = G(rnd((Cprev,), 425) * 0.3 + 1), G(rnd((Cprev,), 426) * 0.2)
    mup, scp, _ = ops.group_norm_stats(xp, gp, relu_first=pro == ops.PRO_RELU_GN, want_rstd=True)
    gn = (mup, scp, bp)
  pad = (k - 1) // 2
  kw = dict(padding=((pad, pad), (pad, pad)), prologue=pro, gn=gn, math=math_)
  # (3 x 3: the twin also unlocks the fused-tap kernel -- another summation order, tested on its own;
  #  the per-tap plan is pinned here so that both launches are the same kernel)
  try:
    via_twin = ops_bwd.conv2d_wgrad(xp, dx, (k, k, Cprev, C), plans=1, **kw)
    ops_bwd.WGRAD_DY_TWIN = False
    via_f32 = ops_bwd.conv2d_wgrad(xp, dx, (k, k, Cprev, C), plans=1, **kw)
  finally:
    ops_bwd.WGRAD_DY_TWIN = True
  assert torch.equal(via_twin, via_f32), float((via_twin - via_f32).abs().max())
  assert float(via_twin.abs().max()) > 0


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,Cin,Cout,pro', [
    (2, 12, 16, 64, 64, ops.PRO_GN_RELU),       # whole patches, the 64-column tile (k-steps split between wave sets)
    (3, 17, 17, 128, 128, ops.PRO_GN_RELU),     # ragged patches in both directions
    (1, 34, 34, 256, 256, ops.PRO_GN_RELU),
    (2, 9, 7, 96, 160, ops.PRO_NONE),           # partial channel / column tiles, one patch column
    (5, 4, 8, 64, 32, ops.PRO_GN_RELU),         # one patch per image
    (20, 17, 17, 512, 512, ops.PRO_GN_RELU),    # the C3 stage-4 layer (32 tiles, few patches per chunk)
])
def test_conv_wgrad_3x3_fused_taps(N, H, W, Cin, Cout, pro, math_):
  """``wgrad3x3.hip``: all nine taps of a 3 x 3 / stride 1 / pad 1 kernel gradient in one workgroup, the
  reduction in 4 x 8-pixel patches, dy in the engine's element type.  Versus the rounded-operand
  restatement in float64 (the tolerance of ``test_conv_wgrad_bf16``) and versus the per-tap kernel
  (same rounded operands, another summation order)."""
  from oracle import encoder as o_enc
  rnd_ = o_enc.bf16_round if math_ == 'bf16' else o_enc.fp16_round
  hd = ops_bwd.HALF_DTYPE[math_]
  x = rnd((N, H, W, Cin), 431) + 0.1
  gamma, beta = rnd((Cin,), 432) * 0.3 + 1, rnd((Cin,), 433) * 0.2
  dyh = G(rnd((N, H, W, Cout), 434)).to(hd)
  gn = gn_cpu = None
  if pro == ops.PRO_GN_RELU:
    mu, sc = oracle_ops.group_norm_stats(x, gamma, relu_first=False)
    gn, gn_cpu = (G(mu), G(sc), G(beta)), (mu, sc, beta)
  z32 = oracle_ops._prologue(x.numpy().astype(np.float32), pro, gn_cpu, (1.0, 0.0), Cin)
  wd = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
  ref_conv(torch.from_numpy(rnd_(z32)).double(), wd, 1, 1).backward(dyh.cpu().double())
  ref = wd.grad.float()
  kw = dict(padding=((1, 1), (1, 1)), prologue=pro, gn=gn, math=math_)
  got = ops_bwd.conv2d_wgrad(G(x), dyh, (3, 3, Cin, Cout), **kw)
  per_tap = ops_bwd.conv2d_wgrad(G(x), dyh, (3, 3, Cin, Cout), plans=1, **kw)     # (a per-call switch)
  tol = 2e-5 * float(ref.abs().max()) + 1e-5
  helpers.report(f'wgrad 3x3 fused {math_} {N}x{H}x{W} {Cin}->{Cout}', got, ref, atol=tol)
  helpers.report(f'wgrad 3x3 per-tap {math_}', per_tap, ref, atol=tol)
  assert not torch.equal(got, per_tap) or N * H * W <= 32     # (two kernels: the A/B switch switches)


@pytest.mark.parametrize('M,C,listed', [(70001, 128, True), (5000, 96, False), (300007, 256, True), (1234, 160, True),
                                        (4099, 1024, False), (9, 4, True)])
def test_colsum_kernels(M, C, listed):
  """Column sums (bias gradients) dense and over a row list with a device-side count: the vector kernel (column
  quads x row phases, four rows in flight) where C / 4 divides 256, the per-column kernel elsewhere; fixed order
  (two runs agree bit for bit)."""
  g = torch.Generator().manual_seed(460)
  a = torch.randn(M, C, generator=g)
  if listed:
    mask = torch.rand(M, generator=g) < 0.45
    index, count = ops.compact_rows(G(mask))
    got = ops_bwd.colsum(G(a), rows=index, row_count=count)
    again = ops_bwd.colsum(G(a), rows=index, row_count=count)
    ref = a[mask].double().sum(0).float()
  else:
    got, again = ops_bwd.colsum(G(a)), ops_bwd.colsum(G(a))
    ref = a.double().sum(0).float()
  assert torch.equal(got, again)
  helpers.report(f'colsum {M}x{C}', got, ref, atol=3e-6 * math.sqrt(M) + 1e-5)


@pytest.mark.parametrize('N,H,W,C', [(2, 37, 41, 64), (1, 16, 16, 8), (3, 33, 17, 72), (1, 5, 3, 4)])
def test_max_pool_bwd_tiled_kernel(N, H, W, C):
  """The tiled 3 x 3 / 2 max-pool VJP (a workgroup reads each window once and keeps the 9-bit set of positions
  that receive its gradient): ties go to the first maximum in scan order (vs torch autograd in float64), and
  with NaNs planted the result is bit-identical to the per-pixel kernel (reached through a channel count that
  is not a multiple of four: the op is channel-wise)."""
  g = torch.Generator().manual_seed(440)
  x = torch.randint(-3, 4, (N, H, W, C), generator=g).float()            # many ties
  Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
  dy = torch.randn(N, Ho, Wo, C, generator=g)
  xd = x.double().requires_grad_(True)
  F.max_pool2d(xd.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).backward(dy.double())
  got = ops_bwd.max_pool_3x3s2_bwd(G(x), G(dy))
  helpers.report(f'maxpool bwd tiled {N}x{H}x{W}x{C}', got, xd.grad.float(), atol=1e-6)
  xn = x + torch.randn(N, H, W, C, generator=g) * 0.1
  xn[torch.rand(N, H, W, C, generator=g) < 0.05] = float('nan')
  xn[0, :3, :3] = float('nan')                                            # a whole window of NaNs
  tiled = ops_bwd.max_pool_3x3s2_bwd(G(xn), G(dy))
  c1 = C - 1                                                              # -> the scalar per-pixel kernel
  plain = ops_bwd.max_pool_3x3s2_bwd(G(xn[..., :c1].contiguous()), G(dy[..., :c1].contiguous()))
  assert torch.equal(tiled[..., :c1], plain)


@pytest.mark.parametrize('k,stride,pad', [(1, 1, 0), (3, 1, 1), (3, 2, 1), (1, 2, 0)])
def test_conv_dgrad_via_engine(k, stride, pad):
  from snap_amd import autograd as ag
  N, H, W, Cin, Cout = 2, 11, 10, 64, 128
  x = rnd((N, H, W, Cin), 6).double().requires_grad_(True)
  w = rnd((k, k, Cin, Cout), 7, 1 / math.sqrt(k * k * Cin))
  y = ref_conv(x, w.double(), stride, pad)
  dy = rnd(tuple(y.shape), 8)
  y.backward(dy.double())
  got = ag.conv_dgrad(G(dy), G(w), (N, H, W, Cin), stride, ((pad, pad), (pad, pad)))
  helpers.report(f'dgrad k{k} s{stride}', got, x.grad.float(), atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,C,Cnext,k,mode,acc', [
    (3, 17, 17, 64, 256, 1, ops.PRO_GN_RELU, False),     # tiles straddle images
    (2, 13, 11, 128, 128, 3, ops.PRO_GN_RELU, True),     # 3 x 3, accumulated second gradient
    (5, 12, 12, 256, 64, 1, ops.PRO_RELU_GN, False),
    (2, 34, 34, 512, 128, 1, ops.PRO_GN_RELU, True),
])
def test_groupnorm_vjp_statistics_from_the_dgrad_epilogue(N, H, W, C, Cnext, k, mode, acc, math_, monkeypatch):
  """The first pass of the GroupNorm VJP (sums of the gated gradient and of gradient x normalised input per
  (image, channel)) taken in the epilogue of the half-input data-gradient launch that writes that gradient
  (``ops.conv2d(gn_bwd_stats=)``): dx / dgamma / dbeta agree with the stand-alone pass to summation order, the
  gradient tensor itself is bit-identical, and a different x (or mode) makes the VJP ignore the sums."""
  hd = ops_bwd.HALF_DTYPE[math_]
  monkeypatch.setattr(ops, 'USE_SPLITK', False)          # (a split-K launch leaves the statistics to the stand-alone pass)
  x = G(rnd((N, H, W, C), 470) * 1.5 + 0.4)
  gamma, beta = G(rnd((C,), 471) * 0.3 + 1), G(rnd((C,), 472) * 0.2)
  mu, sc, rstd = ops.group_norm_stats(x, gamma, relu_first=mode == ops.PRO_RELU_GN, want_rstd=True)
  dyh = G(rnd((N, H, W, Cnext), 473)).to(hd)                     # gradient w.r.t. the NEXT conv's output (a half twin)
  w_rot = G(rnd((k, k, Cnext, C), 474, 1 / math.sqrt(k * k * Cnext)))
  res = G(rnd((N, H, W, C), 475)) if acc else None
  pad = (k - 1) // 2
  kw = dict(padding=((pad, pad), (pad, pad)), residual=res, math=math_)
  dz_plain = ops.conv2d(dyh, w_rot, **kw)
  dz = ops.conv2d(dyh, w_rot, gn_bwd_stats=(x, mu, rstd, gamma, beta, mode), **kw)
  assert torch.equal(dz, dz_plain) and getattr(dz, '_snap_gnb_partial', None) is not None
  add = G(rnd((N, H, W, C), 476))
  fused = ops_bwd.group_norm_bwd(x, dz, mu, rstd, gamma, beta, mode, add=add, half=math_)
  monkeypatch.setattr(ops_bwd, 'USE_GNB_STATS', False)
  plain = ops_bwd.group_norm_bwd(x, dz, mu, rstd, gamma, beta, mode, add=add, half=math_)
  monkeypatch.setattr(ops_bwd, 'USE_GNB_STATS', True)
  for nm, a, b in zip(('dx', 'dgamma', 'dbeta'), fused, plain):
    helpers.report(f'gn vjp fused stats {nm}', a, b, atol=2e-5 * float(b.abs().max()) + 1e-6)
  assert not torch.equal(fused[1], plain[1]) or C * H * W < 4096          # (another summation order: really fused)
  # stale sums are not trusted: another x tensor / another mode -> the stand-alone pass (bit-identical to `plain`)
  x2 = x.clone()
  again = ops_bwd.group_norm_bwd(x2, dz, mu, rstd, gamma, beta, mode, add=add, half=math_)
  assert all(torch.equal(a, b) for a, b in zip(again, plain))


@pytest.mark.parametrize('math_', ['f32', 'bf16'])
@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 11, 10, 64, 128), (3, 8, 8, 256, 512), (1, 17, 17, 100, 64)])
def test_strided_1x1_dgrad_on_the_coarse_grid(N, H, W, Cin, Cout, math_, monkeypatch):
  """The data gradient of a 1 x 1 / stride 2 projection: a GEMM over the Ho x Wo pixels that carry a gradient,
  added into every second pixel of the accumulated tensor -- the same products in the same order as the
  zero-dilated launch over all H x W pixels (bit-identical), with and without an accumulated gradient."""
  from snap_amd import autograd as ag
  monkeypatch.setattr(ops, 'MATMUL_PRECISION', math_)
  Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
  dy, w = G(rnd((N, Ho, Wo, Cout), 450)), G(rnd((1, 1, Cin, Cout), 451, 1 / math.sqrt(Cin)))
  C4 = (Cin + 3) // 4 * 4
  acc = G(rnd((N, H, W, C4), 452))
  pad0 = ((0, 0), (0, 0))
  for accumulate in (None, acc):
    new = ag.conv_dgrad(dy, w, (N, H, W, Cin), 2, pad0, accumulate=accumulate)
    monkeypatch.setattr(ag, 'STRIDED_1X1_DGRAD', False)
    old = ag.conv_dgrad(dy, w, (N, H, W, Cin), 2, pad0, accumulate=accumulate)
    monkeypatch.setattr(ag, 'STRIDED_1X1_DGRAD', True)
    assert new.shape == old.shape and torch.equal(new, old), float((new - old).abs().max())
  assert torch.equal(acc, G(rnd((N, H, W, C4), 452)))                      # (not overwritten without the flag)
  own = acc.clone()
  out = ag.conv_dgrad(dy, w, (N, H, W, Cin), 2, pad0, accumulate=own, accumulate_inplace=True)
  assert out.data_ptr() == own.data_ptr() and torch.equal(out, new)


# -- GroupNorm backward -------------------------------------------------------------------
@pytest.mark.parametrize('mode', [ops.PRO_GN_RELU, ops.PRO_RELU_GN])
@pytest.mark.parametrize('C,HW', [(64, (9, 7)), (256, (5, 6)), (2048, (3, 2))])
def test_group_norm_bwd(mode, C, HW):
  N = 3
  x = rnd((N, *HW, C), 10) * 1.5 + 0.4
  gamma, beta = rnd((C,), 11) * 0.3 + 1, rnd((C,), 12) * 0.2
  dz = rnd((N, *HW, C), 13)
  add = rnd((N, *HW, C), 14)
  xd = x.double().requires_grad_(True)
  gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
  z = ref_gn(xd, gd, bd, 32, mode == ops.PRO_RELU_GN, mode == ops.PRO_GN_RELU)
  z.backward(dz.double())
  mu, sc, rstd = ops.group_norm_stats(G(x), G(gamma), relu_first=mode == ops.PRO_RELU_GN,
                                      want_rstd=True)
  dx, dgamma, dbeta = ops_bwd.group_norm_bwd(G(x), G(dz), mu, rstd, G(gamma), G(beta), mode,
                                             add=G(add))
  helpers.report('gn dx', dx, (xd.grad + add.double()).float(), atol=3e-5, rtol=1e-4)
  helpers.report('gn dgamma', dgamma, gd.grad.float(), atol=2e-4, rtol=1e-4)
  helpers.report('gn dbeta', dbeta, bd.grad.float(), atol=2e-4, rtol=1e-4)


def test_small_backward_ops():
  # weight standardisation
  w = rnd((3, 3, 16, 24), 20) * 0.3 + 0.1
  dws = rnd((3, 3, 16, 24), 21)
  wd = w.double().requires_grad_(True)
  u = wd - wd.mean((0, 1, 2), keepdim=True)
  (u / torch.sqrt((u * u).mean((0, 1, 2), keepdim=True) + 1e-10)).backward(dws.double())
  helpers.report('wstd bwd', ops_bwd.weight_standardize_bwd(G(w), G(dws)), wd.grad.float(),
                 atol=1e-4, rtol=1e-4)
  # max pool
  x = rnd((2, 9, 11, 8), 22)
  xd = x.double().requires_grad_(True)
  y = F.max_pool2d(xd.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
  dy = rnd(tuple(y.shape), 23)
  y.backward(dy.double())
  helpers.report('maxpool bwd', ops_bwd.max_pool_3x3s2_bwd(G(x), G(dy)), xd.grad.float(), atol=1e-6)
  # bilinear x2 transpose
  p = rnd((2, 5, 4, 8), 24).double().requires_grad_(True)
  up = F.interpolate(p.permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=False)
  dyu = rnd((2, 10, 8, 8), 25)
  up.permute(0, 2, 3, 1).backward(dyu.double())
  helpers.report('upsample bwd', ops_bwd.upsample2x_bwd(G(dyu)), p.grad.float(), atol=1e-5)
  # epilogue gate + column sums
  dy2, y2 = rnd((300, 64), 26), rnd((300, 64), 27)
  mask = torch.rand(300, generator=torch.Generator().manual_seed(28)) > 0.3
  ref = dy2 * (y2 > 0) * mask[:, None]
  helpers.report('epilogue bwd', ops_bwd.epilogue_bwd(G(dy2), G(y2), G(mask), relu=True), ref, atol=0)
  a = rnd((5000, 96), 29)
  helpers.report('colsum', ops_bwd.colsum(G(a)), a.double().sum(0).float(), atol=2e-4)


# -- lift / BEV ------------------------------------------------------------------------------
def _lift_reference(f, cam, Rt, pts, K, fd, nb, opts, obs_override=None):
  """float64 torch restatement of lift + pool_multiview_features (streetview_encoder.py:69-178) with
  the projection geometry (taps, weights, visibility) taken from the numpy oracle.  Returns
  (f as a float64 leaf, observations [N, Kv, C], pooled [N, channels], extras)."""
  from oracle import lift as o_lift
  weighted, use_var, minmax = opts
  V = f.shape[1]
  cams = oracle_ops.unpack_cameras(cam, True)
  T = oracle_ops.unpack_transforms(Rt)
  p2d, vis, depth, rays = o_lift.project_points_to_views(T, cams, pts.numpy())
  if K > 0:
    idx, _ = o_lift.view_selection(pts.numpy(), T, vis, K)
    p2d, vis, depth, rays = (o_lift.gather_batched_observations(a, idx) for a in (p2d, vis, depth, rays))
  else:
    idx = np.broadcast_to(np.arange(V), vis.shape).copy()
  fdbl = f.double().requires_grad_(True)
  h, w = f.shape[2:4]
  pt = torch.tensor(p2d[0]).double() - 0.5                     # [N,Kv,2]
  if K > 0:
    pt = torch.minimum(torch.maximum(pt, torch.zeros(2, dtype=torch.float64)),
                       torch.tensor([h - 1.0, w - 1.0], dtype=torch.float64))
  lo = torch.floor(pt)
  w1 = pt - lo
  w0 = 1 - w1
  i0 = lo[..., 0].long().clamp(0, h - 1); i1 = (lo[..., 0].long() + 1).clamp(0, h - 1)
  j0 = lo[..., 1].long().clamp(0, w - 1); j1 = (lo[..., 1].long() + 1).clamp(0, w - 1)
  vi = torch.tensor(idx[0]).long()
  img = fdbl[0]
  val = ((w0[..., 0] * w0[..., 1])[..., None] * img[vi, i0, j0]
         + (w0[..., 0] * w1[..., 1])[..., None] * img[vi, i0, j1]
         + (w1[..., 0] * w0[..., 1])[..., None] * img[vi, i1, j0]
         + (w1[..., 0] * w1[..., 1])[..., None] * img[vi, i1, j1])   # [N,Kv,C]
  visb = torch.tensor(vis[0])
  anyv = visb.any(-1)
  extras = dict(vis=visb, depth=torch.tensor(depth[0]).double(), rays=torch.tensor(rays[0]).double())
  feats = val[..., :fd] if obs_override is None else obs_override
  if weighted:
    scales = val[..., fd:]
    dpt = extras['depth'].clamp(1.0, 16.0)
    tt = torch.log(dpt / 1.0) / math.log(16.0)
    c = (0.5 + tt * (nb - 1)) - 0.5
    fl = torch.floor(c)
    wb = c - fl
    b0 = fl.long().clamp(0, nb - 1); b1 = (fl.long() + 1).clamp(0, nb - 1)
    score = (1 - wb) * torch.gather(scales, -1, b0[..., None])[..., 0] + wb * torch.gather(
        scales, -1, b1[..., None])[..., 0]
    sm = torch.where(visb, score, torch.full_like(score, -math.inf))
    wgt = torch.softmax(torch.where(anyv[:, None], sm, torch.zeros_like(sm)), -1)
    wgt = torch.where(visb, wgt, torch.zeros_like(wgt))
  else:
    cnt = visb.sum(-1, keepdim=True).clamp(min=1)
    wgt = visb.double() / cnt
  mean = (wgt[..., None] * feats).sum(1)
  stats = [mean]
  if use_var:
    stats.append((wgt[..., None] * (feats - mean[:, None]) ** 2).sum(1))
  if minmax:
    big = torch.full_like(feats, math.inf)
    stats.append(torch.where(visb[..., None], feats, -big).amax(1))
    stats.append(torch.where(visb[..., None], feats, big).amin(1))
  if weighted:
    stats.append(torch.where(anyv, sm.max(-1).values, torch.zeros_like(anyv, dtype=torch.float64))[:, None])
  pooled = torch.cat(stats, -1)
  pooled = torch.where(anyv[:, None], pooled, torch.zeros_like(pooled))
  return fdbl, val, pooled, extras


LIFT_BWD_OPTS = [(True, True, False), (False, True, False), (True, False, False), (True, True, True),
                 (False, False, True)]


@pytest.mark.parametrize('opts', LIFT_BWD_OPTS, ids=['default', 'unweighted', 'novar', 'minmax', 'unweighted_novar_minmax'])
@pytest.mark.parametrize('K,V', [(0, 3), (2, 4)])
def test_lift_pool_bwd(K, V, opts):
  """Reference: torch autograd through a differentiable restatement that takes the
  projection geometry (taps, weights, visibility) from the numpy oracle.  Every option of
  pool_multiview_features (weighted / use_variance / add_minmax)."""
  import test_gpu_kernels as tk
  weighted, use_var, minmax = opts
  fd, nb = 16, 4
  f, cam, Rt, pts = tk._lift_scene(1, V, 10, 12, fd, nb, 1500, seed=60 + V)
  if not weighted:
    f = f[..., :fd].contiguous()
  if minmax:
    f = torch.round(f * 4) / 4            # ties among the views' observations do occur
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0),
            weighted=weighted, use_variance=use_var, add_minmax=minmax)
  nch = ops.pooled_channels(fd, weighted, use_var, minmax)
  stride = ops.pooled_stride(fd, weighted, use_var, minmax)
  dpooled = rnd((1, 1500, stride), 61)
  dpooled[..., nch:] = 0
  fdbl, _, pooled, _ = _lift_reference(f, cam, Rt, pts, K, fd, nb, opts)
  pooled.backward(dpooled[0, :, :nch].double())
  ref = fdbl.grad.float()
  tol = 2e-4 * float(ref.abs().max()) + 1e-6
  # both forms of the VJP: the deterministic one (records -> stable sort by pixel -> gather; the
  # default) must also be BITWISE reproducible; the scatter form uses float atomics
  got = ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw)
  helpers.report('lift bwd (deterministic)', got, ref, atol=tol)
  for _ in range(3):
    assert torch.equal(ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw), got)
  if opts == (True, True, False):
    prev, ops_bwd.DETERMINISTIC_LIFT_BWD = ops_bwd.DETERMINISTIC_LIFT_BWD, False
    try:
      got_s = ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw)
    finally:
      ops_bwd.DETERMINISTIC_LIFT_BWD = prev
    helpers.report('lift bwd (scatter)', got_s, ref, atol=tol)
    # the default options take the BATCHED record producer (lane = voxel for the geometry, lane =
    # channel quad for the gradients); the half-wave-per-voxel producer it replaces must agree with
    # it to rounding (same records, another summation order inside d w_k)
    prev, ops_bwd.LIFT_BWD_UNBATCHED = ops_bwd.LIFT_BWD_UNBATCHED, True
    try:
      got_u = ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw)
    finally:
      ops_bwd.LIFT_BWD_UNBATCHED = prev
    helpers.report('lift bwd (half-wave-per-voxel producer)', got_u, ref, atol=tol)
    assert float((got_u - got).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-7


@pytest.mark.parametrize('K,V,fd,nb', [(0, 4, 128, 32), (1, 3, 128, 32), (3, 5, 32, 8)])
def test_lift_pool_bwd_batched_producer_real_widths(K, V, fd, nb):
  """The batched record producer at the model's widths (feature_dim 128 + 32 depth bins: every lane
  owns a channel quad; feature_dim 32: lanes 8.. idle), all views / one / three of five selected,
  25 000 voxels (ragged last workgroup): vs the half-wave-per-voxel producer and bitwise
  repeatable."""
  import test_gpu_kernels as tk
  N = 25000 + 37
  f, cam, Rt, pts = tk._lift_scene(2, V, 16, 20, fd, nb, N, seed=70 + V)
  kw = dict(K=K, fisheye=True, feature_dim=fd, num_bins=nb, depth_min_max=(1.0, 16.0))
  stride = ops.pooled_stride(fd, True, True, False)
  nch = ops.pooled_channels(fd, True, True, False)
  dpooled = rnd((2, N, stride), 71)
  dpooled[..., nch:] = 0
  got = ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw)
  assert torch.equal(ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw), got)
  prev, ops_bwd.LIFT_BWD_UNBATCHED = ops_bwd.LIFT_BWD_UNBATCHED, True
  try:
    ref = ops_bwd.lift_pool_bwd(G(f), G(cam), G(Rt), G(pts), G(dpooled), **kw)
  finally:
    ops_bwd.LIFT_BWD_UNBATCHED = prev
  scale = float(ref.abs().max())
  assert scale > 0
  assert float((got - ref).abs().max()) <= 2e-5 * scale, float((got - ref).abs().max()) / scale


@pytest.mark.parametrize('use_var,minmax', [(True, False), (False, True)])
@pytest.mark.parametrize('K,V', [(0, 3), (2, 4)])
def test_depth_mlp_fusion_bwd(K, V, use_var, minmax):
  """The depth_mlp fusion (streetview_encoder.py:263-267) as autograd nodes: observations ->
  per-observation MLP (+ residual) -> pooling; gradients w.r.t. the image features and the MLP
  parameters against a float64 torch restatement."""
  import test_gpu_kernels as tk
  from snap_amd import autograd as ag
  fd = 16
  f, cam, Rt, pts = tk._lift_scene(1, V, 10, 12, fd, 4, 1200, seed=80 + V)
  f = f[..., :fd].contiguous()
  w0 = rnd((fd + 4, 32), 81, 1 / math.sqrt(fd + 4)); b0 = rnd((32,), 82, 0.1)
  w1 = rnd((32, fd), 83, 1 / math.sqrt(32.0)); b1 = rnd((fd,), 84, 0.1)
  nch = ops.pooled_channels(fd, False, use_var, minmax)
  stride = ops.pooled_stride(fd, False, use_var, minmax)
  dpooled = rnd((1, 1200, stride), 85)
  dpooled[..., nch:] = 0
  # reference
  fdbl, val, _, ex = _lift_reference(f, cam, Rt, pts, K, fd, 0, (False, use_var, minmax))
  logd = torch.log10(ex['depth'].clamp(0.1, 100))
  rays = torch.where(ex['vis'][..., None], ex['rays'], torch.zeros_like(ex['rays']))
  x = torch.cat([val, logd[..., None], rays], -1)
  W = [t.double().requires_grad_(True) for t in (w0, b0, w1, b1)]
  corrected = val + torch.relu(x @ W[0] + W[1]) @ W[2] + W[3]
  _, _, pooled, _ = _lift_reference(f, cam, Rt, pts, K, fd, 0, (False, use_var, minmax), obs_override=corrected)
  pooled.backward(dpooled[0, :, :nch].double())
  # the autograd nodes
  fg = G(f).requires_grad_(True)
  Wg = [G(t).requires_grad_(True) for t in (w0, b0, w1, b1)]
  common = dict(K=K, fisheye=True, feature_dim=fd)
  obs, feat, _ = ag.lift_observations(fg, G(cam), G(Rt), G(pts), **common)
  hdn = ag.dense(obs, Wg[0], Wg[1], relu=True)
  out = ag.dense(hdn, Wg[2], Wg[3], residual=feat)
  pg, _ = ag.lift_pool_observations(out, tuple(f.shape), G(cam), G(Rt), G(pts), use_variance=use_var,
                                    add_minmax=minmax, **common)
  pg.backward(G(dpooled))
  ref = fdbl.grad.float()
  helpers.report('depth_mlp d f_images', fg.grad, ref, atol=3e-4 * float(ref.abs().max()) + 1e-6)
  for name, got, want in zip(('w0', 'b0', 'w1', 'b1'), Wg, W):
    helpers.report(f'depth_mlp d {name}', got.grad, want.grad.float(),
                   atol=3e-4 * float(want.grad.abs().max()) + 1e-6)


@pytest.mark.parametrize('pooling', ['max', 'sum', 'mean'])
def test_vertical_pool_bwd(pooling):
  vol = rnd((2, 6, 5, 7, 32), 70)
  valid = torch.rand((2, 6, 5, 7), generator=torch.Generator().manual_seed(71)) > 0.5
  valid[0, 0] = False
  dplane = rnd((2, 6, 5, 32), 72)
  vd = vol.double().requires_grad_(True)
  anyv = valid.any(-1)
  if pooling == 'max':
    out = torch.where(valid[..., None], vd, torch.full_like(vd, -math.inf)).amax(-2)
    out = torch.where(anyv[..., None], out, torch.zeros_like(out))
  else:
    out = (vd * valid[..., None]).sum(-2)
    if pooling == 'mean':
      out = out / valid.sum(-1).clamp(min=1)[..., None]
  out.backward(dplane.double())
  got = ops_bwd.vertical_pool_bwd(G(vol), G(valid), G(dplane), pooling)
  helpers.report('vpool bwd ' + pooling, got, vd.grad.float(), atol=1e-6)


@pytest.mark.parametrize('Z,D,ties', [(7, 32, False), (60, 128, False), (64, 64, True), (12, 128, True)])
def test_vertical_pool_max_bwd_from_the_forward_record(Z, D, ties):
  """Max pooling in a training step: the forward also records where every maximum sits
  (snap_vertical_pool_max_arg_f32) and the VJP writes dvol from that record without reading the volume --
  bit for bit the two-pass kernel, including shared maxima (the gradient is divided), columns without
  a valid level, -inf and NaN entries."""
  from snap_amd import autograd as ag
  g = torch.Generator().manual_seed(700 + Z)
  vol = torch.randn((3, 5, 4, Z, D), generator=g)
  if ties:
    vol = (vol * 2).round() / 2               # many equal maxima
  vol[0, 1, 2, Z // 2, :8] = float('nan')
  vol[1, 0, 0, :, 3] = -math.inf
  valid = torch.rand((3, 5, 4, Z), generator=g) > 0.4
  valid[0, 0] = False
  # a column whose ONLY valid level holds -inf in a channel: that level is the recorded maximum (count 1)
  valid[2, 1, 1] = False
  valid[2, 1, 1, Z // 3] = True
  vol[2, 1, 1, Z // 3, 5] = -math.inf
  dplane = torch.randn((3, 5, 4, D), generator=g)
  plane0, pv0 = ops.vertical_pool(G(vol), G(valid), 'max')
  plane1, pv1, arg = ops.vertical_pool(G(vol), G(valid), 'max', want_arg=True)
  assert arg is not None
  assert torch.equal(plane0.view(torch.int32), plane1.view(torch.int32)) and torch.equal(pv0, pv1)
  want = ops_bwd.vertical_pool_bwd(G(vol), G(valid), G(dplane), 'max')
  got = ops_bwd.vertical_pool_bwd(G(vol), G(valid), G(dplane), 'max', arg=arg)
  assert torch.equal(got.view(torch.int32), want.view(torch.int32))
  if ties:
    assert int((arg[1] > 1).sum()) > 0          # (the shared-maximum path ran)
  # and through autograd
  vg = G(vol).requires_grad_(True)
  out, _ = ag.vertical_pool(vg, G(valid), 'max')
  out.backward(G(dplane))
  assert torch.equal(vg.grad.view(torch.int32), want.view(torch.int32))
  # other poolings have no record
  assert ops.vertical_pool(G(vol), G(valid), 'sum', want_arg=True)[2] is None


@pytest.mark.parametrize('nplanes', [1, 2])
def test_plane_fuse_match_bwd(nplanes):
  D, Dm, M = 64, 16, 140
  planes = [rnd((M, D), 80 + i) for i in range(nplanes)]
  valids = [torch.rand(M, generator=torch.Generator().manual_seed(90 + i)) > 0.3 for i in range(nplanes)]
  if nplanes == 2:
    valids[1] = None
  planes[0] = planes[0] * valids[0][:, None]
  Wm, bm = rnd((D, Dm), 85, 0.2), rnd((Dm,), 86, 0.05)
  dmat = rnd((M, Dm), 87)
  pd = [p.double().requires_grad_(True) for p in planes]
  Wd, bd = Wm.double().requires_grad_(True), bm.double().requires_grad_(True)
  vs = torch.stack([torch.ones(M, dtype=torch.bool) if v is None else v for v in valids], -1)
  st = torch.stack(pd, -2)
  anyv = vs.any(-1)
  fused = torch.where(vs[..., None], st, torch.full_like(st, -math.inf)).amax(-2)
  fused = torch.where(anyv[:, None], fused, torch.zeros_like(fused))
  y = fused @ Wd + bd
  z = y / y.norm(dim=-1, keepdim=True)
  (z * anyv[:, None]).backward(dmat.double())
  dplanes, dy = ops_bwd.plane_fuse_match_bwd([G(p) for p in planes],
                                             [None if v is None else G(v) for v in valids], 'max',
                                             G(Wm), G(bm), True, 1e-5, G(dmat))
  for i in range(nplanes):
    helpers.report(f'dplane{i}', dplanes[i], pd[i].grad.float(), atol=2e-5, rtol=1e-4)
  fused_g, _, _ = ops.plane_fuse_match([G(p) for p in planes],
                                       [None if v is None else G(v) for v in valids], 'max',
                                       G(Wm), G(bm))
  dW = ops_bwd.conv2d_wgrad(fused_g.reshape(1, 1, M, D), dy.reshape(1, 1, M, Dm), (1, 1, D, Dm))
  helpers.report('dWm', dW.reshape(D, Dm), Wd.grad.float(), atol=1e-4, rtol=1e-4)
  helpers.report('dbm', ops_bwd.colsum(dy), bd.grad.float(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('pooling,Dm,normalize,with_dfused', [('max', 32, True, False), ('mean', 16, True, True),
                                                              ('sum', 32, False, True)])
def test_plane_fuse_match_bwd_persistent_kernel_for_128_channels(pooling, Dm, normalize, with_dfused):
  """From 8192 cells of 128 channels the VJP takes the persistent kernel (head columns in registers, the
  transposed head in LDS): bit for bit what the per-cell kernel gives on the same cells in launches below
  the threshold, and against torch autograd for max pooling."""
  D, M = 128, 8192 + 517
  planes = [rnd((M, D), 280 + i) for i in range(3)]
  valids = [torch.rand(M, generator=torch.Generator().manual_seed(290 + i)) > 0.4 for i in range(3)]
  valids[2] = None if pooling == 'sum' else valids[2]
  planes[0][17] = planes[1][17]                 # (a cell whose maximum is shared)
  Wm, bm = rnd((D, Dm), 285, 0.2), rnd((Dm,), 286, 0.05)
  dmat = rnd((M, Dm), 287)
  dfu = rnd((M, D), 288) if with_dfused else None
  gv = [None if v is None else G(v) for v in valids]
  dplanes, dy = ops_bwd.plane_fuse_match_bwd([G(p) for p in planes], gv, pooling, G(Wm), G(bm), normalize, 1e-5,
                                             G(dmat), None if dfu is None else G(dfu))
  for lo in range(0, M, 4096):
    hi = min(lo + 4096, M)
    dp0, dy0 = ops_bwd.plane_fuse_match_bwd([G(p[lo:hi]) for p in planes],
                                            [None if v is None else G(v[lo:hi]) for v in valids], pooling,
                                            G(Wm), G(bm), normalize, 1e-5, G(dmat[lo:hi]),
                                            None if dfu is None else G(dfu[lo:hi]))
    assert torch.equal(dy0.view(torch.int32), dy[lo:hi].view(torch.int32))
    for a, b in zip(dp0, dplanes):
      assert torch.equal(a.view(torch.int32), b[lo:hi].view(torch.int32))
  if pooling == 'max' and not with_dfused:
    pd = [p.double().requires_grad_(True) for p in planes]
    vs = torch.stack([torch.ones(M, dtype=torch.bool) if v is None else v for v in valids], -1)
    st = torch.stack(pd, -2)
    anyv = vs.any(-1)
    fused = torch.where(vs[..., None], st, torch.full_like(st, -math.inf)).amax(-2)
    fused = torch.where(anyv[:, None], fused, torch.zeros_like(fused))
    y = fused @ Wm.double() + bm.double()
    z = y / y.norm(dim=-1, keepdim=True)
    (z * anyv[:, None]).backward(dmat.double())
    for i in range(3):
      helpers.report(f'dplane{i}', dplanes[i], pd[i].grad.float(), atol=3e-5, rtol=1e-4)


# -- pose head --------------------------------------------------------------------------------
@pytest.mark.parametrize('mask_oob', [False, True])
def test_pose_score_bwd(mask_oob):
  B, Nq, X, Y, P = 2, 30, 24, 20, 300
  rng = np.random.default_rng(100)
  sim = torch.tensor(rng.random((B, Nq, X, Y), dtype=np.float32))
  cell = 0.25
  poses = torch.tensor(np.stack([rng.uniform(-3, 3, (B, P)), rng.uniform(-1, X * cell + 1, (B, P)),
                                 rng.uniform(-1, Y * cell + 1, (B, P))], -1).astype(np.float32))
  q_xy = torch.tensor(rng.uniform(-1.5, 1.5, (B, Nq, 2)).astype(np.float32))
  valid_q = torch.tensor(rng.random((B, Nq)) > 0.2)
  mv = torch.tensor(rng.random((B, X, Y)) > 0.1)
  dscores = rnd((B, P), 101)
  sd = sim.double().requires_grad_(True)
  c, s = torch.cos(poses[..., 0].double()), torch.sin(poses[..., 0].double())
  qx, qy = q_xy[..., 0].double(), q_xy[..., 1].double()
  u = (c[:, :, None] * qx[:, None] - s[:, :, None] * qy[:, None] + poses[..., 1].double()[:, :, None]) / cell
  v = (s[:, :, None] * qx[:, None] + c[:, :, None] * qy[:, None] + poses[..., 2].double()[:, :, None]) / cell
  cu, cv = u - 0.5, v - 0.5
  fu, fv = torch.floor(cu), torch.floor(cv)
  wu1, wv1 = cu - fu, cv - fv
  i0 = fu.long().clamp(0, X - 1); i1 = (fu.long() + 1).clamp(0, X - 1)
  j0 = fv.long().clamp(0, Y - 1); j1 = (fv.long() + 1).clamp(0, Y - 1)
  bi = torch.arange(B)[:, None, None]
  ni = torch.arange(Nq)[None, None, :]
  val = ((1 - wu1) * (1 - wv1) * sd[bi, ni, i0, j0] + (1 - wu1) * wv1 * sd[bi, ni, i0, j1]
         + wu1 * (1 - wv1) * sd[bi, ni, i1, j0] + wu1 * wv1 * sd[bi, ni, i1, j1])
  ok = valid_q[:, None, :].expand(B, P, Nq)
  if mask_oob:
    inb = (u >= 0) & (u < X) & (v >= 0) & (v < Y)
    tv = mv[bi, i0, j0] & mv[bi, i0, j1] & mv[bi, i1, j0] & mv[bi, i1, j1]
    ok = ok & inb & tv
  (val * ok).sum(-1).backward(dscores.double())
  got = ops_bwd.pose_score_bwd(G(dscores), G(poses), G(q_xy), G(valid_q), G(mv), tuple(sim.shape),
                               cell, mask_oob=mask_oob)
  helpers.report('pose_score bwd', got, sd.grad.float(), atol=2e-4, rtol=1e-4)
  if not mask_oob:
    # the default is the fixed-point (exactly associative) accumulation: bitwise reproducible, and
    # at least as close to float64 as the float-atomics form; cotangents of any magnitude
    for _ in range(3):
      again = ops_bwd.pose_score_bwd(G(dscores), G(poses), G(q_xy), G(valid_q), G(mv), tuple(sim.shape),
                                     cell, mask_oob=False)
      assert torch.equal(again, got)
    big = ops_bwd.pose_score_bwd(G(dscores * 2.0 ** 40), G(poses), G(q_xy), G(valid_q), G(mv),
                                 tuple(sim.shape), cell, mask_oob=False)
    assert torch.equal(big, got * 2.0 ** 40)            # (the scale is a power of two: exact)
    prev, ops_bwd.DETERMINISTIC_POSE_BWD = ops_bwd.DETERMINISTIC_POSE_BWD, False
    try:
      flt = ops_bwd.pose_score_bwd(G(dscores), G(poses), G(q_xy), G(valid_q), G(mv), tuple(sim.shape),
                                   cell, mask_oob=False)
    finally:
      ops_bwd.DETERMINISTIC_POSE_BWD = prev
    helpers.report('pose_score bwd (float atomics)', flt, sd.grad.float(), atol=2e-4, rtol=1e-4)
    # a NaN / Inf cotangent must REACH dsim (trainer.py:260-277: the non-finite step skip and the
    # DynamicScale back-off look at the gradients): the fixed-point conversion must not turn it into
    # a finite number.  Scene 0 poisoned, scene 1 clean: scene 1's planes stay bit-identical.
    for poison in (float('nan'), float('inf'), -float('inf')):
      ds = dscores.clone()
      ds[0, 17] = poison
      bad = ops_bwd.pose_score_bwd(G(ds), G(poses), G(q_xy), G(valid_q), G(mv), tuple(sim.shape),
                                   cell, mask_oob=False)
      assert not bool(torch.isfinite(bad[0][G(valid_q)[0]]).all()), poison
      assert bool((bad[0][~G(valid_q)[0]] == 0).all())
      assert torch.equal(bad[1], got[1])


def test_confidence_head_bwd():
  """VJP of where(valid, log_sigmoid(f . w + b), 0) (bev_mapper.py:154-157,292-295)."""
  from snap_amd import autograd as ag
  M, D = 700, 64
  f = rnd((M, D), 300); w = rnd((D, 1), 301, 0.3); b = torch.tensor([0.2])
  valid = torch.rand(M, generator=torch.Generator().manual_seed(302)) > 0.2
  g = rnd((M,), 303)
  fd, wd, bd = (t.double().requires_grad_(True) for t in (f, w, b))
  out = torch.where(valid, torch.nn.functional.logsigmoid((fd @ wd)[:, 0] + bd[0]), torch.zeros(M, dtype=torch.float64))
  out.backward(g.double())
  fg, wg, bg = (G(t).requires_grad_(True) for t in (f, w, b))
  ag.confidence_head(fg, G(valid), wg, bg).backward(G(g))
  helpers.report('confidence d features', fg.grad, fd.grad.float(), atol=1e-5)
  helpers.report('confidence d kernel', wg.grad, wd.grad.float(), atol=2e-4)
  helpers.report('confidence d bias', bg.grad, bd.grad.float(), atol=2e-4)


def test_similarity_with_confidence_weights_bwd():
  """add_confidence_query (bev_localizer.py:165-168): sim = relu(fq . fm) exp(T) w[b, n] with
  w = layers.masked_softmax(confidence, valid points): gradients of fq, fm, T and the confidence."""
  from snap_amd import autograd as ag
  B, Nq, X, Y, Dm = 2, 50, 12, 16, 32
  fq = F.normalize(rnd((B, Nq, Dm), 310), dim=-1)
  fm = F.normalize(rnd((B, X, Y, Dm), 311), dim=-1)
  conf = rnd((B, Nq), 312)
  valid = torch.rand((B, Nq), generator=torch.Generator().manual_seed(313)) > 0.2
  nv = valid.sum(-1).float()
  temp = torch.tensor(2.0)
  dsim = rnd((B, Nq, X, Y), 314)
  fqd, fmd, td, cd = (t.double().requires_grad_(True) for t in (fq, fm, temp, conf))
  wref = torch.softmax(torch.where(valid, cd, torch.full_like(cd, -math.inf)), -1)
  simr = torch.relu(torch.einsum('bnd,bxyd->bnxy', fqd, fmd)) * torch.exp(td) * wref[..., None, None]
  simr.backward(dsim.double())
  fqg, fmg, tg, cg = (G(t).requires_grad_(True) for t in (fq, fm, temp, conf))
  wg, _ = ag.masked_softmax_rows(cg, G(valid))
  sim, _, _, _ = ag.sim_softmax_weighted(fqg, fmg, tg, wg, True, G(nv))
  helpers.report('weighted sim', sim, simr.detach().float(), atol=1e-6, rtol=1e-4)
  sim.backward(G(dsim))
  for name, got, want in (('fq', fqg, fqd), ('fm', fmg, fmd), ('temperature', tg, td), ('confidence', cg, cd)):
    helpers.report(f'weighted sim d {name}', got.grad, want.grad.float(),
                   atol=3e-4 * float(want.grad.abs().max()) + 1e-7)


def test_similarity_bwd():
  from snap_amd import autograd as ag
  B, Nq, X, Y, Dm = 2, 50, 12, 16, 32
  fq = F.normalize(rnd((B, Nq, Dm), 110), dim=-1)
  fm = F.normalize(rnd((B, X, Y, Dm), 111), dim=-1)
  nv = torch.tensor([48.0, 50.0])
  temp = torch.tensor(2.0)
  dsim = rnd((B, Nq, X, Y), 112)
  fqd, fmd, td = (t.double().requires_grad_(True) for t in (fq, fm, temp))
  sim = torch.relu(torch.einsum('bnd,bijd->bnij', fqd, fmd)) * torch.exp(td) / nv.double()[:, None, None, None]
  sim.backward(dsim.double())
  simg, _, _, _ = ops.sim_softmax(G(fq), G(fm), float(torch.exp(temp)), True, G(nv))
  dfq, dfm, dtemp = ag.similarity_bwd(G(dsim).clone(), simg, G(fq), G(fm), float(torch.exp(temp)),
                                      True, G(nv))
  helpers.report('dfq', dfq, fqd.grad.float(), atol=2e-5, rtol=1e-4)
  helpers.report('dfm', dfm, fmd.grad.float(), atol=2e-5, rtol=1e-4)
  assert abs(float(dtemp) - float(td.grad)) < 1e-3 * max(1.0, abs(float(td.grad)))


@pytest.mark.parametrize('cin,cs,pro', [(65, 68, ops.PRO_NONE), (33, 36, ops.PRO_RELU), (65, 65, ops.PRO_NONE)])
def test_dense_autograd_ragged_channels(cin, cs, pro):
  """Dense over the first `cin` of `cs` stored channels (the fusion-MLP input is
  mean|var|score = 2D+1 wide): dx must be zero past cin, dw/dbias match torch."""
  from snap_amd import autograd as ag
  M, Cout = 300, 64
  x = rnd((M, cs), 120)
  w = rnd((cin, Cout), 121, 1 / math.sqrt(cin))
  b = rnd((Cout,), 122)
  dy = rnd((M, Cout), 123)
  xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
  z = xd[:, :cin]
  z = torch.relu(z) if pro == ops.PRO_RELU else z
  torch.relu(z @ wd + bd).backward(dy.double())
  xg, wg, bg = (G(t).requires_grad_(True) for t in (x, w, b))
  y = ag.dense(xg, wg, bg, cin=cin, prologue=pro, relu=True)
  y.backward(G(dy))
  helpers.report('dx', xg.grad, xd.grad.float(), atol=3e-5, rtol=1e-5)
  helpers.report('dw', wg.grad, wd.grad.float(), atol=2e-4, rtol=1e-5)
  helpers.report('db', bg.grad, bd.grad.float(), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize('layers_,in_dim,stride,relu_in', [((64, 32), 65, 68, False), ((128,), 257, 260, False),
                                                            ((256, 128), 257, 260, True)])
def test_masked_rows_mlp_gradients_match_dense_path(layers_, in_dim, stride, relu_in, monkeypatch):
  """Row-list MLP (observed voxels only) vs the dense autograd MLP with the row mask on its
  last layer: same outputs, same parameter / input gradients (summation order aside)."""
  from snap_amd.models import layers
  from snap_amd.utils import config_dict
  cfg = config_dict.ConfigDict(dict(layers=layers_, activation='relu', apply_input_activation=relu_in))
  mlp = layers.MLP(cfg, in_dim=in_dim)
  gen = torch.Generator().manual_seed(11)
  params = helpers.params_to_device(mlp.init_params(gen, 'cpu'), 'cuda')
  for i in range(len(layers_)):
    params[f'Dense_{i}']['bias'] = torch.randn(layers_[i], generator=gen).cuda()
  M = 70001
  x = torch.randn(M, stride, generator=gen).cuda()
  mask = (torch.rand(M, generator=gen) < 0.6).cuda()
  dy = torch.randn(M, layers_[-1], generator=gen).cuda()

  def run(min_rows):
    monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', min_rows)
    leaves = [x.clone().requires_grad_(True)]
    p = {}
    for i in range(len(layers_)):
      k = params[f'Dense_{i}']['kernel'].clone().requires_grad_(True)
      b = params[f'Dense_{i}']['bias'].clone().requires_grad_(True)
      p[f'Dense_{i}'] = {'kernel': k, 'bias': b}
      leaves += [k, b]
    y = mlp(p, leaves[0], train=True, row_mask=mask)
    y.backward(dy)
    return y.detach(), [t.grad for t in leaves]

  y_d, g_d = run(1 << 30)     # dense autograd path
  y_c, g_c = run(0)           # row-list path
  assert torch.equal(y_d, y_c)
  names = ['dx'] + [f'd{n}{i}' for i in range(len(layers_)) for n in ('W', 'b')]
  for nm, a, b in zip(names, g_c, g_d):
    scale = float(b.abs().max())
    helpers.report(nm, a, b, atol=2e-5 * max(scale, 1.0), rtol=1e-4)
  # masked rows receive exactly zero input gradient
  assert bool((g_c[0][~mask] == 0).all())


@pytest.mark.parametrize('math_', ['bf16', 'fp16'])
@pytest.mark.parametrize('layers_,in_dim,stride,relu_in', [((256, 128), 257, 260, False), ((64, 32), 65, 68, True)])
def test_masked_rows_mlp_half_hidden_tensors_keep_the_bits(layers_, in_dim, stride, relu_in, math_, monkeypatch):
  """Training-precision engines: the hidden activation and the gradient w.r.t. it live ONLY in the
  engine's element type (``autograd.MASKED_MLP_HALF``: epilogue half output, half-input GEMMs by
  LDS-DMA, half-operand kernel-gradient loaders, half ReLU gate).  Every consumer of those tensors
  rounds them to that type anyway: outputs, input gradient, both kernel gradients and the last bias
  gradient are BIT-IDENTICAL to the f32-tensor formulation; the first layer's bias gradient sums the
  rounded instead of the unrounded gated gradient (checked to 1e-2 of its scale)."""
  from snap_amd import autograd as ag
  from snap_amd.models import layers
  from snap_amd.utils import config_dict
  cfg = config_dict.ConfigDict(dict(layers=layers_, activation='relu', apply_input_activation=relu_in))
  mlp = layers.MLP(cfg, in_dim=in_dim)
  gen = torch.Generator().manual_seed(12)
  params = helpers.params_to_device(mlp.init_params(gen, 'cpu'), 'cuda')
  for i in range(len(layers_)):
    params[f'Dense_{i}']['bias'] = torch.randn(layers_[i], generator=gen).cuda()
  M = 50001
  x = torch.randn(M, stride, generator=gen).cuda()
  x[:, in_dim:] = 0
  mask = (torch.rand(M, generator=gen) < 0.6).cuda()
  dy = torch.randn(M, layers_[-1], generator=gen).cuda()
  monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', 0)
  monkeypatch.setattr(ops, 'MATMUL_PRECISION', math_)

  def run(half):
    monkeypatch.setattr(ag, 'MASKED_MLP_HALF', half)
    leaves = [x.clone().requires_grad_(True)]
    p = {}
    for i in range(len(layers_)):
      k = params[f'Dense_{i}']['kernel'].clone().requires_grad_(True)
      b = params[f'Dense_{i}']['bias'].clone().requires_grad_(True)
      p[f'Dense_{i}'] = {'kernel': k, 'bias': b}
      leaves += [k, b]
    y = mlp(p, leaves[0], train=True, row_mask=mask)
    y.backward(dy)
    return y.detach(), [t.grad for t in leaves]

  y_f, g_f = run(False)
  y_h, g_h = run(True)
  assert torch.equal(y_f, y_h)
  for nm, a, b in zip(('dx', 'dW0', 'db0', 'dW1', 'db1'), g_h, g_f):
    if nm == 'db0':
      assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max()), nm
    elif nm == 'dx' and in_dim % 128 == 1:
      # (that channel's data gradient is a row-wise dot product of the gate pass on the half path --
      #  ``MASKED_MLP_DX_TAIL`` --, a column of the GEMM on the other; the padding columns stay zero)
      c = in_dim - 1
      assert torch.equal(a[:, :c], b[:, :c]), (nm, float((a[:, :c] - b[:, :c]).abs().max()))
      assert float((a[:, c] - b[:, c]).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, nm
      assert bool((a[:, in_dim:] == 0).all()) and bool((b[:, in_dim:] == 0).all())
      monkeypatch.setattr(ag, 'MASKED_MLP_DX_TAIL', False)
      _, g_t = run(True)
      monkeypatch.setattr(ag, 'MASKED_MLP_DX_TAIL', True)
      assert torch.equal(g_t[0], b), 'dx without the tail pass'
    elif nm == 'dW0' and in_dim % 128 == 1:
      # (the one channel above the 128-channel tiles is a weighted column sum of the gate pass on the
      #  half path, a narrow GEMM launch on the other: same rounded operands, another summation order)
      assert torch.equal(a[:-1], b[:-1]), (nm, float((a[:-1] - b[:-1]).abs().max()))
      assert float((a[-1] - b[-1]).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, nm
    else:
      assert torch.equal(a, b), (nm, float((a - b).abs().max()))
  assert float(g_h[0].abs().max()) > 0 and bool((g_h[0][~mask] == 0).all())


@pytest.mark.parametrize('math_', ['f32', 'bf16'])
def test_masked_rows_mlp_with_an_empty_row_list(math_, monkeypatch):
  """No observed voxel at all (a query that sees nothing): the row-list GEMMs, the wide kernel-gradient plan, the
  gate pass with its weighted sums / tail column and the vector column sums all run over ZERO rows -- outputs and
  every gradient are exact zeros, nothing is read out of bounds (M large enough for the wide plan)."""
  from snap_amd.models import layers
  from snap_amd.utils import config_dict
  cfg = config_dict.ConfigDict(dict(layers=(256, 128), activation='relu', apply_input_activation=False))
  mlp = layers.MLP(cfg, in_dim=257)
  gen = torch.Generator().manual_seed(13)
  params = helpers.params_to_device(mlp.init_params(gen, 'cpu'), 'cuda')
  M = 70000
  x = torch.randn(M, 260, generator=gen).cuda().requires_grad_(True)
  mask = torch.zeros(M, dtype=torch.bool, device='cuda')
  monkeypatch.setattr(layers.MLP, 'COMPACT_MIN_ROWS', 0)
  monkeypatch.setattr(ops, 'MATMUL_PRECISION', math_)
  p = {k: {n: t.clone().requires_grad_(True) for n, t in v.items()} for k, v in params.items()}
  y = mlp(p, x, train=True, row_mask=mask)
  assert bool((y == 0).all())
  y.backward(torch.randn(M, 128, generator=gen).cuda())
  assert bool((x.grad == 0).all())
  for v in p.values():
    for n, t in v.items():
      assert t.grad is not None and bool((t.grad == 0).all()), n
  # one observed row: the same kernels over a single row
  mask[12345] = True
  x.grad = None
  y = mlp(p, x, train=True, row_mask=mask)
  y.backward(torch.ones(M, 128, device='cuda'))
  assert bool((x.grad[~mask] == 0).all()) and float(x.grad[12345].abs().max()) > 0
  assert bool((x.grad[12345, 257:] == 0).all())


@pytest.mark.parametrize('mode', ['softmax', 'weighted'])
def test_vertical_pool_conf_bwd(mode):
  from snap_amd import autograd as ag
  g = torch.Generator().manual_seed(140)
  lead, Z, D = (2, 5, 4), 60, 128
  vol = torch.randn(*lead, Z, D, generator=g)
  valid = torch.rand(*lead, Z, generator=g) < 0.5
  valid[0, 0, 0] = False
  valid[1, 2, 3] = True
  w = torch.randn(D, 1, generator=g) * 0.3
  b = torch.randn(1, generator=g)
  dplane = torch.randn(*lead, D, generator=g)
  vd, wd, bd = (t.double().requires_grad_(True) for t in (vol, w, b))
  s = (vd @ wd)[..., 0] + bd[0]
  if mode == 'weighted':
    s = F.logsigmoid(s)
  any_ = valid.any(-1, keepdim=True)
  where = torch.where(any_, valid, torch.ones_like(valid))
  p = torch.softmax(s.masked_fill(~where, -float('inf')), -1)
  p = torch.where(valid, p, torch.zeros_like(p))
  plane = (vd * p[..., None]).sum(-2) * any_.double()
  plane.backward(dplane.double())
  vg, wg, bg = (G(t).requires_grad_(True) for t in (vol, w, b))
  out, pv, sc, wt = ag.vertical_pool_conf(vg, G(valid), wg, bg, mode == 'weighted')
  helpers.report('plane', out, plane.detach().float(), atol=2e-5, rtol=2e-5)
  out.backward(G(dplane))
  helpers.report('dvol', vg.grad, vd.grad.float(), atol=2e-5, rtol=1e-4)
  helpers.report('dw', wg.grad, wd.grad.float(), atol=2e-4, rtol=1e-4)
  helpers.report('db', bg.grad, bd.grad.float(), atol=2e-4, rtol=1e-4)


# -- ViT pieces (no reference ViT exists; fp reference = torch fp64 autograd of the published ops) --
@pytest.mark.parametrize('M,C', [(70, 768), (33, 192), (5, 1024)])
def test_layer_norm_bwd(M, C):
  x = rnd((M, C), 301) * 1.3 + 0.2
  gamma, beta = rnd((C,), 302) * 0.3 + 1, rnd((C,), 303) * 0.1
  dy = rnd((M, C), 304)
  xd, gd, bd = (t.double().requires_grad_(True) for t in (x, gamma, beta))
  F.layer_norm(xd, (C,), gd, bd, eps=1e-6).backward(dy.double())
  dx, dgamma, dbeta = ops_bwd.layer_norm_bwd(G(x), G(dy), G(gamma))
  helpers.report('ln dx', dx, xd.grad.float(), atol=2e-5, rtol=1e-4)
  helpers.report('ln dgamma', dgamma, gd.grad.float(), atol=1e-4, rtol=1e-4)
  helpers.report('ln dbeta', dbeta, bd.grad.float(), atol=1e-4, rtol=1e-4)


def test_gelu_fwd_bwd():
  x = rnd((257, 64), 305) * 2.5
  dy = rnd((257, 64), 306)
  xd = x.double().requires_grad_(True)
  y = F.gelu(xd, approximate='tanh')
  y.backward(dy.double())
  helpers.report('gelu', ops.gelu(G(x)), y.detach().float(), atol=2e-6, rtol=1e-5)
  helpers.report('gelu bwd', ops_bwd.gelu_bwd(G(x), G(dy)), xd.grad.float(), atol=5e-6, rtol=1e-5)


@pytest.mark.parametrize('B,N,H', [(2, 200, 2), (1, 512, 3), (1, 65, 1)])
def test_attention_bwd(B, N, H):
  """dqkv of the bf16 attention VJP vs torch fp64 autograd of softmax(q k^T / 8) v on the
  bf16-rounded operands.  bf16-class tolerance: 2e-2 of each gradient's range (P, dS and dO are
  rounded to bf16 before their matrix-core products)."""
  from oracle import encoder as o_enc
  qkv = rnd((B, N, 3, H, 64), 310 + N)
  qkv[:, :, 0] *= 1.5
  dout = rnd((B, N, H * 64), 311 + N)
  out, lse = ops.attention(G(qkv), want_lse=True)
  dqkv = ops_bwd.attention_bwd(G(qkv), out, G(dout), lse)
  r = torch.from_numpy(o_enc.bf16_round(qkv.numpy())).double().requires_grad_(True)
  q, k, v = (r[:, :, i].permute(0, 2, 1, 3) for i in range(3))
  ref = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B, N, H * 64)
  ref.backward(dout.double())
  helpers.report('attention fwd (lse path)', out, ref.detach().float(),
                 atol=1e-2 * float(ref.detach().abs().max()), rtol=0)
  for i, name in enumerate(('dq', 'dk', 'dv')):
    want = r.grad[:, :, i].float()
    helpers.report(f'attention {name} B{B} N{N} H{H}', dqkv[:, :, i], want,
                   atol=2e-2 * float(want.abs().max()), rtol=0)
  # the saved statistic: base-2 log-sum-exp of the scaled scores
  s = torch.einsum('bhqd,bhkd->bhqk', q, k).detach() * 0.125
  want_lse = torch.logsumexp(s, -1) / math.log(2.0)
  helpers.report('attention lse', lse, want_lse.float(), atol=2e-2, rtol=0)


def test_semantic_embed_table_gradients():
  """d tables of the semantic-raster embedding (one-hot^T @ dy on the wgrad engine) vs torch
  autograd of the indexing expression of semantic_raster_encoder.py:63-79."""
  from snap_amd import autograd as ag
  g = torch.Generator().manual_seed(41)
  N, nr, no, E = 7, 3, 4, 8
  idx_road, idx_other = [0, 2, 5], [1, 3, 4, 6]
  rasters = torch.rand((2, 11, 9, N), generator=g) < 0.5
  t_road = torch.randn((nr, E), generator=g)
  t_other = torch.randn((2 * no, E), generator=g)
  dy = torch.randn((2, 11, 9, (1 + no) * E), generator=g)
  tr, to = t_road.double().requires_grad_(True), t_other.double().requires_grad_(True)
  label = torch.argmax(rasters[..., idx_road].int(), dim=-1)
  f_road = tr[label]
  lab_o = torch.arange(no) + rasters[..., idx_other].long()
  f_other = to[lab_o].reshape(2, 11, 9, no * E)
  (torch.cat([f_road, f_other], -1) * dy.double()).sum().backward()
  a, b = G(t_road).requires_grad_(True), G(t_other).requires_grad_(True)
  out = ag.semantic_embed(G(rasters), idx_road, idx_other, a, b)
  ga, gb = torch.autograd.grad((out * G(dy)).sum(), [a, b])
  helpers.report('d table_road', ga, tr.grad.float(), atol=1e-4, rtol=1e-5)
  helpers.report('d table_other', gb, to.grad.float(), atol=1e-4, rtol=1e-5)
