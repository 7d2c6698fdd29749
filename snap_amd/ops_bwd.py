"""Host-side wrappers of the backward (training-path) entry points of the C ABI.

Same conventions as :mod:`snap_amd.ops`: GPU tensors only, no fallback.
"""
import ctypes
import weakref

import numpy as np
import torch

from snap_amd import _lib
from snap_amd import ops
from snap_amd.ops import _f32, _mask, _p, _region, _stream, POOLING

# The lift's VJP: records -> stable sort by image pixel -> gather (bitwise reproducible, no
# atomics); False = the scatter form with hardware float atomics (order-dependent sums).
DETERMINISTIC_LIFT_BWD = True
# The pose-scoring VJP accumulates its score planes in 64-bit fixed point (exactly associative);
# False = float LDS atomics (order-dependent sums).
DETERMINISTIC_POSE_BWD = True
# tests / tools: the half-wave-per-voxel record producer of the deterministic lift VJP (A/B)
LIFT_BWD_UNBATCHED = False


def _conv_desc(x_shape, w_shape, stride, padding, prologue, in_affine, cs=None):
  N, H, W, Cs = x_shape
  KH, KW, Cin, Cout = w_shape
  (pt, pb), (pl, pr) = padding
  Ho = (H + pt + pb - KH) // stride + 1
  Wo = (W + pl + pr - KW) // stride + 1
  return _lib.SnapConvDesc(
      N, H, W, Cin, Cs if cs is None else cs, KH, KW, stride, pt, pl, Ho, Wo, Cout, Cout,
      prologue, 0, float(in_affine[0]), float(in_affine[1]),
  ), (N, Ho, Wo, Cout)


WGRAD_NO_WIDE, WGRAD_NO_FUSED3 = 1 << 8, 1 << 9      # SnapConvDesc.tile_hint bits (snap_hip.h)


def conv2d_wgrad(x, dy, w_shape, *, stride=1, padding=((0, 0), (0, 0)), prologue=ops.PRO_NONE,
                 gn=None, in_affine=(1.0, 0.0), rows_z=None, rows_dy=None, row_count=None,
                 math=None, x_channel_offset=0, plans=3):
  """dw [KH,KW,Cin,Cout] = im2col(prologue(x))^T dy  (MFMA; deterministic split-M).

  plans (tools / tests, per call): bit 0 = the wide flat plan, bit 1 = the fused-tap 3 x 3 plan may be
  chosen where they apply (default both); a cleared bit pins the per-tap 128 x 128 tiles.
  rows_z / rows_dy / row_count: row lists over flat [1,1,M,C] operands (masked MLP).
  math: 'f32' | 'bf16' | 'fp16' (None = ``ops.MATMUL_PRECISION``), as ``ops.conv2d``.
  x_channel_offset (multiple of 4, f32 x): the kernel's Cin input channels start at that channel of
  x's rows (x keeps its row stride): the gradient of a channel slice without copying it."""
  math = ops.precision() if math is None else math
  if math in ops.SPLIT_PARTS:
    math = 'f32'         # the split engine has no weight-gradient kernel: exact f32 (trainer 'bf16x3')
  if math not in ('f32', 'bf16', 'fp16'):
    raise ValueError(f'conv2d_wgrad: math={math!r}')
  lib = _lib.load()
  x_half = x.dtype in (torch.bfloat16, torch.float16)
  if (WGRAD_DY_TWIN and math in HALF_DTYPE and not x_half and dy.dtype == torch.float32 and x.shape[-1] % 4 == 0
      and w_shape[2] >= 4 and w_shape[3] % 4 == 0 and (gn is None or w_shape[2] % 4 == 0)):
    # the gradient came out of the GroupNorm VJP with its rounded twin (``half_twin``): the engine rounds dy to
    # that type anyway -- read the twin (half the bytes, byte permutes instead of conversions)
    twin = half_twin(dy, math)
    if twin is not None:
      dy = twin
  dy_half = dy.dtype in (torch.bfloat16, torch.float16)
  if x_half or dy_half:
    # one operand already in the engine's element type (masked MLP: hidden activations / inter-layer
    # gradients): 2-byte loads and byte permutes instead of f32 loads and conversions
    want = HALF_DTYPE.get(math)
    if want is None or (x_half and dy_half) or (x_half and (x.dtype != want or prologue != ops.PRO_NONE)) or \
       (dy_half and dy.dtype != want):
      raise ValueError('conv2d_wgrad: a half-precision operand needs the matching engine (one operand, prologue NONE for x)')
    ops._chk(x, x.dtype, 'x'); ops._chk(dy, dy.dtype, 'dy')
  else:
    _f32(x, 'x'); _f32(dy, 'dy')
  d, yshape = _conv_desc(x.shape, w_shape, stride, padding, prologue, in_affine)
  d.tile_hint = (0 if plans & 1 else WGRAD_NO_WIDE) | (0 if plans & 2 else WGRAD_NO_FUSED3)
  if tuple(dy.shape) != yshape:
    raise ValueError(f'conv2d_wgrad: dy {tuple(dy.shape)} vs {yshape}')
  if x_channel_offset:
    if x_half or x_channel_offset % 4 or x_channel_offset + min(w_shape[2], 4) > x.shape[-1] or gn is not None:
      raise ValueError('conv2d_wgrad: x_channel_offset needs an f32 x, a multiple of 4 inside the row, no GroupNorm')
  for t, nm in ((rows_z, 'rows_z'), (rows_dy, 'rows_dy'), (row_count, 'row_count')):
    if t is not None:
      ops._chk(t, torch.int32, nm)
  mu = sc = beta = None
  if prologue in (ops.PRO_GN_RELU, ops.PRO_RELU_GN):
    mu, sc, beta = gn
  wsb = lib.snap_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=x.device)
  dw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
  KH, KW, Cin, Cout = w_shape
  M = yshape[0] * yshape[1] * yshape[2]
  kfl = 2.0 * KH * KW * Cin * Cout
  flops = kfl * M if row_count is None else (lambda: kfl * int(row_count.item()))
  bf16 = math in ops.HALF_MATH and x.shape[-1] % 4 == 0 and Cin >= 4
  code = (2 if math == 'fp16' else 1) if bf16 else 0          # SNAP_MATH_F16 / _BF16 / _F32
  with _region(('conv_wgrad_fp16' if code == 2 else 'conv_wgrad_bf16') if bf16 else 'conv_wgrad', flops,
               4.0 * (x.numel() + dy.numel()),
               tag=f'M{M}_K{KH}x{KW}x{Cin}_N{Cout}_s{stride}_p{prologue}' + ('_dyh' if dy_half else '') + ('_xh' if x_half else '')):
    st = lib.snap_conv2d_wgrad_half_f32(
        ctypes.byref(d), ctypes.c_void_p(x.data_ptr() + 4 * int(x_channel_offset)), _p(dy), _p(dw), _p(mu), _p(sc),
        _p(beta), 0, _p(ws),
        ws.numel() * 4, _p(rows_z), _p(rows_dy), _p(row_count), code, int(x_half), int(dy_half), _stream(),
    )
  _lib.check(st, 'snap_conv2d_wgrad_half_f32')
  return dw


# f32 gradient tensor -> its half-precision twin (bf16 / f16, same shape), written by the kernel
# that produced the gradient (``group_norm_bwd(half=...)``): the producing layer's data-gradient
# convolution reads the twin instead (both operands by LDS-DMA, half the bytes).  Keyed by data_ptr,
# checked against a weak reference, one entry per live gradient.
_HALF_TWINS = {}
HALF_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16}
USE_HALF_TWINS = True
WGRAD_DY_TWIN = True        # (tests / tools: False keeps the kernel gradients on the f32 dy)
USE_GNB_STATS = True        # (tests / tools: False = the GroupNorm VJP always takes its own statistics pass)


def half_twin(t, math):
  """The twin of gradient tensor ``t`` in the element type of ``math`` ('bf16' | 'fp16'), or None."""
  ref = _HALF_TWINS.get(t.data_ptr())
  owner = None if ref is None else ref()
  held = None if owner is None else getattr(owner, '_snap_half_twin', None)
  if held is None:
    return None
  twin, ver = held
  if (owner.data_ptr() != t.data_ptr() or owner.shape != t.shape or owner._version != ver
      or twin.dtype != HALF_DTYPE.get(math) or not t.is_contiguous()):
    return None
  return twin


def dense_wgrad_tail(cin, Cs):
  """(main, tail) channel split of ``dense_wgrad_rows`` for a [*, Cs] input with cin live channels,
  or None where the single launch stays."""
  tail = cin % 128
  if cin > 128 and 0 < tail <= 4 and Cs % 4 == 0 and Cs >= cin - tail + 4:
    return cin - tail, tail
  return None


def dense_wgrad_rows(x2, g, cin, H, *, prologue, rows_z, rows_dy, row_count, tail_row=None):
  """dW [cin, H] = prologue(x2[rows, :cin])^T g over a row list (the masked MLP's first layer).  A
  channel count just above a multiple of the engine's 128-channel tile (the fusion MLP: 257 = mean +
  variance + the view score) would spend a whole extra tile on its last 1-4 channels -- a third of the
  launch at 257; the tail quad goes through a narrow launch of its own instead (same sums per
  element, the row stride of x2 carries the zero padding of the quad)."""
  M, Cs = x2.shape
  split = dense_wgrad_tail(cin, Cs)
  if split is not None:
    main, tail = split
    dw_main = conv2d_wgrad(x2.reshape(1, 1, M, Cs), g.reshape(1, 1, M, H), (1, 1, main, H), prologue=prologue,
                           rows_z=rows_z, rows_dy=rows_dy, row_count=row_count).reshape(main, H)
    if tail_row is not None and tail == 1:      # (the caller took the one tail channel in its gate pass)
      return torch.cat([dw_main, tail_row.reshape(1, H)], 0)
    dw_tail = conv2d_wgrad(x2.reshape(1, 1, M, Cs), g.reshape(1, 1, M, H), (1, 1, 4, H), prologue=prologue,
                           rows_z=rows_z, rows_dy=rows_dy, row_count=row_count,
                           x_channel_offset=main).reshape(4, H)
    return torch.cat([dw_main, dw_tail[:tail]], 0)
  return conv2d_wgrad(x2.reshape(1, 1, M, Cs), g.reshape(1, 1, M, H), (1, 1, cin, H), prologue=prologue,
                      rows_z=rows_z, rows_dy=rows_dy, row_count=row_count).reshape(cin, H)


def group_norm_bwd(x, dz, mu, rstd, gamma, beta, mode, *, groups=32, add=None, half=None):
  """VJP of the fused GroupNorm(+ReLU) prologue.  Returns dx, dgamma, dbeta.  half ('bf16' | 'fp16'):
  dx is also written rounded to that type (``half_twin(dx, half)`` finds it)."""
  lib = _lib.load()
  _f32(x, 'x'); _f32(dz, 'dz'); _f32(mu, 'mu'); _f32(rstd, 'rstd')
  N, H, W, C = x.shape
  wsb = lib.snap_group_norm_bwd_workspace_bytes(N, H * W, C, groups)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=x.device)
  dx = torch.empty_like(x)
  dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
  dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
  twin = None
  if half in HALF_DTYPE and USE_HALF_TWINS and C % 8 == 0:
    twin = torch.empty(x.shape, dtype=HALF_DTYPE[half], device=x.device)
  # the statistics pass may already have been taken by the convolution that wrote dz (ops.conv2d(gn_bwd_stats=))
  stats, tile_rows = None, 0
  held = getattr(dz, '_snap_gnb_partial', None) if USE_GNB_STATS else None
  if (held is not None and held[2] == x.data_ptr() and held[3] == x._version and held[4] == int(mode)
      and dz.shape == x.shape and H * W >= held[1] > 0):
    stats, tile_rows = held[0], held[1]
  with _region('group_norm_bwd', 0.0, (16.0 if stats is None else 8.0) * x.numel()
               + (2.0 * x.numel() if twin is not None else 0.0)):
    st = lib.snap_group_norm_bwd_stats_f32(
        _p(x), _p(dz), _p(add), _p(dx), N, H * W, C, groups, _p(mu), _p(rstd), _p(gamma),
        _p(beta), mode, _p(dgamma), _p(dbeta), 0, _p(ws), ws.numel() * 4,
        _p(twin), 0 if twin is None else (2 if half == 'fp16' else 1), _p(stats), int(tile_rows), _stream(),
    )
  _lib.check(st, 'snap_group_norm_bwd_stats_f32')
  if twin is not None:
    # the twin lives exactly as long as its f32 tensor (an attribute of it; the table only holds weak
    # references): keeping it in the table would hold a whole step's twins -- gigabytes -- alive into
    # the next step and push the caching allocator into fresh hipMallocs in the middle of a run
    if len(_HALF_TWINS) > 256:
      for k in [k for k, v in _HALF_TWINS.items() if v() is None]:
        del _HALF_TWINS[k]
    dx._snap_half_twin = (twin, dx._version)
    _HALF_TWINS[dx.data_ptr()] = weakref.ref(dx)
  return dx, dgamma, dbeta


def weight_standardize_bwd(w, dws, eps=1e-10):
  lib = _lib.load()
  _f32(w, 'w'); _f32(dws, 'dws')
  dw = torch.empty_like(w)
  K = w.shape[0] * w.shape[1] * w.shape[2]
  st = lib.snap_weight_standardize_bwd_f32(_p(w), _p(dws), _p(dw), K, w.shape[3], eps, _stream())
  _lib.check(st, 'snap_weight_standardize_bwd_f32')
  return dw


def max_pool_3x3s2_bwd(x, dy):
  lib = _lib.load()
  _f32(x, 'x'); _f32(dy, 'dy')
  N, H, W, C = x.shape
  dx = torch.empty_like(x)
  st = lib.snap_max_pool_3x3s2_bwd_f32(_p(x), _p(dy), _p(dx), N, H, W, C, _stream())
  _lib.check(st, 'snap_max_pool_3x3s2_bwd_f32')
  return dx


def upsample2x_bwd(dy):
  lib = _lib.load()
  _f32(dy, 'dy')
  N, Ho, Wo, C = dy.shape
  dprev = torch.empty((N, Ho // 2, Wo // 2, C), dtype=torch.float32, device=dy.device)
  st = lib.snap_upsample2x_bwd_f32(_p(dy), _p(dprev), N, Ho // 2, Wo // 2, C, _stream())
  _lib.check(st, 'snap_upsample2x_bwd_f32')
  return dprev


def epilogue_bwd(dy, y=None, row_mask=None, relu=False):
  """dy gated by the forward epilogue: * [y > 0] (ReLU) * row_mask."""
  lib = _lib.load()
  _f32(dy, 'dy')
  C = dy.shape[-1]
  M = dy.numel() // C
  out = torch.empty_like(dy)
  st = lib.snap_epilogue_bwd_f32(_p(dy), _p(y), _p(row_mask), _p(out), M, C, int(relu), _stream())
  _lib.check(st, 'snap_epilogue_bwd_f32')
  return out


def epilogue_bwd_colsum(dy, y=None, row_mask=None, relu=False, row_count=None, wsum=None, dtail=None):
  """``epilogue_bwd`` and the column sums of its output (over the first *row_count rows) in one
  pass -> (gated dy, [C] sums); falls back to the two passes for widths the kernel does not take.
  dy / y in bf16 / f16 (both): the half kernel, gated dy in the same type.  wsum (half kernel only) =
  (x2 [rows, Cs] f32, channel, row list or None, relu): a third result, sum_r x2[rows[r], channel] *
  out[r, :] with the weight rounded to the element type -- one extra input channel's kernel-gradient
  row of the layer in front, in the same pass.  dtail (with wsum, C == 256) = (w_row [C] f32, dx [rows, Cs]
  f32): also that channel's DATA gradient, dx[rows[r], channel : channel + 4] = (out[r, :] . round(w_row),
  0, 0, 0) (``channel + 4 == Cs``: the zero padding of the row)."""
  lib = _lib.load()
  if dtail is not None and wsum is None:
    raise ValueError('epilogue_bwd_colsum: dtail goes with wsum')
  if wsum is not None and dy.dtype not in (torch.bfloat16, torch.float16):
    raise ValueError('epilogue_bwd_colsum: wsum goes with half tensors')
  if dy.dtype in (torch.bfloat16, torch.float16):
    C = dy.shape[-1]
    M = dy.numel() // C
    q = C // 4 if C < 1024 else 256
    if row_mask is not None or C % 4 or 256 % q or (C > 1024 and C % 1024) or (y is not None and y.dtype != dy.dtype):
      raise ValueError('epilogue_bwd_colsum: half tensors take no row mask, matching types, C % 4 == 0')
    ops._chk(dy, dy.dtype, 'dy')
    if y is not None:
      ops._chk(y, y.dtype, 'y')
    out = torch.empty_like(dy)
    wsb = lib.snap_colsum_workspace_bytes(M, C) * (2 if wsum is not None else 1)
    ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=dy.device)
    sums = torch.empty(C, dtype=torch.float32, device=dy.device)
    kind = 2 if dy.dtype == torch.float16 else 1
    if wsum is None:
      st = lib.snap_epilogue_bwd_colsum_half(_p(dy), _p(y), _p(out), M, C, int(relu), _p(row_count), _p(sums),
                                             _p(ws), ws.numel() * 4, kind, _stream())
      _lib.check(st, 'snap_epilogue_bwd_colsum_half')
      return out, sums
    x2, channel, wrows, wrelu = wsum
    _f32(x2, 'wsum source')
    if wrows is not None:
      ops._chk(wrows, torch.int32, 'wsum rows')
    wout = torch.empty(C, dtype=torch.float32, device=dy.device)
    wt = dt = None
    dstride = 0
    if dtail is not None:
      w_row, dx = dtail
      _f32(w_row, 'dtail weights'); _f32(dx, 'dtail dx')
      dstride = int(dx.shape[-1])
      if w_row.numel() != C or C != 256 or int(channel) + 4 != dstride or dstride % 4:
        raise ValueError('epilogue_bwd_colsum: dtail needs C == 256 and the channel in the last quad of dx rows')
      wt, dt = ctypes.c_void_p(w_row.data_ptr()), ctypes.c_void_p(dx.data_ptr() + 4 * int(channel))
    st = lib.snap_epilogue_bwd_colsum_wsum_tail_half(
        _p(dy), _p(y), _p(out), M, C, int(relu), _p(row_count), _p(sums), _p(ws), ws.numel() * 4, kind,
        ctypes.c_void_p(x2.data_ptr() + 4 * int(channel)), _p(wrows), int(x2.shape[-1]), int(bool(wrelu)),
        _p(wout), wt, dt, dstride, _stream())
    _lib.check(st, 'snap_epilogue_bwd_colsum_wsum_tail_half')
    return out, sums, wout
  _f32(dy, 'dy')
  C = dy.shape[-1]
  M = dy.numel() // C
  q = C // 4 if C < 1024 else 256
  if C % 4 or 256 % q or (C > 1024 and C % 1024):
    out = epilogue_bwd(dy, y, row_mask, relu=relu)
    return out, colsum(out, row_count=row_count)
  out = torch.empty_like(dy)
  wsb = lib.snap_colsum_workspace_bytes(M, C)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=dy.device)
  sums = torch.empty(C, dtype=torch.float32, device=dy.device)
  st = lib.snap_epilogue_bwd_colsum_f32(_p(dy), _p(y), _p(row_mask), _p(out), M, C, int(relu),
                                        _p(row_count), _p(sums), _p(ws), ws.numel() * 4, _stream())
  _lib.check(st, 'snap_epilogue_bwd_colsum_f32')
  return out, sums


def colsum(a, rows=None, row_count=None):
  """Column sums of a [..., C] -> [C]  (bias gradients); optionally over a row list."""
  lib = _lib.load()
  _f32(a, 'a')
  C = a.shape[-1]
  M = a.numel() // C
  wsb = lib.snap_colsum_workspace_bytes(M, C)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=a.device)
  out = torch.empty(C, dtype=torch.float32, device=a.device)
  st = lib.snap_colsum_rows_f32(_p(a), M, C, _p(rows), _p(row_count), _p(out), 0, _p(ws),
                                ws.numel() * 4, _stream())
  _lib.check(st, 'snap_colsum_rows_f32')
  return out


def _aligned_ws(nbytes, device):
  ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
  return ws, ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)


def lift_pool_bwd(f_images, cam, Rt, points, dpooled, *, K, fisheye, feature_dim, num_bins,
                  depth_min_max, max_view_distance=None, weighted=True, use_variance=True,
                  add_minmax=False):
  """VJP of ``ops.lift_pool`` w.r.t. f_images, every fusion option (the non-default ones on the
  deterministic form only)."""
  lib = _lib.load()
  _f32(f_images, 'f_images'); _f32(dpooled, 'dpooled')
  B, V, h, w, C = f_images.shape
  N = points.shape[1]
  default = weighted and use_variance and not add_minmax
  d = _lib.SnapLiftDesc(
      B, V, h, w, C, feature_dim, num_bins if weighted else 0, N, K, int(fisheye), dpooled.shape[-1],
      float(depth_min_max[0]), float(depth_min_max[1]),
      -1.0 if max_view_distance is None else float(max_view_distance),
      int(weighted), int(use_variance), int(add_minmax),
  )
  d.tune_flags = int(bool(LIFT_BWD_UNBATCHED))
  df = torch.empty_like(f_images)
  wsb = (lib.snap_lift_pool_bwd_det_workspace_bytes(ctypes.byref(d))
         if (DETERMINISTIC_LIFT_BWD or not default) else 0)
  if not wsb and not default:
    raise ValueError('lift_pool_bwd: the non-default fusion options need the deterministic form '
                     '(<= 8 selected views, <= 32 depth bins)')
  if wsb:
    # records -> stable sort by image pixel -> gather: no atomics, bitwise reproducible
    ws, wsp = _aligned_ws(wsb, f_images.device)
    with _region('lift_pool_bwd', 0.0, 4.0 * (f_images.numel() * 2 + dpooled.numel())):
      st = lib.snap_lift_pool_bwd_det_f32(
          ctypes.byref(d), _p(f_images), _p(cam), _p(Rt), _p(points), _p(dpooled), _p(df), wsp, wsb,
          _stream()
      )
    _lib.check(st, 'snap_lift_pool_bwd_det_f32')
    return df
  with _region('lift_pool_bwd', 0.0, 4.0 * (f_images.numel() * 2 + dpooled.numel())):
    st = lib.snap_lift_pool_bwd_f32(
        ctypes.byref(d), _p(f_images), _p(cam), _p(Rt), _p(points), _p(dpooled), _p(df), _stream()
    )
  _lib.check(st, 'snap_lift_pool_bwd_f32')
  return df


def lift_pool_observations_bwd(obs_feat, f_shape, cam, Rt, points, dpooled, *, K, fisheye, feature_dim,
                               max_view_distance=None, use_variance=True, add_minmax=False):
  """VJP of ``ops.lift_pool_observations`` w.r.t. the observations [B, N, S, fd]."""
  lib = _lib.load()
  _f32(obs_feat, 'obs_feat'); _f32(dpooled, 'dpooled')
  B, N = points.shape[:2]
  d = ops._obs_desc(f_shape, N, K, fisheye, feature_dim, max_view_distance, use_variance, add_minmax,
                    dpooled.shape[-1])
  dobs = torch.empty_like(obs_feat)
  with _region('lift_pool_bwd', 0.0, 4.0 * (obs_feat.numel() * 2 + dpooled.numel())):
    st = lib.snap_lift_pool_observations_bwd_f32(ctypes.byref(d), _p(cam), _p(Rt), _p(points), _p(obs_feat),
                                                 _p(dpooled), _p(dobs), _stream())
  _lib.check(st, 'snap_lift_pool_observations_bwd_f32')
  return dobs


def lift_observations_bwd(dobs, f_shape, cam, Rt, points, *, K, fisheye, feature_dim, max_view_distance=None):
  """VJP of ``ops.lift_observations`` w.r.t. f_images: dobs [B, N, S, fd] = the gradient of the
  feature part of ``obs`` plus that of ``obs_feat`` (depth and ray carry none) -> [B, V, h, w, fd]."""
  lib = _lib.load()
  _f32(dobs, 'dobs')
  B, N = points.shape[:2]
  d = ops._obs_desc(f_shape, N, K, fisheye, feature_dim, max_view_distance, True, False, 0)
  wsb = lib.snap_lift_observations_bwd_workspace_bytes(ctypes.byref(d))
  if not wsb:
    raise ValueError('lift_observations_bwd: unsupported shape (> 8 selected views)')
  ws, wsp = _aligned_ws(wsb, dobs.device)
  df = torch.empty(tuple(f_shape), dtype=torch.float32, device=dobs.device)
  with _region('lift_pool_bwd', 0.0, 4.0 * (dobs.numel() + df.numel())):
    st = lib.snap_lift_observations_bwd_f32(ctypes.byref(d), _p(cam), _p(Rt), _p(points), _p(dobs), _p(df),
                                            wsp, wsb, _stream())
  _lib.check(st, 'snap_lift_observations_bwd_f32')
  return df


def vertical_pool_bwd(vol, valid, dplane, pooling='max', arg=None):
  """arg = (argz, ties) of ``ops.vertical_pool(want_arg=True)``: max pooling without a pass over vol."""
  lib = _lib.load()
  _f32(vol, 'vol'); _mask(valid, 'valid'); _f32(dplane, 'dplane')
  Z, D = vol.shape[-2:]
  M = vol.numel() // (Z * D)
  dvol = torch.empty_like(vol)
  if arg is not None:
    if pooling != 'max':
      raise ValueError('vertical_pool_bwd: arg goes with max pooling')
    argz, ties = arg
    ops._chk(argz, torch.uint8, 'argz'); ops._chk(ties, torch.uint8, 'ties')
    if argz.numel() != M * D or ties.numel() != M * D:
      raise ValueError('vertical_pool_bwd: argz / ties must be [..., D]')
    with _region('vertical_pool_bwd', 0.0, 4.0 * dvol.numel()):
      st = lib.snap_vertical_pool_max_bwd_arg_f32(_p(vol), _p(valid), _p(argz), _p(ties), _p(dplane), _p(dvol),
                                                  M, Z, D, _stream())
    _lib.check(st, 'snap_vertical_pool_max_bwd_arg_f32')
    return dvol
  with _region('vertical_pool_bwd', 0.0, 12.0 * dvol.numel()):
    st = lib.snap_vertical_pool_bwd_f32(
        _p(vol), _p(valid), _p(dplane), _p(dvol), M, Z, D, POOLING[pooling], _stream()
    )
  _lib.check(st, 'snap_vertical_pool_bwd_f32')
  return dvol


def plane_fuse_match_bwd(planes, valids, pooling, Wm, bm, normalize, eps, dmatching, dfused=None):
  """Returns (dplanes list, dy [M,Dm] or None)."""
  lib = _lib.load()
  n = len(planes)
  D = planes[0].shape[-1]
  M = planes[0].numel() // D
  dev = planes[0].device
  dplanes = [torch.empty_like(p) for p in planes]
  pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in planes])
  vv = (ctypes.c_void_p * n)(*[None if v is None else _mask(v, 'valid').data_ptr() for v in valids])
  dp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in dplanes])
  dy = None
  Dm = 0
  if dmatching is not None:
    _f32(dmatching, 'dmatching')
    Dm = Wm.shape[1]
    dy = torch.empty((M, Dm), dtype=torch.float32, device=dev)
  st = lib.snap_plane_fuse_match_bwd_f32(
      ctypes.cast(pp, ctypes.c_void_p), ctypes.cast(vv, ctypes.c_void_p),
      ctypes.cast(dp, ctypes.c_void_p), n, M, D, POOLING[pooling], _p(Wm), _p(bm), Dm,
      int(normalize), eps, _p(dmatching), _p(dfused), _p(dy), _stream(),
  )
  _lib.check(st, 'snap_plane_fuse_match_bwd_f32')
  return dplanes, dy


# Gradient buffers an op of this module allocated and handed to autograd, which their one consumer
# may overwrite in place instead of cloning first (data_ptr -> weak reference).  Autograd passes a
# single incoming gradient through unchanged (same storage); a sum of several gradients is a new
# tensor and never matches.  One-time: ``take_scratch`` removes the entry.
# ONLY inside ``owning_scratch_grads()``: the hand-over is safe when nobody else can see the
# gradient tensor -- the trainer's own backward pass (no retain_grad, no tensor hooks on intermediate
# gradients, no retain_graph).  A caller who differentiates the model himself (torch.autograd.grad with
# an intermediate as input, hooks, a second backward over a retained graph) gets the clone.
import contextlib
import weakref
_SCRATCH_GRADS = {}
_SCRATCH_OWNERS = 0


@contextlib.contextmanager
def owning_scratch_grads():
  """The enclosed backward pass owns every gradient tensor it produces (``trainer._forward_backward``)."""
  global _SCRATCH_OWNERS
  _SCRATCH_OWNERS += 1
  try:
    yield
  finally:
    _SCRATCH_OWNERS -= 1
    if _SCRATCH_OWNERS == 0:
      _SCRATCH_GRADS.clear()          # marks nobody consumed do not outlive the pass


def mark_scratch(t):
  if _SCRATCH_OWNERS <= 0:
    return t
  if len(_SCRATCH_GRADS) > 64:
    _SCRATCH_GRADS.clear()
  _SCRATCH_GRADS[t.data_ptr()] = (weakref.ref(t), t.numel())
  return t


def take_scratch(t):
  """True if ``t`` IS a buffer registered by ``mark_scratch`` (the caller may then modify it in place)."""
  hit = _SCRATCH_GRADS.pop(t.data_ptr(), None)
  if hit is None or not t.is_contiguous():
    return False
  ref = hit[0]()
  return ref is not None and hit[1] == t.numel() and ref.data_ptr() == t.data_ptr()


def pose_score_bwd(dscores, poses, q_xy, valid_q, map_valid, sim_shape, cell_size, mask_oob=False):
  lib = _lib.load()
  _f32(dscores, 'dscores'); _f32(poses, 'poses'); _f32(q_xy, 'q_xy'); _mask(valid_q, 'valid_q')
  B, Nq, X, Y = sim_shape
  P = poses.shape[1]
  wsb = lib.snap_pose_score_bwd_workspace_bytes(B, P)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=poses.device)
  dsim = mark_scratch(torch.empty(sim_shape, dtype=torch.float32, device=poses.device))
  with _region('pose_score_bwd', 0.0, 4.0 * dsim.numel()):
    st = lib.snap_pose_score_bwd_ex_f32(
        _p(dscores), _p(poses), _p(q_xy), _p(valid_q), _p(map_valid), B, Nq, X, Y, P,
        float(cell_size), int(mask_oob), 0 if DETERMINISTIC_POSE_BWD else 1, _p(dsim), _p(ws),
        ws.numel() * 4, _stream(),
    )
  _lib.check(st, 'snap_pose_score_bwd_ex_f32')
  return dsim


def sim_bwd_prepare_(dsim, sim, clip_negative, coef):
  """In place G = dsim * [sim>0] * coef[b]; returns sum(dsim * sim) per scene [B]."""
  lib = _lib.load()
  _f32(dsim, 'dsim'); _f32(sim, 'sim'); _f32(coef, 'coef')
  B = dsim.shape[0]
  per_scene = dsim.numel() // B
  nparts = 512
  partial = torch.empty((B, nparts), dtype=torch.float32, device=dsim.device)
  st = lib.snap_sim_bwd_prepare_f32(
      _p(dsim), _p(sim), B, per_scene, int(clip_negative), _p(coef), _p(partial), nparts, _stream()
  )
  _lib.check(st, 'snap_sim_bwd_prepare_f32')
  return partial.to(torch.float64).sum(-1)


def vertical_pool_conf_bwd(vol, valid, w, bias, weights, dplane, log_sigmoid_scores):
  """Returns (dvol, dw [D], dbias [1])."""
  lib = _lib.load()
  _f32(vol, 'vol'); _mask(valid, 'valid'); _f32(weights, 'weights'); _f32(dplane, 'dplane')
  Z, D = vol.shape[-2:]
  M = vol.numel() // (Z * D)
  rows = lib.snap_vertical_pool_conf_bwd_partial_rows(M)
  partial = torch.empty((rows, D + 4), dtype=torch.float32, device=vol.device)
  dvol = torch.empty_like(vol)
  st = lib.snap_vertical_pool_conf_bwd_f32(
      _p(vol), _p(valid), _p(w), _p(bias), _p(weights), _p(dplane), M, Z, D,
      int(log_sigmoid_scores), _p(dvol), _p(partial), _stream(),
  )
  _lib.check(st, 'snap_vertical_pool_conf_bwd_f32')
  sums = colsum(partial)
  return dvol, sums[:D].contiguous(), sums[D:D + 1].contiguous()


def layer_norm_bwd(x, dy, gamma, eps=1e-6):
  """VJP of ``ops.layer_norm``: dx, dgamma, dbeta (deterministic column sums)."""
  lib = _lib.load()
  _f32(x, 'x'); _f32(dy, 'dy'); _f32(gamma, 'gamma')
  C = x.shape[-1]
  M = x.numel() // C
  wsb = lib.snap_layer_norm_bwd_workspace_bytes(M, C)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=x.device)
  dx = torch.empty_like(x)
  dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
  dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
  with _region('layer_norm_bwd', 0.0, 12.0 * x.numel()):
    st = lib.snap_layer_norm_bwd_f32(_p(x), _p(dy), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), M, C,
                                     float(eps), _p(ws), ws.numel() * 4, _stream())
  _lib.check(st, 'snap_layer_norm_bwd_f32')
  return dx, dgamma, dbeta


def gelu_bwd(x, dy):
  lib = _lib.load()
  _f32(x, 'x'); _f32(dy, 'dy')
  dx = torch.empty_like(x)
  with _region('gelu_bwd', 0.0, 12.0 * x.numel()):
    st = lib.snap_gelu_bwd_f32(_p(x), _p(dy), _p(dx), x.numel(), _stream())
  _lib.check(st, 'snap_gelu_bwd_f32')
  return dx


def attention_bwd(qkv, out, dout, lse, scale=None):
  """VJP of ``ops.attention`` w.r.t. qkv (bf16 matrix cores, two atomic-free kernels)."""
  lib = _lib.load()
  _f32(qkv, 'qkv'); _f32(out, 'out'); _f32(dout, 'dout'); _f32(lse, 'lse')
  B, N, _, H, D = qkv.shape
  scale = D ** -0.5 if scale is None else float(scale)
  dqkv = torch.empty_like(qkv)
  delta = torch.empty_like(lse)
  with _region('attention_bwd', 10.0 * B * H * N * N * D, 8.0 * qkv.numel()):
    st = lib.snap_attention_bwd_bf16_f32(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv),
                                         B, N, H, D, scale, _stream())
  _lib.check(st, 'snap_attention_bwd_bf16_f32')
  return dqkv


def confidence_head_bwd(features, valid, kernel, bias, dconf):
  """VJP of ``ops.confidence_head``: (d features, d kernel [D], d bias [1])."""
  lib = _lib.load()
  _f32(features, 'features'); _f32(kernel, 'kernel'); _f32(bias, 'bias'); _f32(dconf, 'dconf')
  if valid is not None:
    _mask(valid, 'valid')
  D = features.shape[-1]
  M = features.numel() // D
  df = torch.empty_like(features)
  prod = torch.empty((M, D), dtype=torch.float32, device=features.device)
  dsv = torch.empty((M, 4), dtype=torch.float32, device=features.device)
  st = lib.snap_confidence_head_bwd_f32(_p(features), _p(valid), _p(kernel), _p(bias), _p(dconf), M, D,
                                        _p(df), _p(prod), _p(dsv), _stream())
  _lib.check(st, 'snap_confidence_head_bwd_f32')
  return df, colsum(prod), colsum(dsv)[:1]


def sim_bwd_prepare_rows_(dsim, sim, clip_negative, row_coef):
  """In place G = dsim * [sim > 0] * row_coef[b, n]; returns rowdot [B, Nq] = sum_cells dsim * sim."""
  lib = _lib.load()
  _f32(dsim, 'dsim'); _f32(sim, 'sim'); _f32(row_coef, 'row_coef')
  B, Nq, X, Y = sim.shape
  rowdot = torch.empty((B, Nq), dtype=torch.float32, device=sim.device)
  st = lib.snap_sim_bwd_prepare_rows_f32(_p(dsim), _p(sim), B, Nq, X * Y, int(clip_negative), _p(row_coef),
                                         _p(rowdot), _stream())
  _lib.check(st, 'snap_sim_bwd_prepare_rows_f32')
  return rowdot


def masked_softmax_rows_bwd(weights, dweights):
  lib = _lib.load()
  _f32(weights, 'weights'); _f32(dweights, 'dweights')
  B, N = weights.shape
  dx = torch.empty_like(weights)
  st = lib.snap_masked_softmax_rows_bwd_f32(_p(weights), _p(dweights), B, N, _p(dx), _stream())
  _lib.check(st, 'snap_masked_softmax_rows_bwd_f32')
  return dx


_ADAM_ITEM = np.dtype([('p', np.uint64), ('g', np.uint64), ('m', np.uint64), ('v', np.uint64),
                       ('n', np.int64), ('block_begin', np.int64)])


def adam_update_(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, apply_flag=None):
  """optax.adam (bias-corrected, eps outside the sqrt) over every tensor of the lists in ONE launch;
  in place on params / m / v (trainer.py:236-243).  ``step`` counts from 1.  apply_flag: optional
  0-d f32 DEVICE tensor -- the update is skipped unless it is > 0 (the non-finite step skip)."""
  lib = _lib.load()
  quads = [q for q in zip(params, grads, m, v) if q[0].numel()]       # (empty tensors: nothing to do)
  if not quads:
    return
  items = np.zeros(len(quads), dtype=_ADAM_ITEM)
  blk = 0
  keep = []
  seen = set()
  for i, (p, g, mi, vi) in enumerate(quads):
    _f32(p, 'param'); _f32(mi, 'm'); _f32(vi, 'v')
    g = _f32(g if g.is_contiguous() else g.contiguous(), 'grad')
    keep.append(g)
    if not (p.numel() == g.numel() == mi.numel() == vi.numel()):
      raise ValueError('adam_update_: tensor sizes differ')
    if p.data_ptr() in seen:
      raise ValueError('adam_update_: a parameter is listed twice (it would be updated twice)')
    seen.add(p.data_ptr())
    items[i] = (p.data_ptr(), g.data_ptr(), mi.data_ptr(), vi.data_ptr(), p.numel(), blk)
    blk += lib.snap_adam_multi_blocks(p.numel())
  table = ops.upload_table(items, quads[0][0].device)
  if apply_flag is not None:
    _f32(apply_flag, 'apply_flag')
  st = lib.snap_adam_multi_f32(_p(table), len(quads), blk, float(lr), float(b1), float(b2), float(eps),
                               int(step), _p(apply_flag), _stream())
  _lib.check(st, 'snap_adam_multi_f32')
  # the kernel writes through raw pointers: tell torch (and every cache keyed on ``_version``:
  # ops.host_exp's prefetched exp(temperature), base._WSTD_CACHE, the packed-weight images) that
  # params / m / v changed in place, as the torch._foreach_* path does
  for p, _, mi, vi in quads:
    torch.autograd.graph.increment_version(p)
    torch.autograd.graph.increment_version(mi)
    torch.autograd.graph.increment_version(vi)
