"""Exhaustive (x, y, theta) pose voting (``snap/models/pose_exhaustive_voting.py``).

``template_matching`` has two formulations of the same correlation:

* ``method='fft'`` (voting_fft.hip; the default where the geometry fits): FFT products of the
  templates and the edge-padded map, channel pairs packed as complex numbers, the overlap count the
  same way and rounded to the nearest integer -- ~1e-3 of the direct form's multiply-adds, scores
  within ~1e-6 (relative to the largest score) of the direct sum, the -inf mask identical;
* ``method='direct'``: the dense contraction (2*R*(2H-1)(2W-1)*H*W*D flop) on the MFMA conv engine
  with the R rotated templates as an (H x W x D) -> R filter bank over the edge-padded map -- the
  reference's own formulation (``jax.scipy.signal.convolve``, :86-91) and the checker of the other.

Rotation, padding and the -inf / normalise pass are small HIP kernels (voting.hip).
"""
import math

import numpy as np
import torch

from snap_amd import ops
from snap_amd.models import types
from snap_amd.utils import geometry


def get_grid_center_transform(grid, device=None):
  """corner_t_center (pose_exhaustive_voting.py:31-34)."""
  center = torch.tensor((np.asarray(grid.extent_meters) / 2).astype(np.float32), device=device)
  return geometry.Transform2D(torch.zeros((), device=device), center)


def _template_transforms(num_rotations, grid, device):
  """templates_t_grid = corner_t_center @ rot(theta) @ corner_t_center^-1 (:44-50)."""
  angles = torch.tensor(
      np.linspace(0, np.pi * 2, num_rotations, endpoint=False).astype(np.float32)
  )
  c = get_grid_center_transform(grid)
  n = len(angles)
  corner = geometry.Transform2D(c.angle.expand(n), c.t.expand(n, 2))
  rot = geometry.Transform2D(angles, torch.zeros(n, 2))
  t = corner @ rot @ corner.inv
  tfm = torch.stack([torch.cos(t.angle), torch.sin(t.angle), t.t[:, 0], t.t[:, 1]], -1)
  return tfm.to(device=device, dtype=torch.float32).contiguous()


def sample_query_templates(features, valid, num_rotations, grid, _engine=False):
  """Rotate a BEV by ``num_rotations`` angles (pose_exhaustive_voting.py:37-69)."""
  if num_rotations % 4 != 0:
    raise ValueError('num_rotations must be divisible by 4')
  if features.shape[0] != features.shape[1]:
    raise ValueError('the rot90 completion requires a square BEV')
  tfm = _template_transforms(num_rotations, grid, features.device)
  H, W = features.shape[:2]
  out = ops.rotate_templates(
      features.contiguous(), valid.contiguous(), tfm[: num_rotations // 4].contiguous(),
      num_rotations, grid.cell_size,
      want_tw=_engine and not _stacked(num_rotations, (H, W)),   # (the stacked path reads `templates`)
  )
  if _engine:
    return out
  return out[0], out[1]


# Shift-stacking factor of the correlation GEMM (0 / 1 = plain direct form).  With R = 36 templates
# the GEMM has 36 output columns -- 56 % of a 64-wide MFMA tile; stacking the S x S shifted copies
# of every template as extra filters (and striding the correlation by S) gives R S^2 = 576 columns
# = 9 full tiles for 2.4 % more multiply-adds: the same products, summed in the same k order.
STACK_SHIFT = 4
STACK_MIN_CELLS = 64 * 64      # below this the plain form is already launch-bound
# 'auto': the frequency-domain form from FFT_MIN_CELLS map cells on (below, the direct form is a
# handful of launch-bound kernels), 'fft' / 'direct': forced.  Per call: the ``method`` argument.
# Two behaviours of the frequency-domain form differ from the direct form and are why 'direct' stays
# selectable: (1) the transforms are GLOBAL -- one NaN / Inf anywhere in the map features (valid cell or
# not: the reference does not mask them out of the scores either) or in the query plane makes every
# score of every rotation NaN, where the direct form poisons only the placements that overlap the bad
# cell; (2) its workspace grows with R * D * N^2 (0.9 GB of first-axis spectra at R = 36, 256^2, D = 32):
# 'auto' falls back to the direct form beyond FFT_WORKSPACE_BUDGET bytes.
VOTING_METHOD = 'auto'
FFT_MIN_CELLS = 64 * 64
FFT_WORKSPACE_BUDGET = 16 << 30


def _use_fft(method, R, q_hw, D, m_hw):
  method = method or VOTING_METHOD
  if method not in ('auto', 'fft', 'direct'):
    raise ValueError(f'voting method {method!r}: expected auto | fft | direct')
  if method == 'direct':
    return False
  ok = ops.voting_fft_supported(R, q_hw[0], q_hw[1], D, m_hw[0], m_hw[1])
  if method == 'fft':
    if not ok:
      raise ValueError(f'voting method fft: map {m_hw} is beyond the transform sizes of voting_fft.hip')
    return True
  return (ok and m_hw[0] * m_hw[1] >= FFT_MIN_CELLS
          and ops.voting_fft_workspace_bytes(R, q_hw[0], q_hw[1], D, m_hw[0], m_hw[1]) <= FFT_WORKSPACE_BUDGET)


def _stacked(R, q_hw):
  S = STACK_SHIFT
  return S > 1 and q_hw[0] * q_hw[1] >= STACK_MIN_CELLS and (R * S * S) % 4 == 0


def _correlate(mp, tw, R, q_hw, templates=None):
  """``tw`` [H,W,D,R] and / or ``templates`` [R,H,W,D]: the same filter bank in two layouts."""
  H, W = q_hw
  S = STACK_SHIFT
  Ho, Wo = mp.shape[0] - H + 1, mp.shape[1] - W + 1
  if not _stacked(R, q_hw):
    if tw is None:
      tw = templates.permute(1, 2, 3, 0).contiguous()
    return ops.conv2d(mp[None], tw)[0]                  # [Ho, Wo, R]
  A4, B4 = -(-Ho // S), -(-Wo // S)
  pb = max(0, S * (A4 - 1) + (H + S - 1) - mp.shape[0])  # zero rows only cropped outputs can see
  pr = max(0, S * (B4 - 1) + (W + S - 1) - mp.shape[1])
  tws_shape = (H + S - 1, W + S - 1, mp.shape[-1], R * S * S)
  presplit = (ops.precision() == 'bf16x3' and ops.USE_PRESPLIT_VOTING and mp.shape[-1] % 16 == 0
              and (R * S * S) % 192 == 0
              and ops.conv2d_presplit_supported((1,) + tuple(mp.shape), tws_shape, S, ((0, pb), (0, pr))))
  tws = None
  if presplit and templates is not None and ops.FUSED_TEMPLATE_PACK:
    # the bank goes straight into the engine's weight image (no f32 bank, no pack pass)
    tws = ops.pack_stacked_templates_split(templates, S)
  if tws is None:
    tws = ops.stack_templates(templates, S, 'rhwd') if templates is not None else ops.stack_templates(tw, S)
  if presplit:
    # the correlation as ONE large GEMM on the pre-split engine: the map is split into its two
    # bf16 parts once (instead of once per tap and column tile inside the K loop), both operands
    # travel by LDS-DMA, 256 x 192 tiles (R S^2 = 576 = three column tiles); same products, same
    # k order, same bits as the split engine's im2col body.  Template banks beyond the engine's
    # 32-bit offsets (matching_dim 64 at 256^2, queries of ~360^2 cells) keep the plain-input
    # launch below, which drops to the f32 engine where the split weight image does not fit
    raw4 = ops.conv2d(ops.presplit(mp[None]), tws, stride=S, padding=((0, pb), (0, pr)), ps_tile=3)[0]
  else:
    raw4 = ops.conv2d(mp[None], tws, stride=S, padding=((0, pb), (0, pr)))[0]   # [A4, B4, R*S*S]
  del tws
  raw = raw4.reshape(A4, B4, R, S, S).permute(0, 3, 1, 4, 2).reshape(A4 * S, B4 * S, R)
  return raw[:Ho, :Wo].contiguous()


def _overlap_count(mvp, cw, R, q_hw):
  """cnt[a, b, r] = sum_ij cw[i, j, r] * mvp[a+i, b+j]  (pose_exhaustive_voting.py:93-101).

  Plain form: a 1-channel correlation on the f32 engine's scalar path.  Large maps with
  W % 32 == 0: the template columns are folded into 32 input channels (j = 32 t + c) and the
  output columns into 32 images (b = 32 bq + br), X[br][row, col, c] = mvp[row, 32 col + br + c],
  which makes it a KH x (W/32) conv with Cin = 32 -- run on the bf16 engine.  EXACT: both
  operands are 0 / 1 (exact in bf16), every product is 0 / 1 and the f32 accumulator counts at
  most H W < 2^24 of them, so the result is the same integers."""
  H, W = q_hw
  if not _stacked(R, q_hw) or W % 32:
    return ops.conv2d(mvp[None, :, :, None].contiguous(), cw)[0]
  Hp, Wp = mvp.shape
  Ho, Wo = Hp - H + 1, Wp - W + 1
  nbq = -(-Wo // 32)                       # output column groups
  ncol = nbq + W // 32 - 1                 # decimated columns a group can touch
  need = 32 * (ncol - 1) + 31 + 31 + 1     # widest index + 1
  mv = torch.nn.functional.pad(mvp, (0, max(0, need - Wp)))
  win = mv.unfold(1, 32, 1)[:, :32 * ncol]                       # win[row, s, c] = mv[row, s + c]
  # (0 / 1 are exact in bf16: the folded image is written in the engine's element type, and the launch
  #  moves both operands by LDS-DMA -- conv_bf16_xh_kernel -- instead of converting f32 in its loop)
  X = win.reshape(Hp, ncol, 32, 32).permute(2, 0, 1, 3).to(torch.bfloat16).contiguous()   # [br, row, col, c]
  wq = cw.reshape(H, W // 32, 32, R)                               # [i, t, c, r], j = 32 t + c
  out = ops.conv2d(X, wq, math='bf16')                             # [32, Ho, nbq, R]
  return out.permute(1, 2, 0, 3).reshape(Ho, nbq * 32, R)[:, :Wo].contiguous()


def _match(tw, cw, tcount, R, q_hw, m, m_valid, min_overlap, templates=None):
  H, W = q_hw
  mp, mvp = ops.pad_map(m.contiguous(), m_valid.contiguous())
  raw = _correlate(mp, tw, R, q_hw, templates)          # [Ho, Wo, R]
  cnt = None
  if min_overlap is not None:
    cnt = _overlap_count(mvp, cw, R, q_hw)
  thr = 0.0 if min_overlap is None else min_overlap * H * W
  return ops.template_finalize(raw, cnt, tcount, R, thr, use_overlap=min_overlap is not None)


def template_matching(q, q_valid, m, m_valid, do_padding=True, min_overlap=0.05, method=None):
  """pose_exhaustive_voting.py:72-104 with explicit templates q [R,H,W,D]."""
  R, H, W, D = q.shape
  if do_padding and _use_fft(method, R, (H, W), D, tuple(m.shape[:2])):
    tcount = q_valid.sum((-1, -2)).to(torch.float32)
    thr = 0.0 if min_overlap is None else min_overlap * H * W
    return ops.voting_fft(q.contiguous(), q_valid.contiguous(), m.contiguous(), m_valid.contiguous(), tcount,
                          thr, use_overlap=min_overlap is not None)
  cw = q_valid.flip(1, 2).permute(1, 2, 0).to(torch.float32)[:, :, None, :].contiguous()
  tcount = q_valid.sum((-1, -2)).to(torch.float32)
  if not do_padding:
    # :82-86: no edge padding and jax.scipy.signal.convolve(mode='full') == the same
    # cross-correlation over the map extended by ZEROS.  The reference's overlap test pads the
    # validity mask regardless of do_padding (:93-96), which gives it a [4H-3, 4W-3] count
    # against [2H-1, 2W-1] scores: only min_overlap=None is a working combination there.
    if min_overlap is not None:
      raise ValueError('template_matching(do_padding=False) needs min_overlap=None '
                       '(the reference\'s shapes disagree otherwise)')
    mp = torch.nn.functional.pad(m, (0, 0, W - 1, W - 1, H - 1, H - 1)).contiguous()
    raw = _correlate(mp, None, R, (H, W), q.contiguous())
    return ops.template_finalize(raw, None, tcount, R, 0.0, use_overlap=False)
  return _match(None, cw, tcount, R, (H, W), m, m_valid, min_overlap, templates=q.contiguous())


def exhaustive_pose_voting(plane_q, plane_map, num_rotations, grid, conf_q=None, method=None):
  """pose_exhaustive_voting.py:107-124 -> scores [R, 2H-1, 2W-1]."""
  feats_q = plane_q.features
  if conf_q is not None:
    feats_q = feats_q * conf_q[..., None]
  H, W = feats_q.shape[:2]
  fft = _use_fft(method, num_rotations, (H, W), feats_q.shape[-1], tuple(plane_map.features.shape[:2]))
  if fft:
    # sample_query_templates runs INSIDE the first transform (rows interpolated on the fly, rot90
    # quadrants as index maps): the [R, H, W, D] template tensor is never written
    if num_rotations % 4 != 0:
      raise ValueError('num_rotations must be divisible by 4')
    if H != W:
      raise ValueError('the rot90 completion requires a square BEV')
    tfm = _template_transforms(num_rotations, grid, feats_q.device)[: num_rotations // 4].contiguous()
    return ops.voting_fft_rotated(feats_q.contiguous(), plane_q.valid.contiguous(), tfm, grid.cell_size,
                                  plane_map.features.contiguous(), plane_map.valid.contiguous(),
                                  num_rotations, 0.05)
  templates, tvalid, tw, cw, tcount = sample_query_templates(
      feats_q, plane_q.valid, num_rotations, grid, _engine=True
  )
  return _match(tw, cw, tcount, num_rotations, (H, W), plane_map.features, plane_map.valid, 0.05,
                templates=templates)


def exhaustive_index_to_tfm(index, grid, num_rotations):
  """pose_exhaustive_voting.py:127-137 (keeps the reference's +0.5 cell offset)."""
  index = torch.as_tensor(index)
  dev = index.device
  extent = torch.tensor(grid.extent, device=dev)
  xy_cell = ((index[1:] - extent + 1 + 0.5) * grid.cell_size).to(torch.float32)
  angle = (index[0] * 2 * math.pi / num_rotations).to(torch.float32)
  m_t_q_center = geometry.Transform2D(-angle, xy_cell)
  c = get_grid_center_transform(grid, dev)
  return c @ m_t_q_center @ c.inv


def exhaustive_tfm_to_index(m_t_q_corner, grid, num_rotations):
  """pose_exhaustive_voting.py:140-149."""
  dev = m_t_q_corner.t.device
  c = get_grid_center_transform(grid, dev)
  m_t_q_center = c.inv @ m_t_q_corner @ c
  k = (-m_t_q_center.angle / (math.pi * 2) % 1) * num_rotations
  ij = (m_t_q_center.t / grid.cell_size) + torch.tensor(grid.extent, device=dev) - 1.5
  return torch.cat([k[..., None], ij], -1)
