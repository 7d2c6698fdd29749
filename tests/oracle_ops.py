"""The oracle expressed through the ``snap_amd.ops`` API (TEST INFRASTRUCTURE).

Every function here has the signature of its namesake in ``snap_amd/ops.py`` but
computes on the CPU with the numpy oracle (``/oracle``).  Two uses:
  * ``-m gpu`` parity tests: run ``snap_amd.ops.X`` on the GPU and ``oracle_ops.X``
    on the same inputs, compare;
  * ``-m "not gpu"`` host-logic tests: the ``oracle_backend`` fixture of
    ``conftest.py`` monkeypatches ``snap_amd.ops`` with these so the module code
    (pytrees, configs, shapes) can be exercised without a GPU.
The product never imports this file.
"""
import numpy as np
import torch

from oracle import bev as o_bev
from oracle import encoder as o_enc
from oracle import geometry as o_geo
from oracle import grids as o_grids
from oracle import lift as o_lift
from oracle import pose as o_pose
from oracle import vit as o_vit

PRO_NONE, PRO_AFFINE, PRO_GN_RELU, PRO_RELU_GN, PRO_RELU = 0, 1, 2, 3, 4
SIM_CHUNK = 64
DTYPE = np.float32  # set to np.float64 for a high-precision reference


def _np(t, dtype=None):
  if t is None:
    return None
  a = t.detach().cpu().numpy()
  if dtype is not None and a.dtype.kind == 'f':
    a = a.astype(dtype)
  return a


def _t(a, like=None, dtype=None):
  out = torch.as_tensor(np.ascontiguousarray(a))
  if dtype is not None:
    out = out.to(dtype)
  elif out.dtype == torch.float64:
    out = out.to(torch.float32)
  if like is not None:
    out = out.to(like.device)
  return out


# -- encoder -------------------------------------------------------------------
def _prologue(x, prologue, gn, in_affine, cin):
  x = x[..., :cin]
  if prologue == PRO_AFFINE:
    return x * x.dtype.type(in_affine[0]) + x.dtype.type(in_affine[1])
  if prologue == PRO_RELU:
    return np.maximum(x, 0)
  if prologue in (PRO_GN_RELU, PRO_RELU_GN):
    mu, sc, beta = (_np(g, x.dtype) for g in gn)
    mu = mu.reshape(x.shape[0], 1, 1, -1)
    sc = sc.reshape(x.shape[0], 1, 1, -1)
    beta = beta.reshape(1, 1, 1, -1)
    if prologue == PRO_GN_RELU:
      return np.maximum((x - mu) * sc + beta, 0)
    return (np.maximum(x, 0) - mu) * sc + beta
  return x


def conv2d(x, w, *, stride=1, padding=((0, 0), (0, 0)), cin=None, prologue=PRO_NONE,
           gn=None, in_affine=(1.0, 0.0), bias=None, relu=False, residual=None,
           up_prev=None, row_mask=None, emit_gn_stats=None, math=None, gelu=False):
  # emit_gn_stats: a speed hint of the HIP path (statistics out of the epilogue); no-op here.
  # math='bf16': both operands rounded to bf16 after the f32 prologue, products summed in
  # float64 (the engine accumulates in f32: the test tolerance covers the summation order).
  xn, wn = _np(x, DTYPE), _np(w, DTYPE)
  cin = wn.shape[2] if cin is None else cin
  cs = xn.shape[-1]
  xn = _prologue(xn, prologue, gn, in_affine, cin)
  if math in ('bf16', 'fp16') and cs % 4 == 0 and cin >= 4:   # (other shapes run on the f32 engine)
    rnd_ = o_enc.fp16_round if math == 'fp16' else o_enc.bf16_round
    xn = rnd_(xn.astype(np.float32)).astype(np.float64)
    wn = rnd_(wn.astype(np.float32)).astype(np.float64)
    y = o_enc.conv2d(xn, wn, (stride, stride), padding).astype(DTYPE)
    xn = None
  if xn is not None:
    y = o_enc.conv2d(xn, wn, (stride, stride), padding)
  if bias is not None:
    y = y + _np(bias, DTYPE)
  if residual is not None:
    y = y + _np(residual, DTYPE)
  if up_prev is not None:
    y = y + o_enc.resize_bilinear_x2(_np(up_prev, DTYPE))
  if relu:
    y = np.maximum(y, 0)
  if gelu:
    y = o_vit.gelu_tanh(y)
  if row_mask is not None:
    y = np.where(_np(row_mask).reshape(*y.shape[:-1], 1), y, 0)
  return _t(y, x)


def dense(x, kernel, bias=None, *, cin=None, prologue=PRO_NONE, relu=False, row_mask=None,
          math=None, gelu=False, residual=None):
  lead = x.shape[:-1]
  M = int(np.prod(lead)) if len(lead) else 1
  y = conv2d(
      x.reshape(1, 1, M, x.shape[-1]), kernel.reshape(1, 1, *kernel.shape),
      cin=cin if cin is not None else kernel.shape[0], prologue=prologue, bias=bias,
      relu=relu, row_mask=row_mask, math=math, gelu=gelu,
      residual=None if residual is None else residual.reshape(1, 1, M, kernel.shape[1]),
  )
  return y.reshape(*lead, kernel.shape[1])


def semantic_embed(rasters, idx_road, idx_other, table_road, table_other):
  r = _np(rasters).astype(bool)
  tr, to = _np(table_road, DTYPE), _np(table_other, DTYPE)
  label = np.argmax(r[..., list(idx_road)], axis=-1)
  f_road = tr[label]
  lab_o = np.arange(len(idx_other)) + r[..., list(idx_other)].astype(int)
  f_other = to[lab_o].reshape(*r.shape[:-1], -1)
  return _t(np.concatenate([f_road, f_other], -1), table_road)


def layer_norm(x, gamma, beta, eps=1e-6):
  return _t(o_vit.layer_norm(_np(x, DTYPE), _np(gamma, DTYPE), _np(beta, DTYPE), eps), x)


def attention(qkv, scale=None, bf16_operands=True):
  return _t(o_vit.attention(_np(qkv, np.float64), scale, bf16_operands).astype(DTYPE), qkv)


def weight_standardize(w, eps=1e-10):
  return _t(o_enc.standardize(_np(w, DTYPE), (0, 1, 2), eps), w)


def weight_standardize_multi(ws, eps=1e-10):
  return [weight_standardize(w, eps) for w in ws]


def group_norm_stats(x, gamma, *, groups=32, eps=1e-5, relu_first=False):
  xn = _np(x, DTYPE)
  if relu_first:
    xn = np.maximum(xn, 0)
  N, H, W, C = xn.shape
  g = xn.reshape(N, H * W, groups, C // groups)
  mean = g.mean(axis=(1, 3))
  var = np.square(g - mean[:, None, :, None]).mean(axis=(1, 3))
  rstd = 1.0 / np.sqrt(var + DTYPE(eps))
  cpg = C // groups
  mu = np.repeat(mean, cpg, axis=1)
  sc = np.repeat(rstd, cpg, axis=1) * _np(gamma, DTYPE).reshape(1, C)
  return _t(mu, x), _t(sc, x)


def group_norm_apply(x, mu, sc, beta, mode):
  xn = _np(x, DTYPE)
  return _t(_prologue(xn, mode, (mu, sc, beta), None, xn.shape[-1]), x)


def max_pool_3x3s2(x):
  return _t(o_enc.max_pool(_np(x, DTYPE)), x)


# -- lift ------------------------------------------------------------------------
def pooled_stride(feature_dim, weighted=True, use_variance=True, add_minmax=False):
  ch = feature_dim * (1 + int(use_variance) + 2 * int(add_minmax)) + int(weighted)
  return (ch + 3) // 4 * 4


def unpack_cameras(cam, fisheye):
  c = _np(cam, DTYPE)
  if fisheye:
    return o_geo.FisheyeCamera(c[..., 0:2], c[..., 2:4], c[..., 4:6], c[..., 6:9], c[..., 9])
  return o_geo.Camera(c[..., 0:2], c[..., 2:4], c[..., 4:6])


def unpack_transforms(Rt):
  r = _np(Rt, DTYPE)
  return o_geo.Transform3D(r[..., :9].reshape(*r.shape[:-1], 3, 3), r[..., 9:12])


def _split_rows(rows):
  """[..., C] f32 -> the split-bf16 hand-over layout of SnapLiftDesc.out_split in an f32 container:
  [..., ceil(C / 16) slabs][hi | lo][16] bf16 (hi = bf16(v) RNE, lo = bf16(v - hi))."""
  import torch
  t = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.float32))
  C = t.shape[-1]
  ks = (C + 15) // 16
  pad = torch.zeros(*t.shape[:-1], ks * 16)
  pad[..., :C] = t
  hi = pad.bfloat16()
  lo = (pad - hi.float()).bfloat16()
  both = torch.stack([hi.reshape(*t.shape[:-1], ks, 16), lo.reshape(*t.shape[:-1], ks, 16)], dim=-2)
  return both.contiguous().view(torch.int16).reshape(*t.shape[:-1], ks * 32).view(torch.float32).numpy()


def _unsplit_rows(x):
  """inverse of _split_rows (hi + lo; the residual below the second part is dropped)."""
  import torch
  t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))
  ks = t.shape[-1] // 16
  parts = t.view(torch.int16).view(torch.bfloat16).reshape(*t.shape[:-1], ks, 2, 16).float()
  return (parts[..., 0, :] + parts[..., 1, :]).reshape(*t.shape[:-1], ks * 16).numpy()


def _tap_records(f, p2d, vis, idx, scores, selective):
  """The tap records of lift.hip (SnapLiftDesc tap records) for the voxels with ONE visible
  observation: byte offset of tap (i0, j0) | bit 8: i1 != i0, bit 9: j1 != j0 | wi1 | wj1 | score.
  Taps as streetview_encoder.py:93-105 (selective: the point clipped) / grids.interpolate_nd (all
  views: every tap index clipped); the weights come from the unclipped-index coordinate."""
  B, V, h, w, C = f.shape
  recs = np.zeros(vis.shape[:2] + (8,), np.int32)
  bi, ni = np.nonzero(vis.sum(-1) == 1)
  ss = vis[bi, ni].argmax(-1)
  view = idx[bi, ni, ss] if idx is not None else ss
  c = p2d[bi, ni, ss].astype(np.float32) - np.float32(0.5)            # [n, 2] (i, j)
  size = np.array([h, w], np.float32)
  if selective:
    c = np.maximum(np.minimum(c, size - 1), 0)
  fl = np.floor(c)
  w1 = (c - fl).astype(np.float32)
  lo = np.clip(fl, 0, size - 1).astype(np.int64)
  hi = np.clip(fl + 1, 0, size - 1).astype(np.int64)
  off = ((((bi * V + view) * h + lo[:, 0]) * w + lo[:, 1]) * (C * 4)).astype(np.uint32)
  flags = ((hi[:, 0] != lo[:, 0]).astype(np.uint32) << 8) | ((hi[:, 1] != lo[:, 1]).astype(np.uint32) << 9)
  recs[bi, ni, 0] = off.view(np.int32)
  recs[bi, ni, 1] = flags.view(np.int32)
  recs[bi, ni, 2] = w1[:, 0].view(np.int32)
  recs[bi, ni, 3] = w1[:, 1].view(np.int32)
  recs[bi, ni, 4] = scores[bi, ni, ss].astype(np.float32).view(np.int32)
  return recs


def _rows_from_tap_records(f_images, recs, fd, sel):
  """What mlp2_pool_kernel<.., GATHER> stages for a record: mean = the four-tap blend (weight order
  of lift.hip phase B), variance = 0, score from the record.  [M, 2 fd + 1]; rows outside ``sel``
  (no / several observations: their records are uninitialised) are zero."""
  f = _np(f_images, np.float32)
  B, V, h, w, C = f.shape
  sel = np.asarray(sel).reshape(-1)
  r = np.ascontiguousarray(_np(recs)).reshape(-1, 8)[sel]
  px = np.ascontiguousarray(r[:, 0]).view(np.uint32).astype(np.int64) // (C * 4)
  di = (r[:, 1] >> 8) & 1
  dj = (r[:, 1] >> 9) & 1
  wi1, wj1, sc = (np.ascontiguousarray(r[:, k]).view(np.float32) for k in (2, 3, 4))
  wi0, wj0 = 1 - wi1, 1 - wj1
  ff = f.reshape(-1, C)[:, :fd]
  a00, a01, a10, a11 = ff[px], ff[px + dj], ff[px + di * w], ff[px + di * w + dj]
  mean = (((wi0 * wj0)[:, None] * a00 + (wi0 * wj1)[:, None] * a01) + (wi1 * wj0)[:, None] * a10) + (wi1 * wj1)[:, None] * a11
  out = np.zeros((sel.size, 2 * fd + 1), np.float32)
  out[sel] = np.concatenate([mean, np.zeros_like(mean), sc[:, None]], -1)
  return out


def lift_pool(f_images, cam, Rt, points, *, K, fisheye, feature_dim, num_bins,
              depth_min_max, max_view_distance=None, weighted=True, use_variance=True,
              add_minmax=False, grid_yz=None, valid_rows_only=False, out_split=False,
              class_rows=False, tap_records=False):
  f = _np(f_images, DTYPE)
  cams = unpack_cameras(cam, fisheye)
  T = unpack_transforms(Rt)
  pts = _np(points, DTYPE)
  p2d, vis, depth, _ = o_lift.project_points_to_views(T, cams, pts)
  min_distance = idx = None
  if K > 0:
    idx, min_distance = o_lift.view_selection(pts, T, vis, K)
    p2d, vis, depth = (
        o_lift.gather_batched_observations(a, idx) for a in (p2d, vis, depth)
    )
    f_proj = o_lift.interpolate_views_selective(f, p2d, idx)
  else:
    f_proj = o_lift.interpolate_views_all(f, p2d)
  feats, scales = f_proj[..., :feature_dim], f_proj[..., feature_dim:]
  scores = o_lift.interpolate_depth_score(scales, depth, depth_min_max) if weighted else None
  pooled, valid = o_lift.pool_multiview_features(feats, vis, scores, add_minmax, use_variance)
  if max_view_distance is not None and min_distance is not None:
    valid = valid & (min_distance <= max_view_distance)
  nvis = vis.sum(-1)
  if out_split:
    # the hand-over format of the fused MLP / pool kernel; what the kernel does not write is NaN here
    out = _split_rows(pooled).copy()
    if class_rows:
      nv = feature_dim // 16
      out[nvis == 1, 16 * nv:32 * nv] = np.nan       # single observation: the variance slabs
  else:
    stride = (pooled.shape[-1] + 3) // 4 * 4
    out = np.zeros(pooled.shape[:-1] + (stride,), pooled.dtype)
    out[..., : pooled.shape[-1]] = pooled
  if valid_rows_only:
    out[nvis == 0] = np.nan
  if class_rows:
    classes = np.where(valid, np.where(nvis > 1, 2, 1), 0).astype(np.uint8)
    if tap_records:                # one observation: no row at all, a tap record instead
      assert out_split and valid_rows_only and weighted
      out[nvis == 1] = np.nan
      recs = _tap_records(f, p2d, vis, idx, scores, K > 0)
      return (_t(out, f_images), _t(valid, f_images), _t(classes, f_images, dtype=torch.uint8),
              _t(recs, f_images, dtype=torch.int32))
    return _t(out, f_images), _t(valid, f_images), _t(classes, f_images, dtype=torch.uint8)
  return _t(out, f_images), _t(valid, f_images)


def project_points(cam, Rt, points, fisheye):
  p2d, vis, depth, _ = o_lift.project_points_to_views(
      unpack_transforms(Rt), unpack_cameras(cam, fisheye), _np(points, DTYPE)
  )
  return _t(p2d, cam), _t(vis, cam), _t(depth, cam)


# -- BEV ---------------------------------------------------------------------------
def vertical_pool(vol, valid, pooling='max'):
  out = o_bev.vertical_pooling({'pooling': pooling}, _np(vol, DTYPE), _np(valid).astype(bool))
  return _t(out['features'], vol), _t(out['valid'], vol)


def mlp2_pool_supported(cin, hidden, out_dim):
  return True


def mlp2_pool_max(x, row_mask, w0, b0, w1, b1, *, cin, Z, relu_in=False, x_split=False,
                  zero_slabs=None, gather=None):
  xx = _np(x, DTYPE)
  if x_split:
    xx = _unsplit_rows(xx)
  xx = xx[:, :cin].copy()
  m = _np(row_mask)
  if gather is not None:          # class-1 rows do not exist: blended from their tap records
    one = m.reshape(-1) == 1
    xx[one] = _rows_from_tap_records(gather[0], gather[1], (cin - 1) // 2, one)[one].astype(xx.dtype)
  if zero_slabs is not None:      # row classes: class 1 is exactly zero over the slab range
    lo, n = zero_slabs
    xx[m == 1, 16 * lo:16 * (lo + n)] = 0
  m = m.astype(bool)
  xx[~m] = 0                       # (rows the lift did not write)
  if relu_in:
    xx = np.maximum(xx, 0)
  hid = np.maximum(xx @ _np(w0, DTYPE) + _np(b0, DTYPE), 0)
  vol = hid @ _np(w1, DTYPE) + _np(b1, DTYPE)
  vol = np.where(m[:, None], vol, 0).reshape(-1, Z, vol.shape[-1])
  out = o_bev.vertical_pooling({'pooling': 'max'}, vol, m.reshape(-1, Z))
  return _t(out['features'], x), _t(out['valid'], x)


def plane_fuse_match(planes, valids, pooling='max', Wm=None, bm=None, normalize=True,
                     eps=1e-5, want_fused=True):
  feats = np.stack([_np(p, DTYPE) for p in planes], axis=-2)
  vals = np.stack(
      [np.ones(planes[0].shape[:-1], bool) if v is None else _np(v).astype(bool)
       for v in valids], axis=-1,
  )
  out = o_bev.vertical_pooling({'pooling': pooling}, feats, vals)
  matching = None
  if Wm is not None:
    f = out['features'] @ _np(Wm, DTYPE) + _np(bm, DTYPE)
    if normalize:
      f = o_enc.normalize(f, eps=eps)
    matching = _t(np.where(out['valid'][..., None], f, 0), planes[0])
  fused = _t(out['features'], planes[0]) if want_fused else None
  return fused, _t(out['valid'], planes[0]), matching


# -- pose ------------------------------------------------------------------------------
def _raw_sim(fq, fm, scale, clip):
  x = np.einsum('bnd,bijd->bnij', fq, fm)
  if clip:
    x = np.maximum(x, 0)
  return x * fq.dtype.type(scale)


def sim_softmax(fq, fm, scale, clip_negative, num_valid, want_prob=False,
                want_rowstats=False, row_weight=None, math=None):
  q, m = _np(fq, DTYPE), _np(fm, DTYPE)
  nv = _np(num_valid, DTYPE)[:, None, None, None]
  if row_weight is not None:      # confidence weights replace 1 / num_valid (bev_localizer.py:165-172)
    nv = 1.0 / np.maximum(_np(row_weight, DTYPE), 1e-38)[:, :, None, None]
    nv = np.where(_np(row_weight, DTYPE)[:, :, None, None] > 0, nv, np.inf)
  x = _raw_sim(q, m, scale, clip_negative)
  B, Nq, X, Y = x.shape
  XY = X * Y
  NC = (XY + SIM_CHUNK - 1) // SIM_CHUNK
  flat = np.full((B, Nq, NC * SIM_CHUNK), -np.inf, x.dtype)
  flat[..., :XY] = x.reshape(B, Nq, XY)
  ch = flat.reshape(B, Nq, NC, SIM_CHUNK)
  cm = ch.max(-1)
  cs = np.exp(ch - cm[..., None]).sum(-1)
  stats = np.stack([cm, cs], -1)
  M = x.max(axis=(-1, -2), keepdims=True)
  e = np.exp(x - M)
  T = e.sum(axis=(-1, -2), keepdims=True)
  prob = _t(e / T / nv, fq) if want_prob else None
  rowstats = (
      _t(np.concatenate([M.reshape(B, Nq, 1), T.reshape(B, Nq, 1)], -1), fq)
      if (want_prob or want_rowstats) else None
  )
  return _t(x / nv, fq), _t(stats, fq), prob, rowstats


def ransac_sample(fq, fm, chunk_stats, scale, clip_negative, S, seed=0, uniforms=None, row_cdf=None,
                  sim=None, row_unscale=None):
  """Two-level inverse-CDF sampler in float64 (same scheme as pose.hip)."""
  q, m = _np(fq, np.float64), _np(fm, np.float64)
  B, Nq, _ = q.shape
  X, Y = m.shape[1:3]
  if uniforms is None:
    u = np.random.default_rng(seed).random((B, S, 2))
  else:
    u = _np(uniforms, np.float64)
  corr = np.zeros((B, S, 3), np.int32)
  for b in range(B):
    n = np.minimum((u[b, :, 0].astype(np.float32) * np.float32(Nq)).astype(np.int64), Nq - 1)
    if row_cdf is not None:
      c = _np(row_cdf, np.float64)[b]
      n = np.minimum(np.searchsorted(c, u[b, :, 0] * c[-1], side='right'), Nq - 1)
    for s in range(S):
      x = np.einsum('d,ijd->ij', q[b, n[s]], m[b])
      if clip_negative:
        x = np.maximum(x, 0)
      x = (x * scale).reshape(-1)
      e = np.exp(x - x.max())
      cdf = np.cumsum(e)
      target = u[b, s, 1] * cdf[-1]
      cell = min(int(np.searchsorted(cdf, target, side='right')), X * Y - 1)
      corr[b, s] = (n[s], cell // Y, cell % Y)
  return _t(corr, fq, torch.int32)


def poses_from_corr(corr, q_xy, P, retries, cell_size):
  c = _np(corr).astype(np.int64)
  xy = _np(q_xy, DTYPE)
  B = c.shape[0]
  Hm = int(c[..., 1].max()) + 1
  Wm = int(c[..., 2].max()) + 1
  grid = o_grids.Grid2D((Hm, Wm), cell_size)
  out = np.zeros((B, P, 3), DTYPE)
  for b in range(B):
    tf = o_pose.poses_from_correspondences(c[b], xy[b], P, retries, grid)
    out[b, :, 0] = tf.angle
    out[b, :, 1:] = tf.t
  return _t(out, q_xy)


def pose_score(sim, poses, q_xy, valid_q, map_valid, cell_size, mask_oob=False):
  s, p, xy = _np(sim, DTYPE), _np(poses, DTYPE), _np(q_xy, DTYPE)
  vq = _np(valid_q).astype(bool)
  B, Nq, X, Y = s.shape
  mv = np.ones((B, X, Y), bool) if map_valid is None else _np(map_valid).astype(bool)
  grid = o_grids.Grid2D((X, Y), cell_size)
  out = np.stack([
      o_pose.pose_scoring_many(
          o_geo.Transform2D(p[b, :, 0], p[b, :, 1:]), s[b], xy[b], vq[b], mv[b], grid,
          mask_oob,
      )
      for b in range(B)
  ])
  return _t(out, sim)


def pose_score_window_supported(X, Y, radius_cells):
  return True


def pose_score_window(sim, poses, centers, radius_cells, q_xy, valid_q, cell_size):
  """The windowed scoring is ``pose_score`` on a promise: every pose of a scene maps every query point to
  within ``radius_cells`` cells of where the scene's centre pose maps it.  The twin CHECKS the promise
  (the host's radius bound, pose_estimation.grid_refinement_batched) before it scores."""
  p, c, xy = _np(poses, np.float64), _np(centers, np.float64), _np(q_xy, np.float64)

  def image(th, t):          # [B, P] angles, [B, P, 2] translations -> [B, P, Nq, 2] in cells
    co, si = np.cos(th)[..., None], np.sin(th)[..., None]
    x = co * xy[:, None, :, 0] - si * xy[:, None, :, 1] + t[..., 0:1]
    y = si * xy[:, None, :, 0] + co * xy[:, None, :, 1] + t[..., 1:2]
    return np.stack([x, y], -1) / cell_size
  far = np.abs(image(p[..., 0], p[..., 1:]) - image(c[:, None, 0], c[:, None, 1:])).max()
  assert far <= radius_cells, (far, radius_cells)
  return pose_score(sim, poses, q_xy, valid_q, None, cell_size)


def refine_lattice(init, offs_r, offs_p):
  i, r, p = _np(init, DTYPE), _np(offs_r, DTYPE), _np(offs_p, DTYPE)
  rr, xx, yy = np.meshgrid(r, p, p, indexing='ij')
  offs = o_geo.Transform2D(rr.reshape(-1), np.stack([xx.reshape(-1), yy.reshape(-1)], -1))
  out = []
  for b in range(i.shape[0]):
    n = offs.angle.shape[0]
    base = o_geo.Transform2D(np.full(n, i[b, 0], DTYPE), np.tile(i[b, 1:], (n, 1)))
    tf = base @ offs
    out.append(np.concatenate([tf.angle[:, None], tf.t], -1))
  return _t(np.stack(out).astype(DTYPE), init)


def argmax_rows(scores, start=0):
  s = _np(scores)
  return _t(np.argmax(s[:, start:], axis=-1).astype(np.int32), scores, torch.int32)


# -- exhaustive voting ---------------------------------------------------------------
def rotate_templates(feat, valid, tfm, num_rotations, cell_size):
  f, v, tf = _np(feat, DTYPE), _np(valid).astype(bool), _np(tfm, DTYPE)
  H, W, D = f.shape
  grid = o_grids.Grid2D((H, W), cell_size)
  grid_xy = grid.index_to_xyz(grid.grid_index(), DTYPE).reshape(-1, 2)
  quarter, qv = [], []
  for r in range(num_rotations // 4):
    c, s, tx, ty = tf[r]
    xy = np.stack(
        [(c * grid_xy[:, 0] - s * grid_xy[:, 1]) + tx,
         (s * grid_xy[:, 0] + c * grid_xy[:, 1]) + ty], -1,
    )
    uv = (xy / DTYPE(cell_size)).astype(DTYPE)
    val, ok = o_grids.interpolate_nd(f, uv, v)
    quarter.append(np.where(ok[..., None], val, 0).reshape(H, W, D))
    qv.append(ok.reshape(H, W))
  quarter, qv = np.stack(quarter), np.stack(qv)
  templates = np.concatenate([np.rot90(quarter, k, axes=(2, 1)) for k in range(4)], 0)
  tvalid = np.concatenate([np.rot90(qv, k, axes=(2, 1)) for k in range(4)], 0)
  tw = np.transpose(templates, (1, 2, 3, 0))
  cw = np.transpose(tvalid[:, ::-1, ::-1], (1, 2, 0))[:, :, None, :].astype(DTYPE)
  tcount = tvalid.sum((-1, -2)).astype(DTYPE)
  return (_t(templates, feat), _t(tvalid, feat), _t(tw, feat), _t(cw, feat), _t(tcount, feat))


def pad_map(m, mvalid):
  mn, mv = _np(m, DTYPE), _np(mvalid).astype(bool)
  H, W = mn.shape[:2]
  mp = np.pad(mn, ((H - 1,) * 2, (W - 1,) * 2, (0, 0)), mode='edge')
  mvp = np.pad(mv.astype(DTYPE), ((H - 1,) * 2, (W - 1,) * 2))
  return _t(mp, m), _t(mvp, m)


def template_finalize(raw, cnt, tcount, R, threshold, use_overlap=True):
  r = np.transpose(_np(raw, DTYPE)[..., :R], (2, 0, 1))
  if use_overlap:
    c = np.transpose(_np(cnt, DTYPE)[..., :R], (2, 0, 1))
    r = np.where(c > threshold, r, -np.inf)
  with np.errstate(divide='ignore', invalid='ignore'):
    r = r / _np(tcount, DTYPE)[:, None, None]
  return _t(r.astype(DTYPE), raw)


ALL_OPS = [
    'conv2d', 'dense', 'weight_standardize', 'weight_standardize_multi', 'group_norm_stats', 'group_norm_apply',
    'max_pool_3x3s2', 'pooled_stride', 'lift_pool', 'project_points', 'vertical_pool', 'mlp2_pool_max',
    'plane_fuse_match', 'sim_softmax', 'ransac_sample', 'poses_from_corr', 'pose_score', 'pose_score_window',
    'pose_score_window_supported',
    'refine_lattice', 'argmax_rows', 'rotate_templates', 'pad_map', 'template_finalize',
]
