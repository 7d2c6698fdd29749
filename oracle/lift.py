"""Oracle (test infrastructure): camera-ray lift of image features to a voxel grid.

Restates ``snap/models/streetview_encoder.py`` (the per-voxel projective
gather, view selection, depth-score interpolation and multi-view pooling).
"""
import itertools

import numpy as np

from oracle import encoder
from oracle import grids


def project_points_to_views(scene_t_view, camera, points):
  """streetview_encoder.py:42-65 (batched over B and V).

  Args:
    scene_t_view: Transform3D with batch shape [B, V].
    camera: (Fisheye)Camera with batch shape [B, V].
    points: [B, N, 3] in the scene frame.
  Returns:
    p2d [B, N, V, 2] in (i, j) = (row, col) order, vis [B, N, V],
    depth [B, N, V], rays [B, N, V, 3].
  """
  dtype = points.dtype
  points_view = scene_t_view.inv @ points[:, None]  # [B, V, N, 3]
  points_view = points_view.astype(dtype)
  depth = points_view[..., -1]
  distance = np.linalg.norm(points_view, axis=-1, keepdims=True)
  rays = points_view / np.clip(distance, 1e-5, None)
  p2d, vis = camera.world2image(points_view)
  p2d = p2d[..., ::-1]  # xy -> ij
  sw = lambda a: np.swapaxes(a, 1, 2)
  return sw(p2d).astype(dtype), sw(vis), sw(depth), sw(rays).astype(dtype)


def view_selection(points, scene_t_view, vis, num):
  """streetview_encoder.py:127-138.

  ``jax.lax.top_k(-dist, k)``: k smallest distances, ties (including several
  +inf) resolved towards the lowest view index.
  Returns indices [B, N, K] (int), min_dist [B, N].
  """
  diff = points[..., None, :] - scene_t_view.t[..., None, :, :]  # B,N,V,3
  dist = np.linalg.norm(diff, axis=-1)
  dist = np.where(vis, dist, np.inf).astype(points.dtype)
  min_dist = dist.min(axis=-1)
  indices = np.argsort(dist, axis=-1, kind='stable')[..., :num]
  return indices, min_dist


def gather_batched_observations(x, indices):
  """streetview_encoder.py:66: x [B, N, V, ...] indexed per (b, n) by [K]."""
  idx = indices.reshape(indices.shape + (1,) * (x.ndim - 3))
  return np.take_along_axis(x, idx, axis=2)


def interpolate_views_all(f_images, p2d):
  """streetview_encoder.py:69-76.  f_images [B,V,h,w,D], p2d [B,N,V,2] -> [B,N,V,D]."""
  B, V = f_images.shape[:2]
  out = np.empty(p2d.shape[:3] + (f_images.shape[-1],), f_images.dtype)
  for b in range(B):
    for v in range(V):
      out[b, :, v], _ = grids.interpolate_nd(f_images[b, v], p2d[b, :, v])
  return out


def interpolate_views_selective(f_images, p2d, index):
  """streetview_encoder.py:80-105.

  f_images [B,V,h,w,D]; p2d [B,N,K,2] (ij); index [B,N,K] -> [B,N,K,D].
  The point is clipped to [0, size-1] (after the half-pixel shift) and the
  upper tap ``lower+1`` may equal ``size``: its weight is then exactly zero and
  XLA clamps the gather index, so it is a no-op.
  """
  dtype = f_images.dtype
  B, V, h, w, D = f_images.shape
  size = np.array([h, w], dtype)
  point = np.maximum(np.minimum(p2d.astype(dtype) - 0.5, size - 1), 0)
  lower = np.floor(point).astype(np.int64)
  upper = lower + 1
  w_upper = (point - lower).astype(dtype)
  w_lower = 1 - w_upper
  weights = [w_lower, w_upper]
  coords = [lower, upper]
  bidx = np.arange(B).reshape(B, 1, 1)
  out = None
  for i, j in itertools.product(range(2), repeat=2):
    wgt = weights[i][..., 0] * weights[j][..., 1]
    ci = np.clip(coords[i][..., 0], 0, h - 1)
    cj = np.clip(coords[j][..., 1], 0, w - 1)
    contrib = wgt[..., None] * f_images[bidx, index, ci, cj]
    out = contrib if out is None else out + contrib
  return out


def interpolate_depth_score(score_scales, depth, depth_min_max):
  """streetview_encoder.py:109-124.  score_scales [...,S], depth [...] -> [...]."""
  dtype = score_scales.dtype
  num_bins = score_scales.shape[-1]
  min_, max_ = depth_min_max
  depth = np.clip(depth, dtype.type(min_), dtype.type(max_))
  t = np.log(depth / dtype.type(min_)) / np.log(dtype.type(max_ / min_))
  index = 0.5 + t * (num_bins - 1)
  # interpolate_nd on a 1-D array: coordinate (index - 0.5), taps clipped.
  c = (index - 0.5).astype(dtype)
  lo = np.floor(c)
  w_hi = (c - lo).astype(dtype)
  w_lo = 1 - w_hi
  i_lo = np.clip(lo.astype(np.int64), 0, num_bins - 1)
  i_hi = np.clip(lo.astype(np.int64) + 1, 0, num_bins - 1)
  s_lo = np.take_along_axis(score_scales, i_lo[..., None], -1)[..., 0]
  s_hi = np.take_along_axis(score_scales, i_hi[..., None], -1)[..., 0]
  return w_lo * s_lo + w_hi * s_hi


def pool_multiview_features(
    feats, valid, scores=None, add_minmax=True, use_variance=True
):
  """streetview_encoder.py:141-178.  feats [...,V,D], valid [...,V], scores [...,V]."""
  dtype = feats.dtype
  valid_any = valid.any(-1)
  valid_ = np.where(valid_any[..., None], valid, True)[..., None]  # [...,V,1]
  if scores is None:
    cnt = valid_.sum(-2)
    mean_ = (feats * valid_).sum(-2) / cnt
    var_ = (((feats - mean_[..., None, :]) ** 2) * valid_).sum(-2) / cnt
  else:
    ctype = np.float64 if dtype == np.float64 else np.float32
    s = scores.astype(ctype)[..., None]
    s_masked = np.where(valid_, s, -np.inf)
    # jax.nn.softmax(..., where=valid_, initial=0): the shift is
    # max(initial, max over valid entries) = max(0, .)  (streetview_encoder.py:157-159).
    m = np.maximum(s_masked.max(-2, keepdims=True), 0)
    e = np.where(valid_, np.exp(s_masked - m), 0)
    weights = e / e.sum(-2, keepdims=True)
    weights = np.where(valid_, weights, 0)
    mean_ = np.sum(weights * feats, axis=-2)
    var_ = np.sum(weights * (feats - mean_[..., None, :]) ** 2, axis=-2)
    mean_ = mean_.astype(dtype)
    var_ = var_.astype(dtype)
  stats = [mean_]
  if use_variance:
    stats.append(var_)
  if add_minmax:
    stats.append(np.where(valid_, feats, -np.inf).max(-2))
    stats.append(np.where(valid_, feats, np.inf).min(-2))
  if scores is not None:
    stats.append(np.where(valid_, scores[..., None], -np.inf).max(-2))
  stats = np.where(
      valid_any[..., None], np.concatenate(stats, -1), 0
  ).astype(dtype)
  return stats, valid_any


def streetview_encoder(params, config, data):
  """streetview_encoder.py:217-287 with do_weighted_fusion (the default path).

  data: images [B,V,H,W,3], camera (batch [B,V]), T_view2scene (batch [B,V]),
  xyz_query [B, ..., 3].
  """
  images = data['images']
  B, V = images.shape[:2]
  if data.get('image_feature_pyr') is None:
    pyr_b = [
        encoder.image_encoder(
            params['image_encoder'], config['image_encoder'], images[b]
        )
        for b in range(B)
    ]
    f_image_pyr = dict(
        features=[
            np.stack([p['features'][l] for p in pyr_b])
            for l in range(len(pyr_b[0]['features']))
        ],
        strides=[
            np.stack([p['strides'][l] for p in pyr_b])
            for l in range(len(pyr_b[0]['strides']))
        ],
    )
  else:
    f_image_pyr = data['image_feature_pyr']
  f_images = f_image_pyr['features'][-1]
  dtype = f_images.dtype
  feature_stride = f_image_pyr['strides'][-1][0]
  cameras = data['camera'].scale((1 / feature_stride[::-1]).astype(dtype))
  scene_t_view = data['T_view2scene']
  pred = {'image_feature_pyramid': f_image_pyr}

  weighted = bool(config['do_weighted_fusion'])
  if weighted:
    proj_config = dict(
        layers=(config['feature_dim'] + config['num_scale_bins'],),
        apply_input_activation=config['proj_mlp']['apply_input_activation'],
    )
    f_images = encoder.mlp(params['proj_mlp'], proj_config, f_images)
    pred['scores_images'] = f_images[..., -config['num_scale_bins']:]

  xyz = data['xyz_query']
  xyz_flat = xyz.reshape(len(xyz), -1, 3)
  p2d_views, visible, depth, rays = project_points_to_views(
      scene_t_view, cameras, xyz_flat
  )
  k_vs = config['top_k_view_selection']
  if k_vs and f_images.shape[1] > k_vs:
    view_indices, min_distance = view_selection(
        xyz_flat, scene_t_view, visible, k_vs
    )
    p2d_views, visible, depth, rays = (
        gather_batched_observations(x, view_indices)
        for x in (p2d_views, visible, depth, rays)
    )
    f_proj = interpolate_views_selective(f_images, p2d_views, view_indices)
  else:
    f_proj = interpolate_views_all(f_images, p2d_views)
    min_distance = None

  fd = config['feature_dim']
  if weighted:
    f_proj, scores_scales = f_proj[..., :fd], f_proj[..., fd:]
    scores_proj = interpolate_depth_score(
        scores_scales, depth, config['depth_min_max']
    )
  else:
    scores_proj = None      # streetview_encoder.py:261-262
    if config.get('depth_mlp') is not None:
      # streetview_encoder.py:263-267: a per-observation MLP on [features, log10 depth, ray]
      log_depth = np.log10(np.clip(depth, 0.1, 100)).astype(dtype)
      rays_v = np.where(visible[..., None], rays, 0).astype(dtype)
      f_proj_depth = np.concatenate([f_proj, log_depth[..., None], rays_v], -1)
      f_proj = f_proj + encoder.mlp(params['depth_mlp'], config['depth_mlp'], f_proj_depth)
  f_pooled, valid = pool_multiview_features(
      f_proj,
      visible,
      scores_proj,
      config['fusion_add_minmax'],
      config['fusion_use_variance'],
  )
  if config.get('max_view_distance') is not None and min_distance is not None:
    valid = valid & (min_distance <= config['max_view_distance'])
  pred['_pooled'] = f_pooled  # oracle-only extra: the k1-k5 output
  f_grid = encoder.mlp(params['fusion_mlp'], config['fusion'], f_pooled)
  f_grid = np.where(valid[..., None], f_grid, 0).astype(dtype)
  grid_shape = (-1, *xyz.shape[-4:-1])
  f_grid = f_grid.reshape(*grid_shape, f_grid.shape[-1])
  valid = valid.reshape(grid_shape)
  pred['feature_volume'] = dict(features=f_grid, valid=valid)
  return pred
