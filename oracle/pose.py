"""Oracle (test infrastructure): query-vs-map similarity, pose sampling and scoring.

Restates ``snap/models/bev_localizer.py`` and ``snap/models/pose_estimation.py``.
Random sampling cannot reproduce JAX's threefry stream: every function that
draws samples in the reference takes the drawn uniforms / indices explicitly.
"""
import numpy as np

from oracle import geometry
from oracle import grids


def build_query_frustum_grid(
    cell_size, depth, filter_points_in_fov=False, hfov_deg=None,
    dtype=np.float32,
):
  """bev_localizer.py:36-55."""
  width = 3 * depth // 2
  grid = grids.Grid2D.from_extent_meters((width, depth), cell_size)
  grid_p_view = np.array([width / 2, 0.0], dtype)
  qgrid_xy_p = grid.index_to_xyz(grid.grid_index(), dtype)
  q_xy_p = qgrid_xy_p - grid_p_view
  if filter_points_in_fov:
    angle = np.arctan2(q_xy_p[..., 0], q_xy_p[..., 1])
    max_angle = hfov_deg / 2
    q_xy_p = q_xy_p[np.abs(angle) < np.deg2rad(max_angle)][:, None]
  return grid, grid_p_view, q_xy_p


def similarity(f_p_q, map_features, valid_points, temperature=None,
               clip_negative_scores=True, conf_weights=None):
  """bev_localizer.py:157-173.

  f_p_q [B,Nq,D], map_features [B,X,Y,D], valid_points [B,Nq].
  Returns sim_points, prob_points [B,Nq,X,Y] (float32-or-better).
  """
  sim = np.einsum('...nd,...ijd->...nij', f_p_q, map_features)
  if clip_negative_scores:
    sim = np.maximum(sim, 0)
  ctype = np.float64 if sim.dtype == np.float64 else np.float32
  sim = sim.astype(ctype)
  if temperature is not None:
    sim = sim * np.exp(ctype(temperature))
  m = sim.max(axis=(-1, -2), keepdims=True)
  e = np.exp(sim - m)
  prob = e / e.sum(axis=(-1, -2), keepdims=True)
  if conf_weights is not None:
    prob = prob * conf_weights
    sim = sim * conf_weights
  else:
    num_valid = np.clip(valid_points.sum(-1), 1, None)[:, None, None, None]
    sim = sim / num_valid.astype(ctype)
    prob = prob / num_valid.astype(ctype)
  return sim.astype(ctype), prob.astype(ctype)


def interpolate_score_maps(scores, points, valid):
  """pose_estimation.py:49-60.  scores [N,H,W], points [N,2], valid [H,W]."""
  N = scores.shape[0]
  vals = np.empty(N, scores.dtype)
  val_ok = np.empty(N, bool)
  # vectorised over N: each point samples its own map.
  H, W = scores.shape[1:]
  size = np.array([H, W])
  inb = np.all((points >= 0) & (points < size), -1)
  c = points - 0.5
  lo = np.floor(c)
  w_hi = (c - lo).astype(scores.dtype)
  w_lo = 1 - w_hi
  lo = lo.astype(np.int64)
  i0 = np.clip(lo[:, 0], 0, H - 1)
  i1 = np.clip(lo[:, 0] + 1, 0, H - 1)
  j0 = np.clip(lo[:, 1], 0, W - 1)
  j1 = np.clip(lo[:, 1] + 1, 0, W - 1)
  n = np.arange(N)
  vals = (
      (w_lo[:, 0] * w_lo[:, 1]) * scores[n, i0, j0]
      + (w_lo[:, 0] * w_hi[:, 1]) * scores[n, i0, j1]
      + (w_hi[:, 0] * w_lo[:, 1]) * scores[n, i1, j0]
      + (w_hi[:, 0] * w_hi[:, 1]) * scores[n, i1, j1]
  )
  taps_ok = valid[i0, j0] & valid[i0, j1] & valid[i1, j0] & valid[i1, j1]
  val_ok = inb & taps_ok
  return vals, val_ok


def pose_scoring(j_t_i, scores_points_all, i_xy_points, valid_points, valid_j,
                 grid, mask_out_of_bounds):
  """pose_estimation.py:63-82 for ONE pose (Transform2D with shape ())."""
  dtype = scores_points_all.dtype
  j_uv_points = ((j_t_i @ i_xy_points) / dtype.type(grid.cell_size)).astype(dtype)
  scores_points, valid_j_points = interpolate_score_maps(
      scores_points_all, j_uv_points, valid_j
  )
  if mask_out_of_bounds:
    valid_points = valid_points & valid_j_points
  return np.sum(valid_points * scores_points, axis=-1)


def pose_scoring_many(j_t_i, scores_points_all, i_xy_points, valid_points,
                      valid_j, grid, mask_out_of_bounds, chunk=256):
  """pose_estimation.py:208 (vmap over poses), vectorised in chunks.

  j_t_i: Transform2D with shape [P].  Returns scores [P].
  """
  dtype = scores_points_all.dtype
  N, H, W = scores_points_all.shape
  P = j_t_i.angle.shape[0]
  out = np.empty(P, dtype)
  n = np.arange(N)[None]
  for s in range(0, P, chunk):
    tf = j_t_i[s:s + chunk]
    uv = ((tf @ i_xy_points[None]) / dtype.type(grid.cell_size)).astype(dtype)
    inb = np.all((uv >= 0) & (uv < np.array([H, W])), -1)
    c = uv - dtype.type(0.5)
    lo = np.floor(c)
    w_hi = (c - lo).astype(dtype)
    w_lo = 1 - w_hi
    lo = lo.astype(np.int64)
    i0 = np.clip(lo[..., 0], 0, H - 1)
    i1 = np.clip(lo[..., 0] + 1, 0, H - 1)
    j0 = np.clip(lo[..., 1], 0, W - 1)
    j1 = np.clip(lo[..., 1] + 1, 0, W - 1)
    vals = (
        (w_lo[..., 0] * w_lo[..., 1]) * scores_points_all[n, i0, j0]
        + (w_lo[..., 0] * w_hi[..., 1]) * scores_points_all[n, i0, j1]
        + (w_hi[..., 0] * w_lo[..., 1]) * scores_points_all[n, i1, j0]
        + (w_hi[..., 0] * w_hi[..., 1]) * scores_points_all[n, i1, j1]
    )
    vp = valid_points[None]
    if mask_out_of_bounds:
      ok = valid_j[i0, j0] & valid_j[i0, j1] & valid_j[i1, j0] & valid_j[i1, j1]
      vp = vp & inb & ok
    out[s:s + chunk] = np.sum(vp * vals, axis=-1)
  return out


def kabsch_algorithm_2d(i_p, j_p):
  """pose_estimation.py:100-123.  i_p, j_p [N,2] -> (Transform2D i_t_j, valid, rssd)."""
  mu_i = i_p.mean(0)
  mu_j = j_p.mean(0)
  i_p = i_p - mu_i
  j_p = j_p - mu_j
  covariance = np.einsum('ji,jk->ik', i_p, j_p)
  u, s, vh = np.linalg.svd(covariance)
  sign = np.sign(np.linalg.det(u @ vh))
  u = u * np.array([1, sign], u.dtype)
  s = s * np.array([1, sign], s.dtype)
  valid = s[1] > 1e-16 * s[0]
  error = np.sum(np.sum(i_p**2 + j_p**2, axis=1)) - 2 * np.sum(s)
  rssd = np.sqrt(np.clip(error, 0, None))
  i_r_j = u @ vh
  i_p_j = mu_i - i_r_j @ mu_j
  return geometry.Transform2D.from_R(i_r_j, i_p_j), valid, rssd


def poses_from_correspondences(indices, i_xy_p, num_poses, num_retries, grid):
  """pose_estimation.py:146-165: the deterministic part of sample_transforms_ransac.

  indices: [num_poses*num_retries*2, 3] = (query point n, map cell i, map cell j).
  Returns Transform2D [num_poses] (map_t_query).
  """
  dtype = i_xy_p.dtype
  pool_shape = (num_poses, num_retries, 2, 2)
  i_xy_pool = i_xy_p[indices[..., 0]].reshape(pool_shape)
  j_xy_pool = grid.index_to_xyz(indices[..., 1:], dtype).reshape(pool_shape)
  if num_retries > 1:
    d_i = np.linalg.norm(np.diff(i_xy_pool, axis=-2).squeeze(-2), axis=-1)
    d_j = np.linalg.norm(np.diff(j_xy_pool, axis=-2).squeeze(-2), axis=-1)
    ratio = np.maximum(
        d_i / np.clip(d_j, 1e-5, None), d_j / np.clip(d_i, 1e-5, None)
    )
    sel = np.argmin(ratio, axis=-1)
    p = np.arange(num_poses)
    i_xy_pool = i_xy_pool[p, sel]
    j_xy_pool = j_xy_pool[p, sel]
  else:
    i_xy_pool = i_xy_pool.squeeze(1)
    j_xy_pool = j_xy_pool.squeeze(1)
  angles = np.empty(num_poses, dtype)
  ts = np.empty((num_poses, 2), dtype)
  for k in range(num_poses):
    tf, _, _ = kabsch_algorithm_2d(j_xy_pool[k], i_xy_pool[k])
    angles[k] = tf.angle
    ts[k] = tf.t
  return geometry.Transform2D(angles, ts)


def sample_correspondences(prob_points, uniforms):
  """pose_estimation.py:139-145: ``jax.random.choice(p=prob)`` by inverse CDF.

  jax draws ``r = total * (1 - U)`` and returns ``searchsorted(cumsum(p), r)``.
  Here the uniforms are given; float64 CDF.  Returns [S,3] unravelled indices.
  """
  shape = prob_points.shape
  cdf = np.cumsum(prob_points.reshape(-1).astype(np.float64))
  r = cdf[-1] * (1 - uniforms.astype(np.float64))
  flat = np.clip(np.searchsorted(cdf, r), 0, cdf.size - 1)
  return np.stack(np.unravel_index(flat, shape), -1)


def refinement_offsets(dtype=np.float32):
  """pose_estimation.py:178-193: the 41x41x41 (rot, x, y) offset lattice."""
  delta_p, delta_r, range_p, range_r = 0.2, 0.25, 4, 5
  slice_p = slice(-range_p, range_p + delta_p, delta_p)
  slice_r = slice(-range_r, range_r + delta_r, delta_r)
  offsets = np.mgrid[slice_r, slice_p, slice_p]
  shape = offsets.shape[1:]
  offsets = offsets.reshape(3, -1).T.astype(dtype)
  tf = geometry.Transform2D(np.deg2rad(offsets[..., 0]), offsets[..., 1:])
  return tf, shape


def grid_refinement(j_t_i_init, scores_points_all, i_xy_points, valid_points,
                    valid_j, grid, mask_out_of_bounds):
  """pose_estimation.py:168-205 for one scene."""
  dtype = scores_points_all.dtype
  offs, shape = refinement_offsets(dtype)
  init = geometry.Transform2D(
      np.broadcast_to(j_t_i_init.angle, offs.angle.shape),
      np.broadcast_to(j_t_i_init.t, offs.t.shape),
  )
  samples = init @ offs
  samples = geometry.Transform2D(
      samples.angle.astype(dtype), samples.t.astype(dtype)
  )
  scores = pose_scoring_many(
      samples, scores_points_all, i_xy_points, valid_points, valid_j, grid,
      mask_out_of_bounds,
  )
  best = int(np.argmax(scores))
  return samples[best], scores.reshape(shape)


def loss_metrics(pred, m_t_q_gt_3d, threshold_remove_accurate_poses=None):
  """bev_localizer.py:244-278 (numpy; pred fields as numpy / oracle structs)."""
  scores = pred['scores_poses']
  m_t_q_gt = geometry.Transform2D.from_Transform3D(m_t_q_gt_3d)
  gt_b = geometry.Transform2D(m_t_q_gt.angle[..., None], m_t_q_gt.t[..., None, :])
  samples_t_gt = pred['map_t_query_samples'].inv @ gt_b
  dr_samples, dt_samples = samples_t_gt.magnitude()
  if threshold_remove_accurate_poses is not None:
    dr_min, dt_min = threshold_remove_accurate_poses
    remove = (dr_samples < dr_min) & (dt_samples < dt_min)
    remove[..., 0] = False
    scores = np.where(remove, -np.inf, scores)
  m = scores.max(-1, keepdims=True)
  lse = m[..., 0] + np.log(np.exp(scores - m).sum(-1))
  nll = -(scores[..., 0] - lse)
  losses = {'localization/nll': nll, 'total': nll}
  dr, dt = (pred['map_t_query'].inv @ m_t_q_gt).magnitude()
  metrics = {
      'loc/err_max_position': dt,
      'loc/err_max_rotation': dr,
      'loc/recall_top1': np.argmax(pred['scores_poses'], axis=-1) == 0,
  }
  for t in [0.5, 1, 2, 5]:
    metrics[f'loc/recall_max_{t}m'] = dt < t
    metrics[f'loc/recall_max_{t}°'] = dr < t
  for dt_thresh, dr_thresh in [(0.5, 1), (1, 2), (2, 4)]:
    recall = (dr_samples < dr_thresh) & (dt_samples < dt_thresh)
    metrics[f'loc/recall_samples_{dt_thresh}m_{dr_thresh}°'] = np.mean(
        recall[..., 1:], axis=-1
    )
  return losses, metrics
