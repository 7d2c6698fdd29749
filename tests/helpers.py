"""Shared test helpers: tiny configs, parameter conversion, comparison reports."""
import copy

import numpy as np
import torch

from oracle import geometry as o_geo
from snap_amd.configs import defaults


# Device the `-m gpu` tests run on.  Always the GPU inside the suite; only the developer
# plugin tools/dryrun_plugin.py (never loaded by tests/, bench.py or the driver) rebinds it.
DEVICE = 'cuda'


def tiny_localizer_config(num_pose_samples=64, retries=2, top_k=2, feature_dim=32,
                          matching_dim=8, num_bins=8, depth=(1, 1), width=0.5,
                          refine=False, aerial=True):
  """A BEVLocalizer config small enough for the numpy oracle (seconds)."""
  cfg = defaults.bev_localizer()
  cfg.filter_points_in_fov = True
  cfg.num_pose_samples = num_pose_samples
  cfg.num_pose_sampling_retries = retries
  cfg.do_grid_refinement = refine
  cfg.query_frustum_depth = 3.2
  mods = ('streetview', 'aerial') if aerial else ('streetview',)
  cfg.bev_mapper = defaults.bev_mapper(mods)
  m = cfg.bev_mapper
  m.matching_dim = matching_dim
  m.scene_z_height = 2.4
  m.scene_z_offset = 1.6
  sv = m.streetview_encoder
  sv.feature_dim = feature_dim
  sv.num_scale_bins = num_bins
  sv.top_k_view_selection = top_k
  sv.fusion.layers = (2 * feature_dim, feature_dim)
  sv.depth_min_max = (0.5, 8.0)
  sv.image_encoder.output_dim = feature_dim
  sv.image_encoder.encoder.depth = list(depth)
  sv.image_encoder.encoder.width = width
  sv.image_encoder.encoder.limit_num_blocks = len(depth)
  if aerial:
    a = m.aerial_encoder
    a.output_dim = feature_dim
    a.encoder.depth = list(depth)
    a.encoder.width = width
    a.encoder.limit_num_blocks = len(depth)
  return cfg


def params_to_numpy(params, dtype=np.float32):
  if isinstance(params, dict):
    return {k: params_to_numpy(v, dtype) for k, v in params.items()}
  return params.detach().cpu().numpy().astype(dtype)


def params_to_device(params, device):
  if isinstance(params, dict):
    return {k: params_to_device(v, device) for k, v in params.items()}
  return params.to(device)


def scene_to_oracle(scene, dtype=np.float32):
  """torch scene dict -> numpy dict with oracle camera / transform structs."""
  def n(t):
    return t.detach().cpu().numpy().astype(dtype)
  cam = scene['camera']
  T = scene['T_view2scene']
  out = {
      'images': n(scene['images']),
      'camera': o_geo.FisheyeCamera(n(cam.wh), n(cam.f), n(cam.c), n(cam.k_radial), n(cam.max_fov)),
      'T_view2scene': o_geo.Transform3D(n(T.R), n(T.t)),
  }
  if 'rasters' in scene:
    out['rasters'] = {}
    if 'rgb' in scene['rasters']:
      out['rasters']['rgb'] = n(scene['rasters']['rgb'])
    if 'semantics' in scene['rasters']:
      out['rasters']['semantics'] = scene['rasters']['semantics'].detach().cpu().numpy().astype(bool)
  if 'xyz_query' in scene:
    out['xyz_query'] = n(scene['xyz_query'])
  return out


def batch_to_oracle(batch, dtype=np.float32):
  out = {'map': scene_to_oracle(batch['map'], dtype), 'query': scene_to_oracle(batch['query'], dtype)}
  if 'T_query2map' in batch:
    T = batch['T_query2map']
    out['T_query2map'] = o_geo.Transform3D(
        T.R.detach().cpu().numpy().astype(dtype), T.t.detach().cpu().numpy().astype(dtype)
    )
  return out


def batch_to_device(batch, device):
  def mv(x):
    if isinstance(x, dict):
      return {k: mv(v) for k, v in x.items()}
    if hasattr(x, 'to'):
      return x.to(device)
    return x
  return mv(batch)


def report(name, got, want, atol, rtol=0.0):
  """Assert closeness with a diagnostic message (max error, location, fraction bad)."""
  got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
  want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
  assert got.shape == want.shape, f'{name}: shape {got.shape} vs {want.shape}'
  if got.dtype == bool or want.dtype == bool or got.dtype.kind in 'iu':
    bad = got != want
    assert not bad.any(), (
        f'{name}: {bad.sum()} / {bad.size} mismatches; first at '
        f'{np.argwhere(bad)[0].tolist()}: got {got[bad][0]} want {want[bad][0]}'
    )
    return
  g = got.astype(np.float64)
  w = want.astype(np.float64)
  both_inf = np.isinf(g) & np.isinf(w) & (np.sign(g) == np.sign(w))
  both_nan = np.isnan(g) & np.isnan(w)
  err = np.where(both_inf | both_nan, 0.0, np.abs(g - w))
  err = np.where(np.isnan(err), np.inf, err)
  tol = atol + rtol * np.abs(np.where(np.isfinite(w), w, 0))
  bad = err > tol
  if bad.any():
    i = np.unravel_index(np.argmax(np.where(np.isfinite(err), err, 1e30)), err.shape)
    raise AssertionError(
        f'{name}: {bad.sum()} / {bad.size} beyond tol (atol={atol}, rtol={rtol}); '
        f'max err {err[i]:.3e} at {tuple(int(k) for k in i)}: got {g[i]:.6e} want {w[i]:.6e}; '
        f'ref |max| {np.nanmax(np.abs(np.where(np.isfinite(w), w, 0))):.3e}'
    )


def assert_same_argmax(name, got_scores, want_scores, got_index=None, rel=2e-6):
  """Row-wise argmax parity.

  The kernel's argmax must equal the oracle's.  The only tolerated deviation is an
  fp32 near-tie: the oracle's score AT the kernel's index must then be within
  `rel` (a few fp32 ulps of the summed magnitude) of the oracle's maximum -- two
  different summation orders cannot order such values consistently.
  """
  g = got_scores.detach().cpu().numpy() if isinstance(got_scores, torch.Tensor) else np.asarray(got_scores)
  w = np.asarray(want_scores.detach().cpu().numpy() if isinstance(want_scores, torch.Tensor) else want_scores)
  gi = np.argmax(g, -1)
  if got_index is not None:
    k = got_index.detach().cpu().numpy() if isinstance(got_index, torch.Tensor) else np.asarray(got_index)
    assert (k == gi).all(), f'{name}: reported index {k} != argmax of reported scores {gi}'
  wi = np.argmax(w, -1)
  rows = np.arange(w.shape[0])
  exact = gi == wi
  gap = (w[rows, wi] - w[rows, gi]) / np.maximum(np.abs(w[rows, wi]), 1e-30)
  assert (exact | (gap <= rel)).all(), (
      f'{name}: argmax {gi} vs oracle {wi}; oracle score gap {gap} exceeds near-tie bound {rel}'
  )
  # how often the near-tie escape was taken is part of the result: printed, counted for the
  # session summary (conftest.py), and returned so that a test can demand zero
  escapes = int((~exact).sum())
  NEAR_TIE['rows'] += int(exact.size)
  NEAR_TIE['escapes'] += escapes
  print(f'[argmax] {name}: {exact.size - escapes} of {exact.size} rows identical, '
        f'{escapes} near-tie escape(s) (oracle gap <= {rel})')
  return escapes


NEAR_TIE = {'rows': 0, 'escapes': 0}


def assert_template_validity_mismatches_on_borders(name, got_valid, want_valid, cell_size, tol=1e-4):
  """Rotated-template validity [R, H, W] (pose_exhaustive_voting.py:37-69) must equal the oracle's
  EXCEPT where a float64 evaluation of the rotated coordinate puts the cell within `tol` cells of
  a decision boundary: the grid border (0 <= u < H) or a change of the bilinear tap pair
  (u - 0.5 integral).  Returns the number of such boundary flips (0 in every test so far)."""
  got = got_valid.detach().cpu().numpy() if isinstance(got_valid, torch.Tensor) else np.asarray(got_valid)
  want = np.asarray(want_valid)
  assert got.shape == want.shape
  mism = np.argwhere(got != want)
  if len(mism) == 0:
    return 0
  R, H, W = got.shape
  RQ = R // 4
  c = np.array([H * cell_size / 2.0, W * cell_size / 2.0])
  for rr, di, dj in mism:
    k, r0 = divmod(int(rr), RQ)
    # source cell of the first-quadrant template whose rot90 copy lands on (di, dj)
    si, sj = [(di, dj), (H - 1 - dj, di), (H - 1 - di, W - 1 - dj), (dj, H - 1 - di)][k]
    th = 2.0 * np.pi * r0 / R
    p = np.array([(si + 0.5) * cell_size, (sj + 0.5) * cell_size]) - c
    q = np.array([np.cos(th) * p[0] - np.sin(th) * p[1], np.sin(th) * p[0] + np.cos(th) * p[1]]) + c
    u, v = q / cell_size
    d = min(abs(u), abs(u - H), abs(v), abs(v - W),
            abs((u - 0.5) - round(u - 0.5)), abs((v - 0.5) - round(v - 0.5)))
    assert d <= tol, (f'{name}: validity differs at template {rr} cell ({di}, {dj}) whose float64 '
                      f'coordinate ({u:.6f}, {v:.6f}) is {d:.2e} cells from any decision boundary')
  print(f'[validity] {name}: {len(mism)} boundary flip(s) of {got.size}')
  return len(mism)


def assert_validity_mismatches_on_borders(name, got_valid, want_valid, scene, xyz_query, stride,
                                          tol=2e-4, max_view_distance=None):
  """Boolean voxel validity must equal the oracle's EXCEPT where a float64 re-projection shows the
  voxel sitting on a visibility boundary of some view: within `tol` feature-map pixels of an image
  edge, within `tol` of the near plane (z = eps), or within `tol` (relative) of the fisheye FoV
  limit -- the only places where two correct fp32 evaluations of streetview_encoder.py:42-65 may
  disagree.  `scene` is the oracle-side scene dict (camera, T_view2scene); `xyz_query` [B,...,3];
  `stride` = (sy, sx) of the feature level the lift samples.  Returns the mismatch count."""
  got = np.asarray(got_valid.detach().cpu().numpy() if hasattr(got_valid, 'detach') else got_valid, bool)
  want = np.asarray(want_valid, bool)
  assert got.shape == want.shape, f'{name}: shape {got.shape} vs {want.shape}'
  mism = got != want
  n = int(mism.sum())
  if n == 0:
    return 0
  B = got.shape[0]
  f8 = lambda a: np.asarray(a, np.float64)
  cam, T = scene['camera'], scene['T_view2scene']
  cam64 = o_geo.FisheyeCamera(f8(cam.wh), f8(cam.f), f8(cam.c), f8(cam.k_radial), f8(cam.max_fov))
  cam64 = cam64.scale(1.0 / f8(stride)[::-1])
  R, t = f8(T.R), f8(T.t)
  pts = f8(xyz_query).reshape(B, -1, 3)
  flat = mism.reshape(B, -1)
  worst = 0.0
  for b in range(B):
    idx = np.nonzero(flat[b])[0]
    if len(idx) == 0:
      continue
    p = pts[b, idx]                                             # [n, 3]
    pv = np.einsum('vji,vnj->vni', R[b], p[None] - t[b][:, None])   # R^T (p - t): [V, n, 3]
    z = pv[..., 2]
    zc = np.clip(z, cam64.eps, None)
    xy = pv[..., :2] / zc[..., None]
    radius = np.sqrt((xy ** 2).sum(-1))
    theta = np.arctan(np.maximum(radius, 1e-12))
    k = cam64.k_radial[b][:, None, :]
    offset = sum(k[..., i] * theta ** (2 * (i + 1)) for i in range(3))
    dist = np.where(radius < cam64.eps, 1.0, (offset + 1) * theta / np.maximum(radius, 1e-12))
    uv = xy * dist[..., None] * cam64.f[b][:, None, :] + cam64.c[b][:, None, :]
    wh = cam64.wh[b][:, None, :]
    lim = np.tan(0.5 * cam64.max_fov[b])[:, None]
    margin = np.minimum.reduce([
        np.abs(uv[..., 0]), np.abs(uv[..., 0] - wh[..., 0]),
        np.abs(uv[..., 1]), np.abs(uv[..., 1] - wh[..., 1]),
        np.abs(z - cam64.eps) * 1e3,                           # near plane, in units of eps
        np.abs(radius - lim) / np.maximum(lim, 1e-12) * 1e2,   # FoV limit (relative, scaled)
    ])                                                         # [V, n]
    m = margin.min(0)
    if max_view_distance is not None:
      d = np.sqrt(((p[None] - t[b][:, None]) ** 2).sum(-1))    # [V, n]
      m = np.minimum(m, np.abs(d - max_view_distance).min(0))
    worst = max(worst, float(m.max()))
  print(f'[parity] {name:38s} validity mismatches {n} of {got.size}, farthest from a visibility '
        f'boundary: {worst:.2e} (tol {tol:.0e})')
  assert worst <= tol, (f'{name}: a voxel whose validity differs from the oracle is {worst:.3e} away '
                        f'from every visibility boundary (tol {tol})')
  return n
