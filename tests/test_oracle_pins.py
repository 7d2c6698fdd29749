"""Pins the numpy oracle (CPU only).

The reference ships no tests / golden vectors and cannot be imported here (SURVEY
8c), so the oracle is pinned against the documented models of the third-party
primitives the reference calls (scipy.ndimage.map_coordinates,
scipy.signal.convolve, torch.nn.functional on CPU) and against analytic
known-answer tests.
"""
import numpy as np
import pytest
import scipy.ndimage as ndi
import scipy.signal as sig
import torch
import torch.nn.functional as F

from oracle import bev as o_bev
from oracle import encoder as o_enc
from oracle import geometry as o_geo
from oracle import grids as o_grids
from oracle import lift as o_lift
from oracle import pose as o_pose
from oracle import voting as o_voting


# -- grids.interpolate_nd vs scipy ------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_interpolate_nd_matches_scipy(dtype):
  rng = np.random.default_rng(0)
  a = rng.standard_normal((7, 9, 3)).astype(dtype)
  pts = rng.uniform(-1.5, 10.5, (400, 2)).astype(dtype)
  pts[:8] = [[0, 0], [7, 9], [0.5, 0.5], [6.5, 8.5], [0.25, 8.9], [6.999, 0.001], [3.5, 4.5], [7, 0]]
  v, ok = o_grids.interpolate_nd(a, pts)
  ref = np.stack(
      [ndi.map_coordinates(a[..., c], (pts - 0.5).T, order=1, mode='nearest') for c in range(3)], -1
  )
  np.testing.assert_allclose(v, ref, atol=1e-6 if dtype == np.float32 else 1e-13)
  inb = np.all((pts >= 0) & (pts < np.array([7, 9])), -1)
  assert (ok == inb).all()


def test_interpolate_nd_nan_mask_validity():
  """A tap with zero weight still invalidates (0 * nan = nan), as scipy does."""
  rng = np.random.default_rng(1)
  a = rng.standard_normal((6, 6, 2))
  valid = np.ones((6, 6), bool)
  valid[2, 3] = False
  pts = np.array([[2.5, 3.5], [2.5, 2.5], [1.5, 3.5], [3.5, 3.5], [4.0, 4.0], [2.0, 3.5]])
  _, ok = o_grids.interpolate_nd(a, pts, valid)
  nan_mask = np.where(valid, 0.0, np.nan)
  ref = ~np.isnan(ndi.map_coordinates(nan_mask, (pts - 0.5).T, order=1, mode='nearest'))
  assert (ok == ref).all()
  # (2.5, 2.5) sits exactly on cell (2,2): its upper taps (weight 0) touch (2,3).
  assert not ok[1]


def test_interpolate_nd_1d_depth_bins():
  s = np.arange(8, dtype=np.float64) ** 2
  idx = np.array([0.5, 1.0, 7.5, 7.9, 3.25])
  v, _ = o_grids.interpolate_nd(s[:, None], idx[:, None])
  ref = ndi.map_coordinates(s, (idx - 0.5)[None], order=1, mode='nearest')
  np.testing.assert_allclose(v[:, 0], ref)


# -- exhaustive voting vs scipy ----------------------------------------------------
def test_template_matching_matches_scipy_convolve():
  rng = np.random.default_rng(2)
  H = W = 8
  R, D = 4, 3
  q = rng.standard_normal((R, H, W, D))
  qv = rng.random((R, H, W)) > 0.2
  m = rng.standard_normal((H, W, D))
  mv = rng.random((H, W)) > 0.1
  s = o_voting.template_matching(q, qv, m, mv)
  # literal restatement with scipy of pose_exhaustive_voting.py:83-103
  mp = np.pad(m, ((H - 1,) * 2, (W - 1,) * 2, (0, 0)), mode='edge')
  scores = np.stack([
      sum(sig.convolve(q[r, ::-1, ::-1, d], mp[..., d], mode='valid', method='direct')
          for d in range(D))
      for r in range(R)
  ])
  mvp = np.pad(mv.astype(float), ((H - 1,) * 2, (W - 1,) * 2))
  nv = np.stack([
      sig.convolve(qv[r].astype(float), mvp, mode='valid', method='direct') for r in range(R)
  ])
  ref = np.where(nv > 0.05 * H * W, scores, -np.inf) / qv.sum((-1, -2), keepdims=True)
  assert (np.isfinite(s) == np.isfinite(ref)).all()
  fin = np.isfinite(ref)
  np.testing.assert_allclose(s[fin], ref[fin], atol=1e-12)


def test_template_matching_without_padding_matches_scipy_full_convolve():
  """pose_exhaustive_voting.py:86 mode='full' (do_padding=False): the oracle's zero-extended correlation
  against the literal scipy.signal.convolve(mode='full'); with an overlap threshold the reference's
  count has the padded shape and cannot broadcast (:93-101): the oracle raises."""
  rng = np.random.default_rng(12)
  R, H, W, Hm, Wm, D = 3, 6, 7, 9, 5, 2
  q = rng.standard_normal((R, H, W, D))
  qv = rng.random((R, H, W)) > 0.2
  m = rng.standard_normal((Hm, Wm, D))
  mv = rng.random((Hm, Wm)) > 0.1
  s = o_voting.template_matching(q, qv, m, mv, do_padding=False, min_overlap=None)
  ref = np.stack([
      sum(sig.convolve(q[r, ::-1, ::-1, d], m[..., d], mode='full', method='direct') for d in range(D))
      for r in range(R)
  ]) / qv.sum((-1, -2), keepdims=True)
  assert s.shape == ref.shape == (R, Hm + H - 1, Wm + W - 1)
  np.testing.assert_allclose(s, ref, atol=1e-12)
  with pytest.raises(ValueError):
    o_voting.template_matching(q, qv, m, mv, do_padding=False)


@pytest.mark.parametrize('H,W,Hm,Wm,R,D', [(16, 16, 16, 16, 8, 8), (24, 24, 24, 24, 12, 32), (22, 22, 22, 22, 7, 34),
                                           (10, 13, 12, 9, 12, 16), (43, 43, 43, 43, 4, 8), (30, 30, 30, 30, 4, 4)])
@pytest.mark.parametrize('overlap', [0.05, None])
def test_fft64_template_matching_equals_the_oracle(H, W, Hm, Wm, R, D, overlap):
  """tests/fft_reference.py (float64 numpy / scipy FFT correlation, the checker of the frequency-domain
  voting at 256 x 256 cells) == oracle/voting.template_matching (direct sliding-window sum) at the small
  geometries the GPU test runs: -inf mask exact, finite scores to 1e-10 (both float64 here)."""
  import fft_reference
  rng = np.random.default_rng(500 + H + R + D)
  t = rng.standard_normal((R, H, W, D))
  tv = rng.random((R, H, W)) > 0.2
  t = t * tv[..., None]
  fm = rng.standard_normal((Hm, Wm, D))
  vm = rng.random((Hm, Wm)) > 0.1
  want = o_voting.template_matching(t, tv, fm, vm, min_overlap=overlap)
  got = fft_reference.template_matching_fft64(t, tv, fm, vm, min_overlap=overlap)
  assert got.shape == want.shape
  fw, fg = np.isfinite(want), np.isfinite(got)
  assert (fw == fg).all()
  if overlap is not None:
    assert 0 < int((~fw).sum()) and (want[~fw] == got[~fg]).all()
  np.testing.assert_allclose(got[fg], want[fw], atol=1e-10)


def test_exhaustive_identity_known_answer():
  """identity pose => argmax at (0, H-1, W-1)."""
  rng = np.random.default_rng(3)
  H = 16
  g = o_grids.Grid2D((H, H), 0.5)
  f = rng.standard_normal((H, H, 4)).astype(np.float32)
  v = np.ones((H, H), bool)
  sc = o_voting.exhaustive_pose_voting(dict(features=f, valid=v), dict(features=f, valid=v), 8, g)
  assert np.unravel_index(np.argmax(sc), sc.shape) == (0, H - 1, H - 1)


def _smooth_field(rng, n, d, size):
  """A smooth random function R^2 -> R^d (sum of Gaussians), evaluable anywhere."""
  centres = rng.uniform(0, size, (n, 2))
  amps = rng.standard_normal((n, d))
  def f(xy):
    d2 = ((xy[..., None, :] - centres) ** 2).sum(-1)
    return np.exp(-d2 / (2 * 1.2**2)) @ amps
  return f


def test_exhaustive_planted_pose_known_answer():
  """q(u) = m(T u) with a centre-frame rotation by -2 pi k / R and a shift of s
  cells => argmax (k, H-1+s_x, W-1+s_y)  (probed KAT)."""
  rng = np.random.default_rng(4)
  H, R, k, s = 40, 36, 5, np.array([3, -4])
  cell = 0.5
  g = o_grids.Grid2D((H, H), cell)
  field = _smooth_field(rng, 60, 6, H * cell)
  xy = g.index_to_xyz(g.grid_index()).reshape(-1, 2)
  m = field(xy).reshape(H, H, -1)
  c = o_voting.get_grid_center_transform(g, np.float64)
  centre_tf = o_geo.Transform2D(np.asarray(-2 * np.pi * k / R), s * cell)
  m_t_q = c @ centre_tf @ c.inv
  q = field(m_t_q @ xy).reshape(H, H, -1)
  v = np.ones((H, H), bool)
  sc = o_voting.exhaustive_pose_voting(dict(features=q, valid=v), dict(features=m, valid=v), R, g)
  assert np.unravel_index(np.argmax(sc), sc.shape) == (k, H - 1 + s[0], H - 1 + s[1])
  # the reference's index<->transform helpers are mutually consistent.
  idx = np.array([k, H - 1 + s[0], H - 1 + s[1]])
  tf = o_voting.exhaustive_index_to_tfm(idx, g, R, np.float64)
  back = o_voting.exhaustive_tfm_to_index(tf, g, R, np.float64)
  np.testing.assert_allclose(back, idx, atol=1e-6)


# -- encoder primitives vs torch (CPU) -----------------------------------------------
def test_conv_pool_resize_groupnorm_match_torch():
  rng = np.random.default_rng(5)
  x = rng.standard_normal((2, 11, 9, 8)).astype(np.float32)
  w = rng.standard_normal((3, 3, 8, 5)).astype(np.float32)
  xt = torch.tensor(x).permute(0, 3, 1, 2)
  wt = torch.tensor(w).permute(3, 2, 0, 1)
  for stride, pad in [(1, 1), (2, 1), (2, 0)]:
    got = o_enc.conv2d(x, w, (stride, stride), ((pad, pad), (pad, pad)))
    ref = F.conv2d(xt, wt, stride=stride, padding=pad).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, ref, atol=2e-5)
  got = o_enc.max_pool(x)
  ref = F.max_pool2d(xt, 3, 2, 1).permute(0, 2, 3, 1).numpy()
  np.testing.assert_array_equal(got, ref)
  got = o_enc.resize_bilinear_x2(x)
  ref = F.interpolate(xt, scale_factor=2, mode='bilinear', align_corners=False)
  np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), atol=1e-6)
  x32 = rng.standard_normal((2, 5, 6, 64)).astype(np.float32) * 3 + 1
  gamma = rng.standard_normal(64).astype(np.float32)
  beta = rng.standard_normal(64).astype(np.float32)
  got = o_enc.group_norm(x32, gamma, beta)
  ref = F.group_norm(torch.tensor(x32).permute(0, 3, 1, 2), 32, torch.tensor(gamma),
                     torch.tensor(beta), eps=1e-5).permute(0, 2, 3, 1).numpy()
  np.testing.assert_allclose(got, ref, atol=2e-5)


def test_weight_standardisation_statistics():
  rng = np.random.default_rng(6)
  w = (rng.standard_normal((3, 3, 16, 8)) * 0.3 + 0.2).astype(np.float32)
  ws = o_enc.standardize(w, (0, 1, 2), 1e-10)
  np.testing.assert_allclose(ws.mean((0, 1, 2)), 0, atol=1e-6)
  np.testing.assert_allclose((ws**2).mean((0, 1, 2)), 1, atol=1e-5)


def test_pad_to_multiple_quirk():
  x = np.zeros((1, 64, 60, 3), np.float32)
  assert o_enc.pad_to_multiple(x, 32).shape == (1, 96, 64, 3)  # 64 -> +32 (quirk), 60 -> 64


# -- geometry -----------------------------------------------------------------------
def test_transform_identities():
  rng = np.random.default_rng(7)
  a = o_geo.Transform2D(rng.uniform(-3, 3, 5), rng.standard_normal((5, 2)))
  b = o_geo.Transform2D(rng.uniform(-3, 3, 5), rng.standard_normal((5, 2)))
  p = rng.standard_normal((5, 4, 2))
  np.testing.assert_allclose((a @ b) @ p, a @ (b @ p), atol=1e-12)
  ident = a @ a.inv
  np.testing.assert_allclose(ident.angle, 0, atol=1e-12)
  np.testing.assert_allclose(ident.t, 0, atol=1e-12)
  dr, dt = o_geo.Transform2D(np.array([np.deg2rad(350.0)]), np.array([[3.0, 4.0]])).magnitude()
  np.testing.assert_allclose(dr, 10.0)
  np.testing.assert_allclose(dt, 5.0)
  th = 0.7
  R3 = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
  t3 = o_geo.Transform3D(R3, np.array([1.0, 2.0, 3.0]))
  t2 = o_geo.Transform2D.from_Transform3D(t3)
  np.testing.assert_allclose(t2.angle, th)
  np.testing.assert_allclose(t2.t, [1.0, 2.0])
  np.testing.assert_allclose((t3 @ t3.inv).R, np.eye(3), atol=1e-12)


def test_fisheye_projection_properties():
  cam = o_geo.FisheyeCamera(np.array([100.0, 80.0]), np.array([60.0, 60.0]),
                            np.array([50.0, 40.0]), np.zeros(3), np.asarray(np.deg2rad(115.0)))
  p = np.array([[0, 0, 5.0], [0.0, 0.0, -1.0], [50.0, 0, 1.0], [1.0, 0.5, 4.0]])
  uv, ok = cam.world2image(p)
  np.testing.assert_allclose(uv[0], [50.0, 40.0])      # optical axis -> principal point
  assert ok[0] and not ok[1] and not ok[2]             # behind camera / outside FoV
  r = np.hypot(0.25, 0.125)                            # equidistant model: r_d = atan(r)
  np.testing.assert_allclose(np.hypot(*(uv[3] - [50, 40])) / 60.0, np.arctan(r), rtol=1e-12)


# -- pose ---------------------------------------------------------------------------
def test_kabsch_two_points_recovers_transform():
  rng = np.random.default_rng(8)
  for _ in range(20):
    tf = o_geo.Transform2D(np.asarray(rng.uniform(-3.1, 3.1)), rng.standard_normal(2))
    i_p = rng.standard_normal((2, 2)) * 3
    j_p = tf @ i_p
    est, valid, rssd = o_pose.kabsch_algorithm_2d(j_p, i_p)
    np.testing.assert_allclose(np.exp(1j * est.angle), np.exp(1j * tf.angle), atol=1e-9)
    np.testing.assert_allclose(est.t, tf.t, atol=1e-9)
    assert rssd < 1e-6


def test_frustum_grid_default_sizes():
  grid, p, q = o_pose.build_query_frustum_grid(0.2, 16.0, True, 72.0)
  assert grid.extent == (120, 80) and q.shape == (4652, 1, 2)
  np.testing.assert_allclose(p, [12.0, 0.0])


def test_refinement_lattice_is_41_cubed():
  tf, shape = o_pose.refinement_offsets()
  assert shape == (41, 41, 41) and tf.angle.shape == (68921,)
  np.testing.assert_allclose(tf.t.min(), -4.0, atol=1e-6)
  np.testing.assert_allclose(np.rad2deg(tf.angle.max()), 5.0, atol=1e-5)


def test_pose_scoring_prefers_planted_pose():
  """Scores from sim maps peaked at T(q_n) are maximal at T."""
  rng = np.random.default_rng(9)
  X = Y = 32
  cell = 0.25
  grid = o_grids.Grid2D((X, Y), cell)
  gt = o_geo.Transform2D(np.asarray(0.4), np.array([4.0, 3.5]))
  q_xy = rng.uniform(-1.5, 1.5, (40, 2))
  target = (gt @ q_xy) / cell
  ii, jj = np.meshgrid(np.arange(X) + 0.5, np.arange(Y) + 0.5, indexing='ij')
  sim = np.exp(-((ii[None] - target[:, 0, None, None]) ** 2
                 + (jj[None] - target[:, 1, None, None]) ** 2) / 4.0)
  poses = o_geo.Transform2D(
      np.concatenate([[0.4], rng.uniform(-3, 3, 200)]),
      np.concatenate([[[4.0, 3.5]], rng.uniform(0, 8, (200, 2))]),
  )
  sc = o_pose.pose_scoring_many(poses, sim, q_xy, np.ones(40, bool), np.ones((X, Y), bool), grid, False)
  assert np.argmax(sc) == 0
  one = o_pose.pose_scoring(poses[0], sim, q_xy, np.ones(40, bool), np.ones((X, Y), bool), grid, False)
  np.testing.assert_allclose(one, sc[0], rtol=1e-12)


def test_view_selection_ties_and_few_visible():
  pts = np.zeros((1, 2, 3))
  T = o_geo.Transform3D(np.tile(np.eye(3), (1, 4, 1, 1)),
                        np.array([[[1.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [0.5, 0, 0]]]))
  vis = np.array([[[True, True, True, False], [False, False, False, False]]])
  idx, md = o_lift.view_selection(pts, T, vis, 3)
  assert idx[0, 0].tolist() == [0, 1, 2]     # tie -> lowest index first; invisible last
  assert idx[0, 1].tolist() == [0, 1, 2]     # nothing visible: lowest indices
  assert md[0, 0] == 1.0 and np.isinf(md[0, 1])


def test_pool_multiview_all_invalid_is_zero():
  feats = np.ones((2, 3, 4))
  valid = np.array([[True, False, True], [False, False, False]])
  scores = np.array([[0.1, 5.0, -0.2], [1.0, 2.0, 3.0]])
  stats, any_ = o_lift.pool_multiview_features(feats, valid, scores, False, True)
  assert any_.tolist() == [True, False]
  np.testing.assert_allclose(stats[1], 0)
  np.testing.assert_allclose(stats[0, :4], 1)       # mean of ones
  np.testing.assert_allclose(stats[0, 4:8], 0, atol=1e-15)  # variance
  np.testing.assert_allclose(stats[0, 8], 0.1)      # max valid score


def test_vit_oracle_against_independent_torch_implementations():
  """The reference has no ViT (image_encoder.py:103), so oracle/vit.py cannot be pinned to it
  ("parity unpinned" in its header); its primitives are pinned here to torch's own
  implementations of the same published operators."""
  import torch
  from oracle import vit as o_vit
  rng = np.random.default_rng(0)
  p = rng.standard_normal((1, 16, 8))
  t = torch.nn.functional.interpolate(
      torch.from_numpy(p).reshape(1, 4, 4, 8).permute(0, 3, 1, 2), size=(4, 3), mode='bilinear',
      align_corners=False).permute(0, 2, 3, 1).reshape(1, 12, 8).numpy()
  np.testing.assert_allclose(o_vit.resize_posemb(p, (4, 4), (4, 3)), t, atol=1e-12)
  qkv = rng.standard_normal((2, 10, 3, 2, 64))
  tq = torch.from_numpy(qkv)
  ref = torch.nn.functional.scaled_dot_product_attention(
      tq[:, :, 0].permute(0, 2, 1, 3), tq[:, :, 1].permute(0, 2, 1, 3),
      tq[:, :, 2].permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(2, 10, 128).numpy()
  np.testing.assert_allclose(o_vit.attention(qkv), ref, atol=1e-12)
  x = rng.standard_normal((5, 64))
  np.testing.assert_allclose(
      o_vit.gelu_tanh(x), torch.nn.functional.gelu(torch.from_numpy(x), approximate='tanh').numpy(),
      atol=1e-12)
  g, b = rng.standard_normal(64), rng.standard_normal(64)
  np.testing.assert_allclose(
      o_vit.layer_norm(x, g, b),
      torch.nn.functional.layer_norm(torch.from_numpy(x), (64,), torch.from_numpy(g),
                                     torch.from_numpy(b), eps=1e-6).numpy(), atol=1e-12)
  # bf16 rounding restatement == torch's float32 -> bfloat16 conversion
  from oracle import encoder as o_enc
  v = (rng.standard_normal(4096) * 3.3).astype(np.float32)
  np.testing.assert_array_equal(o_enc.bf16_round(v), torch.from_numpy(v).to(torch.bfloat16).float().numpy())


def test_semantic_net_cross_entropies_against_torch():
  """oracle/semantic_net.py's log-softmax / sigmoid cross-entropies == torch.nn.functional's
  (the published definitions optax implements, semantic_net.py:65,98)."""
  import torch
  from oracle import semantic_net as o_sem
  rng = np.random.default_rng(3)
  logits = rng.standard_normal((2, 5, 4, 6)) * 3
  labels = rng.integers(0, 6, (2, 5, 4))
  valid = np.ones((2, 5, 4), bool)
  nll, _ = o_sem.multiclass_crossentropy_metrics(logits, labels, valid, list('abcdef'), None)
  ref = F.cross_entropy(torch.from_numpy(logits).reshape(-1, 6), torch.from_numpy(labels).reshape(-1),
                        reduction='none').reshape(2, -1).mean(-1).numpy()
  np.testing.assert_allclose(nll, ref, rtol=1e-12)
  gt = rng.random((2, 5, 4, 6)) < 0.4
  nll, _ = o_sem.binary_crossentropy_metrics(logits, gt, valid, list('abcdef'), None)
  ref = F.binary_cross_entropy_with_logits(torch.from_numpy(logits), torch.from_numpy(gt).double(),
                                           reduction='none').mean(-1).reshape(2, -1).mean(-1).numpy()
  np.testing.assert_allclose(nll, ref, rtol=1e-12)


# -- pins the oracle does not generate itself (closed forms / hand arithmetic) -----------------
def test_lift_of_an_affine_feature_field_is_the_field_at_the_projected_point():
  """Bilinear interpolation reproduces an affine function exactly, so lifting an image whose
  channel c is a_c * i + b_c * j + c_c must return that function evaluated at the projected
  (continuous, half-pixel-centred) coordinate -- a closed form independent of any tap logic
  (streetview_encoder.py:69-76, grids.py:116-137)."""
  rng = np.random.default_rng(3)
  B, V, h, w, D, N = 1, 2, 12, 16, 3, 200
  a, b, c = rng.standard_normal((3, D))
  ii, jj = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing='ij')   # pixel centres
  img = (a * ii[..., None] + b * jj[..., None] + c).astype(np.float64)
  f_images = np.broadcast_to(img, (B, V, h, w, D)).copy()
  # interior points only (1 px away from the border: no clamped taps)
  p2d = np.stack([rng.uniform(1.0, h - 1.0, (B, N, V)), rng.uniform(1.0, w - 1.0, (B, N, V))], -1)
  got = o_lift.interpolate_views_all(f_images, p2d)
  want = a * p2d[..., 0:1] + b * p2d[..., 1:2] + c
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
  # the selective branch (top-K gather) follows the same closed form
  idx = np.stack([rng.integers(0, V, (B, N)) for _ in range(2)], -1)
  p_sel = np.take_along_axis(p2d, idx[..., None], axis=2)
  got_s = o_lift.interpolate_views_selective(f_images, p_sel, idx)
  np.testing.assert_allclose(got_s, a * p_sel[..., 0:1] + b * p_sel[..., 1:2] + c, rtol=0, atol=1e-12)


def test_two_view_softmax_pooling_by_hand():
  """pool_multiview_features (streetview_encoder.py:141-178) on numbers small enough to do by
  hand: two valid views with scores 0.5 and 1.5, features (1, 2) and (3, 6).
  softmax weights: e^0.5 / (e^0.5 + e^1.5) = 1 / (1 + e) and e / (1 + e)."""
  e = np.e
  w0, w1 = 1 / (1 + e), e / (1 + e)
  feats = np.array([[[1.0, 2.0], [3.0, 6.0], [100.0, -100.0]]])       # third view is invalid
  valid = np.array([[True, True, False]])
  scores = np.array([[0.5, 1.5, 9.0]])
  stats, ok = o_lift.pool_multiview_features(feats, valid, scores, add_minmax=True, use_variance=True)
  mean = np.array([w0 * 1 + w1 * 3, w0 * 2 + w1 * 6])
  var = np.array([w0 * (1 - mean[0]) ** 2 + w1 * (3 - mean[0]) ** 2,
                  w0 * (2 - mean[1]) ** 2 + w1 * (6 - mean[1]) ** 2])
  want = np.concatenate([mean, var, [3.0, 6.0], [1.0, 2.0], [1.5]])    # mean | var | max | min | score_max
  assert ok.tolist() == [True]
  np.testing.assert_allclose(stats[0], want, rtol=1e-13)
  # no valid view at all: zeros and valid = False (the double-where keeps it finite)
  stats0, ok0 = o_lift.pool_multiview_features(feats, np.array([[False, False, False]]), scores, True, True)
  assert ok0.tolist() == [False] and not np.any(stats0)
  # unweighted (scores = None): plain mean / population variance over the valid views
  s2, _ = o_lift.pool_multiview_features(feats, valid, None, add_minmax=False, use_variance=True)
  np.testing.assert_allclose(s2[0], [2.0, 4.0, 1.0, 4.0], rtol=1e-13)


def test_softmax_where_initial_semantics():
  """jax.nn.softmax(x, where=w, initial=0): JAX computes
      unnormalized = exp(x - max(x, where=w, initial=0));  result = unnormalized / sum(unnormalized, where=w)
  (jax/_src/nn/functions.py: `x_max = jnp.max(x, axis, where=where, initial=initial, keepdims=True)`),
  i.e. the shift is max(0, max over the selected entries) -- mathematically the plain softmax over
  the selected entries, numerically a different (never larger-than-needed) shift.  All-negative
  scores therefore use shift 0.  KAT on both the view-pooling and the vertical-pooling restatement."""
  x = np.array([-3.0, -1.0, -2.0, 50.0])
  w = np.array([True, True, True, False])
  want = np.exp(x[:3]) / np.exp(x[:3]).sum()            # shift 0 == plain softmax
  got = o_bev.masked_softmax(x, w)
  np.testing.assert_allclose(got[:3], want, rtol=1e-14)
  assert got[3] == 0.0
  # a positive maximum shifts by it (no overflow at large scores)
  x2 = np.array([700.0, 699.0, -5.0])
  got2 = o_bev.masked_softmax(x2, np.array([True, True, False]))
  np.testing.assert_allclose(got2[:2], [1 / (1 + np.exp(-1.0)), np.exp(-1.0) / (1 + np.exp(-1.0))], rtol=1e-14)
  # the same rule inside pool_multiview_features: weights of two all-negative scores
  feats = np.array([[[1.0], [2.0]]])
  stats, _ = o_lift.pool_multiview_features(feats, np.array([[True, True]]), np.array([[-2.0, -1.0]]), False, False)
  wgt = np.exp([-2.0, -1.0]) / np.exp([-2.0, -1.0]).sum()
  np.testing.assert_allclose(stats[0], [wgt[0] * 1 + wgt[1] * 2, -1.0], rtol=1e-14)
