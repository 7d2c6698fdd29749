"""`-m gpu` tests at BASELINE.json's full sizes, through size-independent properties.

The numpy oracle is too slow at these sizes (minutes to hours), so the checks are
properties the domain offers: exact linearity under power-of-two scaling, planted
poses recovered by the argmax, constant fields reproduced by interpolation,
softmax normalisation, run-to-run bit determinism.
"""
import math

import numpy as np
import pytest
import torch

from snap_amd import ops
from snap_amd.configs import train_localization
from snap_amd.data import synthetic
from snap_amd.models import bev_localizer
from snap_amd.models import pose_exhaustive_voting as pev
from snap_amd.models import types
from snap_amd.utils import grids

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _planted_scene(B, Nq, X, Y, cell, seed):
  """sim planes with one smooth peak each, at the image of the query point under a planted pose."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  q_xy = (torch.rand(B, Nq, 2, generator=g) * 2 - 1) * torch.tensor([10.0, 8.0])
  q_xy[..., 1] = q_xy[..., 1].abs()
  theta = torch.rand(B, generator=g) * 2 * math.pi
  t = torch.stack([torch.full((B,), X * cell * 0.45), torch.full((B,), Y * cell * 0.55)], -1)
  c, s = torch.cos(theta), torch.sin(theta)
  tx = (c[:, None] * q_xy[..., 0] - s[:, None] * q_xy[..., 1] + t[:, None, 0]) / cell
  ty = (s[:, None] * q_xy[..., 0] + c[:, None] * q_xy[..., 1] + t[:, None, 1]) / cell
  ii = (torch.arange(X) + 0.5).to(DEV)
  jj = (torch.arange(Y) + 0.5).to(DEV)
  di = (ii[None, None, :] - tx.to(DEV)[..., None]) ** 2
  dj = (jj[None, None, :] - ty.to(DEV)[..., None]) ** 2
  sim = torch.exp(-(di[..., :, None] + dj[..., None, :]) / 8.0).contiguous()
  gt = torch.cat([theta[:, None], t], -1)
  return sim, q_xy.to(DEV).contiguous(), gt.to(DEV)


@pytest.mark.parametrize('X,Y,Nq,P', [(128, 128, 4652, 10001), (256, 256, 600, 20001)])
def test_pose_score_full_size_properties(X, Y, Nq, P):
  """C2 (128^2, LDS-DMA double-buffered path) and C4 (256^2, band-tiled path)."""
  B, cell = 2, 0.2
  sim, q_xy, gt = _planted_scene(B, Nq, X, Y, cell, seed=1)
  g = torch.Generator(device='cpu').manual_seed(2)
  poses = torch.stack([torch.rand(B, P, generator=g) * 2 * math.pi,
                       torch.rand(B, P, generator=g) * X * cell,
                       torch.rand(B, P, generator=g) * Y * cell], -1).to(DEV)
  planted = [1234, P - 1]
  for b in range(B):
    poses[b, planted[b]] = gt[b]
  valid_q = torch.ones(B, Nq, dtype=torch.bool, device=DEV)
  valid_q[:, ::7] = False
  mv = torch.ones(B, X, Y, dtype=torch.bool, device=DEV)
  s1 = ops.pose_score(sim, poses.contiguous(), q_xy, valid_q, mv, cell)
  # planted pose wins
  assert ops.argmax_rows(s1).cpu().tolist() == planted
  # exact linearity under a power-of-two scale (fp32 scaling by 2 is exact)
  s2 = ops.pose_score((sim * 2).contiguous(), poses.contiguous(), q_xy, valid_q, mv, cell)
  big = s1.abs() > 1e-20          # (denormal tap products are not exactly scalable)
  assert torch.equal(s2[big], (s1 * 2)[big])
  assert torch.allclose(s2, s1 * 2, rtol=1e-5, atol=1e-30)
  # additivity over a split of the query points
  va = valid_q.clone(); va[:, : Nq // 2] = False
  vb = valid_q & ~va
  sa = ops.pose_score(sim, poses.contiguous(), q_xy, va, mv, cell)
  sb = ops.pose_score(sim, poses.contiguous(), q_xy, vb, mv, cell)
  assert torch.allclose(sa + sb, s1, rtol=1e-5, atol=1e-4)
  # masking out-of-bounds can only remove (non-negative) contributions
  sm = ops.pose_score(sim, poses.contiguous(), q_xy, valid_q, mv, cell, mask_oob=True)
  assert bool((sm <= s1 + 1e-4).all())
  # bit determinism
  assert torch.equal(s1, ops.pose_score(sim, poses.contiguous(), q_xy, valid_q, mv, cell))


def test_sim_softmax_full_size_properties():
  B, Nq, X, Y, Dm = 2, 4652, 128, 128, 32
  g = torch.Generator(device='cpu').manual_seed(3)
  fq = torch.nn.functional.normalize(torch.randn(B, Nq, Dm, generator=g), dim=-1).to(DEV)
  fm = torch.nn.functional.normalize(torch.randn(B, X, Y, Dm, generator=g), dim=-1).to(DEV)
  nv = torch.tensor([4000.0, 4652.0], device=DEV)
  scale = math.exp(2.0)
  sim, stats, _, rowstats = ops.sim_softmax(fq, fm, scale, True, nv, want_rowstats=True)
  assert bool((sim >= 0).all())
  # spot-check rows against a dense matmul
  rows = [0, 17, 4651]
  ref = torch.relu(torch.einsum('bnd,bijd->bnij', fq[:, rows], fm)) * scale / nv[:, None, None, None]
  assert torch.allclose(sim[:, rows], ref, rtol=1e-5, atol=1e-7)
  # the chunk statistics encode a proper softmax: sum_c s_c exp(m_c - M) == sum exp(x - M)
  x = sim * nv[:, None, None, None]
  M = x.amax((-1, -2))
  T = torch.exp(x - M[..., None, None]).sum((-1, -2))
  assert torch.allclose(rowstats[..., 0], M, rtol=1e-5, atol=1e-6)
  assert torch.allclose(rowstats[..., 1], T, rtol=1e-4)
  # sampling returns in-range, deterministic correspondences
  S = 10000 * 8 * 2
  c1 = ops.ransac_sample(fq, fm, stats, scale, True, S, seed=5)
  c2 = ops.ransac_sample(fq, fm, stats, scale, True, S, seed=5)
  assert torch.equal(c1, c2)
  assert int(c1[..., 0].max()) < Nq and int(c1[..., 1].max()) < X and int(c1[..., 2].max()) < Y
  assert int(c1.min()) >= 0
  # samples concentrate on high-probability cells: mean prob of sampled cells >> uniform
  p = torch.exp(x - M[..., None, None]) / T[..., None, None]
  bi = torch.arange(B, device=DEV)[:, None].expand(B, S)
  ps = p[bi, c1[..., 0].long(), c1[..., 1].long(), c1[..., 2].long()]
  assert float(ps.mean()) > 1.5 / (X * Y)


def test_exhaustive_voting_identity_full_size():
  """128^2 map, R=36, D=32: 2.46 TFLOP of direct correlation on the MFMA engine."""
  H, D, R = 128, 32, 36
  g = torch.Generator(device='cpu').manual_seed(4)
  f = torch.randn(H, H, D, generator=g).to(DEV)
  plane = types.FeaturePlane(f, torch.ones(H, H, dtype=torch.bool, device=DEV))
  s = pev.exhaustive_pose_voting(plane, plane, R, grids.Grid2D((H, H), 0.2))
  assert s.shape == (R, 2 * H - 1, 2 * H - 1)
  idx = np.unravel_index(int(torch.argmax(s)), s.shape)
  assert tuple(int(i) for i in idx) == (0, H - 1, H - 1)
  # the peak equals the mean squared norm of the features (self-correlation / valid count)
  assert abs(float(s[0, H - 1, H - 1]) - float((f * f).sum() / (H * H))) < 1e-2
  # far corners overlap < 5 % of the template: masked to -inf
  assert float(s[0, 0, 0]) == -math.inf


def test_lift_constant_field_full_size():
  """C2 lift: a constant feature image is reproduced exactly (weights sum to 1), variance 0."""
  B, V, h, w, fd, nb = 1, 4, 128, 128, 128, 32
  meta = synthetic.meta_data(0.2, (25.6, 25.6, 12))
  batch = synthetic.make_batch(B, meta['grid'], V, (512, 512), seed=6, device=DEV, with_aerial=False)
  scene = batch['map']
  cam = scene['camera'].scale(torch.tensor([0.25, 0.25], device=DEV)).packed()
  Rt = scene['T_view2scene'].packed()
  f = torch.empty(B, V, h, w, fd + nb, device=DEV)
  f[..., :fd] = 3.25
  f[..., fd:] = 0.5
  xy = meta['grid'].bev().index_to_xyz(meta['grid'].bev().grid_index(device=DEV).float())
  z = torch.arange(60, device=DEV) * 0.2 - 2.0 + 0.1
  pts = torch.cat([xy[:, :, None, :].expand(128, 128, 60, 2),
                   z[None, None, :, None].expand(128, 128, 60, 1)], -1).reshape(1, -1, 3).contiguous()
  pooled, valid = ops.lift_pool(f, cam, Rt, pts, K=0, fisheye=True, feature_dim=fd, num_bins=nb,
                                depth_min_max=(1.0, 32.0))
  assert 0.05 < float(valid.float().mean()) < 0.95
  pv = pooled[valid]
  assert torch.allclose(pv[:, :fd], torch.full_like(pv[:, :fd], 3.25), atol=2e-6)
  assert float(pv[:, fd:2 * fd].abs().max()) < 1e-9
  assert torch.allclose(pv[:, 2 * fd], torch.full_like(pv[:, 2 * fd], 0.5), atol=1e-6)
  assert float(pooled[~valid].abs().max()) == 0.0


def test_c2_forward_is_deterministic_and_sane():
  """End-to-end C2 step (reduced to 2 scenes to bound memory/time)."""
  cfg = train_localization.get_config().model
  meta = synthetic.meta_data(0.2, (25.6, 25.6, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  variables = loc.init(0, device=DEV)
  batch = synthetic.make_batch(2, meta['grid'], 4, (512, 512), seed=7, device=DEV)
  p1 = loc.apply(variables, batch, rngs={'sampling': 9})
  p2 = loc.apply(variables, batch, rngs={'sampling': 9})
  assert torch.equal(p1['scores_poses'], p2['scores_poses'])
  assert torch.equal(p1['map']['bev_matching'].features, p2['map']['bev_matching'].features)
  assert p1['scores_poses'].shape == (2, 10001)
  assert bool(torch.isfinite(p1['scores_poses']).all())
  vol = p1['map']['streetview']['feature_volume']
  assert vol.features.shape == (2, 128, 128, 60, 128)
  assert float(vol.features[~vol.valid].abs().max()) == 0.0
  m = p1['map']['bev_matching']
  nrm = m.features.norm(dim=-1)
  assert torch.allclose(nrm[m.valid], torch.ones_like(nrm[m.valid]), atol=1e-4)
  assert int(p1['best_index'].max()) < 10000 and int(p1['best_index'].min()) >= 0
  p3 = loc.apply(variables, batch, rngs={'sampling': 10})
  assert not torch.equal(p1['map_t_query_samples'].t[:, 1:], p3['map_t_query_samples'].t[:, 1:])


def test_reference_default_shapes_forward():
  """The reference's own default scene (SURVEY REF): 120 x 160 x 60 voxels (24 x 32 x 12 m at
  0.2 m -- a NON-square map), V = 20 views -> top-K = 4 view selection, 4652 frustum points,
  10 001 pose hypotheses x 8 retries; eval config adds the 41^3 refinement lattice."""
  from snap_amd.configs import eval_localization
  cfg = train_localization.get_config().model
  meta = synthetic.meta_data(0.2, (24, 32, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  variables = loc.init(0, device=DEV)
  batch = synthetic.make_batch(1, meta['grid'], 20, (256, 256), seed=11, device=DEV)
  p1 = loc.apply(variables, batch, rngs={'sampling': 3})
  assert p1['scores_poses'].shape == (1, 10001) and bool(torch.isfinite(p1['scores_poses']).all())
  vol = p1['map']['streetview']['feature_volume']
  assert vol.features.shape == (1, 120, 160, 60, 128)
  assert 0.05 < float(vol.valid.float().mean()) < 1.0
  assert float(vol.features[~vol.valid].abs().max()) == 0.0
  assert p1['query']['bev_matching'].features.shape[1] == 4652
  p2 = loc.apply(variables, batch, rngs={'sampling': 3})
  assert torch.equal(p1['scores_poses'], p2['scores_poses'])
  # eval config: more hypotheses + grid refinement around the best pose
  import copy
  ecfg = copy.deepcopy(cfg)
  ecfg.update(eval_localization.get_config().model)      # (the eval config overrides the train model config)
  assert ecfg.num_pose_samples == 20_000 and ecfg.do_grid_refinement
  eloc = bev_localizer.BEVLocalizer(ecfg, meta['build_config'].scene_config, meta['grid'].bev())
  pe = eloc.apply(variables, batch, rngs={'sampling': 3})
  assert pe['scores_grid_refine'].shape == (1, 41, 41, 41)
  assert bool(torch.isfinite(pe['map_t_query'].t).all())
  # refinement can only improve on the RANSAC winner's score (the lattice contains offset 0)
  best_ransac = pe['scores_poses'][:, 1:].max(dim=1).values
  assert bool((pe['scores_grid_refine'].reshape(1, -1).max(dim=1).values >= best_ransac - 1e-4).all())


def test_bf16_engines_full_size_properties():
  """bf16-operand conv / wgrad engines at the fusion-MLP size of the C3 step (1 M voxel rows,
  K = 257 of a 260-wide row, N = 256): scaling the input by a power of two commutes with the
  bf16 rounding and with every f32 product and sum, so the result must scale bit-exactly; the
  weight gradient of an all-ones dy is the (bf16-rounded) column sum of x, checked against
  torch on the rounded values; and a launch is run-to-run deterministic."""
  from snap_amd import ops_bwd
  g = torch.Generator(device=DEV).manual_seed(7)
  M, Cs, K, N = 1 << 20, 260, 257, 256
  x = torch.randn((1, 1, M, Cs), device=DEV, generator=g)
  w = torch.randn((1, 1, K, N), device=DEV, generator=g) / 16
  b = torch.randn((N,), device=DEV, generator=g)
  y1 = ops.conv2d(x, w, cin=K, math='bf16')
  y2 = ops.conv2d(x * 4.0, w, cin=K, math='bf16')
  assert torch.equal(y1 * 4.0, y2)
  assert torch.equal(y1, ops.conv2d(x, w, cin=K, math='bf16'))
  yb = ops.conv2d(x, w, cin=K, bias=b, relu=True, math='bf16')
  assert torch.equal(yb, torch.relu(y1 + b))
  exact = ops.conv2d(x[:, :, :4096], w, cin=K)                       # f32 engine, first rows
  err = (y1[:, :, :4096] - exact).abs().max() / exact.abs().max()
  assert 1e-5 < float(err) < 2e-2, float(err)                         # bf16-class, and really bf16
  dy = torch.ones((1, 1, M, N), device=DEV)
  dw = ops_bwd.conv2d_wgrad(x, dy, (1, 1, K, N), math='bf16')
  want = x[0, 0, :, :K].to(torch.bfloat16).double().sum(0)
  got = dw[0, 0, :, 0].double()
  assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 0.5   # f32 sums of 1M terms
  assert torch.equal(dw[0, 0, :, 0], dw[0, 0, :, N - 1])              # every column sees the same sum
  assert torch.equal(dw, ops_bwd.conv2d_wgrad(x, dy, (1, 1, K, N), math='bf16'))


def test_attention_full_size_properties():
  """ViT-B/16 attention at the C5 size (20 images x 1024 tokens x 12 heads): softmax rows sum to
  one, so a per-head constant V comes back exactly as that constant (whatever Q and K are); a
  one-hot-sharp softmax (one dominant key per query) returns that key's V row; deterministic."""
  g = torch.Generator(device=DEV).manual_seed(9)
  B, N, H, D = 20, 1024, 12, 64
  qkv = torch.randn((B, N, 3, H, D), device=DEV, generator=g)
  const = torch.randn((H, D), device=DEV, generator=g).to(torch.bfloat16).float()   # bf16-exact values
  qkv[:, :, 2] = const
  out = ops.attention(qkv)
  assert torch.equal(out, ops.attention(qkv))
  want = const.reshape(1, 1, H * D).expand(B, N, H * D)
  assert float((out - want).abs().max()) <= 8e-3 * float(const.abs().max())   # P rounded to bf16: sum(P) = 1 +- 2^-8
  # sharp softmax: q_i = 64 * k_perm(i), unit-norm keys -> key perm(i) wins by a wide margin
  k = torch.nn.functional.normalize(torch.randn((B, N, H, D), device=DEV, generator=g), dim=-1)
  perm = torch.randperm(N, device=DEV, generator=g)
  v = torch.randn((B, N, H, D), device=DEV, generator=g)
  qkv2 = torch.stack([k[:, perm] * 512.0, k, v], dim=2).contiguous()
  out2 = ops.attention(qkv2).reshape(B, N, H, D)
  want2 = v[:, perm].to(torch.bfloat16).float()
  assert float((out2 - want2).abs().max()) <= 2e-2 * float(v.abs().max())
