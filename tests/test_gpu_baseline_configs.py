"""`-m gpu` tests at the EXACT shapes of BASELINE.json's configs (C1 .. C5).

C1 and one whole C2 scene go through the numpy oracle end to end (the oracle needs ~1 s / ~25 s
on the GPU box's host cores); C3 / C4 / C5 are too big for it and are checked through known
answers and size-independent properties.  Every conv / dense engine the bench can select is
held to the same bar: feature maps within 1e-3 (north star), voxel validity equal to the oracle's
except on visibility boundaries, pose argmax identical.
"""
import copy
import math
import os
import sys

import numpy as np
import pytest
import torch

import helpers
from snap_amd import ops
from snap_amd.configs import train_localization
from snap_amd.data import synthetic
from snap_amd.models import bev_localizer
from snap_amd.models import pose_estimation
from snap_amd.models import pose_exhaustive_voting as pev
from snap_amd.models import types
from snap_amd.utils import geometry
from snap_amd.utils import grids

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ENGINES = ['f32', 'bf16x6', 'bf16x3']


# ------------------------------------------------------------------------------------------
# C1: single scene, 1 StreetView view @256 px + aerial tile, 64 x 64 BEV, tiny ResNet
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('math', ENGINES)
def test_c1_exact_shape_end_to_end_vs_oracle(math):
  import test_gpu_model as tgm
  cfg = helpers.tiny_localizer_config(top_k=1, feature_dim=32, matching_dim=8, depth=(1, 1), width=0.5,   # (GroupNorm(32): base width >= 32)
                                      num_pose_samples=256, retries=2)
  cfg.bev_mapper.scene_z_height = 12.0          # 60 height levels, as the paper's grid
  cfg.bev_mapper.scene_z_offset = 4.0
  cfg.query_frustum_depth = 6.4
  pred, ref, ob, meta = tgm._run(cfg, 1, 1, (256, 256), seed=2, math=math, extent=(12.8, 12.8, 12),
                                 want_batch=True)
  sv, rsv = pred['map']['streetview'], ref['map']['streetview']
  assert sv['feature_volume'].features.shape == (1, 64, 64, 60, 32)
  assert sv['image_feature_pyramid'].features[-1].shape[-3:-1] == (64, 64)     # 256 px / stride 4
  helpers.report('C1 image features', sv['image_feature_pyramid'].features[-1],
                 rsv['image_feature_pyramid']['features'][-1], atol=1e-3)
  tgm._check_validity('C1 map voxel validity', pred['map'], ref['map'], ob['map'], cfg)
  tgm._check_validity('C1 query voxel validity', pred['query'], ref['query'], ob['query'], cfg)
  both = sv['feature_volume'].valid.cpu().numpy() == rsv['feature_volume']['valid']
  helpers.report('C1 feature volume', sv['feature_volume'].features.cpu().numpy()[both],
                 rsv['feature_volume']['features'][both], atol=1e-3)
  helpers.report('C1 aerial plane', pred['map']['aerial']['feature_plane'].features,
                 ref['map']['aerial']['feature_plane']['features'], atol=1e-3)
  for side in ('map', 'query'):
    helpers.report(f'C1 {side} bev_matching', pred[side]['bev_matching'].features,
                   ref[side]['bev_matching']['features'], atol=1e-3)
    assert np.array_equal(pred[side]['bev_matching'].valid.cpu().numpy(), ref[side]['bev_matching']['valid'])
  helpers.report('C1 scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  assert helpers.assert_same_argmax('C1 best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                                    got_index=pred['best_index']) == 0   # no near-tie escape


# ------------------------------------------------------------------------------------------
# C2: one WHOLE scene of the headline workload through the oracle (tools/fullsize_parity.py)
# ------------------------------------------------------------------------------------------
def test_c2_whole_scene_vs_oracle_every_engine():
  """4 views @512 px + aerial, 128 x 128 x 60 voxels, ResNet-50 encoders, 10 000 hypotheses:
  ONE oracle run (~25 s on the GPU box) checks all three engines on the same hypotheses."""
  import fullsize_parity
  import test_gpu_model as tgm
  engines = ENGINES + ['bf16x3+plane']      # + the fused fusion-MLP / max-pool kernel (bench default)
  res = fullsize_parity.run(engines, views=4, image=512)
  ref, ob, cfg = res['ref'], res['oracle_batch'], res['cfg']
  # the projection Dense read the 544 -> 512 crop through the row list at this geometry (no 268 MB copy)
  assert all(v == 'rows' for v in res['projection_path'].values()), res['projection_path']
  for m in engines:
    r = res['per_math'][m]
    print(f'[parity] C2 scene, engine {m}: ' + ', '.join(
        f'{k}={v:.2e}' for k, v in r.items() if isinstance(v, float)))
    assert r['image_features_rel_err'] <= 1e-3, (m, r)
    assert (r['feature_volume_rel_err'] is None) == m.endswith('+plane'), (m, r)
    assert m.endswith('+plane') or r['feature_volume_rel_err'] <= 1e-3, (m, r)
    assert r['streetview_plane_rel_err'] <= 1e-3 and r['streetview_plane_valid_equal'], (m, r)
    assert r['aerial_plane_rel_err'] <= 1e-3, (m, r)
    assert r['map_bev_matching_max_abs_err'] <= 1e-3, (m, r)
    assert r['query_bev_matching_max_abs_err'] <= 1e-3, (m, r)
    assert r['scores_poses_rel_err'] <= 1e-3, (m, r)
    assert r['pose_argmax_equal'], (m, r)          # exact: the near-tie escape is not available here
    pred = res['pred'][m]
    tgm._check_validity(f'C2 map voxel validity [{m}]', pred['map'], ref['map'], ob['map'], cfg)
    tgm._check_validity(f'C2 query voxel validity [{m}]', pred['query'], ref['query'], ob['query'], cfg)


def test_c2_full_batch_lift_inside_the_consumer_is_bitwise_the_rows_through_memory_path():
  """The bench's own workload (8 scenes, 4 views @512 px + aerial, 128 x 128 x 60 voxels, R50, plane-only
  mode): tap records + the in-kernel gather (the default) against pooled rows through memory
  (``Tuning(LIFT_IN_CONSUMER=False)``) -- StreetView planes, matching features, similarity-derived scores and
  the pose argmax are the SAME BITS for the map and the query branch (6.8 M + 2.2 M voxel rows)."""
  import bench
  from snap_amd import ops
  loc, cfg, meta, variables, batch = bench.build('c2', torch.device(DEV), 0, materialize_volume=False)
  loc.engine = 'bf16x3'
  preds = {}
  for consumer in (True, False):
    with ops.tuning_scope(LIFT_IN_CONSUMER=consumer):
      preds[consumer] = loc.apply(variables, batch, train=False, rngs={'sampling': 3})
    torch.cuda.synchronize()
  a, b = preds[True], preds[False]
  for side in ('map', 'query'):
    pa, pb = a[side]['streetview']['feature_plane'], b[side]['streetview']['feature_plane']
    assert torch.equal(pa.valid, pb.valid) and 0 < int(pa.valid.sum())
    assert torch.equal(pa.features, pb.features), (side, float((pa.features - pb.features).abs().max()))
    assert torch.equal(a[side]['bev_matching'].features, b[side]['bev_matching'].features)
  assert torch.equal(a['scores_poses'], b['scores_poses']) and torch.equal(a['best_index'], b['best_index'])


# ------------------------------------------------------------------------------------------
# C3: the train_localization default model at its full size, 4 scenes per GPU
# ------------------------------------------------------------------------------------------
def _clone_tree(t):
  return {k: _clone_tree(v) for k, v in t.items()} if isinstance(t, dict) else t.clone()


def test_c3_full_size_train_step():
  """fwd + bwd + Adam on 4 scenes (4 views @512 px + aerial, 128 x 128 x 60 voxels, R50):
  finite loss and gradients, the update moves every parameter group, a repeated step from the
  same state is BITWISE the same step -- loss, gradient norm, every first Adam moment (= 0.1 x the
  gradient, so every gradient) and every updated parameter: no sum of the training path depends
  on execution order --, and the bf16-operand precision tracks the f32 step.  Then the reference's
  LITERAL train configuration (``config.dtype_str = 'float16'``, train_localization.py:93): a model
  built with ``dtype=torch.float16`` (the dtype selects the IEEE-half engine) stepped under
  ``DynamicScale(minimum_scale=256)`` exactly as ``trainer.py:387-397`` pairs them -- finite, within
  the bf16 bounds of the f32 step, the scale unchanged after a finite step, bitwise repeatable."""
  from snap_amd import models, trainer
  cfg = train_localization.get_config().model
  meta = synthetic.meta_data(0.2, (25.6, 25.6, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  dtype16, ds16 = trainer.dtype_and_dynamic_scale(train_localization.get_config().dtype_str)
  assert dtype16 == torch.float16 and ds16 is not None and ds16.minimum_scale == 256
  model16 = models.get_model('bev_localizer')(cfg, meta, dtype16)
  assert model16.engine == 'fp16' and model.engine is None
  params0 = model.flax_model.init(0, device=DEV)['params']
  batch = synthetic.make_batch(4, meta['grid'], 4, (512, 512), seed=31, device=DEV)
  tcfg = train_localization.get_config()
  lr_fn = trainer.make_lr_fn(tcfg.lr_configs['base_learning_rate'], tcfg.num_training_steps)
  out = {}
  for tag, prec in (('a', 'f32'), ('b', 'f32'), ('bf16', 'bf16')):
    state = trainer.TrainState.create(_clone_tree(params0), rng=5)
    state, reduced, logs = trainer.train_step(state, batch, model=model, lr_fn=lr_fn, precision=prec)
    torch.cuda.synchronize()
    out[tag] = (logs, reduced, state)
    assert logs['is_finite'] and math.isfinite(logs['loss']) and math.isfinite(logs['l2_grads'])
    assert logs['l2_grads'] > 0
  la, lb, lh = out['a'][0], out['b'][0], out['bf16'][0]
  assert la['loss'] == lb['loss'], (la['loss'], lb['loss'])
  assert la['l2_grads'] == lb['l2_grads'], (la['l2_grads'], lb['l2_grads'])
  names = [n for n, _ in trainer.flatten_params(params0)]
  differ = [n for n, ma, mb in zip(names, out['a'][2].m, out['b'][2].m) if not torch.equal(ma, mb)]
  assert not differ, f'{len(differ)} of {len(names)} gradients differ between two identical steps: {differ[:5]}'
  pa, pb = dict(trainer.flatten_params(out['a'][2].params)), dict(trainer.flatten_params(out['b'][2].params))
  assert all(torch.equal(pa[n], pb[n]) for n in names)
  assert abs(lh['loss'] - la['loss']) <= 2e-2 * abs(la['loss'])
  assert abs(lh['l2_grads'] - la['l2_grads']) <= 0.15 * la['l2_grads']
  assert out['a'][2].global_step == 1 and out['a'][2].opt_count == 1
  new = dict(trainer.flatten_params(out['a'][2].params))
  old = dict(trainer.flatten_params(params0))
  moved = {n: bool((new[n] != old[n]).any()) for n in new}
  for key in ('bev_mapper/streetview_encoder/image_encoder/encoder/root_block/conv_root/kernel',
              'bev_mapper/streetview_encoder/fusion_mlp/Dense_0/kernel',
              'bev_mapper/aerial_encoder/encoder/conv_root/kernel',
              'bev_mapper/matching_proj/kernel', 'temperature'):
    assert moved[key], key
  assert 'loss/total' in out['a'][1] and math.isfinite(out['a'][1]['loss/total'])
  # float16 + DynamicScale, twice from the same state (no precision= keyword: the model's dtype decides)
  h = {}
  for tag in ('h1', 'h2'):
    state = trainer.TrainState.create(_clone_tree(params0), rng=5, dynamic_scale=ds16)
    state, reduced, logs = trainer.train_step(state, batch, model=model16, lr_fn=lr_fn)
    torch.cuda.synchronize()
    h[tag] = (logs, state)
    assert logs['is_finite'] and math.isfinite(logs['loss']) and logs['l2_grads'] > 0
    assert logs['loss_scale'] == ds16.scale and state.dynamic_scale.fin_steps == 1   # a finite step keeps the scale
    assert state.opt_count == 1
  l16 = h['h1'][0]
  print(f"[C3 fp16] loss {l16['loss']:.6f} (f32 {la['loss']:.6f})  |g| {l16['l2_grads']:.5f} (f32 {la['l2_grads']:.5f})"
        f"  scale {l16['loss_scale']}")
  assert abs(l16['loss'] - la['loss']) <= 2e-2 * abs(la['loss'])
  assert abs(l16['l2_grads'] - la['l2_grads']) <= 0.15 * la['l2_grads']
  assert h['h1'][0]['loss'] == h['h2'][0]['loss'] and h['h1'][0]['l2_grads'] == h['h2'][0]['l2_grads']
  assert all(torch.equal(a, b) for a, b in zip(h['h1'][1].m, h['h2'][1].m))


# ------------------------------------------------------------------------------------------
# C4: eval_localization path -- 256 x 256 map, 36 yaw hypotheses, exhaustive (x, y, theta)
# correlation; pose scoring at 256^2 with all 4652 frustum points, 20 001 hypotheses and the
# 41^3 refinement lattice
# ------------------------------------------------------------------------------------------
def _smooth_plane(H, D, seed):
  """Random features smoothed with a 2-cell Gaussian (a unique correlation peak)."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  f = torch.randn(1, D, H, H, generator=g)
  k = torch.exp(-(torch.arange(-6, 7).float() ** 2) / (2 * 2.0 ** 2))
  k = (k / k.sum())
  f = torch.nn.functional.conv2d(f, k.reshape(1, 1, 13, 1).expand(D, 1, 13, 1), padding=(6, 0), groups=D)
  f = torch.nn.functional.conv2d(f, k.reshape(1, 1, 1, 13).expand(D, 1, 1, 13), padding=(0, 6), groups=D)
  return f[0].permute(1, 2, 0).contiguous()


@pytest.mark.parametrize('method', ['fft', 'direct'])
def test_c4_exhaustive_voting_256_identity_and_planted_pose(method):
  _voting_known_answers(256, method=method)


def test_c4_voting_fft_equals_the_direct_form_at_256():
  """C4 size (256^2 map, R = 36, D = 32): the frequency-domain voting against the direct-form GEMM on
  the same planes -- the -inf mask IDENTICAL (the FFT overlap count is rounded to the integer the
  direct form counts), every finite score within 1e-3 (absolute; scores are O(1) means of unit
  features: the direct form's own split-operand rounding is the larger term), argmax equal."""
  H, D, R = 256, 32, 36
  DEV = helpers.DEVICE
  f = _smooth_plane(H, D, seed=47).to(DEV)
  q = _smooth_plane(H, D, seed=48).to(DEV)
  g = torch.Generator().manual_seed(49)
  vm = (torch.rand(H, H, generator=g) > 0.1).to(DEV)
  vq = (torch.rand(H, H, generator=g) > 0.15).to(DEV)
  q = (q * vq[..., None]).contiguous()
  grid = grids.Grid2D((H, H), 0.2)
  a = pev.exhaustive_pose_voting(types.FeaturePlane(q, vq), types.FeaturePlane(f, vm), R, grid, method='fft')
  b = pev.exhaustive_pose_voting(types.FeaturePlane(q, vq), types.FeaturePlane(f, vm), R, grid, method='direct')
  fa, fb = torch.isfinite(a), torch.isfinite(b)
  assert torch.equal(fa, fb), int((fa != fb).sum())
  assert 0 < int(fa.sum()) < fa.numel()
  err = float((a[fa] - b[fb]).abs().max())
  print(f'[voting fft vs direct @256] max |d| {err:.3e}  max |score| {float(b[fb].abs().max()):.3e}')
  assert err < 1e-3
  assert int(torch.argmax(a)) == int(torch.argmax(b))


def test_c4_voting_fft_against_the_float64_numpy_fft_checker():
  """C4 size (256^2 map, R = 36, D = 32, N = 768 = 3 * 4^4 transform points per axis, partial validity
  on both planes): ``method='fft'`` against tests/fft_reference.py -- a float64 numpy / scipy FFT
  correlation that shares nothing with the HIP formulations and is itself pinned to oracle/voting.py
  at the small geometries (tests/test_oracle_pins.py).  -inf mask EXACT, every finite score within
  2e-5 x max |score|, argmax equal."""
  import fft_reference
  H, D, R = 256, 32, 36
  DEV = helpers.DEVICE
  rng = np.random.default_rng(256)
  t = rng.standard_normal((R, H, H, D)).astype(np.float32)
  tv = rng.random((R, H, H)) > 0.2
  t = t * tv[..., None]
  fm = rng.standard_normal((H, H, D)).astype(np.float32)
  vm = rng.random((H, H)) > 0.1
  vm[:90, :70] = False            # (an unobserved corner: placements with too little overlap exist)
  tv[:, 200:, :] = False
  t = t * tv[..., None]
  got = pev.template_matching(*[torch.tensor(a).to(DEV) for a in (t, tv, fm, vm)], method='fft').cpu().numpy()
  want = fft_reference.template_matching_fft64(t, tv, fm, vm)
  assert got.shape == want.shape == (R, 2 * H - 1, 2 * H - 1)
  fw, fg = np.isfinite(want), np.isfinite(got)
  assert (fw == fg).all(), int((fw != fg).sum())
  assert 0 < int((~fw).sum()) < fw.size and (got[~fg] == -np.inf).all()
  scale = float(np.abs(want[fw]).max())
  err = float(np.abs(got[fg] - want[fw]).max())
  print(f'[voting fft vs float64 numpy.fft @256] max |d| {err:.3e}  max |score| {scale:.3e}  masked {int((~fw).sum())}')
  assert err <= 2e-5 * scale
  assert int(np.argmax(np.where(fg, got, -np.inf))) == int(np.argmax(np.where(fw, want, -np.inf)))


def test_c4_voting_matching_dim_64_beyond_the_presplit_engine():
  """matching_dim 64 at 256^2: the shift-stacked template bank (259 x 259 x 64 per filter) is beyond
  the pre-split engine's 32-bit offsets (``snap_conv2d_presplit_supported`` = 0), so the voting must
  take the plain-input launch instead of raising -- and still find the identity peak."""
  from snap_amd import ops
  assert ops.conv2d_presplit_supported((1, 770, 770, 32), (259, 259, 32, 576), 4)
  assert not ops.conv2d_presplit_supported((1, 770, 770, 64), (259, 259, 64, 576), 4)
  _voting_known_answers(256, D=64, planted=False, method='direct')


def test_c4_voting_fft_matching_dim_64():
  """matching_dim 64 = two 32-channel groups accumulated in the frequency domain."""
  _voting_known_answers(256, D=64, planted=False, method='fft')


def _voting_known_answers(H, D=32, R=36, dev=None, planted=True, method=None):
  """H = W = 256, R = 36, D = 32 (3.94e13 direct-form flops per call).  Identity: the map
  against itself peaks at (0, H-1, W-1).  Planted pose: the query is the map rotated by
  rotation index k = 9 (a quarter turn, exact on the grid) and shifted by s = (3, -4) cells
  => argmax (k, H-1+s_x, W-1+s_y)  (the index convention of pose_exhaustive_voting.py:127-149,
  pinned on the oracle in tests/test_oracle_pins.py)."""
  DEV = dev or helpers.DEVICE
  f = _smooth_plane(H, D, seed=41).to(DEV)
  ones = torch.ones(H, H, dtype=torch.bool, device=DEV)
  grid = grids.Grid2D((H, H), 0.2)
  plane = types.FeaturePlane(f, ones)
  s = pev.exhaustive_pose_voting(plane, plane, R, grid, method=method)
  assert s.shape == (R, 2 * H - 1, 2 * H - 1)
  idx = tuple(int(i) for i in np.unravel_index(int(torch.argmax(s)), s.shape))
  assert idx == (0, H - 1, H - 1)
  assert abs(float(s[0, H - 1, H - 1]) - float((f * f).sum() / (H * H))) < 1e-2 * float((f * f).mean() * D)
  assert float(s[0, 0, 0]) == -math.inf
  tf = pev.exhaustive_index_to_tfm(torch.tensor(idx, device=DEV), grid, R)
  c = pev.get_grid_center_transform(grid, DEV)
  centre = c.inv @ tf @ c
  assert abs(float(centre.angle)) < 1e-6 and float(centre.t.abs().max()) <= 0.5 * 0.2 + 1e-6
  del s
  if not planted:
    return
  # planted pose: q(u) = m(T u), T = centre-frame rotation by -2 pi k / R and a shift of s cells;
  # k = 9 is a quarter turn, so the resampling is an exact index permutation: q = rot90 + roll
  k, sx, sy = 9, 3, -4
  xy = grid.index_to_xyz(grid.grid_index(device=DEV).to(torch.float32)).reshape(-1, 2)
  tfq = c @ geometry.Transform2D(torch.tensor(-2 * math.pi * k / R, device=DEV),
                                 torch.tensor([sx * 0.2, sy * 0.2], device=DEV)) @ c.inv
  src = (tfq @ xy) / 0.2 - 0.5                                   # continuous map index of every query cell
  ij = torch.round(src).to(torch.int64)
  assert float((src - ij).abs().max()) < 1e-3                   # (exact permutation)
  inside = ((ij >= 0) & (ij < H)).all(-1)
  q = torch.zeros(H * H, D, device=DEV)
  q[inside] = f[ij[inside, 0], ij[inside, 1]]
  qplane = types.FeaturePlane(q.reshape(H, H, D).contiguous(), inside.reshape(H, H).contiguous())
  s2 = pev.exhaustive_pose_voting(qplane, plane, R, grid, method=method)
  idx2 = tuple(int(i) for i in np.unravel_index(int(torch.argmax(s2)), s2.shape))
  assert idx2 == (k, H - 1 + sx, H - 1 + sy), idx2
  back = pev.exhaustive_tfm_to_index(pev.exhaustive_index_to_tfm(torch.tensor(idx2, device=DEV), grid, R),
                                     grid, R)
  assert torch.allclose(back.to(torch.float32).cpu(), torch.tensor(idx2, dtype=torch.float32), atol=1e-3)


def test_c4_pose_scoring_256_full_point_set_and_lattice():
  """X = Y = 256, Nq = 4652 (the default frustum's FoV-filtered points), P = 20 001 hypotheses and
  the 41^3 = 68 921-pose refinement lattice (pose_estimation.py:168-205): the planted pose wins
  the scoring; refinement started AT the planted pose returns it (offset 0 is a lattice point:
  cell (20, 20, 20)); started one lattice step away it comes back to it exactly."""
  from test_gpu_fullsize import _planted_scene
  B, X, Y, cell, P = 1, 256, 256, 0.2, 20001
  _, _, q_xy_p = bev_localizer.build_query_frustum_grid(cell, 16.0, True, 72.0)
  Nq = q_xy_p.shape[0]
  assert Nq == 4652
  sim, q_xy, gt = _planted_scene(B, Nq, X, Y, cell, seed=43)
  # the planted scene draws its own points; re-plant on the REAL frustum points
  q_xy = q_xy_p[:, 0].to(DEV)[None].contiguous()
  th, t = gt[:, 0], gt[:, 1:]
  cth, sth = torch.cos(th), torch.sin(th)
  tx = (cth[:, None] * q_xy[..., 0] - sth[:, None] * q_xy[..., 1] + t[:, None, 0]) / cell
  ty = (sth[:, None] * q_xy[..., 0] + cth[:, None] * q_xy[..., 1] + t[:, None, 1]) / cell
  ii = (torch.arange(X, device=DEV) + 0.5)
  sim = torch.exp(-((ii[None, None, :, None] - tx[..., None, None]) ** 2
                    + (ii[None, None, None, :] - ty[..., None, None]) ** 2) / 8.0).contiguous()
  g = torch.Generator(device='cpu').manual_seed(44)
  poses = torch.stack([torch.rand(B, P, generator=g) * 2 * math.pi,
                       torch.rand(B, P, generator=g) * X * cell,
                       torch.rand(B, P, generator=g) * Y * cell], -1).to(DEV)
  poses[0, 777] = gt[0]
  vq = torch.ones(B, Nq, dtype=torch.bool, device=DEV)
  mv = torch.ones(B, X, Y, dtype=torch.bool, device=DEV)
  s = ops.pose_score(sim, poses.contiguous(), q_xy, vq, mv, cell)
  assert s.shape == (B, P) and int(ops.argmax_rows(s)[0]) == 777
  assert torch.equal(s, ops.pose_score(sim, poses.contiguous(), q_xy, vq, mv, cell))
  grid = grids.Grid2D((X, Y), cell)
  init = geometry.Transform2D(gt[:, 0].contiguous(), gt[:, 1:].contiguous())
  ref_t, lat = pose_estimation.grid_refinement_batched(init, sim, q_xy, vq, mv, grid, False)
  assert lat.shape == (B, 41, 41, 41)
  # the lattice through ONE WINDOW per point (ops.pose_score_window: what BEVLocalizer runs, given the
  # frustum's largest point norm) -- the same bits as the general kernels, also for an initial pose near
  # the map's corner (windows clipped at two borders) and with a third of the points invalid
  qn = float(q_xy_p.norm(dim=-1).max())
  vq3 = vq.clone(); vq3[:, ::3] = False
  corner = geometry.Transform2D(torch.tensor([2.5], device=DEV), torch.tensor([[1.0, 50.5]], device=DEV))
  for ini, vv in ((init, vq), (corner, vq), (init, vq3)):
    rw, lw = pose_estimation.grid_refinement_batched(ini, sim, q_xy, vv, mv, grid, False, max_point_norm=qn)
    rg, lg = pose_estimation.grid_refinement_batched(ini, sim, q_xy, vv, mv, grid, False)
    assert torch.equal(lw, lg), float((lw - lg).abs().max())
    assert torch.equal(rw.packed(), rg.packed())
  best = tuple(int(i) for i in np.unravel_index(int(torch.argmax(lat[0])), (41, 41, 41)))
  assert best == (20, 20, 20)
  assert abs(float(lat[0, 20, 20, 20]) - float(s[0, 777])) <= 1e-5 * abs(float(s[0, 777]))
  assert torch.allclose(ref_t.packed(), gt, atol=1e-6)
  # start one lattice step (0.2 m along the query's x axis) away: samples = init @ offset, so the
  # offset (-0.2, 0) of cell (20, 19, 20) lands exactly on the planted pose
  off = geometry.Transform2D(torch.zeros(B, device=DEV), torch.tensor([[0.2, 0.0]], device=DEV))
  init2 = init @ off
  ref2, lat2 = pose_estimation.grid_refinement_batched(init2, sim, q_xy, vq, mv, grid, False)
  best2 = tuple(int(i) for i in np.unravel_index(int(torch.argmax(lat2[0])), (41, 41, 41)))
  assert best2 == (20, 19, 20), best2
  assert torch.allclose(ref2.packed(), gt, atol=1e-5)


# ------------------------------------------------------------------------------------------
# C5: ViT-B/16 StreetView encoder + 128 x 128 BEV (no reference ViT exists: build-only)
# ------------------------------------------------------------------------------------------
def test_c5_vit_full_forward():
  import bench
  loc, cfg, meta, variables, batch = bench.build('c5', torch.device(DEV), 0)
  p1 = loc.apply(variables, batch, train=False, rngs={'sampling': 3})
  p2 = loc.apply(variables, batch, train=False, rngs={'sampling': 3})
  torch.cuda.synchronize()
  assert p1['scores_poses'].shape == (4, 10001) and bool(torch.isfinite(p1['scores_poses']).all())
  assert torch.equal(p1['scores_poses'], p2['scores_poses'])
  pyr = p1['map']['streetview']['image_feature_pyramid']
  assert pyr.features[-1].shape == (4, 4, 32, 32, 128)           # 512 px / patch 16
  vol = p1['map']['streetview']['feature_volume']
  assert vol.features.shape == (4, 128, 128, 60, 128)
  assert float(vol.features[~vol.valid].abs().max()) == 0.0
  m = p1['map']['bev_matching']
  nrm = m.features.norm(dim=-1)
  assert torch.allclose(nrm[m.valid], torch.ones_like(nrm[m.valid]), atol=1e-4)
