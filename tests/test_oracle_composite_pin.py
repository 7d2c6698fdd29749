"""Composite pin of the oracle (CPU, float64): ``oracle/model.py`` end to end against
``tests/torch_reference.py`` -- an independent restatement of the same forward on torch's own
primitives (F.conv2d, F.max_pool2d, F.interpolate, F.grid_sample(border) for the lift taps and
the pose scoring, torch.sort for the view selection).  The per-op pins of test_oracle_pins.py
check the oracle's pieces; this one checks their COMPOSITION on one small scene: three views with
top-2 view selection (so both the selection and the multi-view softmax pooling are live), aerial
tile, fisheye cameras with non-zero radial distortion, 32 x 32 x 12 voxels.  Agreement: 1e-9.
"""
import numpy as np
import torch

import helpers
import torch_reference as tr
from oracle import geometry as o_geo
from oracle import grids as o_grids
from oracle import model as o_model
from snap_amd.data import synthetic
from snap_amd.models import bev_localizer


def _close(name, got, want, tol=1e-9):
  got = got.detach().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
  want = np.asarray(want)
  assert got.shape == want.shape, (name, got.shape, want.shape)
  err = float(np.abs(got - want).max())
  scale = max(float(np.abs(want).max()), 1.0)
  print(f'[composite pin] {name}: max |d| = {err:.2e} (range {scale:.2e})')
  assert err <= tol * scale, f'{name}: {err:.3e} > {tol} x {scale:.3e}'


def test_oracle_model_equals_the_torch_restatement_end_to_end():
  cfg = helpers.tiny_localizer_config(top_k=2, feature_dim=32, matching_dim=8, num_pose_samples=32, retries=1)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  params = helpers.params_to_numpy(loc.init(3, device='cpu')['params'], np.float64)
  batch = synthetic.make_batch(1, meta['grid'], 3, (64, 64), seed=21)
  ob = helpers.batch_to_oracle(batch, np.float64)
  for side in ('map', 'query'):                      # live radial distortion (the synthetic cameras have none)
    ob[side]['camera'].k_radial = ob[side]['camera'].k_radial + np.array([0.03, -0.008, 0.002])
  rng = np.random.default_rng(5)
  P = cfg.num_pose_samples
  X, Y = meta['grid'].extent[:2]
  angles = rng.uniform(0, 2 * np.pi, (1, P))
  ts = rng.uniform(0, X * 0.2, (1, P, 2))
  ref = o_model.bev_localizer(params, cfg, {'streetview_hfov_deg': 72.0}, o_grids.Grid2D((X, Y), 0.2), ob,
                              pose_samples=o_geo.Transform2D(angles, ts), keep_sim=True)
  with torch.no_grad():
    got = tr.bev_localizer(params, cfg, 72.0, (X, Y), 0.2, ob, angles, ts)
  m, rm = got['map'][0], ref['map']
  q, rq = got['query'][0], ref['query']
  _close('image features', m['image_features'], rm['streetview']['image_feature_pyramid']['features'][-1][0])
  assert np.array_equal(m['volume_valid'].numpy(), rm['streetview']['feature_volume']['valid'][0])
  assert 0.05 < float(m['volume_valid'].double().mean()) < 0.95          # both outcomes are exercised
  _close('feature volume', m['volume'], rm['streetview']['feature_volume']['features'][0])
  _close('streetview plane', m['sv_plane'], rm['streetview']['feature_plane']['features'][0])
  _close('aerial plane', m['aerial_plane'], rm['aerial']['feature_plane']['features'][0])
  _close('map matching', m['matching'], rm['bev_matching']['features'][0])
  assert np.array_equal(m['valid'].numpy(), rm['bev_matching']['valid'][0])
  assert np.array_equal(q['volume_valid'].numpy(), rq['streetview']['feature_volume']['valid'][0])
  _close('query matching', q['matching'], rq['bev_matching']['features'][0])
  _close('sim_points', got['sim'][0], ref['_sim_points'][0])
  _close('scores_poses', got['scores'][0], ref['scores_poses'][0])
  assert int(torch.argmax(got['scores'][0][1:])) == int(ref['best_index'][0])
