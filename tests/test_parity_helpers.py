"""CPU checks of the parity helpers the `-m gpu` suites rely on (a helper that cannot fail
proves nothing): the visibility-boundary test of voxel-validity mismatches."""
import numpy as np
import pytest

import helpers
from oracle import bev as o_bev
from oracle import grids as o_grids
from oracle import lift as o_lift
from snap_amd.data import synthetic


def _scene():
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=4)
  ob = helpers.batch_to_oracle(batch)['map']
  xyz = o_bev.build_xyz_query({'scene_z_offset': 4.0, 'scene_z_height': 12.0},
                              o_grids.Grid2D(meta['grid'].extent[:2], 0.2), ob['T_view2scene'])
  return ob, xyz


def _validity(ob, xyz, stride):
  cams = ob['camera'].scale((1 / np.asarray(stride, np.float32)[::-1]))
  _, vis, _, _ = o_lift.project_points_to_views(ob['T_view2scene'], cams, xyz.reshape(len(xyz), -1, 3))
  return vis.any(-1).reshape(xyz.shape[:-1])


def test_validity_mismatch_helper_accepts_boundary_flips_and_rejects_real_ones():
  ob, xyz = _scene()
  stride = (4.0, 4.0)
  want = _validity(ob, xyz, stride)
  assert 0.05 < want.mean() < 0.95
  assert helpers.assert_validity_mismatches_on_borders('same', want, want, ob, xyz, stride) == 0
  # a 20 um shift of every voxel flips only voxels that sit on a visibility boundary ...
  near = _validity(ob, xyz + np.float32(2e-5), stride)
  if (near != want).any():
    helpers.assert_validity_mismatches_on_borders('shift 20um', near, want, ob, xyz, stride, tol=5e-3)
  # ... a 15 cm shift flips voxels well inside / outside the frusta: must be rejected
  far = _validity(ob, xyz + np.float32(0.15), stride)
  assert (far != want).sum() > 10
  with pytest.raises(AssertionError):
    helpers.assert_validity_mismatches_on_borders('shift 15cm', far, want, ob, xyz, stride, tol=5e-3)


def test_template_validity_helper_accepts_boundary_flips_and_rejects_real_ones():
  """helpers.assert_template_validity_mismatches_on_borders: a flip is accepted only where the
  float64 rotated coordinate of the cell lies on a decision boundary."""
  from oracle import voting as o_voting
  H, R, cell = 16, 8, 0.25
  rng = np.random.default_rng(0)
  f = rng.standard_normal((H, H, 4)).astype(np.float32)
  v = np.ones((H, H), bool)
  _, tv = o_voting.sample_query_templates(f, v, R, o_grids.Grid2D((H, H), cell))
  assert helpers.assert_template_validity_mismatches_on_borders('same', tv, tv, cell) == 0
  # rotation 0 maps every cell centre onto itself: u - 0.5 is integral everywhere, i.e. every cell
  # of template 0 sits on a tap-pair boundary -- a flip there is a boundary flip ...
  flip = tv.copy()
  flip[0, 3, 5] = ~flip[0, 3, 5]
  assert helpers.assert_template_validity_mismatches_on_borders('boundary', flip, tv, cell) == 1
  # ... template 1 (45 degrees): an interior cell is far from every boundary -- rejected
  bad = tv.copy()
  bad[1, 8, 7] = ~bad[1, 8, 7]
  with pytest.raises(AssertionError):
    helpers.assert_template_validity_mismatches_on_borders('interior', bad, tv, cell)
