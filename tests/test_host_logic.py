"""CPU tests of the host side: configs, struct maths, the C-ABI surface, the module
API (driven through the test-only oracle backend) and failure behaviour."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import helpers
from oracle import geometry as o_geo
from oracle import grids as o_grids
from oracle import model as o_model
from oracle import pose as o_pose
from oracle import voting as o_voting
from snap_amd import _lib
from snap_amd import models
from snap_amd import ops
from snap_amd.configs import defaults
from snap_amd.configs import eval_localization
from snap_amd.configs import train_localization
from snap_amd.data import synthetic
from snap_amd.models import bev_localizer
from snap_amd.models import pose_exhaustive_voting as pev
from snap_amd.utils import geometry
from snap_amd.utils import grids

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# -- C ABI ---------------------------------------------------------------------------
def _header_symbols():
  text = open(os.path.join(ROOT, 'include', 'snap_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return set(re.findall(r'\b(snap_[a-z0-9_]+)\s*\(', text))


def test_abi_header_and_binding_agree():
  assert _header_symbols() == set(_lib.SIGNATURES)


def test_abi_library_loads_and_exports_every_symbol():
  """No compute calls: only dlopen + dlsym + the introspection entry points."""
  lib = _lib.load()
  raw = ctypes.CDLL(_lib.LIB_PATH)
  for name in _header_symbols():
    assert hasattr(raw, name), f'{name} declared in snap_hip.h but not exported'
  # ... and the reverse inclusion: nothing named snap_* leaves the library undeclared
  import subprocess
  nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
  exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith('snap_')}
  assert exported == _header_symbols(), sorted(exported ^ _header_symbols())
  assert lib.snap_abi_version() == _lib.ABI_VERSION
  assert lib.snap_build_arch() == b'gfx950'
  assert lib.snap_status_string(0) == b'ok'
  assert b'shape' in lib.snap_status_string(-1)


def test_ops_fail_loudly_without_gpu():
  x = torch.zeros(1, 4, 4, 8)
  w = torch.zeros(1, 1, 8, 8)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ops.conv2d(x, w)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ops.vertical_pool(torch.zeros(2, 3, 8), torch.ones(2, 3, dtype=torch.bool))
  with pytest.raises(TypeError):
    ops.conv2d(np.zeros((1, 4, 4, 8), np.float32), w)


def test_missing_library_raises(monkeypatch):
  monkeypatch.setattr(_lib, '_lib', None)
  monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsnap_hip.so')
  with pytest.raises(RuntimeError, match='not built'):
    _lib.load()


def test_entry_point_argument_validation_codes():
  """Bad descriptors are rejected with status codes before any launch (CPU-safe)."""
  lib = _lib.load()
  d = _lib.SnapConvDesc(1, 4, 4, 8, 8, 1, 1, 1, 0, 0, 4, 4, 6, 6, 0, 0, 1.0, 0.0)  # Cout % 4 != 0
  one = ctypes.c_void_p(16)
  assert lib.snap_conv2d_nhwc_f32(ctypes.byref(d), one, one, one, None, None, None, None, None,
                                  None, None, None) == -1
  assert lib.snap_conv2d_nhwc_f32(ctypes.byref(d), None, one, one, None, None, None, None, None,
                                  None, None, None) == -3
  assert lib.snap_vertical_pool_f32(one, one, one, one, 4, 3, 6, 0, None) == -1   # D % 4
  assert lib.snap_vertical_pool_f32(one, one, one, one, 4, 3, 8, 7, None) == -2   # pooling
  assert lib.snap_pose_score_f32(one, one, one, one, None, 1, 4, 8, 8, 10, 0.2, 0, one, one, 0,
                                 None) == -5                                      # workspace
  # round-6 entry points: tap records need classed, pre-split, valid-only rows; the gather needs whole
  # 16-channel slabs of mean | var | score; the scoring window must fit its LDS stage
  dl = _lib.SnapLiftDesc(1, 1, 8, 8, 160, 128, 32, 64, 0, 0, 272, 1.0, 32.0, -1.0, 1, 1, 0)
  dl.out_split, dl.valid_rows_only, dl.class_rows = 1, 1, 0
  assert lib.snap_lift_pool_records_f32(ctypes.byref(dl), one, one, one, one, one, one, one, None) == -2
  assert lib.snap_lift_pool_records_f32(ctypes.byref(dl), one, one, one, one, one, one, None, None) == -3
  wb = lib.snap_conv2d_packed_weights_split_bytes(1, 257, 256, 2), lib.snap_conv2d_packed_weights_split_bytes(1, 256, 128, 2)
  args = lambda fd, cin: (one, 1024, cin, 272, one, one, one, one, one, 4096, 8, 160, fd, one, 0, one, wb[0], one, 256,
                          one, wb[1], one, 128, 4, 256, one, one, None)
  assert lib.snap_mlp2_pool_max_f32(one, 1024, 257, 272, one, one, one, wb[0], one, 256, one, wb[1], one, 128, 0, 7, 4,
                                    256, one, one, None) == -2                       # x_split 7 is not a public mode
  assert lib.snap_mlp2_pool_max_gather_f32(*args(120, 257)) == -1                    # feature_dim % 16
  assert lib.snap_mlp2_pool_max_gather_f32(*args(128, 260)) == -1                    # Cin != 2 fd + 1
  assert lib.snap_pose_score_window_supported(256, 256, 39) == 1
  assert lib.snap_pose_score_window_supported(256, 254, 39) == 0                     # Y % 4
  assert lib.snap_pose_score_window_supported(2048, 2048, 400) == 0                  # beyond the LDS stage
  assert lib.snap_pose_score_window_f32(one, one, one, 400, one, one, 1, 4, 2048, 2048, 10, 0.2, one, one, 1 << 30,
                                        None) == -2


# -- configs ----------------------------------------------------------------------------
def test_conv_dispatch_queries_follow_the_documented_rules():
  """`snap_conv2d_stationary_kind` / `snap_conv2d_tile_rows_ex` / `snap_conv2d_gn_partial_bytes_ex` are pure
  host functions of the descriptor (no launch, no device): the kernel a layer gets and the layout of
  the GroupNorm partial sums it emits, on the C2 layer shapes."""
  lib = _lib.load()

  def query(N, H, W, Cin, Cout, K=1, res=False, mode=0):
    pad = 1 if K == 3 else 0
    d = _lib.SnapConvDesc(N=N, H=H, W=W, Cin=Cin, Cin_stride=Cin, KH=K, KW=K, stride=1, pad_t=pad, pad_l=pad,
                          Ho=H, Wo=W, Cout=Cout, Cout_stride=Cout, prologue=ops.PRO_GN_RELU,
                          epilogue=ops.EPI_RESIDUAL if res else 0, in_scale=1.0, in_shift=0.0,
                          tile_hint=1000000 * mode)
    return (int(lib.snap_conv2d_stationary_kind(ctypes.byref(d), 2)),
            int(lib.snap_conv2d_tile_rows_ex(ctypes.byref(d), 2)),
            int(lib.snap_conv2d_gn_partial_bytes_ex(ctypes.byref(d), 2)))

  # StreetView stage-1 expansion: weight panel resident in LDS, statistics per 32-row slab
  assert query(40, 136, 136, 64, 256, res=True) == (2, 32, 40 * (136 * 136 // 32 + 2) * 256 * 8)
  # stage-3 expansion (Cin = 256: no panel fits): activation tile in registers, the tiled engine's 128-row slabs
  assert query(40, 34, 34, 256, 1024, res=True) == (1, 128, 40 * (34 * 34 // 128 + 2) * 1024 * 8)
  # stage-1 3 x 3: one slab per 30-pixel tile of an image row, all of them live (negative = slabs per image)
  assert query(40, 136, 136, 64, 64, K=3) == (3, -(136 * 5), 40 * 136 * 5 * 64 * 8)
  # below 40 000 rows (aerial encoder), K >= 512, the reductions, narrower 3 x 3 layers: the tiled bodies
  for shape in ((8, 68, 68, 128, 512, 1, True), (40, 17, 17, 512, 2048, 1, True), (40, 136, 136, 256, 64),
                (40, 68, 68, 128, 128, 3)):
    assert query(*shape)[:2] == (0, 128), shape
  # tile_hint modes: 1 = tiled bodies only, 2 = stationary kernels below the row threshold too, 3 = no panel kernels
  assert query(40, 136, 136, 64, 256, res=True, mode=1)[:2] == (0, 128)
  assert query(8, 68, 68, 128, 512, res=True, mode=2)[:2] == (2, 32)
  assert query(40, 136, 136, 64, 256, res=True, mode=3)[0] == 1
  assert query(40, 136, 136, 64, 64, K=3, mode=3)[0] == 0


def test_lazy_feature_volume_is_a_feature_volume():
  """types.LazyFeatureVolume (plane-only mode of the StreetView encoder): a FeatureVolume by
  ``isinstance``, lazy until ``features`` is read, ``dataclasses.replace`` works (and materialises),
  ``discard()`` releases the captured encoder inputs."""
  import dataclasses
  from snap_amd.models import types
  calls = []
  v = types.LazyFeatureVolume(lambda: calls.append(1) or 'F', valid='V')
  assert isinstance(v, types.FeatureVolume) and not v.materialized
  w = v.replace(valid='W')
  assert not w.materialized and w.valid == 'W' and calls == []
  r = dataclasses.replace(v, valid='X')
  assert r.features == 'F' and r.valid == 'X' and isinstance(r, types.FeatureVolume)
  assert v.materialized and calls == [1]
  assert v.replace(features='G').features == 'G'
  d = types.LazyFeatureVolume(lambda: 1 / 0, valid=1).discard()
  assert d.features is None and d.valid == 1


def test_presplit_engine_support_query():
  """`snap_conv2d_presplit_supported` (host-only): the exhaustive voting asks it before it pre-splits
  the map.  C4 (256^2, matching_dim 32, shift-stacked by 4) fits; matching_dim 64 at 256^2 and a
  ~360^2 query do not (32-bit offsets of the weight column tile / the input window): those keep the
  plain-input launch (pose_exhaustive_voting._correlate)."""
  assert ops.conv2d_presplit_supported((1, 770, 770, 32), (259, 259, 32, 576), 4)
  assert not ops.conv2d_presplit_supported((1, 770, 770, 64), (259, 259, 64, 576), 4)
  assert not ops.conv2d_presplit_supported((1, 1082, 1082, 32), (367, 367, 32, 576), 4)
  assert not ops.conv2d_presplit_supported((1, 64, 64, 24), (3, 3, 24, 64))       # Cin % 16
  assert ops.conv2d_presplit_supported((8, 34, 34, 256), (3, 3, 256, 256), 1, ((1, 1), (1, 1)))


def test_default_config_values_match_reference_defaults():
  c = defaults.bev_localizer()
  assert (c.mask_score_out_of_bounds, c.clip_negative_scores, c.add_temperature) == (False, True, True)
  assert c.init_temperature == 2.0 and c.query_frustum_depth == 16.0
  assert c.num_pose_samples is None and c.num_pose_sampling_retries == 1
  m = c.bev_mapper
  assert (m.scene_z_offset, m.scene_z_height, m.matching_dim) == (4.0, 12.0, 32)
  assert tuple(m.scene_z_offset_range) == (-2, 2)
  assert m.pooling.pooling == 'max' and m.modality_fusion.pooling == 'max'
  sv = m.streetview_encoder
  assert (sv.feature_dim, sv.num_scale_bins, sv.top_k_view_selection) == (128, 32, 4)
  assert tuple(sv.depth_min_max) == (1.0, 32.0) and tuple(sv.fusion.layers) == (256, 128)
  assert sv.proj_mlp.apply_input_activation and sv.do_weighted_fusion
  assert sv.image_encoder.encoder.depth == 50 and not sv.image_encoder.encoder.skip_root_block
  assert m.aerial_encoder.encoder.skip_root_block
  t = train_localization.get_config()
  assert t.model.num_pose_samples == 10_000 and t.model.num_pose_sampling_retries == 8
  assert t.model.filter_points_in_fov
  e = eval_localization.get_config()
  assert e.model.num_pose_samples == 20_000 and e.model.do_grid_refinement
  with pytest.raises(AttributeError):
    c.not_a_key = 1          # locked, like ml_collections
  with pytest.raises(ValueError):
    defaults.parse_argument_string('foo=1')
  assert defaults.parse_argument_string('image_encoder=R101')['image_encoder'] == 'R101'
  assert defaults.resnet('R152x2').width == 2


# -- struct maths vs oracle ----------------------------------------------------------------
def test_torch_geometry_matches_oracle():
  rng = np.random.default_rng(0)
  ang = rng.uniform(-3, 3, (2, 5))
  tt = rng.standard_normal((2, 5, 2))
  a = geometry.Transform2D(torch.tensor(ang), torch.tensor(tt))
  oa = o_geo.Transform2D(ang, tt)
  pts = rng.standard_normal((2, 5, 7, 2))
  np.testing.assert_allclose((a @ torch.tensor(pts)).numpy(), oa @ pts, atol=1e-12)
  np.testing.assert_allclose((a @ a.inv).t.numpy(), 0, atol=1e-12)
  np.testing.assert_allclose(a.inv.t.numpy(), oa.inv.t, atol=1e-12)
  dr, dt = a.magnitude()
  odr, odt = oa.magnitude()
  np.testing.assert_allclose(dr.numpy(), odr)
  np.testing.assert_allclose(dt.numpy(), odt)
  assert a[0].shape == (5,) and a[:, None].shape == (2, 1, 5)
  assert geometry.Transform2D.from_packed(a.packed()).t.equal(a.t)
  cam = geometry.FisheyeCamera(
      torch.tensor([[100.0, 80.0]]), torch.tensor([[60.0, 55.0]]), torch.tensor([[50.0, 40.0]]),
      torch.tensor([[0.05, -0.01, 0.002]]), torch.tensor([2.0]))
  ocam = o_geo.FisheyeCamera(*(t.numpy() for t in (cam.wh, cam.f, cam.c, cam.k_radial, cam.max_fov)))
  p3 = rng.standard_normal((1, 50, 3)) * np.array([2, 2, 4]) + np.array([0, 0, 2])
  uv, ok = cam.world2image(torch.tensor(p3))
  ouv, ook = ocam.world2image(p3)
  np.testing.assert_allclose(uv.numpy(), ouv, atol=1e-9)
  assert (ok.numpy() == ook).all()
  assert cam.packed().shape == (1, 11)
  sc = cam.scale(torch.tensor([0.25, 0.5]))
  np.testing.assert_allclose(sc.f.numpy(), [[15.0, 27.5]])


def test_grids_and_frustum_match_oracle():
  g = grids.Grid3D.from_extent_meters((24, 32, 12), 0.2)
  assert g.extent == (120, 160, 60) and g.bev().extent == (120, 160)
  with pytest.raises(ValueError):
    grids.Grid2D.from_extent_meters((1.0, 1.05), 0.2)
  idx = g.bev().grid_index()
  assert idx.shape == (120, 160, 2)
  np.testing.assert_allclose(
      g.bev().index_to_xyz(idx.float()).numpy(),
      o_grids.Grid2D((120, 160), 0.2).index_to_xyz(o_grids.Grid2D((120, 160), 0.2).grid_index()),
      atol=1e-5)
  grid, p, q = bev_localizer.build_query_frustum_grid(0.2, 16.0, True, 72.0)
  og, op, oq = o_pose.build_query_frustum_grid(0.2, 16.0, True, 72.0)
  assert grid.extent == og.extent == (120, 80)
  np.testing.assert_allclose(q.numpy(), oq, atol=1e-6)


def test_exhaustive_index_helpers_match_oracle():
  g = grids.Grid2D((48, 48), 0.5)
  og = o_grids.Grid2D((48, 48), 0.5)
  idx = torch.tensor([5, 50, 43])
  tf = pev.exhaustive_index_to_tfm(idx, g, 36)
  otf = o_voting.exhaustive_index_to_tfm(idx.numpy(), og, 36)
  np.testing.assert_allclose(tf.angle.numpy(), otf.angle, atol=1e-6)
  np.testing.assert_allclose(tf.t.numpy(), otf.t, atol=1e-5)
  back = pev.exhaustive_tfm_to_index(tf, g, 36)
  np.testing.assert_allclose(back.numpy(), idx.numpy(), atol=1e-4)


# -- module API through the oracle backend ---------------------------------------------------
def _tiny(refine=False, **kw):
  cfg = helpers.tiny_localizer_config(refine=refine, **kw)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  return cfg, meta, model


def test_model_registry_and_base_model_contract():
  cfg, meta, model = _tiny()
  assert isinstance(model.flax_model, bev_localizer.BEVLocalizer)
  assert model.default_flax_model_config().bev_mapper.matching_dim == 32
  assert models.get_model('semantic_net').__name__ == 'SemanticNetModel'
  with pytest.raises(KeyError):
    models.get_model('occupancy_net')


def test_param_tree_has_flax_layout():
  cfg, meta, model = _tiny()
  p = model.flax_model.init(0, device='cpu')['params']
  assert set(p) == {'bev_mapper', 'temperature'}
  bm = p['bev_mapper']
  assert set(bm) == {'streetview_encoder', 'vertical_pooling', 'aerial_encoder',
                     'modality_fusion', 'matching_proj'}
  sv = bm['streetview_encoder']
  assert set(sv) == {'image_encoder', 'proj_mlp', 'fusion_mlp'}
  enc = sv['image_encoder']['encoder']
  assert enc['root_block']['conv_root']['kernel'].shape == (7, 7, 3, 32)      # HWIO
  u = enc['block1']['unit01']
  assert set(u) == {'gn1', 'conv1', 'gn2', 'conv2', 'gn3', 'conv3', 'conv_proj'}
  assert u['gn1']['scale'].shape == (1, 1, 1, 32)
  assert u['conv2']['kernel'].shape == (3, 3, 32, 32)
  assert 'conv_root' in bm['aerial_encoder']['encoder']                        # skip_root_block
  dec = sv['image_encoder']['decoder']
  assert set(dec) == {'0_skip_norm', '0_skip_conv', '1_skip_norm', '1_skip_conv'}
  assert sv['fusion_mlp']['Dense_0']['kernel'].shape == (65, 64)               # (in, out)
  assert sv['proj_mlp']['Dense_0']['kernel'].shape == (32, 40)
  assert bm['matching_proj']['kernel'].shape == (32, 8)
  assert float(p['temperature']) == 2.0


def test_forward_pytree_and_oracle_agreement(oracle_backend):
  cfg, meta, model = _tiny(refine=True, num_pose_samples=24)
  loc = model.flax_model
  variables = loc.init({'params': 0, 'sampling': 1}, device='cpu')
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=1)
  pred, state = loc.apply(variables, batch, train=False, rngs={'sampling': 3}, mutable=['batch_stats'],
                          debug=True)
  assert state == {}
  assert {'map', 'query', 'map_t_query_samples', 'scores_poses', 'best_index', 'map_t_query',
          'map_t_query_ransac', 'scores_grid_refine'} <= set(pred)
  assert set(pred['map']) == {'streetview', 'aerial', 'bev_features', 'bev_matching'}
  assert set(pred['query']) == {'streetview', 'bev_features', 'bev_matching'}
  assert 'xyz_query' not in batch['map'] and 'image_feature_pyr' not in batch['map']
  assert pred['map']['streetview']['feature_volume'].features.shape == (2, 32, 32, 12, 32)
  assert pred['query']['bev_matching'].features.shape[0:3:2] == (2, 1)
  assert pred['scores_poses'].shape == (2, 25) and pred['scores_grid_refine'].shape == (2, 41, 41, 41)
  assert pred['map_t_query_samples'].shape == (2, 25)
  # GT pose is prepended at index 0.
  gt = geometry.Transform2D.from_Transform3D(batch['T_query2map'])
  assert torch.allclose(pred['map_t_query_samples'].t[:, 0], gt.t)

  samples = pred['map_t_query_samples']
  ps = o_geo.Transform2D(samples.angle[:, 1:].numpy(), samples.t[:, 1:].numpy())
  ref = o_model.bev_localizer(
      helpers.params_to_numpy(variables['params']), cfg, {'streetview_hfov_deg': 72.0},
      o_grids.Grid2D(meta['grid'].extent[:2], 0.2), helpers.batch_to_oracle(batch), pose_samples=ps)
  helpers.report('bev_matching', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=2e-5)
  helpers.report('scores', pred['scores_poses'], ref['scores_poses'], atol=2e-5)
  helpers.assert_same_argmax('best', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])

  losses, metrics = model.loss_metrics_function(pred, batch, variables['params'])
  rl, rm = o_pose.loss_metrics(
      dict(scores_poses=pred['scores_poses'].numpy(),
           map_t_query_samples=o_geo.Transform2D(samples.angle.numpy(), samples.t.numpy()),
           map_t_query=o_geo.Transform2D(pred['map_t_query'].angle.numpy(), pred['map_t_query'].t.numpy())),
      helpers.batch_to_oracle(batch)['T_query2map'])
  assert set(losses) == {'localization/nll', 'total'}
  np.testing.assert_allclose(losses['total'].numpy(), rl['total'], rtol=1e-5)
  for k, v in rm.items():
    np.testing.assert_allclose(metrics[k].numpy().astype(np.float64), np.asarray(v, np.float64),
                               rtol=1e-4, atol=1e-5, err_msg=k)
  assert 'loc/temperature' in metrics


def test_fused_plane_path_host_plumbing(oracle_backend, monkeypatch):
  """materialize_volume=False on a split engine: the module hands the lift's rows to the fused
  MLP / pool op pre-split, only for observed voxels, CLASSED by observation count (single-observation
  rows without their variance slabs) -- checked here on the CPU through the oracle twins of the ops
  (what the kernels leave unwritten is NaN in the twins): same BEV planes as the reference path."""
  from snap_amd import ops
  cfg, meta, model = _tiny(num_pose_samples=12)
  loc = model.flax_model
  variables = loc.init({'params': 0, 'sampling': 1}, device='cpu')
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=1)
  ref = loc.apply(variables, batch, train=False, rngs={'sampling': 3})
  seen = {}
  lift = ops.lift_pool

  def spy(*a, **kw):
    out = lift(*a, **kw)
    seen.setdefault('kw', []).append({k: kw.get(k) for k in ('valid_rows_only', 'out_split', 'class_rows', 'tap_records')})
    if kw.get('class_rows'):
      seen['classes'] = out[2]
    return out

  monkeypatch.setattr(ops, 'lift_pool', spy)
  monkeypatch.setattr(ops, 'MATMUL_PRECISION', 'bf16x3')
  monkeypatch.setattr(ops, 'pack_weights_split_multi', lambda *a, **k: None)
  cfg2 = helpers.tiny_localizer_config()
  cfg2.num_pose_samples = 12
  cfg2.bev_mapper.materialize_volume = False
  loc2 = bev_localizer.BEVLocalizer(cfg2, meta['build_config'].scene_config, meta['grid'].bev())
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=1)
  got = loc2.apply(variables, batch, train=False, rngs={'sampling': 3})
  assert all(k == {'valid_rows_only': True, 'out_split': True, 'class_rows': True, 'tap_records': True} for k in seen['kw'])
  assert set(np.unique(seen['classes'].numpy())) <= {0, 1, 2} and int((seen['classes'] == 1).sum()) > 0
  vol = got['map']['streetview']['feature_volume']      # lazily produced: the reference's pytree entry
  assert not vol.materialized
  assert vol.features.shape == (2, 32, 32, 12, 32) and vol.materialized
  for side in ('map', 'query'):
    helpers.report(f'{side} bev_matching (fused, classed)', got[side]['bev_matching'].features,
                   ref[side]['bev_matching'].features, atol=2e-5)
    assert torch.equal(got[side]['bev_matching'].valid, ref[side]['bev_matching'].valid)
  helpers.report('scores (fused, classed)', got['scores_poses'], ref['scores_poses'], atol=2e-5)


def test_unsupported_configs_raise_like_the_reference():
  cfg = helpers.tiny_localizer_config()
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  sc = meta['build_config'].scene_config
  bad = helpers.tiny_localizer_config()
  bad.add_confidence_map = True
  with pytest.raises(NotImplementedError):
    bev_localizer.BEVLocalizer(bad, sc, meta['grid'].bev())
  bad = helpers.tiny_localizer_config()
  bad.filter_points_in_fov = False
  with pytest.raises(ValueError):
    bev_localizer.BEVLocalizer(bad, sc, meta['grid'].bev())
  bad = helpers.tiny_localizer_config()
  bad.bev_mapper.streetview_encoder.image_encoder.encoder_name = 'swin'   # ('vit' is a build extension)
  with pytest.raises(ValueError):
    bev_localizer.BEVLocalizer(bad, sc, meta['grid'].bev())
  bad = helpers.tiny_localizer_config()
  bad.bev_mapper.pooling.pooling = 'bogus'
  with pytest.raises(NotImplementedError):
    bev_localizer.BEVLocalizer(bad, sc, meta['grid'].bev())


def test_recover_dense_feature_plane(oracle_backend):
  cfg, meta, model = _tiny()
  loc = model.flax_model
  n = loc.q_xy_p.shape[0]
  from snap_amd.models import types
  sparse = types.FeaturePlane(torch.arange(n * 3, dtype=torch.float32).reshape(n, 1, 3),
                              torch.ones(n, 1, dtype=torch.bool))
  dense = loc.recover_dense_feature_plane(sparse)
  assert dense.features.shape == (*loc.grid_query.extent, 3)
  assert int(dense.valid.sum()) == n


def test_semantic_net_losses_match_the_oracle():
  """SemanticNetModel.loss_metrics_function (host-side torch) vs the numpy restatement of
  semantic_net.py:38-111,225-343 on random logits / masks (runs on the CPU: no kernels)."""
  from oracle import semantic_net as o_sem
  from snap_amd import models
  from snap_amd.data import synthetic
  from snap_amd.models import types
  gt_classes = ('crosswalk', 'sidewalk', 'road', 'terrain', 'building', 'fence', 'pole', 'tree',
                'traffic_sign', 'traffic_light', 'street_light', 'line', 'stopline')
  map_classes = ('sidewalk', 'buildings_raw', 'tree', 'pavedroad')
  model_cls = models.get_model('semantic_net')
  cfg = model_cls.default_flax_model_config()
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  meta['semantic_classes_gt'] = gt_classes
  meta['semantic_map_classes'] = None          # (no semantic INPUT modality in the default config)
  model = model_cls(cfg, meta)
  model.dataset_meta_data = dict(meta, semantic_map_classes=map_classes)
  g = torch.Generator().manual_seed(0)
  B, H, W = 3, 9, 7
  pred = {
      'bev_features': types.FeaturePlane(features=torch.zeros(B, H, W, 4),
                                         valid=torch.rand((B, H, W), generator=g) > 0.2),
      'logits_areas': torch.randn((B, H, W, 5), generator=g) * 2,
      'logits_objects_exclusive': torch.randn((B, H, W, 4), generator=g) * 2,
      'logits_objects_independent': torch.randn((B, H, W, 3), generator=g) * 2,
  }
  pred['bev_features'].valid[2] = False        # an empty mask: masked_mean must return 0, not NaN
  rasters = {'gt_semantics': torch.rand((B, H, W, len(gt_classes)), generator=g) < 0.3,
             'semantics': torch.rand((B, H, W, len(map_classes)), generator=g) < 0.5}
  losses, metrics = model.loss_metrics_function(pred, {'map': {'rasters': rasters}})
  opred = {k: (v.numpy() if k != 'bev_features' else {'valid': v.valid.numpy()}) for k, v in pred.items()}
  olosses, ometrics = o_sem.loss_metrics(
      opred, cfg.to_dict(), gt_classes, map_classes, {k: v.numpy() for k, v in rasters.items()})
  assert set(losses) == set(olosses) and set(metrics) == set(ometrics)
  for k in losses:
    np.testing.assert_allclose(losses[k].numpy(), olosses[k], rtol=2e-5, atol=1e-6, err_msg=k)
  for k in metrics:
    np.testing.assert_allclose(metrics[k].numpy(), ometrics[k], rtol=2e-5, atol=1e-6, err_msg=k)
  assert np.isfinite(losses['total'].numpy()).all() and float(losses['total'][2]) == 0.0
  packed = model.pack_evaluation_metrics(metrics, losses, {'map': {'rasters': rasters}}, pred)
  assert 'gt_counts/road' in packed and 'loss' in packed


def test_reference_batch_adapter():
  """snap_amd.data.adapter: a batch in the reference's loader schema (numpy leaves, struct
  objects or field dicts, optional pmap axis) becomes the batch the models consume."""
  import types as pytypes
  from snap_amd.data import adapter
  from snap_amd.data import synthetic
  from snap_amd.utils import geometry
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  ours = synthetic.make_batch(4, meta['grid'], 3, (32, 32), seed=0)

  def np_struct(s, as_object):
    fields = {f: getattr(s, f).numpy() for f in s._fields}
    return pytypes.SimpleNamespace(**fields) if as_object else fields

  def ref_scene(scene, as_object):
    return {'images': scene['images'].numpy(), 'camera': np_struct(scene['camera'], as_object),
            'T_view2scene': np_struct(scene['T_view2scene'], as_object),
            **({'rasters': {k: v.numpy() for k, v in scene['rasters'].items()}} if 'rasters' in scene else {}),
            'scene_id': ['a', 'b', 'c', 'd']}

  for as_object in (True, False):
    ref = {'map': ref_scene(ours['map'], as_object), 'query': ref_scene(ours['query'], as_object),
           'T_query2map': np_struct(ours['T_query2map'], as_object),
           'batch_mask': np.array([1, 1, 1, 0], bool), 'pair_id': np.arange(4)}
    got = adapter.from_reference_batch(ref)
    assert isinstance(got['map']['camera'], geometry.FisheyeCamera)
    assert torch.equal(got['map']['images'], ours['map']['images'])
    assert torch.equal(got['map']['camera'].k_radial, ours['map']['camera'].k_radial)
    assert torch.equal(got['query']['T_view2scene'].R, ours['query']['T_view2scene'].R)
    assert torch.equal(got['T_query2map'].t, ours['T_query2map'].t)
    assert got['batch_mask'].tolist() == [True, True, True, False]
    assert got['map']['scene_id'] == ['a', 'b', 'c', 'd'] and 'pair_id' in got
  # pmap layout [D, B/D, ...]
  def split(x):
    return x.reshape(2, 2, *x.shape[1:])
  refp = {'query': {'images': split(ours['query']['images'].numpy()),
                    'camera': {f: split(getattr(ours['query']['camera'], f).numpy())
                               for f in ours['query']['camera']._fields},
                    'T_view2scene': {'R': split(ours['query']['T_view2scene'].R.numpy()),
                                     't': split(ours['query']['T_view2scene'].t.numpy())}}}
  merged = adapter.from_reference_batch(refp, device_axis='merge')
  assert torch.equal(merged['query']['images'], ours['query']['images'])
  shard1 = adapter.from_reference_batch(refp, device_axis=1)
  assert torch.equal(shard1['query']['images'], ours['query']['images'][2:])
  assert shard1['batch_mask'].shape == (2,)


def test_sample_transforms_random_is_uniform_in_the_corner_frame():
  """pose_estimation.py:85-97: uniform angles, translations within 2/3 of the extent around the
  grid CENTRE, returned in the corner frame (corner_t_center @ T @ corner_t_center^-1)."""
  import torch
  from snap_amd.models import pose_estimation
  from snap_amd.utils import geometry, grids
  grid = grids.Grid2D((120, 160), 0.2)
  tf = pose_estimation.sample_transforms_random(7, 5000, grid, device='cpu')
  assert tf.angle.shape == (5000,) and tf.t.shape == (5000, 2)
  size = torch.tensor([24.0, 32.0])
  c = geometry.Transform2D(torch.zeros(()), size / 2)
  centre = c.inv @ tf @ c
  ang = torch.remainder(centre.angle, 2 * np.pi)
  assert float(ang.min()) >= 0 and float(ang.max()) < 2 * np.pi + 1e-5
  assert abs(float(ang.mean()) - np.pi) < 0.15
  lim = size * 2 / 3
  assert bool((centre.t.abs() <= lim + 1e-3).all()) and float(centre.t.abs().max()) > 0.9 * float(lim.min())
  again = pose_estimation.sample_transforms_random(7, 5000, grid, device='cpu')
  assert torch.equal(again.t, tf.t) and torch.equal(again.angle, tf.angle)


# ----------------------------------------------------------------------------
# frequency-domain voting: the numpy model and the CPU emulation of the HIP kernel bodies
# ----------------------------------------------------------------------------
def _voting_case(R, H, W, D, seed):
  rng = np.random.default_rng(seed)
  t = rng.standard_normal((R, H, W, D)).astype(np.float32)
  tv = rng.random((R, H, W)) > 0.2
  t = t * tv[..., None]
  m = rng.standard_normal((H, W, D)).astype(np.float32)
  mv = rng.random((H, W)) > 0.1
  return t, tv, m, mv


def test_fft_voting_model_equals_the_direct_form():
  """tools/fft_voting_model.py (the index-exact model of voting_fft.hip: DIF forward / DIT inverse
  mixed-radix stages, digit-reversed spectra, channel pairs and rotation pairs packed as complex
  numbers) against oracle/voting.py's sliding-window sum."""
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  import fft_voting_model as fm
  from oracle import voting as o_voting
  for N in (12, 16, 24, 32, 48, 96, 768):
    x = np.random.default_rng(N).standard_normal((N, 2)) + 1j * np.random.default_rng(N + 1).standard_normal((N, 2))
    X = fm.dif_forward(x, 0)
    ref = np.fft.fft(x, axis=0)
    perm = [int(np.argmin(np.abs(ref[:, 0] - X[k, 0]))) for k in range(N)]
    assert sorted(perm) == list(range(N))                      # a permutation of the spectrum ...
    np.testing.assert_allclose(X, ref[perm], atol=1e-9)
    np.testing.assert_allclose(fm.dit_inverse(X, 0) / N, x, atol=1e-9)   # ... that the inverse undoes
  for (R, H, W, D) in [(4, 8, 8, 6), (9, 5, 7, 4)]:
    t, tv, m, mv = _voting_case(R, H, W, D, 7)
    want = o_voting.template_matching(t, tv, m, mv)
    got = fm.template_matching_fft(t, tv, m, mv)
    fin = np.isfinite(want)
    assert (fin == np.isfinite(got)).all()
    np.testing.assert_allclose(got[fin], want[fin], atol=2e-6)


def test_voting_fft_kernel_bodies_on_the_cpu_emulation():
  """The kernel bodies of snap_amd/csrc/voting_fft_body.h -- the same header hipcc compiles into
  libsnap_hip.so -- built with g++ (tests/emu/voting_fft_emu.cpp: a pthread per GPU thread, a barrier
  per __syncthreads) and run against oracle/voting.py: plan, index arithmetic, workspace carving and
  launch sequence checked without a GPU (odd R, two channel groups with a partial one, a rectangular
  map; with and without the overlap mask)."""
  import ctypes
  import subprocess
  from oracle import voting as o_voting
  build = os.path.join(ROOT, 'tests', '_build')
  os.makedirs(build, exist_ok=True)
  so = os.path.join(build, 'libvoting_fft_emu.so')
  src = os.path.join(ROOT, 'tests', 'emu', 'voting_fft_emu.cpp')
  hdr = os.path.join(ROOT, 'snap_amd', 'csrc', 'voting_fft_body.h')
  hdr2 = os.path.join(ROOT, 'snap_amd', 'csrc', 'rotate_sample.h')
  if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
    subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread',
                    '-I' + os.path.join(ROOT, 'snap_amd', 'csrc'), src, '-o', so], check=True)
  lib = ctypes.CDLL(so)
  lib.emu_voting_fft_workspace_bytes.restype = ctypes.c_size_t
  lib.emu_voting_fft_workspace_bytes.argtypes = [ctypes.c_int] * 6
  lib.emu_voting_fft_f32.restype = ctypes.c_int
  lib.emu_voting_fft_f32.argtypes = ([ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 +
                                     [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int])
  for (R, H, W, D, nt, overlap) in [(4, 8, 8, 6, 32, 0.05), (3, 6, 6, 34, 16, 0.05), (5, 5, 7, 4, 32, None)]:
    t, tv, m, mv = _voting_case(R, H, W, D, 11 + R)
    nb = lib.emu_voting_fft_workspace_bytes(R, H, W, D, H, W)
    assert nb > 0
    ws = np.zeros(nb + 256, np.uint8)
    off = (-ws.ctypes.data) % 256
    out = np.full((R, 2 * H - 1, 2 * W - 1), np.nan, np.float32)
    tvb, mvb = tv.astype(np.uint8), mv.astype(np.uint8)
    tc = tv.sum((-1, -2)).astype(np.float32)
    rc = lib.emu_voting_fft_f32(t.ctypes.data, tvb.ctypes.data, m.ctypes.data, mvb.ctypes.data, tc.ctypes.data,
                                R, H, W, D, H, W, float(0.0 if overlap is None else overlap * H * W),
                                int(overlap is not None), ws.ctypes.data + off, out.ctypes.data, nt)
    assert rc == 0
    want = o_voting.template_matching(t, tv, m, mv, min_overlap=overlap)
    fin = np.isfinite(want)
    assert (fin == np.isfinite(out)).all()
    np.testing.assert_allclose(out[fin], want[fin], atol=5e-6)


def test_voting_fft_rotated_entry_on_the_cpu_emulation():
  """exhaustive_pose_voting's frequency-domain entry (templates sampled inside the first transform,
  rot90 quadrants as index maps, masks + counts from the validity-only pass) on the CPU emulation vs
  oracle/voting.py: sample_query_templates + template_matching."""
  import ctypes
  build = os.path.join(ROOT, 'tests', '_build')
  so = os.path.join(build, 'libvoting_fft_emu.so')
  if not os.path.exists(so):
    test_voting_fft_kernel_bodies_on_the_cpu_emulation()
  lib = ctypes.CDLL(so)
  lib.emu_voting_fft_workspace_bytes.restype = ctypes.c_size_t
  lib.emu_voting_fft_workspace_bytes.argtypes = [ctypes.c_int] * 6
  lib.emu_voting_fft_rotated_f32.restype = ctypes.c_int
  lib.emu_voting_fft_rotated_f32.argtypes = ([ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2 +
                                             [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                                                   ctypes.c_int])
  rng = np.random.default_rng(5)
  for (H, R, D) in [(8, 8, 6), (10, 12, 4)]:
    cell = 0.25
    vq = rng.random((H, H)) > 0.15
    fq = rng.standard_normal((H, H, D)).astype(np.float32) * vq[..., None]
    fm = rng.standard_normal((H, H, D)).astype(np.float32)
    vm = rng.random((H, H)) > 0.1
    t_w, tv_w = o_voting.sample_query_templates(fq, vq, R, o_grids.Grid2D((H, H), cell))
    want = o_voting.template_matching(t_w, tv_w, fm, vm)
    tfm = pev._template_transforms(R, grids.Grid2D((H, H), cell), 'cpu')[: R // 4].contiguous().numpy()
    nb = lib.emu_voting_fft_workspace_bytes(R, H, H, D, H, H)
    ws = np.zeros(nb + 256, np.uint8)
    off = (-ws.ctypes.data) % 256
    out = np.full((R, 2 * H - 1, 2 * H - 1), np.nan, np.float32)
    vqb, vmb = vq.astype(np.uint8), vm.astype(np.uint8)
    rc = lib.emu_voting_fft_rotated_f32(fq.ctypes.data, vqb.ctypes.data, tfm.ctypes.data, cell, fm.ctypes.data,
                                        vmb.ctypes.data, R, H, D, H, H, 0.05, ws.ctypes.data + off,
                                        out.ctypes.data, 32)
    assert rc == 0
    fin = np.isfinite(want)
    assert (fin == np.isfinite(out)).all()
    np.testing.assert_allclose(out[fin], want[fin], atol=5e-6)


def test_train_step_rejects_a_precision_that_contradicts_the_models_engine():
  """``Module.apply`` enters the model's own engine scope, so an explicit ``precision=`` that
  differs from ``model.engine`` would be silently ignored -- it raises instead (before any compute); an
  fp16 model stepping without a DynamicScale warns."""
  import types as pytypes
  import warnings
  from snap_amd import trainer
  state = trainer.TrainState.create({'w': torch.zeros(3)})
  model = pytypes.SimpleNamespace(engine='bf16x3')
  with pytest.raises(ValueError, match='contradicts the engine'):
    trainer.train_step(state, {}, model=model, lr_fn=lambda s: 1e-3, precision='f32')
  with pytest.raises(ValueError, match='precision='):
    trainer.train_step(state, {}, model=model, lr_fn=lambda s: 1e-3, precision='no-such-engine')
  half = pytypes.SimpleNamespace(engine='fp16')
  with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter('always')
    with pytest.raises(Exception):                      # (no GPU / no flax_model: the step itself cannot run here)
      trainer.train_step(state, {}, model=half, lr_fn=lambda s: 1e-3)
  assert any('dynamic_scale' in str(w.message) for w in rec)


def test_tuning_object_scopes_and_module_aliases():
  """Every tuning / test switch of the kernel wrappers lives in ONE ``ops.Tuning`` object: per-thread scopes
  (``ops.tuning_scope`` / ``ops.engine_scope(engine, tuning=...)``), the module attributes are aliases (read:
  the tuning in force; write: the process default), unknown switches raise."""
  import threading
  assert ops.tuning() is ops._DEFAULT_TUNING and ops.CONV_TILE is None and ops.LIFT_IN_CONSUMER is True
  with ops.tuning_scope(CONV_TILE='64x64', MLP_POOL_WIDE=True) as t:
    assert ops.CONV_TILE == '64x64' and ops.MLP_POOL_WIDE is True and ops.tuning() is t
    seen = {}
    th = threading.Thread(target=lambda: seen.update(tile=ops.CONV_TILE))     # another thread: the default
    th.start(); th.join()
    assert seen['tile'] is None
    with ops.engine_scope('bf16x3', tuning=ops.Tuning(CONV_NO_HALO=True)):
      assert ops.precision() == 'bf16x3' and ops.CONV_NO_HALO is True and ops.CONV_TILE is None
    assert ops.CONV_TILE == '64x64' and ops.CONV_NO_HALO is False
  assert ops.CONV_TILE is None and ops.MLP_POOL_WIDE is False
  ops.CONV_RS_FORCE = True                       # (module attribute = the process default)
  try:
    assert ops.tuning().CONV_RS_FORCE is True and repr(ops.tuning()) == 'Tuning(CONV_RS_FORCE=True)'
  finally:
    ops.CONV_RS_FORCE = False
  with pytest.raises(TypeError):
    ops.Tuning(NO_SUCH_SWITCH=1)
  with pytest.raises(TypeError):
    with ops.tuning_scope(NO_SUCH_SWITCH=1):
      pass
  # no switch is left as a plain module global (they would shadow nothing, and nothing would read them)
  assert not [k for k in ops._TUNING_DEFAULTS if k in vars(ops)]
