"""`-m gpu` end-to-end parity: BEVLocalizer forward on HIP vs the numpy oracle."""
import numpy as np
import pytest
import torch

import helpers
from oracle import geometry as o_geo
from oracle import grids as o_grids
from oracle import model as o_model
from snap_amd.data import synthetic
from snap_amd.models import bev_localizer

pytestmark = pytest.mark.gpu


def _run(cfg, B, V, img, seed, refine=False, math='f32', extent=(6.4, 6.4, 12), want_batch=False):
  """One BEVLocalizer forward on the GPU (conv / dense engine `math`) and through the oracle."""
  from snap_amd import ops
  dev = torch.device(helpers.DEVICE)
  meta = synthetic.meta_data(0.2, extent)
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  variables = loc.init(seed, device='cpu')
  batch = synthetic.make_batch(B, meta['grid'], V, img, seed=seed + 1)
  prev = ops.MATMUL_PRECISION
  ops.MATMUL_PRECISION = math
  try:
    pred = loc.apply(
        {'params': helpers.params_to_device(variables['params'], dev)},
        helpers.batch_to_device(batch, dev), train=False, rngs={'sampling': 11}, debug=True,
    )
    if dev.type == 'cuda':
      torch.cuda.synchronize()
  finally:
    ops.MATMUL_PRECISION = prev
  samples = pred['map_t_query_samples']
  ps = o_geo.Transform2D(samples.angle[:, 1:].cpu().numpy(), samples.t[:, 1:].cpu().numpy())
  ref = o_model.bev_localizer(
      helpers.params_to_numpy(variables['params']), cfg, {'streetview_hfov_deg': 72.0},
      o_grids.Grid2D(meta['grid'].extent[:2], 0.2), helpers.batch_to_oracle(batch),
      pose_samples=ps, keep_sim=True,
  )
  if want_batch:
    return pred, ref, helpers.batch_to_oracle(batch), meta
  return pred, ref


def _check_validity(name, pred_side, ref_side, scene, cfg):
  """Voxel validity: identical to the oracle except on visibility boundaries (see helper)."""
  sv, rsv = pred_side['streetview'], ref_side['streetview']
  stride = np.asarray(rsv['image_feature_pyramid']['strides'][-1]).reshape(-1, 2)[0]
  xyz = ref_side.get('_xyz_query')
  return helpers.assert_validity_mismatches_on_borders(
      name, sv['feature_volume'].valid, rsv['feature_volume']['valid'], scene, xyz, stride,
      max_view_distance=cfg.bev_mapper.streetview_encoder.get('max_view_distance'))


# every engine the bench can select must hold the same parity bar
@pytest.mark.parametrize('math', ['f32', 'bf16x6', 'bf16x3'])
@pytest.mark.parametrize('top_k,V', [(2, 3), (4, 3)])
def test_localizer_forward_parity(top_k, V, math):
  cfg = helpers.tiny_localizer_config(top_k=top_k)
  pred, ref, ob, _ = _run(cfg, 2, V, (64, 64), seed=0, math=math, want_batch=True)
  sv, rsv = pred['map']['streetview'], ref['map']['streetview']
  # feature maps: north-star tolerance 1e-3 (fp32); observed ~1e-5.
  helpers.report('image features', sv['image_feature_pyramid'].features[-1],
                 rsv['image_feature_pyramid']['features'][-1], atol=1e-3)
  vg = sv['feature_volume'].valid.cpu().numpy()
  vw = rsv['feature_volume']['valid']
  mism = vg != vw
  _check_validity('map voxel validity', pred['map'], ref['map'], ob['map'], cfg)
  _check_validity('query voxel validity', pred['query'], ref['query'], ob['query'], cfg)
  helpers.report('feature volume', sv['feature_volume'].features.cpu().numpy()[~mism],
                 rsv['feature_volume']['features'][~mism], atol=1e-3)
  helpers.report('aerial plane', pred['map']['aerial']['feature_plane'].features,
                 ref['map']['aerial']['feature_plane']['features'], atol=1e-3)
  helpers.report('map bev_matching', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=1e-3)
  helpers.report('query bev_matching', pred['query']['bev_matching'].features,
                 ref['query']['bev_matching']['features'], atol=1e-3)
  helpers.report('sim_points', pred['sim_points'], ref['_sim_points'], atol=1e-5, rtol=1e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  # pose argmax: identical to the oracle's (up to exact fp32 near-ties, see helper).
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


def test_localizer_without_feature_volume():
  """bev_mapper.materialize_volume = False on the bf16x3 engine: the fusion MLP and the vertical
  max pooling run as one kernel (ops.mlp2_pool_max); the plane is bit-identical to the unfused
  chain's and everything downstream holds the same parity bar."""
  cfg = helpers.tiny_localizer_config(top_k=2)
  full, ref, ob, _ = _run(cfg, 2, 3, (64, 64), seed=4, math='bf16x3', want_batch=True)
  cfg.bev_mapper.materialize_volume = False
  pred, _ = _run(cfg, 2, 3, (64, 64), seed=4, math='bf16x3')
  for side in ('map', 'query'):
    sv = pred[side]['streetview']
    # the pytree keeps the reference's entry (streetview_encoder.py:282-286): not computed by the
    # step, produced by the unfused chain on first access -- the volume of the materialising run
    vol = sv['feature_volume']
    assert not vol.materialized
    assert torch.equal(vol.features, full[side]['streetview']['feature_volume'].features)
    assert vol.materialized and torch.equal(vol.valid, full[side]['streetview']['feature_volume'].valid)
    _check_validity(f'{side} voxel validity', pred[side], ref[side], ob[side], cfg)
    fp, fp_full = sv['feature_plane'], full[side]['streetview']['feature_plane']
    assert torch.equal(fp.valid, fp_full.valid)
    assert torch.equal(fp.features, fp_full.features), float((fp.features - fp_full.features).abs().max())
    helpers.report(f'{side} bev_matching', pred[side]['bev_matching'].features,
                   ref[side]['bev_matching']['features'], atol=1e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


def test_localizer_with_query_confidence_parity():
  """add_confidence_query (bev_localizer.py:165-168 + the Dense(1) head of bev_mapper.py:154-157):
  log-sigmoid confidences, masked-softmax point weights instead of 1 / num_valid, parity of the
  weighted similarity / scores / argmax with the oracle."""
  cfg = helpers.tiny_localizer_config(top_k=2)
  cfg.add_confidence_query = True
  pred, ref = _run(cfg, 2, 3, (64, 64), seed=8)
  assert cfg.bev_mapper.add_confidence                      # set by the localizer, as the reference's setup
  helpers.report('query bev_confidence', pred['query']['bev_confidence'], ref['query']['bev_confidence'],
                 atol=1e-4, rtol=1e-4)
  assert float(pred['query']['bev_confidence'].max()) <= 0.0
  helpers.report('sim_points (weighted)', pred['sim_points'], ref['_sim_points'], atol=1e-6, rtol=2e-3)
  helpers.report('prob_points (weighted)', pred['prob_points'], ref['_prob_points'], atol=1e-9, rtol=2e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-4, rtol=2e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


@pytest.mark.parametrize('pooling', ['softmax', 'weighted'])
def test_learned_modality_fusion_parity(pooling):
  """modality_fusion.pooling = 'softmax' / 'weighted' (bev_mapper.py:246-252 reuses
  VerticalPooling, :63-78, over the stacked StreetView / aerial planes)."""
  cfg = helpers.tiny_localizer_config(top_k=2)
  cfg.bev_mapper.modality_fusion.pooling = pooling
  pred, ref = _run(cfg, 2, 3, (64, 64), seed=9)
  helpers.report(f'bev_features ({pooling} fusion)', pred['map']['bev_features'].features,
                 ref['map']['bev_features']['features'], atol=1e-3)
  assert np.array_equal(pred['map']['bev_features'].valid.cpu().numpy(), ref['map']['bev_features']['valid'])
  helpers.report('map bev_matching', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=1e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


def test_unweighted_minmax_fusion_parity():
  """do_weighted_fusion=False (no proj_mlp, plain mean / variance over the views, no score
  channel) + fusion_add_minmax=True: streetview_encoder.py:141-178,259-262."""
  cfg = helpers.tiny_localizer_config(top_k=2)
  sv = cfg.bev_mapper.streetview_encoder
  sv.do_weighted_fusion = False
  sv.fusion_add_minmax = True
  pred, ref, ob, _ = _run(cfg, 2, 3, (64, 64), seed=10, want_batch=True)
  assert 'scores_images' not in pred['map']['streetview']
  _check_validity('map voxel validity', pred['map'], ref['map'], ob['map'], cfg)
  vol, rvol = pred['map']['streetview']['feature_volume'], ref['map']['streetview']['feature_volume']
  both = vol.valid.cpu().numpy() == rvol['valid']
  helpers.report('feature volume (unweighted + minmax)', vol.features.cpu().numpy()[both],
                 rvol['features'][both], atol=1e-3)
  helpers.report('map bev_matching', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=1e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


@pytest.mark.parametrize('top_k,V', [(2, 3), (4, 3)])
def test_depth_mlp_fusion_parity(top_k, V):
  """do_weighted_fusion=False + depth_mlp (streetview_encoder.py:214-216, 263-267): the
  observations leave the lift un-pooled, a per-observation MLP on [features, log10 depth, viewing
  ray] is added to them, then the plain mean / variance pooling.  Top-K (V > K) and all-views
  branches."""
  from snap_amd.configs import defaults
  cfg = helpers.tiny_localizer_config(top_k=top_k)
  sv = cfg.bev_mapper.streetview_encoder
  sv.do_weighted_fusion = False
  dm = defaults.mlp()
  dm.layers = (24, sv.feature_dim)
  sv.depth_mlp = dm
  pred, ref, ob, _ = _run(cfg, 2, V, (64, 64), seed=12, want_batch=True)
  _check_validity('map voxel validity', pred['map'], ref['map'], ob['map'], cfg)
  vol, rvol = pred['map']['streetview']['feature_volume'], ref['map']['streetview']['feature_volume']
  both = vol.valid.cpu().numpy() == rvol['valid']
  helpers.report('feature volume (depth_mlp)', vol.features.cpu().numpy()[both], rvol['features'][both],
                 atol=1e-3)
  helpers.report('map bev_matching', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=1e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])


def test_localizer_grid_refinement_parity():
  cfg = helpers.tiny_localizer_config(refine=True, num_pose_samples=32)
  pred, ref = _run(cfg, 1, 3, (64, 64), seed=3)
  assert pred['scores_grid_refine'].shape == (1, 41, 41, 41)
  helpers.report('scores_grid_refine', pred['scores_grid_refine'], ref['scores_grid_refine'],
                 atol=1e-3, rtol=1e-3)
  helpers.assert_same_argmax(
      'grid refinement argmax', pred['scores_grid_refine'].reshape(1, -1),
      ref['scores_grid_refine'].reshape(1, -1))
  got = int(torch.argmax(pred['scores_grid_refine'].reshape(-1)))
  if got == int(np.argmax(ref['scores_grid_refine'].reshape(-1))):
    helpers.report('refined pose', pred['map_t_query'].packed(),
                   np.concatenate([ref['map_t_query'].angle[:, None], ref['map_t_query'].t], -1),
                   atol=1e-5)


def test_output_pytree_keys():
  """Key-for-key pytree of bev_localizer.py:130-220 / bev_mapper.py:254-296."""
  cfg = helpers.tiny_localizer_config(refine=True, num_pose_samples=16)
  pred, _ = _run(cfg, 1, 3, (64, 64), seed=5)
  base = {'map', 'query', 'map_t_query_samples', 'scores_poses', 'best_index', 'map_t_query',
          'map_t_query_ransac', 'scores_grid_refine'}
  assert base <= set(pred)
  assert set(pred['map']) == {'streetview', 'aerial', 'bev_features', 'bev_matching'}
  assert set(pred['query']) == {'streetview', 'bev_features', 'bev_matching'}
  assert set(pred['map']['streetview']) == {
      'image_feature_pyramid', 'scores_images', 'feature_volume', 'vertical_pooling',
      'feature_plane'}
  assert pred['scores_poses'].shape == (1, 17)
  assert pred['map_t_query_samples'].shape == (1, 17)
  # the same pytree in the plane-only mode of the bench (bev_mapper.materialize_volume = False)
  cfg.bev_mapper.materialize_volume = False
  plane, _ = _run(cfg, 1, 3, (64, 64), seed=5, math='bf16x3')
  assert set(plane) >= base and set(plane['map']['streetview']) == set(pred['map']['streetview'])
  vol = plane['map']['streetview']['feature_volume']
  assert vol.features.shape == pred['map']['streetview']['feature_volume'].features.shape
  assert vol.valid.shape == vol.features.shape[:-1]


def test_evaluator_contract_on_tiny_model(tmp_path):
  """eval_step / eval_on_batches / dump round trip (snap/evaluator.py:57-109,205-238)."""
  from snap_amd import evaluator
  from snap_amd import models
  dev = torch.device('cuda')
  cfg = helpers.tiny_localizer_config(num_pose_samples=64, retries=2)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  params = helpers.params_to_device(model.flax_model.init(0, device='cpu')['params'], dev)
  batches = []
  for s in (1, 2):
    b = helpers.batch_to_device(synthetic.make_batch(3, meta['grid'], 2, (64, 64), seed=s), dev)
    b['batch_mask'] = torch.tensor([True, s == 1, True], device=dev)
    batches.append(b)
  res = evaluator.eval_on_batches(model, params, batches, rng=7)
  want = {'error_max_meter', 'error_max_deg', 'recall_top1', 'pose_score_max', 'overlap',
          'time_delta_days', 'closest_map_view_meter', 'closest_map_view_deg', 'loss'}
  assert set(res) == want
  assert all(v.shape == (5,) for v in res.values())               # 3 + 2 valid examples
  for k in want - {'overlap', 'time_delta_days'}:                  # dataset-only fields are NaN
    assert np.isfinite(res[k]).all(), k
  assert set(np.unique(res['recall_top1'])) <= {0.0, 1.0}
  assert (res['closest_map_view_meter'] >= 0).all() and (res['error_max_deg'] <= 180).all()
  evaluator.write_eval_dump(tmp_path / 'osaka', res, cfg)
  back, _ = evaluator.read_eval_dump(tmp_path / 'osaka')
  np.testing.assert_array_equal(back['loss'], res['loss'])
  th, rec = evaluator.compute_recall(res['error_max_meter'], 5.0)
  assert rec[-1] == 100.0 * np.mean(res['error_max_meter'] < 5.0)


def test_batches_in_flight_same_bits():
  """snap_amd.pipeline: batches on alternating HIP streams give the bits of one batch at a time --
  predictions of the localiser AND the evaluator's rows (order included)."""
  from snap_amd import evaluator, models, pipeline
  dev = torch.device('cuda')
  cfg = helpers.tiny_localizer_config(num_pose_samples=64, retries=2)
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  params = helpers.params_to_device(model.flax_model.init(0, device='cpu')['params'], dev)
  batches = []
  for s in range(5):
    b = helpers.batch_to_device(synthetic.make_batch(3, meta['grid'], 2, (64, 64), seed=10 + s), dev)
    b['batch_mask'] = torch.tensor([True, s % 2 == 0, True], device=dev)
    batches.append(b)
  ref = evaluator.eval_on_batches(model, params, batches, rng=3, in_flight=1)
  for n in (2, 3):
    got = evaluator.eval_on_batches(model, params, batches, rng=3, in_flight=n)
    assert set(got) == set(ref)
    for k in ref:
      np.testing.assert_array_equal(got[k], ref[k], err_msg=f'{k} with {n} batches in flight')
  # the raw predictions, twice through a ring of two
  def run(n):
    ring = pipeline.BatchesInFlight(n, dev)
    outs = []
    for rep in range(2):
      for i, b in enumerate(batches):
        with ring.slot(len(outs)):
          p = model.flax_model.apply({'params': params}, b, train=False, rngs={'sampling': 100 + i})
          outs.append((p['scores_poses'], p['best_index'], p['map_t_query'].packed()))
    ring.join()
    torch.cuda.synchronize()
    return outs
  a, c = run(1), run(2)
  for (s0, b0, t0), (s1, b1, t1) in zip(a, c):
    assert torch.equal(s0, s1) and torch.equal(b0, b1) and torch.equal(t0, t1)
  with pytest.raises(ValueError):
    pipeline.BatchesInFlight(0, dev)


def _tiny_vit_config():
  from snap_amd.configs import defaults
  cfg = defaults.image_encoder('vit')
  cfg.encoder.hidden_size = 128
  cfg.encoder.num_heads = 2
  cfg.encoder.num_layers = 2
  cfg.encoder.mlp_dim = 256
  cfg.encoder.posemb_grid = (4, 4)
  cfg.output_dim = 32
  return cfg


@pytest.mark.parametrize('precision,tol', [('f32', 3e-3), ('bf16', 3e-2)])
def test_vit_encoder_matches_oracle(precision, tol):
  """encoder_name='vit' (BASELINE.json configs[4]; no reference ViT exists): the HIP forward
  vs the float64 numpy restatement of the published architecture, same weights.  Tolerances
  relative to the feature range: f32 GEMMs leave only the bf16 attention products; with bf16
  GEMM operands everything is bf16-class."""
  import numpy as np
  from oracle import vit as o_vit
  from snap_amd.models import image_encoder
  cfg = _tiny_vit_config()
  cfg.encoder.matmul_precision = precision
  enc = image_encoder.ImageEncoder(cfg)
  params = enc.init_params(torch.Generator().manual_seed(3), 'cpu')
  g = torch.Generator().manual_seed(4)
  for name in ('bias',):                                   # non-trivial biases / norms
    params['encoder']['embedding'][name] = torch.randn(128, generator=g) * 0.1
  img = torch.rand((3, 64, 40, 3), generator=g)           # 40 -> padded to 48: grid 4 x 3
  pyr = enc(helpers.params_to_device(params, 'cuda'), img.cuda())
  got = pyr.features[0].cpu().numpy()
  assert got.shape == (3, 4, 3, 32) and list(pyr.strides[0]) == [16, 16]

  def to_np(t):
    return {k: to_np(v) for k, v in t.items()} if isinstance(t, dict) else t.numpy().astype(np.float64)
  ocfg = dict(cfg.encoder.to_dict())
  padded = np.pad(img.numpy().astype(np.float64), ((0, 0), (0, 0), (0, 8), (0, 0)))
  want = o_vit.vit_encoder(to_np(params['encoder']), ocfg, padded)[:, :4, :3]
  err = float(np.abs(got - want).max()) / float(np.abs(want).max())
  print(f'vit encoder {precision}: max err / range = {err:.2e}')
  assert err < tol, err


def test_localizer_runs_with_vit_streetview_encoder():
  """The whole localisation path with a (tiny) ViT as the StreetView image encoder."""
  dev = torch.device('cuda')
  cfg = helpers.tiny_localizer_config(num_pose_samples=48, retries=2)
  vit_cfg = _tiny_vit_config()
  vit_cfg.output_dim = cfg.bev_mapper.streetview_encoder.image_encoder.output_dim
  cfg.bev_mapper.streetview_encoder.image_encoder = vit_cfg
  from snap_amd import models
  from snap_amd.data import synthetic
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  model = models.get_model('bev_localizer')(cfg, meta)
  variables = model.flax_model.init(0, device='cpu')
  params = helpers.params_to_device(variables['params'], dev)
  batch = helpers.batch_to_device(synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=1), dev)
  pred = model.flax_model.apply({'params': params}, batch, train=False, rngs={'sampling': 5})
  assert bool(torch.isfinite(pred['scores_poses']).all())
  f = pred['map']['streetview']['image_feature_pyramid'].features[-1]
  assert f.shape[-3:-1] == (4, 4)


SEM_CLASSES = ('sidewalk', 'buildings_raw', 'pavedroad', 'crosswalk', 'trees', 'poles')


def _semantic_config():
  from snap_amd.configs import defaults
  cfg = helpers.tiny_localizer_config(top_k=2)
  sem = defaults.semantic_raster_encoder()
  sem.embedding_dim = 4
  sem.encoder.output_dim = cfg.bev_mapper.aerial_encoder.output_dim
  sem.encoder.encoder.depth = [1, 1]
  sem.encoder.encoder.width = 0.5
  sem.encoder.encoder.limit_num_blocks = 2
  cfg.bev_mapper.semantic_encoder = sem
  return cfg


def test_semantic_embed_kernel_matches_oracle():
  from oracle import bev as o_bev
  from snap_amd import ops
  g = torch.Generator().manual_seed(31)
  rasters = torch.rand((2, 9, 7, len(SEM_CLASSES)), generator=g) < 0.4
  rasters[0, 0, 0] = False                                   # no class set at all
  t_road = torch.randn((3, 8), generator=g)
  t_other = torch.randn((6, 8), generator=g)
  idx_road = [i for i, c in enumerate(SEM_CLASSES) if c in o_bev.SURFEL_ROAD_CLASSES]
  idx_other = [i for i, c in enumerate(SEM_CLASSES) if c not in o_bev.SURFEL_ROAD_CLASSES]
  got = ops.semantic_embed(rasters.cuda(), idx_road, idx_other, t_road.cuda(), t_other.cuda())
  want = o_bev.semantic_raster_embed(
      {'embeddings_surfel_road': {'embedding': t_road.numpy()},
       'embeddings_other_classes': {'embedding': t_other.numpy()}}, SEM_CLASSES, rasters.numpy())
  helpers.report('semantic embed', got, want, atol=0.0)       # a pure gather: bit-exact


def test_localizer_parity_with_semantic_modality():
  """StreetView + aerial + semantic-raster modality (bev_mapper.py:125-129,214-223,273-278)."""
  from snap_amd.data import synthetic
  from snap_amd.models import bev_localizer
  from oracle import geometry as o_geo, grids as o_grids, model as o_model
  cfg = _semantic_config()
  dev = torch.device('cuda')
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev(),
                                   semantic_map_classes=SEM_CLASSES)
  variables = loc.init(3, device='cpu')
  assert 'semantic_encoder' in variables['params']['bev_mapper']
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=4, semantic_classes=SEM_CLASSES)
  pred = loc.apply({'params': helpers.params_to_device(variables['params'], dev)},
                   helpers.batch_to_device(batch, dev), train=False, rngs={'sampling': 11}, debug=True)
  samples = pred['map_t_query_samples']
  ps = o_geo.Transform2D(samples.angle[:, 1:].cpu().numpy(), samples.t[:, 1:].cpu().numpy())
  ocfg = cfg.to_dict()
  ocfg['bev_mapper']['_semantic_map_classes'] = SEM_CLASSES
  ref = o_model.bev_localizer(
      helpers.params_to_numpy(variables['params']), ocfg, {'streetview_hfov_deg': 72.0},
      o_grids.Grid2D(meta['grid'].extent[:2], 0.2), helpers.batch_to_oracle(batch),
      pose_samples=ps, keep_sim=True)
  helpers.report('semantic plane', pred['map']['semantic']['feature_plane'].features,
                 ref['map']['semantic']['feature_plane']['features'], atol=1e-3)
  helpers.report('map bev_matching (3 modalities)', pred['map']['bev_matching'].features,
                 ref['map']['bev_matching']['features'], atol=1e-3)
  helpers.report('scores_poses', pred['scores_poses'], ref['scores_poses'], atol=1e-3, rtol=1e-3)
  helpers.assert_same_argmax('best_index', pred['scores_poses'][:, 1:], ref['scores_poses'][:, 1:],
                             got_index=pred['best_index'])
  assert 'semantic' not in pred['query']


@pytest.mark.parametrize('decoder', ['mlp', 'resnet_stage'])
def test_semantic_net_forward_parity(decoder):
  """SemanticNet (semantic_net.py:123-199): BEV mapper + decoder -> logits, vs the oracle."""
  from oracle import semantic_net as o_sem
  from snap_amd import models
  model_cls = models.get_model('semantic_net')
  cfg = model_cls.default_flax_model_config()
  cfg.bev_mapper = helpers.tiny_localizer_config(top_k=2).bev_mapper
  cfg.decoder_type = decoder
  cfg.decoder_dim = 128 if decoder == 'resnet_stage' else 64
  cfg.resnet_num_units = 2
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  meta['semantic_classes_gt'] = ('crosswalk', 'sidewalk', 'road', 'terrain', 'building', 'fence',
                                 'pole', 'tree', 'traffic_sign', 'traffic_light', 'street_light')
  model = model_cls(cfg, meta)
  net = model.flax_model
  variables = net.init(5, device='cpu')
  batch = synthetic.make_batch(2, meta['grid'], 3, (64, 64), seed=6)
  dev = torch.device('cuda')
  pred = net.apply({'params': helpers.params_to_device(variables['params'], dev)},
                   helpers.batch_to_device(batch, dev), train=False)
  ref = o_sem.semantic_net(helpers.params_to_numpy(variables['params']), cfg.to_dict(),
                           o_grids.Grid2D(meta['grid'].extent[:2], 0.2), helpers.batch_to_oracle(batch))
  for k in ('logits_areas', 'logits_objects_exclusive', 'logits_objects_independent'):
    helpers.report(f'{decoder} {k}', pred[k], ref[k], atol=1e-3)
  assert pred['logits_areas'].shape[-1] == 5 and pred['logits_objects_exclusive'].shape[-1] == 4
  # loss + gradients through the whole net (training path)
  g = torch.Generator().manual_seed(7)
  gt = torch.rand((2, *pred['logits_areas'].shape[1:3], len(meta['semantic_classes_gt'])), generator=g) < 0.3
  data = helpers.batch_to_device(batch, dev)
  data['map']['rasters']['gt_semantics'] = gt.to(dev)
  from snap_amd import trainer
  params = helpers.params_to_device(variables['params'], dev)
  leaves = [t for _, t in trainer.flatten_params(params)]
  for t in leaves:
    t.requires_grad_(True)
  with torch.enable_grad():
    p2 = net.apply({'params': params}, data, train=True, rngs={'sampling': 3})
    losses, metrics = model.loss_metrics_function(p2, data)
    grads = torch.autograd.grad(losses['total'].mean(), leaves, allow_unused=True)
  assert all(g_ is not None and bool(torch.isfinite(g_).all()) for g_ in grads)
  assert 'semantics/accuracy' in metrics and float(losses['total'].mean()) > 0


@pytest.mark.parametrize('engine', ['f32', 'bf16x3'])
def test_projection_gathers_the_cropped_image_features(engine):
  """``StreetViewEncoder._project``: the projection Dense reads the encoder's CROPPED features through
  the conv engine's row list instead of a copied crop -- bit for bit what the copy gives, for the map
  slice (offset 0) and the query slice (an offset into the joint batch) of one padded tensor."""
  from snap_amd import ops
  dev = torch.device('cuda')
  cfg = helpers.tiny_localizer_config()
  meta = synthetic.meta_data(0.2, (6.4, 6.4, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  sv = loc.bev_mapper.streetview_encoder
  p = helpers.params_to_device(loc.init(0, device='cpu')['params'], dev)['bev_mapper']['streetview_encoder']['proj_mlp']
  C = sv.proj_mlp.in_dim
  g = torch.Generator(device='cpu').manual_seed(9)
  padded = torch.randn(5 * 3 + 5, 24, 20, C, generator=g).to(dev)          # B V + B images of 24 x 20 (padded)
  crop = padded[:, :16, :13]                                                # (the encoder's crop: a view)
  fm = crop[:15].reshape(5, 3, 16, 13, C)
  fq = crop[15:].reshape(5, 1, 16, 13, C)
  with ops.engine_scope(engine):
    for f in (fm, fq):
      assert not f.is_contiguous() and f._base is not None
      got = sv._project(p, f, False)
      assert sv.last_projection_path == 'rows'          # (the row-list path ran, not the copying fallback)
      want = sv.proj_mlp(p, f.contiguous(), False)
      assert got.shape == want.shape
      assert torch.equal(got, want), float((got - want).abs().max())
      assert float(got.abs().max()) > 0
