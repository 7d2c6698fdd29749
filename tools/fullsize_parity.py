"""One FULL-SIZE scene of the headline workload (C2: 4 views @512 px + aerial, 128x128x60 voxels,
ResNet-50 encoders, 10 001 pose hypotheses x 8 retries) through the HIP path and through the
numpy oracle on the host cores; prints one JSON line with the deviations and the argmax check.

  python tools/fullsize_parity.py [--out profiles/r01_c2_fullsize_parity.json]

The oracle needs minutes of CPU time at this size, which is why the pytest suites compare at
reduced sizes and check the full size through properties; this is the direct comparison, run
once per round on the GPU box (the pose samples drawn by the HIP sampler are injected into
the oracle: JAX's / numpy's RNG streams cannot match Philox).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import helpers  # noqa: E402
from oracle import geometry as o_geo  # noqa: E402
from oracle import grids as o_grids  # noqa: E402
from oracle import model as o_model  # noqa: E402
from snap_amd.configs import train_localization  # noqa: E402
from snap_amd.data import synthetic  # noqa: E402
from snap_amd.models import bev_localizer  # noqa: E402


def rel(got, want):
  got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
  return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def compare(pred, ref, args, t_hip, t_cpu):
  sv, rsv = pred['map']['streetview'], ref['map']['streetview']
  vg = sv['feature_volume'].valid.cpu().numpy()
  vw = rsv['feature_volume']['valid']
  mism = vg != vw
  scores_g = pred['scores_poses'].cpu().numpy()
  scores_w = ref['scores_poses']
  gi, wi = int(np.argmax(scores_g[0, 1:])), int(np.argmax(scores_w[0, 1:]))
  out = {
      'workload': f'one C2 scene: {args.views} views @{args.image}px + aerial, 128x128x60 voxels, R50, '
                  f'{scores_g.shape[1] - 1} pose hypotheses',
      'image_features_rel_err': rel(sv['image_feature_pyramid'].features[-1],
                                    rsv['image_feature_pyramid']['features'][-1]),
      'voxel_validity_mismatch_fraction': float(mism.mean()),
      # (engine '<math>+plane': the volume is not materialised -- its vertical max is compared)
      'feature_volume_rel_err': (None if not getattr(sv['feature_volume'], 'materialized', True) else
                                 rel(sv['feature_volume'].features.cpu().numpy()[~mism],
                                     rsv['feature_volume']['features'][~mism])),
      'streetview_plane_rel_err': rel(sv['feature_plane'].features, rsv['feature_plane']['features']),
      'streetview_plane_valid_equal': bool(np.array_equal(sv['feature_plane'].valid.cpu().numpy(),
                                                          rsv['feature_plane']['valid'])),
      'aerial_plane_rel_err': rel(pred['map']['aerial']['feature_plane'].features,
                                  ref['map']['aerial']['feature_plane']['features']),
      'map_bev_matching_max_abs_err': float(np.abs(pred['map']['bev_matching'].features.cpu().numpy()
                                                   - ref['map']['bev_matching']['features']).max()),
      'query_bev_matching_max_abs_err': float(np.abs(pred['query']['bev_matching'].features.cpu().numpy()
                                                     - ref['query']['bev_matching']['features']).max()),
      'scores_poses_rel_err': rel(scores_g, scores_w),
      'pose_argmax_hip': gi, 'pose_argmax_oracle': wi, 'pose_argmax_equal': gi == wi,
      'oracle_score_gap_at_hip_argmax': float((scores_w[0, 1 + wi] - scores_w[0, 1 + gi])
                                              / max(abs(scores_w[0, 1 + wi]), 1e-30)),
      'tolerance': 'north star: feature maps within 1e-3 (unit-norm matching features: absolute), pose argmax identical',
      'hip_seconds_incl_first_call': round(t_hip, 2), 'oracle_seconds': round(t_cpu, 1),
      'host_cores': os.cpu_count(),
  }
  if args.eval:
    lat_g = pred['scores_grid_refine'].cpu().numpy()
    lat_w = ref['scores_grid_refine']
    out['refine_lattice_rel_err'] = rel(lat_g, lat_w)
    out['refine_argmax_hip'] = int(np.argmax(lat_g[0]))
    out['refine_argmax_oracle'] = int(np.argmax(lat_w[0]))
    out['refine_argmax_equal'] = out['refine_argmax_hip'] == out['refine_argmax_oracle']
    tg, tw = pred['map_t_query'], ref['map_t_query']
    out['refined_pose_abs_err'] = [float(abs(float(tg.angle[0]) - float(tw.angle[0]))),
                                   float(np.abs(tg.t[0].cpu().numpy() - tw.t[0]).max())]
  return out


def run(maths=('f32',), views=4, image=512, eval_mode=False, seed=21):
  """HIP forward of one C2 scene per engine in `maths` + ONE oracle run; returns
  {'per_math': {engine: deviations}, 'pred': {engine: pred}, 'ref': oracle pred, ...}.
  An engine name may carry '+plane': bev_mapper.materialize_volume = False (on bf16x3 the fusion
  MLP and the vertical max pooling then run as one kernel, ops.mlp2_pool_max)."""
  from snap_amd import ops
  from snap_amd.utils import geometry as _geo
  dev = torch.device('cuda')
  cfg = train_localization.get_config().model
  if eval_mode:
    from snap_amd.configs import eval_localization
    cfg.update(eval_localization.get_config().model)
  meta = synthetic.meta_data(0.2, (25.6, 25.6, 12))
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  variables = loc.init(0, device='cpu')
  batch = synthetic.make_batch(1, meta['grid'], views, (image, image), seed=seed)
  params_dev = helpers.params_to_device(variables['params'], dev)
  batch_dev = helpers.batch_to_device(batch, dev)
  preds, t_hip, paths = {}, {}, {}
  inject = None
  try:
    for math in maths:
      loc.engine = math.split('+')[0]
      cfg.bev_mapper.materialize_volume = not math.endswith('+plane')
      t0 = time.perf_counter()
      # every engine scores the SAME hypotheses (those the first engine's sampler drew), so that
      # one oracle run checks all of them
      preds[math] = loc.apply({'params': params_dev}, batch_dev, train=False, rngs={'sampling': 11},
                              debug=True, pose_samples=inject)
      torch.cuda.synchronize()
      t_hip[math] = time.perf_counter() - t0
      paths[math] = getattr(loc.bev_mapper.streetview_encoder, 'last_projection_path', None)
      if inject is None:
        smp = preds[math]['map_t_query_samples']
        inject = _geo.Transform2D(smp.angle[:, 1:].contiguous(), smp.t[:, 1:].contiguous())
  finally:
    cfg.bev_mapper.materialize_volume = True
  ps = o_geo.Transform2D(inject.angle.cpu().numpy(), inject.t.cpu().numpy())
  t0 = time.perf_counter()
  ob = helpers.batch_to_oracle(batch)
  ref = o_model.bev_localizer(
      helpers.params_to_numpy(variables['params']), cfg, {'streetview_hfov_deg': 72.0},
      o_grids.Grid2D(meta['grid'].extent[:2], 0.2), ob, pose_samples=ps, keep_sim=False)
  t_cpu = time.perf_counter() - t0

  class _A:   # the fields compare() reads
    pass
  args = _A()
  args.views, args.image, args.eval = views, image, eval_mode
  results = {m: compare(preds[m], ref, args, t_hip[m], t_cpu) for m in maths}
  return {'per_math': results, 'pred': preds, 'ref': ref, 'oracle_batch': ob, 'cfg': cfg, 'projection_path': paths}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--out', default=None)
  ap.add_argument('--views', type=int, default=4)
  ap.add_argument('--image', type=int, default=512)
  ap.add_argument('--math', default='f32',
                  help="comma list of conv engines: f32 | bf16x6 | bf16x3 (ops.MATMUL_PRECISION)")
  ap.add_argument('--eval', action='store_true',
                  help='eval_localization.py overrides: 20 000 hypotheses + the 41^3 refinement lattice')
  ap.add_argument('--seeds', default='21',
                  help='comma list of scene seeds (one oracle run each); more than one: a summary per seed')
  args = ap.parse_args()
  maths = args.math.split(',')
  seeds = [int(x) for x in args.seeds.split(',')]
  if len(seeds) > 1:
    keys = ('image_features_rel_err', 'streetview_plane_rel_err', 'map_bev_matching_max_abs_err',
            'scores_poses_rel_err', 'voxel_validity_mismatch_fraction', 'pose_argmax_equal',
            'refine_argmax_equal')
    out = {'workload': 'one C2 scene per seed', 'per_seed': {}}
    for sd in seeds:
      res = run(maths, args.views, args.image, args.eval, seed=sd)
      out['per_seed'][str(sd)] = {m: {k: r[k] for k in keys if k in r} for m, r in res['per_math'].items()}
    out['all_argmax_equal'] = all(r['pose_argmax_equal'] for s_ in out['per_seed'].values() for r in s_.values())
  else:
    res = run(maths, args.views, args.image, args.eval, seed=seeds[0])
    out = res['per_math'][maths[0]] if len(maths) == 1 else {'per_math': res['per_math']}
  line = json.dumps(out)
  print(line)
  if args.out:
    with open(args.out, 'w') as f:
      f.write(line + '\n')


if __name__ == '__main__':
  main()
