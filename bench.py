"""Benchmark of the SNAP localisation hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one ``BEVLocalizer`` forward (map BEV from 4 StreetView views + aerial,
query BEV, point-vs-map similarity, RANSAC pose sampling, pose scoring, argmax)
over one batch of synthetic scenes resident in HBM.  Workload (BASELINE.json
configs[1], "C2"): 8 scenes per GPU, 4 StreetView views @512 px + aerial tile,
128x128 BEV (25.6 m @ 0.2 m, 60 height levels), ResNet-50 encoders, 10 000 (+1 GT)
pose samples x 8 retries.  Scenes are independent: N GPUs = N ranks, each with its
own 8-scene batch, no data-path collective ("weak" scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings),
including ``roofline`` (dominant kernel, HIP-event timed inside the timed region)
and ``cpu_baseline`` (the numpy oracle timed on the host cores on a bounded sample).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from snap_amd import ops  # noqa: E402
from snap_amd.configs import train_localization  # noqa: E402
from snap_amd.data import synthetic  # noqa: E402
from snap_amd.models import bev_localizer  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (train --precision bf16: operands staged from f32 HBM tensors)
SPLIT_PRODUCTS = {'conv_split_bf16x6': 6, 'conv_split_bf16x3': 3, 'mlp2_pool_bf16x3': 3}
INFER_DTYPE = {
    'f32': 'f32',
    'bf16x6': 'f32 (tensors, accumulation, every non-GEMM op); conv / dense products f32-grade on the bf16 '
              'matrix cores: operands split into 3 bf16 parts, 6 part products per MAC, ~2^-24 per product',
    'bf16x3': 'f32 (tensors, accumulation, every non-GEMM op); conv / dense products on the bf16 matrix '
              'cores: operands split into 2 bf16 parts, 3 part products per MAC, ~2^-17 per product',
}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak

WORKLOADS = {
    # name: (scenes/GPU, views, image px, grid metres, encoder args, pose samples, retries)
    'c2': dict(batch=8, views=4, image=512, grid=(25.6, 25.6, 12), tiny=False,
               desc='C2: 8 scenes/GPU, 4 StreetView views @512px + aerial, 128x128x60 '
                    'voxel BEV @0.2m, ResNet-50 encoders, 10001 pose hypotheses'),
    'c3': dict(batch=4, views=4, image=512, grid=(25.6, 25.6, 12), tiny=False,
               desc='C3: train_localization default model, batch_size 32 over 8 GPUs = 4 scenes/GPU, '
                    '4 StreetView views @512px + aerial, 128x128x60 voxel BEV, ResNet-50 encoders'),
    'c5': dict(batch=4, views=4, image=512, grid=(25.6, 25.6, 12), tiny=False, vit=True,
               desc='C5: ViT-B/16 StreetView image encoder (bf16 matrix-core GEMMs + fused attention; '
                    'not in the reference: build-only), ResNet-50 aerial, 128x128x60 voxel BEV, '
                    'batch 32 over 8 GPUs = 4 scenes/GPU, forward'),
    'c4': dict(batch=1, c4=True, tiny=False,
               desc='C4: eval_localization pose-estimation path on a 256x256 BEV map (D = 32): exhaustive '
                    '(x, y, theta) voting with 36 yaw hypotheses (rotated templates correlated with the edge-padded map, '
                    '[36, 511, 511] scores) + point-vs-map similarity for the 4652 frustum points + scoring '
                    'of 20 001 pose hypotheses + the 41^3 refinement lattice; one scene per step'),
    'tiny': dict(batch=2, views=3, image=64, grid=(6.4, 6.4, 12), tiny=True,
                 desc='tiny plumbing workload (tests only)'),
}


def build(workload, device, rank, materialize_volume=True):
  w = WORKLOADS[workload]
  meta = synthetic.meta_data(0.2, w['grid'])
  if w['tiny']:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    cfg = helpers.tiny_localizer_config()
  else:
    cfg = train_localization.get_config().model
    if w.get('vit'):
      from snap_amd.configs import defaults
      vcfg = defaults.image_encoder('vit')
      vcfg.output_dim = cfg.bev_mapper.streetview_encoder.image_encoder.output_dim
      cfg.bev_mapper.streetview_encoder.image_encoder = vcfg
  cfg.bev_mapper.materialize_volume = bool(materialize_volume)
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, meta['grid'].bev())
  variables = loc.init(0, device='cpu')
  variables = {'params': _to(variables['params'], device)}
  batch = synthetic.make_batch(
      w['batch'], meta['grid'], w['views'], (w['image'], w['image']), seed=100 + rank,
      device=device,
  )
  return loc, cfg, meta, variables, batch


def build_c4(device, rank, method='auto'):
  """Synthetic 256^2 planes for the eval pose-estimation path (unit-norm random
  features smoothed with a 2-cell Gaussian so that the correlation peak is unique)."""
  import math
  from snap_amd.models import pose_exhaustive_voting as pev
  from snap_amd.models import pose_estimation
  from snap_amd.models import types
  from snap_amd.utils import geometry, grids
  H, D, R, P, cell = 256, 32, 36, 20001, 0.2
  g = torch.Generator(device='cpu').manual_seed(300 + rank)

  def plane():
    f = torch.randn(1, D, H, H, generator=g)
    k = torch.exp(-(torch.arange(-6, 7).float() ** 2) / (2 * 2.0 ** 2))
    k = k / k.sum()
    f = torch.nn.functional.conv2d(f, k.reshape(1, 1, 13, 1).expand(D, 1, 13, 1), padding=(6, 0), groups=D)
    f = torch.nn.functional.conv2d(f, k.reshape(1, 1, 1, 13).expand(D, 1, 1, 13), padding=(0, 6), groups=D)
    f = torch.nn.functional.normalize(f[0].permute(1, 2, 0), dim=-1)
    return f.contiguous().to(device)

  fm, fqp = plane(), plane()
  ones = torch.ones(H, H, dtype=torch.bool, device=device)
  grid = grids.Grid2D((H, H), cell)
  _, _, q_xy_p = bev_localizer.build_query_frustum_grid(cell, 16.0, True, 72.0)
  q_xy = q_xy_p[:, 0].to(device)[None].contiguous()                    # [1, 4652, 2]
  Nq = q_xy.shape[1]
  q_norm_max = float(q_xy_p.norm(dim=-1).max())
  fq = torch.nn.functional.normalize(torch.randn(1, Nq, D, generator=g), dim=-1).to(device)
  poses = torch.stack([torch.rand(1, P, generator=g) * 2 * math.pi,
                       torch.rand(1, P, generator=g) * H * cell,
                       torch.rand(1, P, generator=g) * H * cell], -1).to(device).contiguous()
  vq = torch.ones(1, Nq, dtype=torch.bool, device=device)
  nv = torch.full((1,), float(Nq), device=device)
  scale = math.exp(2.0)

  # k15 direct-form flops and algorithmic bytes at 256^2, R = 36, Dm = 32
  algo = dict(voting_flops=2.0 * R * (2 * H - 1) ** 2 * H * H * D,
              voting_bytes=4.0 * (R * H * H * D + H * H * D + R * (2 * H - 1) ** 2) + (R * H * H + H * H),
              scoring_bytes=4.0 * Nq * H * H)

  def step(i):
    # (the outer region times the WHOLE voting call: rotate, pad, stack, correlate, count, finalize)
    with ops._region('exhaustive_voting_total', algo['voting_flops'], algo['voting_bytes']):
      votes = pev.exhaustive_pose_voting(types.FeaturePlane(fqp, ones), types.FeaturePlane(fm, ones), R, grid,
                                         method=method)
    sim, _, _, _ = ops.sim_softmax(fq, fm[None], scale, True, nv)
    scores = ops.pose_score(sim, poses, q_xy, vq, ones[None], cell)
    best = ops.argmax_rows(scores).to(torch.int64)
    init = geometry.Transform2D.from_packed(poses[torch.arange(1, device=device), best])
    # (max_point_norm: the frustum's farthest point, a host constant -- as BEVLocalizer passes it)
    refined, lattice = pose_estimation.grid_refinement_batched(init, sim, q_xy, vq, ones[None], grid, False,
                                                               max_point_norm=q_norm_max)
    return dict(votes=votes, scores_poses=scores, map_t_query=refined, scores_grid_refine=lattice)

  return step, algo


def _to(tree, device):
  if isinstance(tree, dict):
    return {k: _to(v, device) for k, v in tree.items()}
  return tree.to(device)


def source_digest():
  """sha256 (16 hex) over the kernel sources + the op dispatch: ties a committed counter file to the
  build it was collected on (.git does not travel to the GPU box, the sources do)."""
  import glob
  import hashlib
  h = hashlib.sha256()
  files = sorted(glob.glob(os.path.join(ROOT, 'snap_amd', 'csrc', '*.h*')))
  files += [os.path.join(ROOT, 'include', 'snap_hip.h'), os.path.join(ROOT, 'snap_amd', 'ops.py')]
  for f in files:
    with open(f, 'rb') as fh:
      h.update(os.path.basename(f).encode() + b'\0' + fh.read())
  return h.hexdigest()[:16]


TRAFFIC_FILES = ('r06_c2_hbm_traffic.json', 'r05_c2_hbm_traffic.json', 'r04_c2_hbm_traffic.json', 'r03_c2_hbm_traffic.json', 'r02_c2_hbm_traffic.json',
                 'r01_c2_hbm_traffic.json')   # newest round first


def _pmc_traffic_record(kernel, launches, workload, default_config=True):
  """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
  (profiles/r0N_c2_hbm_traffic.json, newest round first: FETCH_SIZE / WRITE_SIZE collected in separate
  passes on this workload, gfx950 x2 read correction applied) + where the figure comes from: the file,
  the git head / source digest / launch count recorded in it, and `traffic_stale` = the file was
  collected on other sources or on a step with another launch count of this family than the step
  timed now (the counters are a separate run by construction; this says when they have aged)."""
  if workload != 'c2' or not default_config:     # (the counter passes ran the default configuration)
    return None
  fam = ('conv_split' if kernel.startswith('conv_split') else
         'mlp2_pool' if kernel.startswith('mlp2_pool') else kernel)
  rec, doc, used = None, None, None
  for name in TRAFFIC_FILES:
    try:
      with open(os.path.join(ROOT, 'profiles', name)) as f:
        doc = json.load(f)
      rec = doc['per_step'].get(fam)
    except (OSError, ValueError, KeyError):
      rec = None
    if rec:
      used = name
      break
  if not rec or not launches:
    return None
  rec_launches = rec.get('body_launches', rec.get('launches'))   # (kernels the event regions count)
  digest_now = source_digest()
  stale = (doc.get('source_digest') != digest_now) or (rec_launches is not None and rec_launches != launches)
  return {
      'traffic': round((rec['hbm_read_bytes'] + rec['hbm_write_bytes']) / launches, 1),
      'traffic_source': {'file': 'profiles/' + used, 'git_head': doc.get('git_head'),
                         'source_digest': doc.get('source_digest'), 'source_digest_now': digest_now,
                         'family_launches_per_step': rec_launches, 'launches_timed': launches},
      'traffic_stale': bool(stale),
  }


def _pmc_traffic(kernel, launches, workload, default_config=True):
  rec = _pmc_traffic_record(kernel, launches, workload, default_config)
  return None if rec is None else rec['traffic']


def _sync(device):
  if device.type == 'cuda':
    torch.cuda.synchronize()


def cpu_baseline_full_scene(cfg, meta, workload):
  """The numpy oracle on ONE WHOLE scene of the workload, end to end (encoders, lift, fusion MLP,
  pooling, fusion, similarity, scoring of all P hypotheses): no extrapolation.  ~23 s on the 256
  host cores of the GPU box (BLAS-threaded), inside the 10-30 s the contract asks for."""
  from oracle import geometry as o_geo
  from oracle import grids as o_grids
  from oracle import model as o_model
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers
  w = WORKLOADS[workload]
  grid = meta['grid']
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, grid.bev())
  params = helpers.params_to_numpy(loc.init(0, device='cpu')['params'])
  batch = synthetic.make_batch(1, grid, w['views'], (w['image'], w['image']), seed=7)
  ob = helpers.batch_to_oracle(batch)
  rng = np.random.default_rng(0)
  P = cfg.num_pose_samples
  X, Y = grid.extent[:2]
  ps = o_geo.Transform2D(rng.uniform(0, 6.28, (1, P)).astype(np.float32),
                         rng.uniform(0, X * grid.cell_size, (1, P, 2)).astype(np.float32))
  t0 = time.perf_counter()
  o_model.bev_localizer(params, cfg, {'streetview_hfov_deg': 72.0}, o_grids.Grid2D((X, Y), grid.cell_size),
                        ob, pose_samples=ps)
  dt = time.perf_counter() - t0
  return {
      'value': 1.0 / dt, 'unit': 'scenes/s', 'cores': os.cpu_count(), 'kind': 'port',
      'sample': ('numpy oracle (CPU restatement of the reference algorithm; JAX unavailable offline), '
                 f'BLAS-threaded: ONE WHOLE scene of the workload end to end ({w["views"]} views + query + aerial, '
                 f'{X}x{Y}x60 voxels, {P + 1} pose hypotheses), no extrapolation'),
      'seconds_measured': round(dt, 2),
  }


def cpu_baseline(cfg, meta, workload, budget_s=40.0):
  """Time the numpy oracle (a CPU restatement of the reference algorithm; JAX is
  not installable offline) on a bounded sample of one scene and extrapolate
  linearly to a whole scene.  Reported, never optimised against."""
  from oracle import bev as o_bev
  from oracle import encoder as o_enc
  from oracle import geometry as o_geo
  from oracle import grids as o_grids
  from oracle import lift as o_lift
  from oracle import pose as o_pose
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers

  w = WORKLOADS[workload]
  grid = meta['grid']
  loc = bev_localizer.BEVLocalizer(cfg, meta['build_config'].scene_config, grid.bev())
  params = helpers.params_to_numpy(loc.init(0, device='cpu')['params'])
  batch = synthetic.make_batch(1, grid, w['views'], (w['image'], w['image']), seed=7)
  ob = helpers.batch_to_oracle(batch)
  mcfg = cfg.bev_mapper
  sv_cfg = mcfg.streetview_encoder
  p_sv = params['bev_mapper']['streetview_encoder']
  V = w['views']
  t = {}

  # (1) two StreetView views through R50 + FPN + proj MLP (time per view); x (V + 1 query view).
  n_enc = min(2, V)
  t0 = time.perf_counter()
  pyr = o_enc.image_encoder(p_sv['image_encoder'], sv_cfg.image_encoder, ob['map']['images'][0, :n_enc])
  f_img = pyr['features'][-1]
  proj_cfg = dict(layers=(sv_cfg.feature_dim + sv_cfg.num_scale_bins,), apply_input_activation=True)
  f_img = o_enc.mlp(p_sv['proj_mlp'], proj_cfg, f_img)
  t['encoder_view'] = (time.perf_counter() - t0) / n_enc
  f_img = f_img[:1]
  # (2) aerial tile through its own R50 (stride 1).
  t0 = time.perf_counter()
  o_enc.image_encoder(params['bev_mapper']['aerial_encoder'], mcfg.aerial_encoder,
                      ob['map']['rasters']['rgb'])
  t['encoder_aerial'] = time.perf_counter() - t0
  # (3) lift + fusion MLP + vertical pooling on a slab of BEV columns.
  X, Y = grid.extent[:2]
  xs = max(1, X // 8)
  data = dict(ob['map'])
  xyz = o_bev.build_xyz_query(mcfg, o_grids.Grid2D((X, Y), grid.cell_size), data['T_view2scene'])
  xyz = xyz[:, :xs]
  f_all = np.repeat(f_img[None], V, axis=1) if f_img.ndim == 4 else f_img
  t0 = time.perf_counter()
  stride = pyr['strides'][-1]
  cams = data['camera'].scale((1 / stride[::-1]).astype(np.float32))
  pts = xyz.reshape(1, -1, 3)
  p2d, vis, depth, _ = o_lift.project_points_to_views(data['T_view2scene'], cams, pts)
  f_proj = o_lift.interpolate_views_all(f_all, p2d)
  fd = sv_cfg.feature_dim
  scores = o_lift.interpolate_depth_score(f_proj[..., fd:], depth, sv_cfg.depth_min_max)
  pooled, valid = o_lift.pool_multiview_features(f_proj[..., :fd], vis, scores, False, True)
  vol = o_enc.mlp(p_sv['fusion_mlp'], sv_cfg.fusion, pooled)
  vol = np.where(valid[..., None], vol, 0).reshape(1, xs, Y, -1, fd)
  o_bev.vertical_pooling(mcfg.pooling, vol, valid.reshape(1, xs, Y, -1))
  t['lift_slab'] = time.perf_counter() - t0
  lift_scale = X / xs
  # query lift: Nq columns
  _, _, q_xy = o_pose.build_query_frustum_grid(grid.cell_size, cfg.query_frustum_depth, True, 72.0)
  Nq = q_xy.shape[0]
  # (4) similarity + softmax for a slice of the query points.
  Dm = mcfg.matching_dim
  rng = np.random.default_rng(0)
  nq_s = min(Nq, 512)
  fq = rng.standard_normal((1, nq_s, Dm)).astype(np.float32)
  fm = rng.standard_normal((1, X, Y, Dm)).astype(np.float32)
  t0 = time.perf_counter()
  sim, _ = o_pose.similarity(fq, fm, np.ones((1, nq_s), bool), 2.0, True)
  t['sim_slice'] = time.perf_counter() - t0
  # (5) pose scoring: a slice of poses against the slice of points.
  P = cfg.num_pose_samples + 1
  p_s = min(P, 1024)
  poses = o_geo.Transform2D(
      rng.uniform(0, 6.28, p_s).astype(np.float32),
      rng.uniform(0, X * grid.cell_size, (p_s, 2)).astype(np.float32),
  )
  qxy = q_xy[:nq_s, 0].astype(np.float32)
  t0 = time.perf_counter()
  o_pose.pose_scoring_many(poses, sim[0], qxy, np.ones(nq_s, bool), np.ones((X, Y), bool),
                           o_grids.Grid2D((X, Y), grid.cell_size), False)
  t['score_slice'] = time.perf_counter() - t0

  Z = xyz.shape[3]
  per_scene = (
      t['encoder_view'] * (V + 1) + t['encoder_aerial']
      + t['lift_slab'] * lift_scale * (1 + Nq / (X * Y))
      + t['sim_slice'] * (Nq / nq_s)
      + t['score_slice'] * (Nq / nq_s) * (P / p_s)
  )
  return {
      'value': 1.0 / per_scene,
      'unit': 'scenes/s',
      'cores': os.cpu_count(),
      'kind': 'port',
      'sample': (
          'numpy oracle (CPU restatement of the reference algorithm; JAX unavailable '
          f'offline), BLAS-threaded: {n_enc} of {V + 1} views through R50+FPN, the aerial R50, '
          f'{xs}/{X} of the BEV columns (x{Z} levels) through lift+fusion-MLP+pooling, '
          f'{nq_s}/{Nq} query points for similarity, {p_s}/{P} poses for scoring; '
          'per-scene time extrapolated linearly'
      ),
      'seconds_measured': round(sum(t.values()) + t['encoder_view'] * (n_enc - 1), 2),
      'stage_seconds': {k: round(v, 3) for k, v in t.items()},
  }


def main(argv=None, emit=True):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
  ap.add_argument('--device', default='cuda')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--dump', default=None, help='write the per-launch HIP-event timings of the profiled step (JSON)')
  ap.add_argument('--no-extra-legs', action='store_true',
                  help='skip the short extra legs (f32_exact, volume_materialized, train_c3, c4, c5)')
  ap.add_argument('--dist-backend', default=None,
                  help='testing only: override the process-group backend (default nccl = RCCL)')
  ap.add_argument('--share-gpu', action='store_true',
                  help='testing only: every rank uses GPU 0 (multi-process plumbing check on a 1-GPU box)')
  ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'fp16', 'bf16x3', 'bf16x6'],
                  help="train mode only: 'bf16' rounds the conv / dense operands to bf16 (f32 "
                       "accumulate) -- the analogue of the reference's float16 train config; 'fp16' is "
                       "that config itself (IEEE-half operands + DynamicScale(minimum_scale=256)); the "
                       "inference headline always runs the exact f32 path")
  ap.add_argument('--math', default='bf16x3', choices=['f32', 'bf16x6', 'bf16x3'],
                  help="infer mode: conv / dense engine.  'f32' = exact f32 MFMA (v_mfma_f32_32x32x2_f32); "
                       "'bf16x6' / 'bf16x3' = f32-grade split-bf16 engine (each f32 operand split into 3 / 2 "
                       "bf16 parts, 6 / 3 part products on v_mfma_f32_32x32x16_bf16, f32 accumulate)")
  ap.add_argument('--voting', default='auto', choices=['auto', 'fft', 'direct'],
                  help="c4: exhaustive-voting formulation ('auto' = frequency domain where the map fits its transform "
                       "sizes; 'direct' = the reference's direct-form correlation on the MFMA conv engine)")
  ap.add_argument('--materialize-volume', action='store_true',
                  help='infer mode: also write the dense [B, X, Y, Z, D] StreetView feature volume (an '
                       'intermediate the localisation outputs do not need; under jit the reference drops it '
                       'as dead code).  Default: only its vertical max, the BEV plane, is produced')
  ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                  help='infer: BEVLocalizer forward (the headline metric); train: one '
                       'snap_amd.trainer.train_step (fwd + bwd + grad all-reduce + Adam)')
  ap.add_argument('--in-flight', type=int, default=None,
                  help='infer mode: batches in flight (snap_amd.pipeline; default 2) -- step i runs on HIP stream '
                       'i %% N, so the dispatch gaps and tail waves of one batch are filled by the kernels of the '
                       'next.  Every step does all of its work and gives the same bits; ms_per_step = elapsed / '
                       'steps.  1 = one batch at a time (also reported as the leg `one_batch_at_a_time`)')
  ap.add_argument('--digest', action='store_true',
                  help='testing: add a float64 digest of every timed step\'s pose scores / best index to the line '
                       '(the same whatever --in-flight is)')
  ap.add_argument('--tune', action='append', default=[], metavar='KNOB=VALUE',
                  help='A/B runs: set a tuning switch of snap_amd.ops (e.g. LIFT_IN_CONSUMER=0); results do not '
                       'depend on any of them beyond summation order')
  args = ap.parse_args(argv)
  for kv in args.tune:
    from snap_amd import ops as _ops
    k, v = kv.split('=', 1)
    if k not in _ops._TUNING_DEFAULTS:
      raise SystemExit(f'--tune: snap_amd.ops.Tuning has no switch {k}')
    cur = getattr(_ops, k)
    setattr(_ops, k, v if (cur is None or isinstance(cur, str)) else type(cur)(int(v)))     # (the process default)

  rank = int(os.environ.get('RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
  use_cuda = args.device == 'cuda'
  if use_cuda:
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
  else:
    device = torch.device('cpu')
  if world > 1:
    dist.init_process_group(args.dist_backend or ('nccl' if use_cuda else 'gloo'))

  if not use_cuda:
    args.math = 'f32'        # (--device cpu exists for the gloo plumbing test only: oracle backend)
  is_c4 = bool(WORKLOADS[args.workload].get('c4'))
  if is_c4:
    if args.mode != 'infer':
      raise SystemExit('--workload c4 is an inference (eval) path')
    c4_step, c4_algo = build_c4(device, rank, args.voting)
    loc = cfg = meta = variables = batch = None
  else:
    loc, cfg, meta, variables, batch = build(
        args.workload, device, rank,
        materialize_volume=args.materialize_volume or args.mode != 'infer')
  scenes_per_rank = WORKLOADS[args.workload]['batch']

  if args.mode == 'train':
    from snap_amd import models, trainer
    # the reference's selection (trainer.py:387-397): config.dtype_str -> dtype -> model + DynamicScale;
    # the f32-grade split engines are float32 models with an explicit engine
    dtype_str = {'fp16': 'float16', 'bf16': 'bfloat16'}.get(args.precision, 'float32')
    dtype, dyn_scale = trainer.dtype_and_dynamic_scale(dtype_str)
    model = models.get_model('bev_localizer')(
        cfg, meta, dtype, engine=args.precision if dtype == torch.float32 else None)
    tcfg = train_localization.get_config()
    lr_fn = trainer.make_lr_fn(tcfg.lr_configs['base_learning_rate'], tcfg.num_training_steps)
    state = trainer.TrainState.create(variables['params'], rng=1000 * rank, dynamic_scale=dyn_scale)
    last_logs = {}

    def step(i):
      nonlocal state
      state, _, logs = trainer.train_step(state, batch, model=model, lr_fn=lr_fn)   # (the model's engine)
      last_logs.update(logs)
      return logs
  elif is_c4:
    def step(i):
      with ops.engine_scope(args.math):  # the direct-form correlation (if selected) runs on this engine
        return c4_step(i)
  else:
    loc.engine = args.math               # the arithmetic is a property of the model (models/base.py)

    def step(i):
      return loc.apply(variables, batch, train=False, rngs={'sampling': 1000 * rank + i})

  def barrier():
    if world > 1:
      dist.barrier()
    _sync(device)

  # what a multi-GPU record must show: the process group really has N ranks on the collective
  # library named, every rank sits on its own device and holds its own shard of scenes
  dist_info = {
      'world_size': world, 'rank_device': str(device),
      'backend': dist.get_backend() if world > 1 else None,
      'rccl_version': ('.'.join(str(v) for v in torch.cuda.nccl.version())
                       if use_cuda and hasattr(torch.cuda, 'nccl') else None),
  }
  if world > 1:
    digest = 0.0 if batch is None else float(
        batch['map']['images'].double().sum() + batch['query']['images'].double().sum())
    mine = torch.tensor([float(rank), float(device.index if use_cuda else -1), digest],
                        dtype=torch.float64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    allr = torch.stack(allr).cpu()
    dist_info['ranks'] = [int(v) for v in allr[:, 0]]
    dist_info['rank_devices'] = [int(v) for v in allr[:, 1]]
    dist_info['shard_digests'] = [round(float(v), 3) for v in allr[:, 2]]

  # Warm-up and timed steps follow the SAME allocation pattern (result dropped before the next
  # step starts): the caching allocator then reaches its steady state in the first warm-up
  # step.  Holding the previous result across a step made the allocator fetch fresh 7.5 GiB
  # segments in the middle of the timed loop (hipMalloc of that size: ~200 ms, seen as one
  # 280 ms step in an otherwise 56 ms run).
  # Batches in flight: step i is enqueued on stream i % N (N = 1: torch's current stream).  Every stream
  # keeps the allocation pattern described above for itself (its previous result dropped before its
  # next step starts); the caching allocator pools blocks per stream.
  from snap_amd import pipeline
  nfl = args.in_flight if args.in_flight is not None else (2 if args.mode == 'infer' else 1)
  nfl = max(1, nfl) if (use_cuda and args.mode == 'infer') else 1
  ring = pipeline.BatchesInFlight(nfl, device)
  preds = [None] * nfl
  digests = []

  def run_step(slot, i, mark=None):
    with ring.slot(slot):
      if mark is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        mark.append(ev)
      preds[slot] = None
      preds[slot] = step(i)
      if mark is not None and args.digest and isinstance(preds[slot], dict) and 'scores_poses' in preds[slot]:
        bi = preds[slot].get('best_index')
        digests.append(torch.stack([preds[slot]['scores_poses'].double().sum(),
                                    (bi if bi is not None else preds[slot]['scores_poses'].argmax(-1)).double().sum()]))

  for i in range(args.warmup * nfl):
    run_step(i % nfl, i)
  # a generational GC pass over the Python heap in the middle of the loop costs 100+ ms
  # (seen as one 280 ms step in an otherwise 56 ms run): collect now, pause it while timing
  import gc
  gc.collect()
  gc.disable()
  barrier()
  prof = None
  marks = []
  t0 = time.perf_counter()
  host_enq = []        # host time per step of this loop: the enqueue cost PLUS the launch queue's back-pressure once it is full
                       # (tools/host_time_probe2.py: the same on an idle GPU = the pure enqueue cost, C2 7.4 ms, C5 8.4, C4 0.44)
  for i in range(args.steps):
    th0 = time.perf_counter()
    if use_cuda and rank == 0 and i == args.steps - 1:
      if nfl > 1:
        torch.cuda.synchronize(device)  # the profiled step runs ALONE (its events time single launches)
      prof = ops.KernelProfiler()   # HIP events around every launch of the last step
      ops.set_profiler(prof)
      # ... with the aerial encoder on the MAIN stream for this one step: next to the StreetView
      # kernels on its side stream, two launches share the GPU and BOTH report inflated durations
      # (their sum counts the shared wall time twice), which understates every family's rate
      overlap_prev, ops.OVERLAP_AERIAL = ops.OVERLAP_AERIAL, False
      wgside_prev, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, False     # (train: kernel gradients too)
    run_step(i % nfl, 100_000 + i, marks if use_cuda else None)
    if i < args.steps - 1:
      host_enq.append(time.perf_counter() - th0)
  ops.set_profiler(None)
  if prof is not None:
    ops.OVERLAP_AERIAL = overlap_prev
    ops.WGRAD_SIDE_STREAM = wgside_prev
  if use_cuda:
    ring.join()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
  barrier()
  elapsed = time.perf_counter() - t0
  gc.enable()
  pred = preds[(args.steps - 1) % nfl]
  # (N batches in flight: a mark is a step's START on its own stream; the period of a stream / N is
  #  the per-step time while N batches share the GPU -- the drained last step is left out)
  if nfl == 1:
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
  else:
    step_ms = [marks[i].elapsed_time(marks[i + nfl]) / nfl for i in range(max(0, len(marks) - 2 - nfl))]
  t = torch.tensor([elapsed], dtype=torch.float64, device=device)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  elapsed = float(t.item())
  ms_per_step = 1e3 * elapsed / args.steps
  value = scenes_per_rank * world * args.steps / elapsed

  out = None
  if rank == 0:
    out = {
        'metric': 'localization_scenes_per_sec' if args.mode == 'infer' else 'train_scenes_per_sec',
        'value': round(value, 3),
        'unit': 'scenes/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3),
        'step_ms': {
            'min': round(min(step_ms), 3), 'median': round(float(np.median(step_ms)), 3),
            'max': round(max(step_ms), 3), 'argmax': int(np.argmax(step_ms)),
        } if step_ms else None,
        'host_enqueue_ms_per_step': round(1e3 * float(np.median(host_enq)), 3) if host_enq else None,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': ('bf16 ViT GEMMs + attention (f32 accumulate), f32 elsewhere' if WORKLOADS[args.workload].get('vit')
                  else INFER_DTYPE[args.math] if args.mode == 'infer'
                  else 'f32' if args.precision == 'f32'
                  else INFER_DTYPE[args.precision] + '; kernel gradients on the exact f32 engine' if args.precision in INFER_DTYPE
                  else 'fp16 (IEEE half) GEMM operands and kernel images, f32 accumulate / master parameters / '
                       'optimizer, DynamicScale loss scaling' if args.precision == 'fp16'
                  else 'bf16 GEMM operands, f32 accumulate / parameters / optimizer'),
        'data': 'synthetic',
        'distributed': dist_info,
        'config': {
            'workload': WORKLOADS[args.workload]['desc'],
            'scenes_per_gpu': scenes_per_rank,
            'batches_in_flight': nfl,
            'global_batch': scenes_per_rank * world,
            'parallelism': (f'scene-sharded x{world}, no data-path collective' if args.mode == 'infer'
                            else f'dp{world}: scene-sharded, RCCL gradient all-reduce'),
            'mode': ('exhaustive voting + similarity + pose scoring + grid refinement (eval path)' if is_c4 else
                     'inference forward (BEVLocalizer.apply, train=False)' if args.mode == 'infer' else
                     f'train_step: forward + backward + gradient all-reduce + Adam ({args.precision})'),
            'feature_volume': ('n/a' if is_c4 else 'materialized' if (args.materialize_volume or args.mode != 'infer')
                               else 'lazy: fusion MLP + vertical max pooling fused into the plane; '
                                    'feature_volume.features is produced on first access (not in the timed step)'),
        },
    }
    if digests:
      out['step_digests'] = [[float(v) for v in d.cpu()] for d in digests]
    if args.mode == 'infer' and not is_c4:
      # share of voxels seen by at least one camera: the fusion MLP multiplies only those
      # rows (the others are masked to zero by the reference too), so the step time
      # depends on it -- stated here so the number can be judged against the data.
      try:
        out['config']['observed_voxel_fraction'] = {
            k: round(float(pred[k]['streetview']['feature_volume'].valid.float().mean()), 4)
            for k in ('map', 'query')
        }
      except (KeyError, TypeError, AttributeError):
        pass
    if prof is not None:
      summ = prof.summary()
      kern = {}
      for name, s in summ.items():
        ms = max(s['ms'], 1e-9)
        kern[name] = {
            'launches': s['launches'], 'ms': round(ms, 3),
            'tflops': round(s['flops'] / ms / 1e9, 2),
            'gbs': round(s['bytes'] / ms / 1e6, 1),
        }
      default_cfg = (args.mode == 'infer' and args.math == 'bf16x3' and not args.materialize_volume)
      dom = max((k for k in summ if k != 'exhaustive_voting_total'), key=lambda k: summ[k]['ms'])
      s = summ[dom]
      if s['flops'] > 0:
        ach = s['flops'] / s['ms'] / 1e9
        peak = PEAK_MFMA_BF16_TFLOPS if dom.endswith(('bf16', 'fp16')) else PEAK_MFMA_F32_TFLOPS   # (f16 MFMA: same rate)
        nprod = SPLIT_PRODUCTS.get(dom)
        if nprod:   # algorithmic (f32-equivalent) flops against the bf16 peak / products per MAC
          peak = round(PEAK_MFMA_BF16_TFLOPS / nprod, 1)
        out['roofline'] = {
            'kernel': dom, 'bound': 'mfma', 'achieved': round(ach, 2),
            'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(ach / peak, 4),
            'traffic': _pmc_traffic(dom, s['launches'], args.workload, default_cfg),
            'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, separate passes; profiles/)',
            'algorithmic_bytes_per_launch': round(s['bytes'] / s['launches'], 1),
            'launches': s['launches'], 'avg_launch_ms': round(s['ms'] / s['launches'], 4),
            'flops_per_step': s['flops'],
        }
        if nprod:
          out['roofline']['note'] = (
              f'achieved = algorithmic f32 flops / time; the engine executes {nprod} bf16 MFMA products per '
              f'MAC, so peak = {PEAK_MFMA_BF16_TFLOPS:.0f} / {nprod} TFLOP/s and frac = matrix-pipe utilisation')
      else:
        ach = s['bytes'] / s['ms'] / 1e6
        out['roofline'] = {
            'kernel': dom, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS,
            'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4),
            'traffic': _pmc_traffic(dom, s['launches'], args.workload, default_cfg),
            'launches': s['launches'], 'avg_launch_ms': round(s['ms'] / s['launches'], 4),
        }
      stamp = _pmc_traffic_record(dom, s['launches'], args.workload, default_cfg)
      if stamp is not None:
        out['roofline']['traffic_source'] = stamp['traffic_source']
        out['roofline']['traffic_stale'] = stamp['traffic_stale']
      if dom.startswith('conv_split') and out['roofline'].get('traffic'):
        # the same family against the HBM roofline: most of its launches are 1x1 convolutions with
        # short reductions, which run at the memory system's pace, not the matrix pipe's
        tb = out['roofline']['traffic'] * s['launches']
        gbs = tb / s['ms'] / 1e6
        out['roofline_conv_hbm'] = {
            'kernel': dom, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': round(gbs / PEAK_HBM_GBS, 4), 'traffic_bytes_per_step': tb,
            'note': 'HBM-side bytes of the family (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, profiles/) / its time'}
      if 'pose_score' in summ:
        s = summ['pose_score']
        ach = s['bytes'] / s['ms'] / 1e6
        out['roofline_pose_corr'] = {
            'kernel': 'pose_score', 'bound': 'hbm', 'achieved': round(ach, 1),
            'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4),
            'traffic': _pmc_traffic('pose_score', s['launches'], args.workload, default_cfg),
            'avg_launch_ms': round(s['ms'] / s['launches'], 4),
            'bytes_per_launch': s['bytes'] / s['launches'],
        }
      out['kernels'] = kern
      out['kernels_note'] = ('HIP-event durations of the LAST timed step, which runs ALONE (the other batch in flight '
                             'is drained first) and with the aerial encoder on its main stream: no two launches share '
                             'the GPU while they are timed; the other steps run ' + str(nfl) + ' batches in flight and overlap '
                             'the aerial encoder with the StreetView encoder on a side stream')
      if is_c4:
        # the direct-form correlation is MFMA-bound (AI ~ 1e5 flop/B); report BOTH the
        # matrix-core fraction on its direct-form flops and the HBM fraction its algorithmic bytes
        # would need (what an FFT formulation would be held to)
        conv = [n for n in summ if n.startswith('conv_')]
        vms = summ.get('exhaustive_voting_total', {}).get('ms', 0.0)
        kern.pop('exhaustive_voting_total', None)
        gbs = c4_algo['voting_bytes'] / vms / 1e6 if vms > 0 else 0.0
        if vms > 0 and 'voting_fft' in summ:
          # frequency-domain formulation: ~1e-3 of the direct form's multiply-adds -- the call is held to the
          # HBM roofline on SURVEY 8(d)'s algorithmic bytes (templates + map + scores = what any formulation
          # must move); `pipeline_hbm_bytes` is what THIS pipeline moves on top (its intermediate spectra)
          H_, D_, R_, N_, ld_ = 256, 32, 36, 768, 512
          pipe = (2 * 128.0 * R_ * N_ * H_                     # X1[r][g][k1][j][8]: written, read (templates are sampled
                                                               #  inside the first transform: no template tensor)
                  + 2 * 128.0 * N_ * (3 * H_ - 2) + 128.0 * N_ * N_   # map: Xm1 written / read, Zm written
                  + 2 * 8.0 * R_ * N_ * ld_                    # Y[r][k1][b]: written, read
                  + 4.0 * R_ * (2 * H_ - 1) ** 2)              # scores
          out['roofline'] = {
              'kernel': 'exhaustive_voting (rotate + voting_fft: vf_slow / vf_fast / vf_inv kernels)',
              'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
              'frac': round(gbs / PEAK_HBM_GBS, 5), 'traffic': None, 'ms': round(vms, 3),
              'formulation': 'frequency domain (voting_fft.hip): 768-point mixed-radix LDS FFTs of 8 channel-pair '
                             'columns, spectra multiplied in digit-reversed order, overlap count by rotation pairs',
              'algorithmic_bytes': c4_algo['voting_bytes'],
              'pipeline_hbm_bytes': pipe, 'pipeline_gbs': round(pipe / vms / 1e6, 1),
              'pipeline_frac': round(pipe / vms / 1e6 / PEAK_HBM_GBS, 4),
              'fft_ms': round(summ['voting_fft']['ms'], 3),
              'flops_direct_form': c4_algo['voting_flops'],
              'direct_form_equivalent_tflops': round(c4_algo['voting_flops'] / vms / 1e9, 1),
          }
          out['roofline_voting_hbm'] = {k: out['roofline'][k] for k in
                                        ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'algorithmic_bytes',
                                         'formulation')}
        elif vms > 0:
          nprod = max([SPLIT_PRODUCTS.get(n, 0) for n in conv] + [0])
          peak = PEAK_MFMA_BF16_TFLOPS / nprod if nprod else PEAK_MFMA_F32_TFLOPS
          ach = c4_algo['voting_flops'] / vms / 1e9
          out['roofline'] = {
              'kernel': 'exhaustive_voting (templates + correlation engine ' + '/'.join(conv) + ')',
              'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
              'frac': round(ach / peak, 4), 'traffic': None, 'ms': round(vms, 3),
              'formulation': 'direct form (as the reference: jax.scipy.signal.convolve), shift-stacked',
              'flops_direct_form': c4_algo['voting_flops'],
          }
          out['roofline_voting_hbm'] = {
              'kernel': 'exhaustive_voting', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS,
              'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5), 'algorithmic_bytes': c4_algo['voting_bytes'],
              'note': 'algorithmic bytes of SURVEY 8(d); only a frequency-domain formulation is HBM-bound'}
      dump = args.dump or os.environ.get('SNAP_BENCH_DUMP')
      if dump:
        with open(dump, 'w') as f:
          json.dump({n: prof.launches(n) for n in summ}, f)
    if (args.mode == 'infer' and not is_c4 and use_cuda and world == 1 and not args.no_extra_legs
        and args.math == 'bf16x3' and not args.materialize_volume):
      # The headline configuration is the split-bf16 engine without the dense feature volume.  The
      # same workload in the two reference-shaped configurations, timed here (short legs, same
      # bracketing) so that the driver's line carries them: exact f32 matrix-core arithmetic with
      # every output written ('f32_exact'), and the headline engine with the dense feature volume
      # written ('volume_materialized').
      def leg(math, steps=5, warmup=2):
        prev = (loc.engine, cfg.bev_mapper.materialize_volume)
        loc.engine, cfg.bev_mapper.materialize_volume = math, True
        try:
          p = None
          for i in range(warmup):
            p = None
            p = step(10_000 + i)
          _sync(device)
          t1 = time.perf_counter()
          for i in range(steps):
            p = None
            p = step(20_000 + i)
          _sync(device)
          dt = time.perf_counter() - t1
          vol = p['map']['streetview']['feature_volume']
          assert getattr(vol, 'materialized', True) and vol.features is not None
        finally:
          loc.engine, cfg.bev_mapper.materialize_volume = prev
        return {'ms_per_step': round(1e3 * dt / steps, 3),
                'scenes_per_sec': round(scenes_per_rank * steps / dt, 3),
                'steps': steps, 'warmup': warmup, 'dtype': INFER_DTYPE[math],
                'feature_volume': 'materialized'}
      pred = None
      preds[:] = [None] * nfl
      if nfl > 1:
        # the same workload, same engine, ONE batch at a time on one stream (what `--in-flight 1` measures)
        p = None
        for i in range(3):
          p = None
          p = step(30_000 + i)
        _sync(device)
        t1 = time.perf_counter()
        for i in range(10):
          p = None
          p = step(40_000 + i)
        _sync(device)
        dt = time.perf_counter() - t1
        p = None
        out['one_batch_at_a_time'] = {'ms_per_step': round(1e2 * dt, 3), 'scenes_per_sec': round(scenes_per_rank * 10 / dt, 3),
                                      'steps': 10, 'warmup': 3, 'batches_in_flight': 1}
      out['f32_exact'] = leg('f32')
      out['volume_materialized'] = leg('bf16x3')
      # The other BASELINE configurations as short driver-timed legs of the same command (same
      # bracketing: warm-up, sync, K steps, sync; each leg builds its own workload and reports its
      # dominant kernel against that kernel's roofline): C3 = one train_step (forward + backward +
      # exchange + Adam) on 4 scenes at the training precision, C4 = the eval pose-estimation path
      # at 256^2 / 36 yaw hypotheses, C5 = the ViT-B/16 encoder build.
      loc = variables = batch = None
      dump_env = os.environ.pop('SNAP_BENCH_DUMP', None)     # (the legs must not overwrite the C2 dump)
      for key, leg_args in (
          ('train_c3', ['--mode', 'train', '--workload', 'c3', '--precision', 'bf16', '--steps', '10', '--warmup', '2']),
          # the reference's literal train config: dtype_str = 'float16' + DynamicScale(minimum_scale=256)
          ('train_c3_fp16', ['--mode', 'train', '--workload', 'c3', '--precision', 'fp16', '--steps', '10', '--warmup', '2']),
          ('c4', ['--workload', 'c4', '--steps', '12', '--warmup', '2']),
          ('c5', ['--workload', 'c5', '--steps', '10', '--warmup', '2'])):
        torch.cuda.empty_cache()
        t1 = time.perf_counter()
        try:
          r = main(leg_args + ['--no-cpu-baseline', '--no-extra-legs'], emit=False)
          out[key] = {
              'ms_per_step': r['ms_per_step'], 'scenes_per_sec': r['value'], 'metric': r['metric'],
              'steps': r['steps'], 'warmup': r['warmup'], 'dtype': r['dtype'],
              'workload': r['config']['workload'], 'mode': r['config']['mode'],
              'roofline': {k: r['roofline'].get(k) for k in
                           ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launches', 'avg_launch_ms', 'ms')
                           if k in r.get('roofline', {})},
              'leg_wall_s': round(time.perf_counter() - t1, 2),
          }
          if 'train_logs' in r:
            out[key]['is_finite'] = r['train_logs'].get('is_finite')
            if 'loss_scale' in r['train_logs']:
              out[key]['loss_scale'] = r['train_logs']['loss_scale']
        except Exception as e:   # a leg must never take the headline down
          out[key] = {'error': repr(e)}
      if dump_env is not None:
        os.environ['SNAP_BENCH_DUMP'] = dump_env
    if args.mode == 'train':
      out['train_logs'] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in last_logs.items()}
    if (world == 1 and not args.no_cpu_baseline and not WORKLOADS[args.workload]['tiny']
        and not WORKLOADS[args.workload].get('vit') and args.mode == 'infer' and not is_c4):
      try:
        # a whole scene where the host is big enough to finish it in ~20-30 s (the GPU box: 256
        # cores); the bounded-sample estimate elsewhere
        fn = cpu_baseline_full_scene if (os.cpu_count() or 1) >= 64 else cpu_baseline
        out['cpu_baseline'] = fn(cfg, meta, args.workload)
      except Exception as e:  # the baseline must never take the bench line down
        out['cpu_baseline'] = {'error': repr(e)}
    if emit:
      print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  return out


if __name__ == '__main__':
  main()
