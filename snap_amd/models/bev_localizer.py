"""BEV localizer: relative pose between a query view and a map scene.

Mirrors ``snap/models/bev_localizer.py:36-278`` (``build_query_frustum_grid``,
``BEVLocalizer``, ``BEVLocalizerModel``).
"""
import math

import numpy as np
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import bev_mapper
from snap_amd.models import pose_estimation
from snap_amd.models import types
from snap_amd.utils import geometry
from snap_amd.utils import grids


def build_query_frustum_grid(cell_size, depth, filter_points_in_fov=False, hfov_deg=None):
  """Gravity-aligned grid bounding the query frustum (bev_localizer.py:36-55).

  Host-side constant (numpy -> fp32 torch on CPU); returns (grid, grid_p_view, q_xy_p).
  """
  width = 3 * depth // 2
  grid = grids.Grid2D.from_extent_meters((width, depth), cell_size)
  grid_p_view = torch.tensor([width / 2, 0.0], dtype=torch.float32)
  qgrid_xy_p = grid.index_to_xyz(grid.grid_index().to(torch.float32))
  q_xy_p = qgrid_xy_p - grid_p_view
  if filter_points_in_fov:
    angle = torch.atan2(q_xy_p[..., 0], q_xy_p[..., 1])
    keep = torch.abs(angle) < math.radians(hfov_deg / 2)
    q_xy_p = q_xy_p[keep][:, None]
  return grid, grid_p_view, q_xy_p


_CONSTS = {}


def _const(values, like):
  """Small constant vector on ``like``'s device, uploaded once (a host -> device copy inside a step stalls the
  launch queue)."""
  key = (tuple(values), like.dtype, like.device)
  t = _CONSTS.get(key)
  if t is None:
    t = _CONSTS[key] = torch.tensor(values, dtype=like.dtype, device=like.device)
  return t


class BEVLocalizer(base.Module):
  """Estimate the relative pose between a pair of overlapping scenes."""

  def __init__(self, config, scene_config, grid_map, semantic_map_classes=None,
               dtype=torch.float32):
    self.config = config
    self.scene_config = scene_config
    self.grid_map = grid_map
    hfov = (
        scene_config['streetview_hfov_deg'] if isinstance(scene_config, dict)
        else scene_config.streetview_hfov_deg
    )
    self.grid_query, self.qgrid_p_q, self.q_xy_p = build_query_frustum_grid(
        grid_map.cell_size, config.query_frustum_depth,
        config.filter_points_in_fov, hfov,
    )
    if self.q_xy_p.dim() != 3 or self.q_xy_p.shape[1] != 1:
      # the reference squeezes axis 2 of [B,Nq,1,2] (bev_localizer.py:151):
      # only the FoV-filtered point list is a valid configuration.
      raise ValueError('filter_points_in_fov=True is required (as in train_localization)')
    if config.add_confidence_map:
      raise NotImplementedError('Map confidence is not yet supported.')
    if config.add_confidence_query or config.add_confidence_map:
      config.bev_mapper.add_confidence = True      # as the reference's setup (bev_localizer.py:90-91)
    self.bev_mapper = bev_mapper.BEVMapper(
        config.bev_mapper, grid_map, semantic_map_classes, dtype
    )
    self.bev_mapper_query = None
    if config.bev_mapper_query is not None:
      self.bev_mapper_query = bev_mapper.BEVMapper(
          config.bev_mapper_query, grid_map, semantic_map_classes, dtype
      )

  def init_params(self, gen, device):
    params = {'bev_mapper': self.bev_mapper.init_params(gen, device)}
    if self.bev_mapper_query is not None:
      params['bev_mapper_query'] = self.bev_mapper_query.init_params(gen, device)
    if self.config.add_temperature:
      params['temperature'] = torch.tensor(
          float(self.config.init_temperature), device=device
      )
    return params

  def recover_dense_feature_plane(self, plane_sparse):
    """Sparse FoV point list -> dense query-frustum plane (bev_localizer.py:110-128)."""
    dev = plane_sparse.features.device
    D = plane_sparse.features.shape[-1]
    feats = torch.zeros((*self.grid_query.extent, D), device=dev)
    valid = torch.zeros(self.grid_query.extent, dtype=torch.bool, device=dev)
    q_xy_p = base.device_const('q_xy_p_flat', dev, lambda: self.q_xy_p.squeeze(1), owner=self)
    idx = self.grid_query.xyz_to_index(
        q_xy_p + base.device_const('qgrid_p_q', dev, lambda: self.qgrid_p_q[:2], owner=self))
    valid[idx[:, 0], idx[:, 1]] = plane_sparse.valid.reshape(len(idx))
    feats[idx[:, 0], idx[:, 1]] = plane_sparse.features.reshape(len(idx), -1)
    return types.FeaturePlane(features=feats, valid=valid)

  def _encode_views_jointly(self, params, data_map, data_query, train, ctx):
    """Map and query views share the StreetView image encoder (same parameters,
    per-image GroupNorm): when their image sizes agree, encode all B*(V+1) views in
    ONE batch (bigger GEMM M, fewer launches) and hand each mapper its slice through
    the ``image_feature_pyr`` input of StreetViewEncoder (streetview_encoder.py:218)."""
    sv = self.bev_mapper.streetview_encoder
    if self.bev_mapper_query is not None or sv is None:
      return
    if 'image_feature_pyr' in data_map or 'image_feature_pyr' in data_query:
      return
    im, iq = data_map['images'], data_query['images']
    if im.shape[2:] != iq.shape[2:]:
      return
    B, V = im.shape[:2]
    Vq = iq.shape[1]
    # (a LIST: the encoder pads each set straight into its slice of the joint batch -- no concatenation pass)
    both = [im.reshape(B * V, *im.shape[2:]), iq.reshape(B * Vq, *iq.shape[2:])]
    pyr = sv.image_encoder(
        params['bev_mapper']['streetview_encoder']['image_encoder'], both, train, ctx=ctx
    )
    data_map['image_feature_pyr'] = types.FeatureImagePyramid(
        features=[f[: B * V].reshape(B, V, *f.shape[1:]) for f in pyr.features],
        strides=pyr.strides,
    )
    data_query['image_feature_pyr'] = types.FeatureImagePyramid(
        features=[f[B * V:].reshape(B, Vq, *f.shape[1:]) for f in pyr.features],
        strides=pyr.strides,
    )

  def _prefetch_scale(self, params):
    """Queue the D2H read of exp(temperature) at the START of an apply, inference and training alike
    (``ops.prefetch_exp``: same value, same bits as a blocking ``float(torch.exp(t))``)."""
    if self.config.add_temperature:
      ops.prefetch_exp(params.get('temperature'))

  def similarity(self, params, f_p_q, plane_map, valid_points, want_prob=False, conf_p=None):
    """bev_localizer.py:157-173: sim_points (+ softmax statistics) on the GPU.  ``conf_p`` [B,Nq]:
    query-point confidences (add_confidence_query, :165-168): their masked softmax weights
    replace the 1 / num_valid normalisation and become the sampler's point distribution."""
    cfg = self.config
    temperature = params['temperature'] if cfg.add_temperature else None
    num_valid = valid_points.sum(-1).clamp(min=1).to(torch.float32)
    fq, fm = f_p_q.contiguous(), plane_map.features.contiguous()
    clip = bool(cfg.clip_negative_scores)
    weights = row_cdf = None
    train_w = conf_p is not None and base.needs_grad(fq, fm, temperature, conf_p)
    if conf_p is not None:
      if train_w:      # the weights carry gradient back into the confidence head
        weights, row_cdf = ag.masked_softmax_rows(conf_p, valid_points)
      else:
        weights, row_cdf = ops.masked_softmax_rows(conf_p.contiguous(), valid_points.contiguous())
    if train_w:
      sim, stats, prob, scale = ag.sim_softmax_weighted(fq, fm, temperature, weights, clip, num_valid,
                                                        want_prob)
      weights, row_cdf = weights.detach(), row_cdf.detach()
    elif base.needs_grad(fq, fm, temperature):
      sim, stats, prob, scale = ag.sim_softmax(fq, fm, temperature, clip, num_valid, want_prob)
    else:
      # exp(temperature) is a kernel ARGUMENT (host scalar).  It was read back at the start of this
      # apply (``_prefetch_scale``: 4 bytes into pinned memory, queued in front of the encoders), so
      # waiting for its event does not drain the stream: the apply has no blocking host sync.
      scale = 1.0 if temperature is None else ops.host_exp(temperature)
      sim, stats, prob, _ = ops.sim_softmax(fq, fm, scale, clip, num_valid, want_prob=want_prob,
                                            row_weight=weights)
    # the sampler sees stop_gradient(prob_points) (bev_localizer.py:178): detached inputs.
    # x = sim * row_unscale: lets the sampler read the selected chunk's scores back from sim
    if weights is not None:
      row_unscale = torch.where(weights > 0, 1.0 / weights.clamp(min=1e-38), torch.zeros_like(weights))
    else:
      row_unscale = num_valid[:, None].expand(-1, fq.shape[1]).contiguous()
    matching = dict(fq=fq.detach(), fm=fm.detach(), chunk_stats=stats, scale=scale, clip=clip,
                    row_cdf=row_cdf, sim=sim.detach(), row_unscale=row_unscale.contiguous())
    return sim, prob, matching

  def __call__(self, params, data, train=False, debug=False, rng=None,
               pose_samples=None):
    cfg = self.config
    ctx = base.ForwardContext()
    dev = data['query']['images'].device
    batch_size = len(data['query']['images'])
    q_xy_p = base.device_const('q_xy_p', dev, lambda: self.q_xy_p, owner=self)[None].expand(
        batch_size, -1, -1, -1)

    pred = {}
    # shallow copies: the mappers add keys (xyz_query, image_feature_pyr) that must
    # not leak into the caller's batch (under jax.jit the reference's dict mutation
    # at bev_mapper.py:196 is likewise invisible to the caller).
    data_map, data_query = dict(data['map']), {**data['query'], 'xy_bev': q_xy_p}
    self._prefetch_scale(params)
    self.bev_mapper.start_aerial(params['bev_mapper'], data_map, train, ctx)   # (second stream)
    try:
      self._encode_views_jointly(params, data_map, data_query, train, ctx)
      pred['map'] = self.bev_mapper(params['bev_mapper'], data_map, train, debug, ctx=ctx, rng=rng)
    finally:
      pending = data_map.pop('_aerial_async', None)
      if pending is not None:      # an exception before the join: no side-stream work is left behind
        torch.cuda.current_stream().wait_event(pending[1])
    mapper_q = self.bev_mapper_query or self.bev_mapper
    params_q = params['bev_mapper_query'] if self.bev_mapper_query is not None else params['bev_mapper']
    pred['query'] = mapper_q(
        params_q, data_query, train, debug, is_query=True, ctx=ctx, rng=rng,
    )

    plane_map = pred['map']['bev_matching']
    plane_q = pred['query']['bev_matching']
    q_xy_p = q_xy_p.squeeze(2).contiguous()
    valid_points = plane_q.valid.reshape(batch_size, -1)
    f_p_q = plane_q.features.reshape(batch_size, -1, plane_q.features.shape[-1])

    conf_p = None
    if cfg.add_confidence_query:
      conf_p = pred['query']['bev_confidence'].reshape(batch_size, -1)
    sim_points, prob_points, matching = self.similarity(
        params, f_p_q, plane_map, valid_points, want_prob=debug, conf_p=conf_p
    )
    if debug:
      pred['sim_points'] = sim_points
      pred['prob_points'] = prob_points

    if pose_samples is None:
      if cfg.num_pose_samples is None:
        raise ValueError('config.num_pose_samples must be set')
      m_t_q, corr = pose_estimation.sample_transforms_ransac_batched(
          rng, matching, q_xy_p, cfg.num_pose_samples,
          cfg.num_pose_sampling_retries, self.grid_map,
      )
      if debug:
        pred['correspondences'] = corr
    else:
      m_t_q = pose_samples
    m_t_q_gt = data.get('T_query2map')
    if m_t_q_gt is not None:
      m_t_q_gt = geometry.Transform2D.from_Transform3D(m_t_q_gt)
      m_t_q = geometry.Transform2D.cat([m_t_q_gt[:, None], m_t_q], 1)
    pred['map_t_query_samples'] = m_t_q

    pred['scores_poses'] = scores = pose_estimation.pose_scoring_many_batched(
        m_t_q, sim_points, q_xy_p, valid_points, plane_map.valid, self.grid_map,
        cfg.mask_score_out_of_bounds,
    )
    start_idx = int(m_t_q_gt is not None)
    pred['best_index'] = best_idx = ops.argmax_rows(scores, start_idx)
    bi = torch.arange(batch_size, device=dev)
    pred['map_t_query'] = m_t_q[bi, best_idx.to(torch.int64) + start_idx]

    if cfg.do_grid_refinement:
      pred['map_t_query_ransac'] = pred['map_t_query']
      pred['map_t_query'], pred['scores_grid_refine'] = (
          pose_estimation.grid_refinement_batched(
              pred['map_t_query'], sim_points, q_xy_p, valid_points,
              plane_map.valid, self.grid_map, cfg.mask_score_out_of_bounds,
              max_point_norm=float(self.q_xy_p.norm(dim=-1).max()),      # (a host constant: the query frustum)
          )
      )
    return pred

  default_config = staticmethod(default_configs.bev_localizer)


class BEVLocalizerModel(base.BaseModel):
  """Trainer-facing wrapper (bev_localizer.py:228-278)."""

  def build_flax_model(self):
    meta = self.dataset_meta_data
    return BEVLocalizer(
        self.config, meta['build_config'].scene_config, meta['grid'].bev(),
        meta.get('semantic_map_classes'), self.dtype,
    )

  @classmethod
  def default_flax_model_config(cls):
    return default_configs.bev_localizer()

  def loss_metrics_function(self, pred, data, model_params=None):
    """NLL over sampled poses + recall metrics (bev_localizer.py:244-278).

    Host-side torch on [B, P] tensors (trivial cost).
    """
    scores = pred['scores_poses']
    m_t_q_gt = geometry.Transform2D.from_Transform3D(data['T_query2map'])
    samples_t_gt = pred['map_t_query_samples'].inv @ m_t_q_gt[..., None]
    dr_samples, dt_samples = samples_t_gt.magnitude()
    if self.config.threshold_remove_accurate_poses is not None:
      dr_min, dt_min = self.config.threshold_remove_accurate_poses
      remove = (dr_samples < dr_min) & (dt_samples < dt_min)
      remove[..., 0] = False
      scores = torch.where(remove, torch.full_like(scores, -math.inf), scores)
    nll = -torch.log_softmax(scores, dim=-1)[..., 0]
    losses = {'localization/nll': nll, 'total': nll}
    dr, dt = (pred['map_t_query'].inv @ m_t_q_gt).magnitude()
    metrics = {
        'loc/err_max_position': dt,
        'loc/err_max_rotation': dr,
        'loc/recall_top1': torch.argmax(pred['scores_poses'], dim=-1) == 0,
    }
    # (the threshold families as one comparison each: the dict entries are rows of the stacked results)
    ts = [0.5, 1, 2, 5]
    tt = _const(ts, dt).reshape(-1, *([1] * dt.dim()))
    rec_t, rec_r = dt[None] < tt, dr[None] < tt
    for i, t in enumerate(ts):
      metrics[f'loc/recall_max_{t}m'] = rec_t[i]
      metrics[f'loc/recall_max_{t}°'] = rec_r[i]
    if self.config.add_temperature and model_params is not None:
      metrics['loc/temperature'] = model_params['temperature'].repeat(len(nll))
    pairs = [(0.5, 1), (1, 2), (2, 4)]
    shape = (-1, *([1] * dt_samples.dim()))
    td = _const([p[0] for p in pairs], dt_samples).reshape(shape)
    tr = _const([p[1] for p in pairs], dr_samples).reshape(shape)
    recall = ((dr_samples[None] < tr) & (dt_samples[None] < td))[..., 1:].to(torch.float32).mean(-1)
    for i, (dt_thresh, dr_thresh) in enumerate(pairs):
      metrics[f'loc/recall_samples_{dt_thresh}m_{dr_thresh}°'] = recall[i]
    return losses, metrics
