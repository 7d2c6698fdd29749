"""Oracle (test infrastructure): end-to-end BEVLocalizer forward in numpy.

Restates ``snap/models/bev_localizer.py:130-220``.  The RANSAC pose samples are
an INPUT (``pose_samples`` or ``sample_indices``) because JAX's threefry stream
cannot be reproduced ("RNG-dependent outputs").
"""
import numpy as np

from oracle import bev
from oracle import geometry
from oracle import pose


def bev_localizer(params, config, scene_config, grid_map, data,
                  pose_samples=None, sample_indices=None, keep_sim=False):
  """Forward pass.

  Args:
    params: nested dict of numpy arrays (Flax layout).
    config: mapping with the keys of ``defaults.bev_localizer()``.
    scene_config: object/dict with ``streetview_hfov_deg``.
    grid_map: oracle.grids.Grid2D.
    data: batch dict (oracle structs for cameras / transforms).
    pose_samples: optional geometry.Transform2D [B,P] -- sampled map_t_query.
    sample_indices: optional int [B, P*retries*2, 3] correspondences, used when
      pose_samples is None.
  """
  hfov = (
      scene_config['streetview_hfov_deg']
      if isinstance(scene_config, dict)
      else scene_config.streetview_hfov_deg
  )
  dtype = data['query']['images'].dtype
  grid_query, qgrid_p_q, q_xy_p0 = pose.build_query_frustum_grid(
      grid_map.cell_size,
      config['query_frustum_depth'],
      config['filter_points_in_fov'],
      hfov,
      dtype,
  )
  if q_xy_p0.ndim == 2:
    q_xy_p0 = q_xy_p0[:, None]
  elif q_xy_p0.ndim == 3 and not config['filter_points_in_fov']:
    pass  # [W, D, 2] dense frustum grid
  B = len(data['query']['images'])
  q_xy_p = np.repeat(q_xy_p0[None], B, axis=0)

  pred = {}
  mapper_cfg = config['bev_mapper']
  pred['map'] = bev.bev_mapper(params['bev_mapper'], mapper_cfg, grid_map, data['map'])
  q_params = params.get('bev_mapper_query', params['bev_mapper'])
  q_cfg = config.get('bev_mapper_query') or mapper_cfg
  pred['query'] = bev.bev_mapper(
      q_params, q_cfg, grid_map, dict(data['query'], xy_bev=q_xy_p)
  )

  plane_map = pred['map']['bev_matching']
  plane_q = pred['query']['bev_matching']
  q_xy = q_xy_p.reshape(B, -1, 2)
  valid_points = plane_q['valid'].reshape(B, -1)
  f_p_q = plane_q['features'].reshape(B, -1, plane_q['features'].shape[-1])

  temperature = params['temperature'] if config['add_temperature'] else None
  conf_weights = None
  if config.get('add_confidence_query'):
    # bev_localizer.py:165-168
    conf_p = pred['query']['bev_confidence'].reshape(B, -1)
    conf_weights = bev.layers_masked_softmax(conf_p, valid_points, -1)[..., None, None]
    pred['_conf_weights'] = conf_weights[..., 0, 0]
  sim, prob = pose.similarity(
      f_p_q, plane_map['features'], valid_points, temperature,
      config['clip_negative_scores'], conf_weights,
  )
  if keep_sim:
    pred['_sim_points'] = sim
    pred['_prob_points'] = prob

  if pose_samples is None:
    angles, ts = [], []
    for b in range(B):
      tf = pose.poses_from_correspondences(
          sample_indices[b], q_xy[b], config['num_pose_samples'],
          config['num_pose_sampling_retries'], grid_map,
      )
      angles.append(tf.angle)
      ts.append(tf.t)
    pose_samples = geometry.Transform2D(np.stack(angles), np.stack(ts))
  m_t_q = pose_samples
  gt3d = data.get('T_query2map')
  if gt3d is not None:
    gt = geometry.Transform2D.from_Transform3D(gt3d)
    m_t_q = geometry.Transform2D(
        np.concatenate([gt.angle[:, None], m_t_q.angle], 1).astype(dtype),
        np.concatenate([gt.t[:, None], m_t_q.t], 1).astype(dtype),
    )
  pred['map_t_query_samples'] = m_t_q

  scores = np.stack([
      pose.pose_scoring_many(
          m_t_q[b], sim[b], q_xy[b], valid_points[b], plane_map['valid'][b],
          grid_map, config['mask_score_out_of_bounds'],
      )
      for b in range(B)
  ])
  pred['scores_poses'] = scores
  start = int(gt3d is not None)
  best = np.argmax(scores[:, start:], axis=-1)
  pred['best_index'] = best
  bi = np.arange(B)
  pred['map_t_query'] = geometry.Transform2D(
      m_t_q.angle[bi, start + best], m_t_q.t[bi, start + best]
  )
  if config['do_grid_refinement']:
    pred['map_t_query_ransac'] = pred['map_t_query']
    ang, tt, sc = [], [], []
    for b in range(B):
      tf, s = pose.grid_refinement(
          pred['map_t_query'][b], sim[b], q_xy[b], valid_points[b],
          plane_map['valid'][b], grid_map, config['mask_score_out_of_bounds'],
      )
      ang.append(tf.angle)
      tt.append(tf.t)
      sc.append(s)
    pred['map_t_query'] = geometry.Transform2D(np.stack(ang), np.stack(tt))
    pred['scores_grid_refine'] = np.stack(sc)
  return pred
