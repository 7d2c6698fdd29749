"""Pose estimation from BEV correspondences (``snap/models/pose_estimation.py``).

Same function names as the reference (``*_batched`` variants operate on a leading
scene axis); the arithmetic runs in pose.hip.
"""
import numpy as np
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.utils import geometry


def pose_scoring_many_batched(
    j_t_i, scores_points_all, i_xy_points, valid_points, valid_j, grid,
    mask_out_of_bounds,
):
  """pose_estimation.py:63-82,208-211.

  j_t_i: Transform2D [B,P]; scores_points_all [B,N,H,W]; i_xy_points [B,N,2];
  valid_points [B,N]; valid_j [B,H,W].  Returns scores [B,P].
  """
  score = ag.pose_score if (torch.is_grad_enabled() and scores_points_all.requires_grad) else ops.pose_score
  return score(
      scores_points_all, j_t_i.packed().detach(), i_xy_points.contiguous(),
      valid_points.contiguous(), valid_j.contiguous(), grid.cell_size,
      mask_oob=mask_out_of_bounds,
  )


def sample_transforms_ransac_batched(
    rng, matching, i_xy_p, num_poses, num_retries, grid, uniforms=None,
):
  """pose_estimation.py:126-165 (vmapped :224).

  The reference samples flat indices of ``prob_points`` with
  ``jax.random.choice(p=...)``; here ``matching`` carries what is needed to sample
  from the same distribution without materialising it:
  ``dict(fq, fm, chunk_stats, scale, clip)`` as produced by
  ``BEVLocalizer.similarity``.  ``rng`` is an integer seed (Philox stream).
  Returns Transform2D [B, num_poses] and the sampled correspondences.
  """
  S = num_poses * num_retries * 2
  corr = ops.ransac_sample(
      matching['fq'], matching['fm'], matching['chunk_stats'], matching['scale'],
      matching['clip'], S, seed=0 if rng is None else int(rng), uniforms=uniforms,
      row_cdf=matching.get('row_cdf'), sim=matching.get('sim'),
      row_unscale=matching.get('row_unscale'),
  )
  poses = ops.poses_from_corr(corr, i_xy_p.contiguous(), num_poses, num_retries, grid.cell_size)
  return geometry.Transform2D.from_packed(poses), corr


def sample_transforms_random(rng, num, grid, device=None):
  """pose_estimation.py:85-97: ``num`` poses uniform in angle and within 2/3 of the grid extent
  around its centre, expressed in the corner frame.  ``rng``: int seed or torch.Generator (CPU);
  the draws come from torch's generator (JAX's threefry stream cannot be reproduced)."""
  gen = rng if isinstance(rng, torch.Generator) else torch.Generator(device='cpu').manual_seed(int(rng))
  angle = torch.rand(num, generator=gen) * (2 * np.pi)
  size = torch.tensor(np.asarray(grid.extent_meters, np.float32))
  t_max = size * 2 / 3
  translation = (torch.rand(num, 2, generator=gen) * 2 - 1) * t_max
  centre = geometry.Transform2D(angle.to(device), translation.to(device))
  corner_t_center = geometry.Transform2D(torch.zeros((), device=device), (size / 2).to(device))
  return corner_t_center @ centre @ corner_t_center.inv


def refinement_offsets(device):
  """The 41 x 41 x 41 (rotation, x, y) lattice of pose_estimation.py:178-184."""
  delta_p, delta_r, range_p, range_r = 0.2, 0.25, 4, 5
  offs_r = np.mgrid[slice(-range_r, range_r + delta_r, delta_r)]
  offs_p = np.mgrid[slice(-range_p, range_p + delta_p, delta_p)]
  from snap_amd.models import base
  return (base.device_const('refine_offs_r', device, lambda: torch.tensor(np.deg2rad(offs_r.astype(np.float32)))),
          base.device_const('refine_offs_p', device, lambda: torch.tensor(offs_p.astype(np.float32))))


def grid_refinement_batched(
    j_t_i_init, scores_points_all, i_xy_points, valid_points, valid_j, grid,
    mask_out_of_bounds, max_point_norm=None,
):
  """pose_estimation.py:168-205 (vmapped :212-214).

  Returns (Transform2D [B], scores [B, 41, 41, 41]).  ``max_point_norm``: an upper bound of |i_xy_points|
  in metres (a host constant of the query frustum); with it the lattice is scored from one small window of
  every point's score plane (``ops.pose_score_window``: the lattice poses move a point by at most
  range_p * sqrt(2) + |point| * range_r around the initial pose's image of it) -- same bits as the general
  kernels, a tenth of the plane traffic.
  """
  dev = scores_points_all.device
  offs_r, offs_p = refinement_offsets(dev)
  init = j_t_i_init.packed()
  samples = ops.refine_lattice(init, offs_r, offs_p)
  X, Y = scores_points_all.shape[-2:]
  radius = None
  if ops.LATTICE_WINDOW and max_point_norm is not None and not mask_out_of_bounds:
    # |R(a0 + da) q + t0 + R(a0) d - (R(a0) q + t0)| <= |d| + |q| * 2 sin(|da| / 2) <= |d| + |q| |da|
    range_p = float(np.abs(np.mgrid[slice(-4, 4 + 0.2, 0.2)]).max())
    range_r = float(np.deg2rad(np.abs(np.mgrid[slice(-5, 5 + 0.25, 0.25)]).max()))
    radius = int(np.ceil((range_p * np.sqrt(2.0) + float(max_point_norm) * range_r) / grid.cell_size * 1.0001)) + 1
    if not ops.pose_score_window_supported(X, Y, radius):
      radius = None
  if radius is not None:
    scores = ops.pose_score_window(
        scores_points_all, samples, init.contiguous(), radius, i_xy_points.contiguous(),
        valid_points.contiguous(), grid.cell_size)
  else:
    scores = ops.pose_score(
        scores_points_all, samples, i_xy_points.contiguous(),
        valid_points.contiguous(), valid_j.contiguous(), grid.cell_size,
        mask_oob=mask_out_of_bounds,
    )
  best = ops.argmax_rows(scores).to(torch.int64)
  B = samples.shape[0]
  refined = samples[torch.arange(B, device=dev), best]
  nr, np_ = offs_r.numel(), offs_p.numel()
  return geometry.Transform2D.from_packed(refined), scores.reshape(B, nr, np_, np_)
