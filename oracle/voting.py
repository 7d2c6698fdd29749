"""Oracle (test infrastructure): exhaustive (x, y, theta) rotated-template correlation.

Restates ``snap/models/pose_exhaustive_voting.py``.
"""
import math

import numpy as np

from oracle import geometry
from oracle import grids


def get_grid_center_transform(grid, dtype=np.float32):
  """pose_exhaustive_voting.py:31-34."""
  center_offset = (np.asarray(grid.extent_meters) / 2).astype(dtype)
  return geometry.Transform2D(np.asarray(0, dtype), center_offset)


def template_transforms(num_rotations, grid, dtype=np.float32):
  """pose_exhaustive_voting.py:44-50: templates_t_grid for all R angles."""
  angles = np.linspace(0, np.pi * 2, num_rotations, endpoint=False).astype(dtype)
  rotated_t_grid = geometry.Transform2D(angles, np.zeros((len(angles), 2), dtype))
  c = get_grid_center_transform(grid, dtype)
  corner_t_center = geometry.Transform2D(
      np.broadcast_to(c.angle, angles.shape), np.broadcast_to(c.t, (len(angles), 2))
  )
  out = corner_t_center @ rotated_t_grid @ corner_t_center.inv
  return geometry.Transform2D(out.angle.astype(dtype), out.t.astype(dtype))


def sample_query_templates(features, valid, num_rotations, grid):
  """pose_exhaustive_voting.py:37-69.  features [H,W,D], valid [H,W]."""
  dtype = features.dtype
  templates_t_grid = template_transforms(num_rotations, grid, dtype)
  grid_xy = grid.index_to_xyz(grid.grid_index(), dtype).reshape(-1, 2)
  nq = num_rotations // 4
  quarter = []
  t_valid = []
  for r in range(nq):
    xy = templates_t_grid[r] @ grid_xy
    uv = (xy / dtype.type(grid.cell_size)).astype(dtype)
    q, v = grids.interpolate_nd(features, uv, valid)
    quarter.append(np.where(v[..., None], q, 0))
    t_valid.append(v)
  quarter = np.stack(quarter).reshape(nq, *grid.extent, features.shape[-1])
  t_valid = np.stack(t_valid).reshape(nq, *grid.extent)
  templates = np.concatenate(
      [np.rot90(quarter, k, axes=(2, 1)) for k in range(4)], 0
  )
  t_valid = np.concatenate(
      [np.rot90(t_valid, k, axes=(2, 1)) for k in range(4)], 0
  )
  return templates.astype(dtype), t_valid


def template_matching(q, q_valid, m, m_valid, do_padding=True, min_overlap=0.05):
  """pose_exhaustive_voting.py:72-104.

  The 'valid'-mode convolution of the flipped template with the edge-padded map
  equals the cross-correlation
      out[r,a,b] = sum_{i,j,d} q[r,i,j,d] * m_pad[a+i, b+j, d].
  Computed here with an explicit sliding-window view (direct form).
  """
  dtype = q.dtype
  R, H, W, D = q.shape
  Hm, Wm = m.shape[:2]
  if do_padding:
    m_pad = np.pad(m, ((Hm - 1,) * 2, (Wm - 1,) * 2, (0, 0)), mode='edge')
  else:
    # :86 mode='full': the true convolution of the flipped template with the UNPADDED map at every
    # overlap = the same correlation over the map zero-extended by (H - 1, W - 1): [Hm+H-1, Wm+W-1].
    # The reference pads the validity mask by (Hm - 1, Wm - 1) whatever the mode (:93-96), so its
    # count has another shape and `jnp.where` fails to broadcast: only min_overlap=None runs there.
    if min_overlap is not None:
      raise ValueError('template_matching(do_padding=False): the overlap count of the reference has '
                       'the padded shape (pose_exhaustive_voting.py:93-101); pass min_overlap=None')
    m_pad = np.pad(m, ((H - 1,) * 2, (W - 1,) * 2, (0, 0)), mode='constant')
  Ho, Wo = m_pad.shape[0] - H + 1, m_pad.shape[1] - W + 1
  scores = np.empty((R, Ho, Wo), dtype)
  win = np.lib.stride_tricks.sliding_window_view(m_pad, (H, W), axis=(0, 1))
  # win: [Ho, Wo, D, H, W]
  for r in range(R):
    scores[r] = np.einsum('abdij,ijd->ab', win, q[r], optimize=True)
  if min_overlap is not None:
    mv = np.pad(
        m_valid.astype(dtype), ((Hm - 1,) * 2, (Wm - 1,) * 2), mode='constant'
    )
    winv = np.lib.stride_tricks.sliding_window_view(mv, (H, W), axis=(0, 1))
    # Reference quirk (pose_exhaustive_voting.py:97-99): q_valid is passed to the
    # true convolution WITHOUT the [::-1, ::-1] flip applied to the features
    # (:90), so the overlap count correlates the map mask with the 180-degree
    # rotated template mask.  Mirrored here.
    qv_flip = q_valid[:, ::-1, ::-1].astype(dtype)
    num_valid = np.einsum('abij,rij->rab', winv, qv_flip, optimize=True)
    valid_score = num_valid > (min_overlap * math.prod(q_valid.shape[-2:]))
    scores = np.where(valid_score, scores, -np.inf).astype(dtype)
  with np.errstate(divide='ignore', invalid='ignore'):
    scores = scores / q_valid.sum((-1, -2), keepdims=True).astype(dtype)
  return scores.astype(dtype)


def exhaustive_pose_voting(plane_q, plane_map, num_rotations, grid, conf_q=None):
  """pose_exhaustive_voting.py:107-124."""
  feats_q = plane_q['features']
  if conf_q is not None:
    feats_q = feats_q * conf_q[..., None]
  templates, t_valid = sample_query_templates(
      feats_q, plane_q['valid'], num_rotations, grid
  )
  return template_matching(
      templates, t_valid, plane_map['features'], plane_map['valid']
  )


def exhaustive_index_to_tfm(index, grid, num_rotations, dtype=np.float32):
  """pose_exhaustive_voting.py:127-137."""
  index = np.asarray(index)
  xy_cell = (
      (index[1:] - np.array(grid.extent) + 1 + 0.5) * grid.cell_size
  ).astype(dtype)
  angle = dtype(index[0] * 2 * np.pi / num_rotations)
  m_t_q_center = geometry.Transform2D(np.asarray(-angle, dtype), xy_cell)
  c = get_grid_center_transform(grid, dtype)
  return c @ m_t_q_center @ c.inv


def exhaustive_tfm_to_index(m_t_q_corner, grid, num_rotations, dtype=np.float32):
  """pose_exhaustive_voting.py:140-149."""
  c = get_grid_center_transform(grid, dtype)
  m_t_q_center = c.inv @ m_t_q_corner @ c
  k = (-m_t_q_center.angle / (np.pi * 2) % 1) * num_rotations
  ij = (m_t_q_center.t / grid.cell_size) + np.array(grid.extent) - 1.5
  return np.concatenate([np.asarray(k)[..., None], ij], -1)
