"""Oracle (test infrastructure): regular grids and N-linear interpolation.

Restates ``snap/utils/grids.py`` (reference file:line cited per function).
"""
import itertools

import numpy as np


class GridND:
  """N-dimensional regular grid (snap/utils/grids.py:33-89)."""

  def __init__(self, extent, cell_size):
    self.extent = tuple(int(e) for e in extent)
    self.cell_size = float(cell_size)

  @classmethod
  def from_extent_meters(cls, extent_meters, cell_size):
    # snap/utils/grids.py:48-57
    extent = tuple(i / cell_size for i in extent_meters)
    if not all(e % 1 == 0 for e in extent):
      raise ValueError(
          f'The metric grid extent {extent_meters} is not divisible '
          f'by the cell size {cell_size}.'
      )
    return cls(tuple(map(int, extent)), cell_size)

  def xyz_to_index(self, xyz):
    # snap/utils/grids.py:59-60
    return np.floor(xyz / self.cell_size).astype(int)

  def index_to_xyz(self, idx, dtype=None):
    # snap/utils/grids.py:62-63 -- half-cell centres.  With `dtype` the arithmetic
    # itself runs in that precision (JAX evaluates it in fp32 by default).
    if dtype is not None:
      dtype = np.dtype(dtype).type
      return (np.asarray(idx).astype(dtype) + dtype(0.5)) * dtype(self.cell_size)
    return (idx + 0.5) * self.cell_size

  @property
  def num_cells(self):
    return int(np.prod(self.extent))

  @property
  def extent_meters(self):
    return np.asarray(self.extent) * self.cell_size

  def grid_index(self):
    # snap/utils/grids.py:87-89
    grid = np.mgrid[tuple(slice(None, e) for e in self.extent)]
    return np.moveaxis(grid, 0, -1)

  def id_to_index(self, ids):
    # snap/utils/grids.py:70-71
    return np.stack(np.unravel_index(ids, self.extent), -1)

  def __eq__(self, other):
    return (
        isinstance(other, GridND)
        and self.extent == other.extent
        and self.cell_size == other.cell_size
    )

  def __repr__(self):
    return f'{type(self).__name__}(extent={self.extent}, cell_size={self.cell_size})'


class Grid2D(GridND):
  pass


class Grid3D(GridND):

  def bev(self):
    # snap/utils/grids.py:105-106
    return Grid2D(self.extent[:2], self.cell_size)


def map_coordinates_linear_nearest(arr, coords):
  """order=1, mode='nearest' map_coordinates over the leading ``n`` dims.

  Restates the behaviour of ``jax.scipy.ndimage.map_coordinates`` as called at
  snap/utils/grids.py:109-113: per dimension ``lo = floor(x)``, upper weight
  ``x - lo``, lower weight ``1 - upper`` (weights from the UNCLIPPED coordinate),
  each tap index clipped to ``[0, size-1]``; the 2^n products are accumulated in
  ``itertools.product`` order (lo/lo, lo/hi, hi/lo, hi/hi).

  Args:
    arr: [s_0..s_{n-1}, ...trailing] array.
    coords: [n, K] coordinates (index space, element centres at integers).
  Returns:
    [K, ...trailing] interpolated values.
  """
  n = coords.shape[0]
  nodes = []
  for d in range(n):
    c = coords[d]
    lo = np.floor(c)
    w_hi = c - lo
    w_lo = 1 - w_hi
    idx = lo.astype(np.int64)
    size = arr.shape[d]
    nodes.append([
        (np.clip(idx, 0, size - 1), w_lo),
        (np.clip(idx + 1, 0, size - 1), w_hi),
    ])
  out = None
  trailing = arr.ndim - n
  for items in itertools.product(*nodes):
    indices = tuple(i for i, _ in items)
    w = items[0][1]
    for _, wi in items[1:]:
      w = w * wi
    w = w.reshape(w.shape + (1,) * trailing).astype(arr.dtype)
    contrib = w * arr[indices]
    out = contrib if out is None else out + contrib
  return out


def interpolate_nd(array, points, valid_array=None):
  """snap/utils/grids.py:116-137.

  Args:
    array: [s_0..s_{n-1}, D].
    points: [K, n] in corner-origin coordinates (cell centres at k + 0.5).
    valid_array: optional bool [s_0..s_{n-1}].
  Returns:
    values [K, D], valid [K].
  """
  n = points.shape[-1]
  size = np.asarray(array.shape[:n])
  valid = np.all((points >= 0) & (points < size), -1)
  coords = np.moveaxis(points - 0.5, -1, 0)
  values = map_coordinates_linear_nearest(array, coords)
  if valid_array is not None:
    # NaN-mask trick (grids.py:131-136): a tap with ZERO weight still
    # invalidates the sample because 0 * nan = nan.
    with np.errstate(invalid='ignore'):
      nan_mask = np.where(valid_array, 0.0, np.nan).astype(array.dtype)
      nan_points = map_coordinates_linear_nearest(nan_mask, coords)
    valid = valid & ~np.isnan(nan_points)
  return values, valid


def argmax_nd(scores, grid):
  # snap/utils/grids.py:140-145
  n = len(grid.extent)
  flat = scores.reshape(*scores.shape[:-n], -1)
  i = np.argmax(flat, axis=-1)
  return grid.id_to_index(i)
