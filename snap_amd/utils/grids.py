"""Regular grids (host-side index maths), mirroring ``snap/utils/grids.py:33-106``.

``interpolate_nd`` / ``argmax_nd`` / ``expectation_nd`` (:116-153) are thin wrappers over
stand-alone HIP entry points (grid_ops.hip); inside the hot path the same interpolation is
fused into lift.hip / pose.hip / voting.hip.
"""
import dataclasses
from typing import Tuple

import numpy as np
import torch


@dataclasses.dataclass(frozen=True)
class GridND:
  """N-dimensional regular grid: ``extent`` cells of ``cell_size`` metres."""

  extent: Tuple[int, ...]
  cell_size: float

  @classmethod
  def from_extent_meters(cls, extent_meters, cell_size):
    extent = tuple(i / cell_size for i in extent_meters)
    if not all(e % 1 == 0 for e in extent):
      raise ValueError(
          f'The metric grid extent {extent_meters} is not divisible '
          f'by the cell size {cell_size}.'
      )
    return cls(tuple(map(int, extent)), cell_size)

  def xyz_to_index(self, xyz):
    return torch.floor(xyz / self.cell_size).to(torch.int64)

  def index_to_xyz(self, idx):
    return (idx + 0.5) * self.cell_size

  @property
  def num_cells(self) -> int:
    return int(np.prod(self.extent))

  @property
  def extent_meters(self) -> np.ndarray:
    return np.asarray(self.extent) * self.cell_size

  def grid_index(self, device=None) -> torch.Tensor:
    """[*extent, N] integer cell indices (``jnp.mgrid`` moved last)."""
    axes = [torch.arange(e, device=device) for e in self.extent]
    return torch.stack(torch.meshgrid(*axes, indexing='ij'), -1)

  def id_to_index(self, ids):
    out = []
    for e in reversed(self.extent):
      out.append(ids % e)
      ids = torch.div(ids, e, rounding_mode='floor')
    return torch.stack(out[::-1], -1)


@dataclasses.dataclass(frozen=True)
class Grid2D(GridND):
  extent: Tuple[int, int]


@dataclasses.dataclass(frozen=True)
class Grid3D(GridND):
  extent: Tuple[int, int, int]

  def bev(self) -> Grid2D:
    return Grid2D(self.extent[:2], self.cell_size)


def interpolate_nd(array, points, valid_array=None, order=1, mode='nearest'):
  """snap/utils/grids.py:116-137.  array [..., D] over an N-D grid, points [K, N] in
  corner-origin coordinates -> (values [K, D], valid [K]).  Only the reference's own
  ``order=1, mode='nearest'`` combination exists (it never calls any other)."""
  if order != 1 or mode != 'nearest':
    raise NotImplementedError("interpolate_nd: only order=1, mode='nearest' (the reference's call)")
  from snap_amd import ops
  return ops.interpolate_nd(array.contiguous(), points.contiguous(),
                            None if valid_array is None else valid_array.contiguous())


def argmax_nd(scores, grid: GridND):
  """snap/utils/grids.py:140-145: index [..., N] of the (first) maximum over the grid axes."""
  from snap_amd import ops
  n = len(grid.extent)
  lead = scores.shape[:-n]
  flat = scores.reshape(-1, int(np.prod(scores.shape[-n:]))).contiguous()
  ids = ops.argmax_rows(flat).to(torch.int64)
  return grid.id_to_index(ids).reshape(*lead, n)


def expectation_nd(pdf, grid: GridND):
  """snap/utils/grids.py:148-153: expected index of an N-D probability tensor."""
  from snap_amd import ops
  return ops.expectation_nd(pdf.contiguous(), grid.extent)
