"""Regular grids (host-side index maths), mirroring ``snap/utils/grids.py:33-106``.

``interpolate_nd`` of the reference (:116-137) has no host-side counterpart here:
every interpolation of the hot path runs inside a HIP kernel (lift.hip, pose.hip,
voting.hip).
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
