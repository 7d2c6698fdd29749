"""Output containers, same names and fields as ``snap/models/types.py:23-44``."""
import dataclasses
from typing import Any, List, Optional


@dataclasses.dataclass
class FeatureVolume:
  """3-D volume of features [..., X, Y, Z, D] with validity mask [..., X, Y, Z]."""

  features: Any
  valid: Optional[Any] = None

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


@dataclasses.dataclass
class FeaturePlane:
  """2-D plane of features [..., X, Y, D] with validity mask [..., X, Y]."""

  features: Any
  valid: Optional[Any] = None

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


@dataclasses.dataclass
class FeatureImagePyramid:
  """Image feature pyramid (coarse -> fine) with per-level stride w.r.t. the input."""

  features: List[Any]
  strides: List[Any]

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)
