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


class LazyFeatureVolume(FeatureVolume):
  """A ``FeatureVolume`` whose ``features`` are produced on first access.

  With ``bev_mapper.materialize_volume = False`` the StreetView encoder produces the BEV plane
  straight from the pooled observations (fusion MLP + vertical max pooling in one kernel) and the
  dense [..., X, Y, Z, D] volume of ``streetview_encoder.py:282-286`` is never written.  The
  output pytree keeps the reference's entry: a consumer that does read ``features`` gets the
  volume of the unfused chain (lift -> fusion MLP -> mask), computed then and remembered.

  It IS a ``FeatureVolume`` (``isinstance`` checks and ``dataclasses.replace`` work; the latter
  reads ``features`` and therefore materialises).  Until ``features`` is read -- or ``discard()``
  is called -- the object keeps the encoder inputs its thunk needs alive (the image features and
  the voxel coordinates: ~0.5 GB at C2); ``discard()`` drops them for consumers that will never
  read the volume."""

  def __init__(self, thunk=None, valid=None, features=None):
    self._thunk = None if features is not None else thunk
    self._features = features
    self.valid = valid

  @property
  def features(self):
    if self._features is None and self._thunk is not None:
      self._features = self._thunk()
      self._thunk = None              # (drops the references to the encoder's inputs)
    return self._features

  @features.setter
  def features(self, value):
    self._features = value
    self._thunk = None

  @property
  def materialized(self):
    return self._features is not None

  # (the dataclass-generated __repr__ / __eq__ of FeatureVolume read ``features``: printing or
  #  comparing a prediction dict would silently run the unfused lift + MLP chain)
  def __repr__(self):
    state = 'materialized' if self._features is not None else ('lazy' if self._thunk is not None else 'discarded')
    shape = None if self.valid is None else tuple(self.valid.shape)
    return f'LazyFeatureVolume({state}, valid shape={shape})'

  def __eq__(self, other):
    return self is other

  __hash__ = object.__hash__

  def discard(self):
    """Release the inputs held for a volume nobody will read (``features`` is then None)."""
    self._thunk = None
    return self

  def replace(self, **kw):
    if 'features' in kw:
      return FeatureVolume(features=kw['features'], valid=kw.get('valid', self.valid))
    out = LazyFeatureVolume(self._thunk, kw.get('valid', self.valid))
    out._features = self._features
    return out


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
