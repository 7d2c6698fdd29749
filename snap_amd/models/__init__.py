"""Top-level register of models (``snap/models/__init__.py:25-40``).

Only the localisation model is on the hot path; ``occupancy_net`` and
``semantic_net`` of the reference are out of scope.
"""
import importlib

BASEPATH = 'snap_amd.models.{}'

MODELS = {
    'bev_localizer': ('bev_localizer', 'BEVLocalizerModel'),
}


def get_class(modulename, classname):
  return getattr(importlib.import_module(BASEPATH.format(modulename)), classname)


def get_model(name):
  """Get a top-level model class by name."""
  if name not in MODELS:
    raise KeyError(f'model {name!r} is not part of the localisation hot path')
  return get_class(*MODELS[name])
