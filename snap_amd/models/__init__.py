"""Top-level register of models (``snap/models/__init__.py:25-40``).

The localisation model is the hot path; ``semantic_net`` reuses its
BEV mapper; ``occupancy_net`` of the reference is out of scope.
"""
import importlib

BASEPATH = 'snap_amd.models.{}'

MODELS = {
    'bev_localizer': ('bev_localizer', 'BEVLocalizerModel'),
    'semantic_net': ('semantic_net', 'SemanticNetModel'),
}


def get_class(modulename, classname):
  return getattr(importlib.import_module(BASEPATH.format(modulename)), classname)


def get_model(name):
  """Get a top-level model class by name."""
  if name not in MODELS:
    raise KeyError(f'model {name!r} is not built (bev_localizer, semantic_net)')
  return get_class(*MODELS[name])
