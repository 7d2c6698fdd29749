"""Minimal stand-in for ``ml_collections.ConfigDict`` (not installable offline).

Supports what ``snap/configs/*.py`` and the model code use: attribute and item
access, ``get``, ``update``, ``lock``/``unlock``, ``placeholder`` (-> ``None``),
``to_dict`` and deep copies.  A real ``ml_collections.ConfigDict`` can be passed
to every snap_amd module instead: only ``cfg.key`` / ``cfg['key']`` /
``cfg.get('key')`` are used.
"""
import copy


def placeholder(_type=None):
  """``config_dict.placeholder(T)``: an unset (None) field."""
  return None


class ConfigDict:
  """Attribute-style nested dictionary with optional key locking."""

  def __init__(self, initial=None, **kwargs):
    object.__setattr__(self, '_fields', {})
    object.__setattr__(self, '_locked', False)
    data = dict(initial or {})
    data.update(kwargs)
    for k, v in data.items():
      self._fields[k] = self._wrap(v)

  @staticmethod
  def _wrap(v):
    if isinstance(v, dict):
      return ConfigDict(v)
    return v

  # -- access -----------------------------------------------------------------
  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    try:
      return self._fields[name]
    except KeyError:
      raise AttributeError(name) from None

  def __setattr__(self, name, value):
    if self._locked and name not in self._fields:
      raise AttributeError(f'ConfigDict is locked; cannot add key {name!r}')
    self._fields[name] = self._wrap(value)

  __getitem__ = __getattr__

  def __getitem__(self, name):
    return self._fields[name]

  def __setitem__(self, name, value):
    self.__setattr__(name, value)

  def __contains__(self, name):
    return name in self._fields

  def __iter__(self):
    return iter(self._fields)

  def __len__(self):
    return len(self._fields)

  def keys(self):
    return self._fields.keys()

  def items(self):
    return self._fields.items()

  def values(self):
    return self._fields.values()

  def get(self, name, default=None):
    return self._fields.get(name, default)

  # -- mutation -----------------------------------------------------------------
  def update(self, other=None, **kwargs):
    data = dict(other.items()) if other is not None else {}
    data.update(kwargs)
    for k, v in data.items():
      cur = self._fields.get(k)
      if isinstance(cur, ConfigDict) and isinstance(v, (dict, ConfigDict)):
        cur.update(v)
      else:
        self.__setattr__(k, v)

  def lock(self):
    object.__setattr__(self, '_locked', True)
    for v in self._fields.values():
      if isinstance(v, ConfigDict):
        v.lock()
    return self

  def unlock(self):
    object.__setattr__(self, '_locked', False)
    for v in self._fields.values():
      if isinstance(v, ConfigDict):
        v.unlock()
    return self

  @property
  def is_locked(self):
    return self._locked

  # -- conversion -----------------------------------------------------------------
  def to_dict(self):
    return {
        k: (v.to_dict() if isinstance(v, ConfigDict) else v)
        for k, v in self._fields.items()
    }

  def copy_and_resolve_references(self):
    return copy.deepcopy(self)

  def __deepcopy__(self, memo):
    new = ConfigDict({k: copy.deepcopy(v, memo) for k, v in self._fields.items()})
    if self._locked:
      new.lock()
    return new

  def __eq__(self, other):
    if isinstance(other, ConfigDict):
      return self.to_dict() == other.to_dict()
    if isinstance(other, dict):
      return self.to_dict() == other
    return NotImplemented

  def __repr__(self):
    return f'ConfigDict({self.to_dict()!r})'


def create(**kwargs):
  """``config_dict.create(**kw)``."""
  return ConfigDict(kwargs)
