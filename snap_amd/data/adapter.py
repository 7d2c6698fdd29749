"""Reference batch -> snap_amd batch (the schema of ``snap/data/loader.py:82-168``).

The reference's tf.data pipeline yields nested dicts of numpy arrays in which cameras and
poses are ``dataclass_array`` structs (``snap/utils/geometry.py``: ``FisheyeCamera{wh, f, c,
k_radial, max_fov}``, ``Transform3D{R, t}``) and, under ``pmap``, every leaf carries an extra
leading local-device axis.  ``from_reference_batch`` turns such a batch (the structs given as
objects with those attributes, or as plain dicts of their fields) into what the models here
consume: torch tensors on ``device`` plus ``snap_amd.utils.geometry`` structs.  Bookkeeping
fields the models never read (``scene_id``, ``latlng``, ``vehicle_type``, ``pair_id``, ...) are
carried through untouched under the same keys.
"""
import numpy as np
import torch

from snap_amd.utils import geometry

_CAMERA_FIELDS = ('wh', 'f', 'c')
_SCENE_TENSORS = ('images', 'xyz_query', 'z_offset', 'xy_bev')


def _field(struct, name):
  return struct[name] if isinstance(struct, dict) else getattr(struct, name)


def _has(struct, name):
  return (name in struct) if isinstance(struct, dict) else hasattr(struct, name)


def _tensor(x, device, dtype=torch.float32, lead=None):
  a = np.asarray(x)
  if lead is not None:                       # drop / select the pmap axis
    a = a.reshape(-1, *a.shape[2:]) if lead == 'merge' else a[lead]
  t = torch.as_tensor(np.ascontiguousarray(a))
  if t.dtype.is_floating_point:
    t = t.to(dtype)
  return t.to(device)


def _camera(cam, device, lead):
  wh, f, c = (_tensor(_field(cam, n), device, lead=lead) for n in _CAMERA_FIELDS)
  if _has(cam, 'k_radial'):
    return geometry.FisheyeCamera(wh, f, c, _tensor(_field(cam, 'k_radial'), device, lead=lead),
                                  _tensor(_field(cam, 'max_fov'), device, lead=lead))
  return geometry.Camera(wh, f, c)


def _transform3d(T, device, lead):
  return geometry.Transform3D(_tensor(_field(T, 'R'), device, lead=lead),
                              _tensor(_field(T, 't'), device, lead=lead))


def _scene(scene, device, lead):
  out = {}
  for k, v in scene.items():
    if k == 'camera':
      out[k] = _camera(v, device, lead)
    elif k == 'T_view2scene':
      out[k] = _transform3d(v, device, lead)
    elif k == 'rasters':
      out[k] = {rk: _tensor(rv, device, lead=lead) for rk, rv in v.items()}   # bool rasters stay bool
    elif k in _SCENE_TENSORS:
      out[k] = _tensor(v, device, lead=lead)
    else:
      out[k] = v
  return out


def from_reference_batch(batch, device='cpu', device_axis=None):
  """``device_axis``: None -- leaves are [B, ...]; 'merge' -- leaves are [D, B, ...] (pmap
  layout), folded to [D*B, ...]; an int -- take that local device's shard."""
  lead = device_axis
  out = {}
  for k, v in batch.items():
    if k in ('map', 'query'):
      out[k] = _scene(v, device, lead)
    elif k == 'T_query2map':
      out[k] = _transform3d(v, device, lead)
    elif k == 'batch_mask':
      out[k] = _tensor(v, device, lead=lead).to(torch.bool)
    elif k in ('overlap', 'time_delta_days'):
      out[k] = _tensor(v, device, lead=lead)
    else:
      out[k] = v
  if 'batch_mask' not in out and 'query' in out:
    out['batch_mask'] = torch.ones(len(out['query']['images']), dtype=torch.bool, device=device)
  return out
