"""Parameter-tree <-> flat-name conversion and checkpoint import / export.

Parameters keep the Flax layout and names of the reference (conv kernels HWIO, Dense
``(in, out)``, GroupNorm ``(1,1,1,C)``; ``bev_mapper/streetview_encoder/image_encoder/
encoder/block1/unit01/conv1/kernel`` ...), so a restored Flax checkpoint maps 1:1 -- no
transposes.  This module is the counterpart of the reference's
``load_pretrained_variables`` hooks (``snap/models/bev_mapper.py:303-315``: restore,
then ``misc.find_nested_dict(state['params'], 'bev_mapper')``; ``resnet.py:223-233``)
for checkpoints exchanged as ``.npz`` files of ``'a/b/c' -> array`` entries (what
``flax.traverse_util.flatten_dict(params, sep='/')`` + ``np.savez`` produces).
"""
import os
from typing import Any, Dict, Optional

import numpy as np
import torch


def flatten(tree: Dict[str, Any], prefix: str = '') -> Dict[str, Any]:
  """Nested dict -> {'a/b/c': leaf}, depth-first in sorted key order."""
  out = {}
  for k in sorted(tree):
    v = tree[k]
    name = f'{prefix}/{k}' if prefix else str(k)
    if isinstance(v, dict):
      out.update(flatten(v, name))
    else:
      out[name] = v
  return out


def unflatten(flat: Dict[str, Any]) -> Dict[str, Any]:
  tree: Dict[str, Any] = {}
  for name, v in flat.items():
    parts = name.split('/')
    node = tree
    for p in parts[:-1]:
      node = node.setdefault(p, {})
      if not isinstance(node, dict):
        raise ValueError(f'{name}: {p} is both a leaf and a sub-tree')
    node[parts[-1]] = v
  return tree


def find_nested_dict(tree: Dict[str, Any], target_key: str) -> Optional[Dict[str, Any]]:
  """First sub-dict stored under ``target_key``, depth-first in insertion order
  (``snap/utils/misc.py:57-66``); None if absent."""
  for k, v in tree.items():
    if isinstance(v, dict):
      if k == target_key:
        return v
      hit = find_nested_dict(v, target_key)
      if hit is not None:
        return hit
  return None


def save_npz(path, params: Dict[str, Any]) -> None:
  flat = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
          for k, v in flatten(params).items()}
  # Written through an open handle (np.savez would append '.npz' to a bare path, so the
  # file would not be found again under the name the caller used) and renamed into place,
  # so a crash mid-save cannot corrupt the only resume checkpoint.
  path = os.fspath(path)
  tmp = f'{path}.tmp.{os.getpid()}'
  with open(tmp, 'wb') as f:
    np.savez(f, **flat)
  os.replace(tmp, path)


def load_npz(path) -> Dict[str, Any]:
  with np.load(path, allow_pickle=False) as z:
    return unflatten({k: torch.from_numpy(np.asarray(z[k])) for k in z.files})


def load_into(template: Dict[str, Any], source: Dict[str, Any], *, strict: bool = True,
              device=None, dtype=torch.float32) -> Dict[str, Any]:
  """A copy of ``template`` (e.g. ``model.init(...)['params']``) with every leaf replaced by
  the same-named leaf of ``source``.  Shapes must match exactly (layouts are identical by
  construction, so a mismatch is a wrong checkpoint, never a transpose).  ``strict``:
  missing or unexpected names raise; otherwise missing leaves keep the template value."""
  want = flatten(template)
  have = flatten(source)
  missing = sorted(set(want) - set(have))
  extra = sorted(set(have) - set(want))
  if strict and (missing or extra):
    raise KeyError(f'checkpoint mismatch: missing {missing[:5]}{"..." if len(missing) > 5 else ""}, '
                   f'unexpected {extra[:5]}{"..." if len(extra) > 5 else ""}')
  out = {}
  for name, ref in want.items():
    if name in have:
      v = have[name]
      v = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
      if tuple(v.shape) != tuple(ref.shape):
        raise ValueError(f'{name}: checkpoint shape {tuple(v.shape)} != model {tuple(ref.shape)}')
      out[name] = v.to(device=device if device is not None else ref.device, dtype=dtype).contiguous()
    else:
      out[name] = ref
  return unflatten(out)


def load_pretrained(template: Dict[str, Any], path, scope: Optional[str] = None, **kw):
  """Restore ``path`` and, like ``BEVMapper.load_pretrained_variables``, take the sub-tree
  stored under ``scope`` (e.g. 'bev_mapper') when given."""
  params = load_npz(path)
  params = params.get('params', params)
  if scope is not None:
    sub = find_nested_dict({'': params}, scope) if scope not in params else params[scope]
    if sub is None:
      raise ValueError(f'No parameters for {scope} in {path}')
    params = sub
  return load_into(template, params, **kw)


def load_bit_resnet(template: Dict[str, Any], path, **kw) -> Dict[str, Any]:
  """BiT ResNet-v2 checkpoint (``.npz``) -> the encoder sub-tree, as ``ResNetV2.load_pretrained_
  variables`` (``snap/models/resnet.py:223-233``) does through ``big_vision``'s ``load_params``.

  ``big_vision`` is not part of the reference; what its loader does for these files is: read
  the ``'a/b/c' -> array`` entries, rebuild the tree, and (for trainer checkpoints) unwrap a
  ``params/`` or ``opt/target/`` prefix.  The BiT parameter tree IS the reference's
  (``root_block/conv_root/kernel``, ``block{i}/unit{jj}/{conv1,conv2,conv3,conv_proj}/kernel``,
  ``.../gn{1,2,3}/{scale,bias}`` with shape (1,1,1,C)), so leaves map 1:1.  The classification
  head of the checkpoint (``norm-pre-head``, ``head``) has no counterpart and is dropped; a
  leaf the encoder needs but the file lacks raises.  ``template`` = ``ResNetV2.init_params``."""
  with np.load(path, allow_pickle=False) as z:
    flat = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
  for prefix in ('opt/target/', 'params/'):
    if flat and all(k.startswith(prefix) for k in flat):
      flat = {k[len(prefix):]: v for k, v in flat.items()}
  want = set(flatten(template))
  dropped = sorted(k for k in flat if k not in want)
  not_head = [k for k in dropped if k.split('/')[0] not in ('head', 'norm-pre-head')]
  if not_head:
    raise KeyError(f'unexpected (non-head) entries in {path}: {not_head[:5]}')
  source = unflatten({k: v for k, v in flat.items() if k in want})
  return load_into(template, source, strict=True, **kw)
