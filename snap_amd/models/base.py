"""Module plumbing: a tiny functional module system + the BaseModel contract.

The reference modules are Flax ``nn.Module``s driven as
``model.init(rngs, batch, train=...)`` / ``model.apply(variables, batch, ...)``
(snap/trainer.py:147-156,211-218; snap/models/base.py:32-67).  Here a module is a
plain object with

  * ``init_params(gen, device) -> dict``  -- nested dict of fp32 tensors in the
    FLAX parameter layout (conv kernels HWIO, Dense kernels (in, out), GroupNorm
    scale/bias (1,1,1,C)), so a Flax checkpoint maps 1:1 with no transposes;
  * ``__call__(params, ...)``              -- the forward pass on HIP kernels.

``Module.init`` / ``Module.apply`` adapt that to the Flax calling convention.
"""
import math
from typing import Any, Callable, Dict, Optional, Tuple

import torch

from snap_amd import ops

Batch = Dict[str, Any]
Predictions = Dict[str, Any]
LossMetricsTuple = Tuple[Dict[str, Any], Dict[str, Any]]


import weakref

# id(kernel tensor) -> (weakref to it, tensor._version, standardised copy).
# StdConv re-standardises its kernel on every call in the reference (it is part of the traced
# graph).  Default here: the same -- every ``apply`` standardises the kernels it uses (all
# kernels of an encoder in ONE launch), cached only within that call (``ForwardContext``).
# A serving deployment may fold this parameter-only transform across calls with
# ``CACHE_STANDARDIZED_WEIGHTS_ACROSS_CALLS = True``: the copy is then reused until the parameter
# is modified in place (``_version`` bump) or dies.  ``bench.py`` leaves it False, so the timed
# step contains the standardisation.
CACHE_STANDARDIZED_WEIGHTS_ACROSS_CALLS = False
_WSTD_CACHE = {}


class ForwardContext:
  """Per-``apply`` scratch: the standardised StdConv kernels of this call."""

  def __init__(self):
    self._std = {}
    from snap_amd import ops
    ops.PACK_EPOCH += 1        # bf16 weight images are per apply (see ops._packed_weights)

  def _lookup(self, kernel):
    hit = self._std.get(id(kernel))
    if hit is not None and hit[0] is kernel:
      return hit[1]
    if CACHE_STANDARDIZED_WEIGHTS_ACROSS_CALLS:
      hit = _WSTD_CACHE.get(id(kernel))
      if hit is not None and hit[0]() is kernel and hit[1] == kernel._version:
        return hit[2]
    return None

  def _store(self, kernel, out):
    self._std[id(kernel)] = (kernel, out)
    if CACHE_STANDARDIZED_WEIGHTS_ACROSS_CALLS:
      if len(_WSTD_CACHE) > 4096:
        _WSTD_CACHE.clear()
      _WSTD_CACHE[id(kernel)] = (weakref.ref(kernel), kernel._version, out)

  def standardized(self, kernel, fn):
    out = self._lookup(kernel)
    if out is None:
      out = fn(kernel)
      self._store(kernel, out)
    return out

  def standardize_all(self, kernels, fn_multi):
    """Standardise every not-yet-known kernel of ``kernels`` with ONE launch."""
    todo = [k for k in kernels if self._lookup(k) is None]
    if todo:
      for k, out in zip(todo, fn_multi(todo)):
        self._store(k, out)


def needs_grad(*tensors):
  """True when autograd must record this op: grad mode is on and some input needs it.
  Inference (plain parameter tensors / torch.no_grad()) takes the direct `ops` path."""
  if not torch.is_grad_enabled():
    return False
  return any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def clear_caches():
  _WSTD_CACHE.clear()


class Module:
  """Base class: Flax-style ``init`` / ``apply`` over explicit parameter dicts."""

  def init_params(self, gen: torch.Generator, device) -> Dict[str, Any]:
    raise NotImplementedError

  def init(self, rngs, *args, device=None, **kwargs):
    """``model.init(rngs, batch, train=False)`` -> ``{'params': ...}``.

    ``rngs`` may be an int seed, a ``torch.Generator`` or a dict with a 'params'
    entry holding either (the Flax ``{'params': key, 'sampling': key}`` shape).
    """
    if isinstance(rngs, dict):
      rngs = rngs.get('params', 0)
    if isinstance(rngs, torch.Generator):
      gen = rngs
    else:
      gen = torch.Generator(device='cpu')
      gen.manual_seed(int(rngs))
    if device is None:
      device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
    return {'params': self.init_params(gen, torch.device(device))}

  def apply(self, variables, *args, rngs=None, mutable=False, **kwargs):
    """``model.apply({'params': p}, batch, train=..., rngs={'sampling': seed})``.

    Returns ``pred`` or ``(pred, new_state)`` when ``mutable`` is truthy (the
    model has no mutable collections: GroupNorm only, so ``new_state == {}``).
    """
    seed = None
    if isinstance(rngs, dict):
      seed = rngs.get('sampling')
    elif rngs is not None:
      seed = rngs
    # the arithmetic of this apply is the MODEL's (BaseModel(dtype=, engine=) sets ``engine`` on its
    # flax_model); a bare Module without one runs on whatever scope / process default is in force
    with ops.engine_scope(getattr(self, 'engine', None)):
      out = self(variables['params'], *args, rng=seed, **kwargs)
    if mutable:
      return out, {}
    return out


# -- initialisers (host side, CPU generator for reproducibility) --------------
def lecun_normal(gen, shape, fan_in, device):
  std = 1.0 / math.sqrt(fan_in)
  return (torch.randn(shape, generator=gen) * std).to(device)


def glorot_uniform(gen, shape, fan_in, fan_out, device):
  lim = math.sqrt(6.0 / (fan_in + fan_out))
  return ((torch.rand(shape, generator=gen) * 2 - 1) * lim).to(device)


def truncated_normal(gen, shape, std, device):
  t = torch.empty(shape)
  torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
  # flax variance_scaling divides by the std of the truncated unit normal.
  return (t * (std / 0.87962566103423978)).to(device)


class BaseModel:
  """Trainer-facing wrapper (snap/models/base.py:32-67)."""

  def __init__(self, config, dataset_meta_data: Dict[str, Any], dtype=torch.float32, engine=None):
    """``dtype`` selects the arithmetic as in the reference (``model_cls(config.model, meta, dtype)``,
    trainer.py:387-397, evaluator.py:179-184): float16 -> IEEE-half operands with f32 accumulation
    (to be trained under ``DynamicScale``: ``trainer.dtype_and_dynamic_scale(config.dtype_str)`` returns the
    pair; ``train_step`` warns when an fp16 model steps without one), bfloat16 -> bf16
    operands, float32 -> an f32-class engine.  ``engine`` names the engine explicitly where the dtype
    leaves a choice: 'f32' (exact f32 matrix instructions) | 'bf16x3' | 'bf16x6' (split-bf16, f32
    grade; what bench.py measures) for float32.  None with float32: the process default
    (``ops.MATMUL_PRECISION``, 'f32' unless a tool changed it)."""
    self.config = config
    self.dataset_meta_data = dataset_meta_data
    self.dtype = dtype
    by_dtype = ops.engine_of_dtype(dtype)
    if engine is not None:
      if engine not in ops.ENGINES:
        raise ValueError(f'engine {engine!r}: expected one of {ops.ENGINES}')
      if by_dtype is not None and engine != by_dtype:
        raise ValueError(f'engine {engine!r} contradicts dtype {dtype} (-> {by_dtype!r})')
      if by_dtype is None and engine in ops.HALF_MATH:
        raise ValueError(f'engine {engine!r} needs dtype float16 / bfloat16, got {dtype}')
    self.engine = engine if engine is not None else by_dtype
    self.flax_model = self.build_flax_model()
    self.flax_model.engine = self.engine

  def loss_metrics_function(self, pred, batch, model_params=None) -> LossMetricsTuple:
    raise NotImplementedError('Subclasses must implement metrics.')

  def build_flax_model(self) -> Module:
    raise NotImplementedError('Subclasses must implement build_flax_model().')

  def default_flax_model_config(self):
    raise NotImplementedError('Subclasses must implement default_flax_model_config().')


_DEVICE_CONSTS = {}


def device_const(key, device, make, owner=None):
  """A small constant tensor uploaded ONCE per (key, device): lattices, query grids and per-config
  scales were re-uploaded on every apply (one pageable host-to-device blit kernel each, ~4 us of
  stream time: the `__amd_rocclr_copyBuffer` rows of the round-2 profile).  ``make()`` builds the
  host tensor; the cached device copy must be treated as read-only.  ``owner``: the object whose
  state the constant derives from (the cache then lives and dies with it); without one the key
  must identify the value globally."""
  device = torch.device(device)
  cache = _DEVICE_CONSTS if owner is None else owner.__dict__.setdefault('_device_consts', {})
  k = (key, device.type, device.index)
  t = cache.get(k)
  if t is None:
    t = make().to(device)
    cache[k] = t
  return t
