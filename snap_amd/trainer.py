"""Data-parallel training step of the localisation model.

Semantics of ``snap/trainer.py:165-295`` (``train_step``): per-rank loss =
masked mean of ``losses['total']``; gradients averaged over ranks (``pmean``);
optional global-norm clipping; Adam; the update is SKIPPED when any gradient is
non-finite on any rank; metrics reduced as (sum, count) pairs.  The Scenic loop
around it (data, checkpoints, logging) is out of scope; this module is the step a
driver calls.  One process per GPU; the exchange goes through ``snap_amd.dist``
(RCCL over xGMI under backend "nccl").
"""
import dataclasses
import math
from typing import Any, Callable, Dict, List, Optional

import torch

from snap_amd import dist as sdist
from snap_amd import ops


def flatten_params(tree, prefix=''):
  return sdist.flatten_tree(tree, prefix)


def make_lr_fn(base_lr: float, num_steps: int, start_decay_step: Optional[int] = None) -> Callable:
  """'constant * cosine_decay' with the decay starting at `start_decay_step`
  (snap/configs/train_localization.py:87-92: half-way, one cosine cycle)."""
  start = num_steps // 2 if start_decay_step is None else start_decay_step
  cycle = max(num_steps - start, 1)

  def lr_fn(step: int) -> float:
    progress = max(0, step - start) / cycle
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * (progress % 1.0)))

  return lr_fn


@dataclasses.dataclass
class DynamicScale:
  """Dynamic loss scaling, the state machine of ``flax.training.dynamic_scale.DynamicScale``
  that the reference attaches to float16 runs (``trainer.py:223-229``: the loss is multiplied
  by ``scale`` before differentiation, the gradients divided by it; ``trainer.py:391``:
  ``DynamicScale(minimum_scale=256)``).  A non-finite gradient halves the scale (not below
  ``minimum_scale``) and the step is skipped; ``growth_interval`` consecutive finite steps
  double it.  The scale is a power of two, so scaling is exact in f32 / bf16 arithmetic: with
  this framework's bf16-operand precision (f32 exponent range) it changes nothing but the
  overflow behaviour -- it exists so that a float16-style training recipe ports unchanged."""
  growth_factor: float = 2.0
  backoff_factor: float = 0.5
  growth_interval: int = 2000
  fin_steps: int = 0
  scale: float = 65536.0
  minimum_scale: Optional[float] = None

  def update(self, is_finite: bool) -> 'DynamicScale':
    f32_max = 3.4028234663852886e38
    if is_finite:
      grow = self.fin_steps == self.growth_interval
      scale = min(self.scale * self.growth_factor, f32_max) if grow else self.scale
      fin_steps = 0 if grow else self.fin_steps + 1
    else:
      scale = self.scale * self.backoff_factor
      if self.minimum_scale is not None:
        scale = max(scale, self.minimum_scale)
      fin_steps = 0
    return dataclasses.replace(self, scale=scale, fin_steps=fin_steps)


@dataclasses.dataclass
class TrainState:
  params: Dict[str, Any]
  m: List[torch.Tensor]
  v: List[torch.Tensor]
  global_step: int = 0
  opt_count: int = 0      # optimizer updates actually applied (skipped steps do not count)
  rng: int = 0
  dynamic_scale: Optional[DynamicScale] = None   # (train_state.dynamic_scale, trainer.py:223)

  @classmethod
  def create(cls, params, rng=0, dynamic_scale=None):
    leaves = [t for _, t in flatten_params(params)]
    return cls(params=params, m=[torch.zeros_like(t) for t in leaves],
               v=[torch.zeros_like(t) for t in leaves], global_step=0, rng=rng,
               dynamic_scale=dynamic_scale)


DTYPES = {'float32': torch.float32, 'float16': torch.float16, 'bfloat16': torch.bfloat16}


def dtype_and_dynamic_scale(dtype_str: str):
  """``config.dtype_str`` -> (dtype, DynamicScale | None), the reference's selection
  (``trainer.py:387-394``): float32 / bfloat16 train unscaled, float16 under
  ``DynamicScale(minimum_scale=256)``; anything else is an error.  Use as

      dtype, ds = trainer.dtype_and_dynamic_scale(config.dtype_str)
      model = model_cls(config.model, meta, dtype)          # dtype selects the engine (models/base.py)
      state = TrainState.create(params, dynamic_scale=ds)"""
  if dtype_str not in DTYPES:
    raise ValueError(f'Unsupported dtype: {dtype_str}')
  dtype = DTYPES[dtype_str]
  return dtype, (DynamicScale(minimum_scale=256) if dtype == torch.float16 else None)


def save_train_state(path, state: TrainState) -> None:
  """Checkpoint for resume (the reference: ``train_utils.save_checkpoint`` of the Flax TrainState,
  ``trainer.py:594-602``): parameters under ``params/...`` (Flax names, so the file doubles as a
  pretrained checkpoint for ``checkpoint.load_pretrained``), Adam moments under ``opt/m|v/...``,
  step and sampling rng as 0-d arrays."""
  from snap_amd.utils import checkpoint
  names = [n for n, _ in flatten_params(state.params)]
  tree = {
      'params': state.params,
      'opt': {'m': checkpoint.unflatten(dict(zip(names, state.m))),
              'v': checkpoint.unflatten(dict(zip(names, state.v)))},
      'global_step': torch.tensor(state.global_step, dtype=torch.int64),
      'opt_count': torch.tensor(state.opt_count, dtype=torch.int64),
      'rng': torch.tensor(state.rng, dtype=torch.int64),
  }
  if state.dynamic_scale is not None:
    tree['dynamic_scale'] = {'scale': torch.tensor(state.dynamic_scale.scale, dtype=torch.float64),
                             'fin_steps': torch.tensor(state.dynamic_scale.fin_steps, dtype=torch.int64)}
  checkpoint.save_npz(path, tree)


def load_train_state(path, template: TrainState) -> TrainState:
  """Restore a ``save_train_state`` file into the structure (names, shapes, devices) of
  ``template`` (``trainer.py:437-440``: ``restore_checkpoint(workdir, train_state)``)."""
  from snap_amd.utils import checkpoint
  tree = checkpoint.load_npz(path)
  names = [n for n, _ in flatten_params(template.params)]
  params = checkpoint.load_into(template.params, tree['params'])
  tmpl_m = checkpoint.unflatten(dict(zip(names, template.m)))
  m = checkpoint.load_into(tmpl_m, tree['opt']['m'])
  v = checkpoint.load_into(tmpl_m, tree['opt']['v'])
  ds = template.dynamic_scale
  if ds is not None and 'dynamic_scale' in tree:
    ds = dataclasses.replace(ds, scale=float(tree['dynamic_scale']['scale']),
                             fin_steps=int(tree['dynamic_scale']['fin_steps']))
  return TrainState(params=params, m=[t for _, t in flatten_params(m)], v=[t for _, t in flatten_params(v)],
                    global_step=int(tree['global_step']), rng=int(tree['rng']),
                    opt_count=int(tree.get('opt_count', tree['global_step'])), dynamic_scale=ds)


FUSED_ADAM = True     # device tensors: one HIP launch over every parameter (optim.hip)


def _adam_update_(leaves, grads, m, v, step, lr, b1=None, b2=None, eps=None, apply_flag=None):
  """optax.adam (bias-corrected, eps outside the sqrt); in place on `leaves`.  apply_flag (0-d f32
  device tensor, fused path only): the update applies only where it is > 0."""
  b1 = ADAM_B1 if b1 is None else b1
  b2 = ADAM_B2 if b2 is None else b2
  eps = ADAM_EPS if eps is None else eps
  if _fusable(leaves, grads, m, v):
    from snap_amd import ops_bwd
    ops_bwd.adam_update_(leaves, grads, m, v, step, lr, b1, b2, eps, apply_flag=apply_flag)
    return
  if apply_flag is not None:
    raise ValueError('_adam_update_: apply_flag needs the fused device path')
  torch._foreach_mul_(m, b1)
  torch._foreach_add_(m, grads, alpha=1 - b1)
  torch._foreach_mul_(v, b2)
  torch._foreach_addcmul_(v, grads, grads, value=1 - b2)
  c1 = 1 - b1 ** step
  c2 = 1 - b2 ** step
  denom = torch._foreach_sqrt(v)
  torch._foreach_div_(denom, math.sqrt(c2))
  torch._foreach_add_(denom, eps)
  torch._foreach_addcdiv_(leaves, m, denom, value=-lr / c1)


ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8      # optax.adam defaults (trainer.py:236-243): ONE set for every path


def _fusable(leaves, grads, m, v):
  """The fused launch takes contiguous f32 device tensors, each listed once."""
  if not (FUSED_ADAM and leaves and leaves[0].is_cuda):
    return False
  seen = set()
  for t4 in zip(leaves, grads, m, v):
    for t in t4:
      if t.dtype != torch.float32 or not t.is_cuda:
        return False
    if not (t4[0].is_contiguous() and t4[2].is_contiguous() and t4[3].is_contiguous()):
      return False
    if t4[0].numel() and t4[0].data_ptr() in seen:
      return False
    seen.add(t4[0].data_ptr())
  return True


def _global_norm(tensors):
  """sqrt(sum ||t||^2): one multi-tensor launch + a tiny fp64 reduction."""
  return torch.linalg.vector_norm(torch.stack(torch._foreach_norm(tensors)).double())


def train_step(state: TrainState, batch, *, model, lr_fn: Callable, max_grad_norm=None,
               group=None, debug=False, overlap_allreduce=True, precision=None):
  """One optimisation step.  Returns (state, metrics, training_logs).

  precision: None (the reference's behaviour: the arithmetic is the MODEL's -- ``model.engine``, chosen
  by the ``dtype`` it was built with, models/base.py; the process default where the model names none)
  or an explicit engine for this step: 'f32' -- every conv / dense (forward, dgrad, wgrad) on the exact f32 matrix
  cores; 'bf16' -- their operands are rounded to bf16 with f32 accumulation (the analogue of
  the reference's ``dtype='float16'`` train config, train_localization.py:25, trainer.py:391;
  bf16 keeps the f32 exponent range, so no DynamicScale loss scaling is needed); 'fp16' -- the
  reference's configuration itself: operands (activations, gradients AND the kernel images the
  engine multiplies: resnet.py:97 param_dtype) rounded to IEEE half, f32 accumulate, to be run with
  ``TrainState.dynamic_scale = DynamicScale(minimum_scale=256)`` (trainer.py:391-392): an overflow
  to inf in the half-precision operands reaches the gradients, the step is skipped and the scale
  halves; 'bf16x3' /
  'bf16x6' -- forward and data-gradient convs / denses on the f32-grade split-bf16 engine
  (``conv_split.hip``: 2 / 3 bf16 parts per operand, f32 accumulate), kernel gradients on the
  exact f32 engine: an f32-class step at well under the f32 engine's cost.  Parameters,
  gradients, Adam state, GroupNorm statistics and all non-GEMM kernels stay f32.

  With more than one rank and ``overlap_allreduce`` the gradient buckets are all-reduced
  while the backward pass is still running (``dist.OverlappedGradReducer``); otherwise one
  bucketed all-reduce follows the backward pass.  Same averaged gradients either way."""
  named = flatten_params(state.params)
  leaves = [t for _, t in named]
  for t in leaves:
    t.requires_grad_(True)
    t.grad = None
  state.rng += 1
  rank = torch.distributed.get_rank(group) if sdist._world(group) > 1 else 0
  sampling_rng = state.rng * 7919 + rank            # bind the stream to the device
  model_engine = getattr(model, 'engine', None)
  if precision is None:
    precision = model_engine
  if precision is not None and precision not in ops.ENGINES:
    raise ValueError(f'train_step: precision={precision!r}')
  if model_engine is not None and precision != model_engine:
    # Module.apply enters the model's own engine scope, which would silently win over this argument
    raise ValueError(f'train_step: precision={precision!r} contradicts the engine the model was built with '
                     f'({model_engine!r}: BaseModel(dtype=..., engine=...)); build the model for the engine '
                     'or pass precision=None')
  if precision == 'fp16' and state.dynamic_scale is None:
    import warnings
    warnings.warn("train_step: IEEE-half engine ('fp16') without TrainState.dynamic_scale -- the reference trains "
                  'float16 under DynamicScale(minimum_scale=256) (trainer.py:391-392); see '
                  'trainer.dtype_and_dynamic_scale', RuntimeWarning, stacklevel=2)
  loss_scale = state.dynamic_scale.scale if state.dynamic_scale is not None else None
  # a per-thread scope; the autograd nodes carry it into the backward thread (autograd._engine_scoped)
  with ops.engine_scope(precision):
    grads, loss, losses, metrics = _forward_backward(state, batch, model, leaves, sampling_rng,
                                                     group, debug, overlap_allreduce, loss_scale)
  for t in leaves:
    t.requires_grad_(False)
    t.grad = None
  logs = {}
  if max_grad_norm is not None:
    gn = _global_norm(grads)
    factor = torch.clamp(max_grad_norm / (gn + 1e-6), max=1.0)
    torch._foreach_mul_(grads, factor)
  # The reference restores the whole opt_state on a skipped step, optax's step and schedule
  # counts included (trainer.py:269-276): bias correction and schedule follow opt_count.
  lr = lr_fn(state.opt_count)
  logs['learning_rate'] = lr
  fin_t = sdist.all_finite_tensor(grads, group).to(torch.float32).reshape(())
  gnorm_t = _global_norm(grads).reshape(())
  device_skip = _fusable(leaves, grads, state.m, state.v)
  if device_skip:
    # ONE host transfer per step, at its end: the update kernel itself reads the finite flag on the
    # device and applies nothing when a gradient is non-finite (the reference does the same inside
    # the traced step, trainer.py:269-276) -- the host need not know before it launches it
    with torch.no_grad():
      _adam_update_(leaves, grads, state.m, state.v, state.opt_count + 1, lr, apply_flag=fin_t)
    is_fin = None
  else:
    head = torch.stack([fin_t.to(torch.float64), gnorm_t]).cpu()
    is_fin = bool(head[0] > 0)
    if is_fin:                                           # otherwise: skip the update
      with torch.no_grad():
        _adam_update_(leaves, grads, state.m, state.v, state.opt_count + 1, lr)
  with torch.no_grad():
    per_example = {k: v.detach() for k, v in metrics.items()}      # (stacked per dtype and widened once by the reduce)
    for k, v in losses.items():
      per_example[f'loss/{k}'] = v.detach()
    keys, means = sdist.reduce_batch_metrics_tensor(per_example, batch['batch_mask'], group)
    tail = torch.cat([torch.stack([_global_norm(leaves).reshape(()), loss.detach().to(torch.float64).reshape(()),
                                   fin_t.to(torch.float64), gnorm_t.to(torch.float64)]),
                      means.to(torch.float64)]).cpu().tolist()
  if is_fin is None:
    is_fin = tail[2] > 0
  if is_fin:
    state.opt_count += 1
  if state.dynamic_scale is not None:
    state.dynamic_scale = state.dynamic_scale.update(is_fin)
    logs['loss_scale'] = state.dynamic_scale.scale
  logs['l2_grads'] = tail[3]
  logs['is_finite'] = is_fin
  logs['l2_params'] = tail[0]
  logs['loss'] = tail[1]
  reduced = dict(zip(keys, tail[4:]))
  state.global_step += 1
  return state, reduced, logs


def _forward_backward(state, batch, model, leaves, sampling_rng, group, debug, overlap_allreduce,
                      loss_scale=None):
  """Loss + (all-reduced) gradients of one batch.  ``loss_scale``: DynamicScale -- differentiate
  loss * scale, return the gradients divided by it (dynamic_scale.value_and_grad)."""
  with torch.enable_grad():
    pred = model.flax_model.apply(
        {'params': state.params}, batch, train=True, rngs={'sampling': sampling_rng},
        mutable=False, debug=debug,
    )
    losses, metrics = model.loss_metrics_function(pred, batch, state.params)
    # mean(where=batch_mask) (trainer.py:221): a non-finite loss on a padding example must
    # not reach the sum (NaN * 0 = NaN), so select instead of multiplying.
    mask = batch['batch_mask'].to(torch.bool)
    total = losses['total']
    loss = torch.where(mask, total, torch.zeros_like(total)).sum() / mask.sum().clamp(min=1)
    objective = loss if loss_scale is None else loss * loss_scale
    from snap_amd import ops_bwd
    # (this backward pass owns its intermediate gradients: VJP nodes may overwrite the buffer they are
    #  handed instead of cloning it -- ops_bwd.owning_scratch_grads)
    with ops_bwd.owning_scratch_grads():
      if sdist._exchanges(group) and overlap_allreduce:
        reducer = sdist.OverlappedGradReducer(leaves, group).attach()
        objective.backward()                             # buckets go out as their grads land
        grads = reducer.finish()                         # jax.lax.pmean(grad, 'batch')
      else:
        grads = torch.autograd.grad(objective, leaves, allow_unused=True)
        grads = [torch.zeros_like(t) if g is None else g.contiguous() for g, t in zip(grads, leaves)]
        sdist.allreduce_mean_(grads, group)              # jax.lax.pmean(grad, 'batch')
    if loss_scale is not None:
      torch._foreach_mul_(grads, 1.0 / loss_scale)
  return grads, loss, losses, metrics
