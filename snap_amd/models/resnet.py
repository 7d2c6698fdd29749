"""BiT ResNet-v2 encoder on the HIP conv engine (``snap/models/resnet.py:34-216``).

Per residual unit the reference runs GroupNorm -> ReLU -> StdConv three times.
Here GroupNorm+ReLU never materialise: ``group_norm_stats`` reads the activation
once for (mean, rstd) and the consumer conv applies normalise+ReLU while staging
its A operand; the residual add is fused in conv3's epilogue.  Kernels keep the
Flax HWIO layout; StdConv standardisation (fp32, eps 1e-10) is a HIP kernel whose
result is cached per forward (map and query passes share parameters).
"""
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base


def get_block_desc(depth):
  """resnet.py:158-167."""
  if isinstance(depth, list):
    depth = tuple(depth)
  return {
      26: [2, 2, 2, 2],
      50: [3, 4, 6, 3],
      101: [3, 4, 23, 3],
      152: [3, 8, 36, 3],
      200: [3, 24, 36, 3],
  }.get(depth, depth)


def _std(ctx, kernel):
  if base.needs_grad(kernel):
    pre = getattr(ctx, 'pre_std', None)      # training: differentiable, never cached
    if pre is not None and id(kernel) in pre:
      return pre[id(kernel)]
    return ag.weight_standardize(kernel)
  return ctx.standardized(kernel, ops.weight_standardize)


def _kernels(tree, out):
  for k, v in tree.items():
    if isinstance(v, dict):
      _kernels(v, out)
    elif k == 'kernel':
      out.append(v)
  return out


def _conv_gn(ctx, x, kernel, gn_p, gn_stats=None, emit=True, presplit=False, **kw):
  """GroupNorm -> ReLU -> StdConv.  Training: one autograd node (statistics inside);
  inference: `gn_stats` may be shared between the convs reading the same input.
  emit: True -> the statistics of the output come out of the epilogue ('both': also those of
  relu(output), for an FPN level that reads it ReLU -> GroupNorm).
  presplit: x has this one consumer (the 3x3 / closing 1x1 conv of a unit): where the engine
  allows, GroupNorm -> ReLU -> split is ONE pass over x (``ops.gn_norm_split``, which also stands
  in for the statistics finalize launch) and the conv runs on the pre-split engine."""
  w = _std(ctx, kernel)
  mode = None if not emit else ('both' if emit == 'both' else 'raw')
  if base.needs_grad(x, w, gn_p['scale'], gn_p['bias']):
    return ag.conv2d(x, w, prologue=ops.PRO_GN_RELU, gn_params=(gn_p['scale'], gn_p['bias']),
                     emit_gn_stats='raw' if emit else None, **kw)
  if presplit and ops.USE_PRESPLIT and gn_stats is None and w.shape[2] % 16 == 0:
    xs = ops.gn_norm_split(x, gn_p['scale'].reshape(-1), gn_p['bias'].reshape(-1))
    if xs is not None:
      return ops.conv2d(xs, w, emit_gn_stats=mode, **kw)
  # the output feeds the next GroupNorm: its statistics come out of this conv's epilogue
  return ops.conv2d(x, w, prologue=ops.PRO_GN_RELU, gn=gn_stats or _gn(x, gn_p),
                    emit_gn_stats=mode, **kw)


SHARE_PROJ_GN = True     # training: a projection unit's conv1 + conv_proj as one autograd node (see residual_unit)
FORK_SHORTCUT = True     # training: see residual_unit (False = autograd adds the two gradients of a unit's input)


def _gn(x, p, relu_first=False):
  mu, sc = ops.group_norm_stats(x, p['scale'].reshape(-1), relu_first=relu_first)
  return mu, sc, p['bias'].reshape(-1)


def residual_unit(ctx, p, x, stride, nmid, last_of_stage=False):
  """Bottleneck unit (resnet.py:103-132).  last_of_stage: the output is also an FPN level's
  input (ReLU -> GroupNorm): the closing conv emits both kinds of statistics."""
  nmid = nmid or x.shape[-1] // 4
  nout = nmid * 4
  train = base.needs_grad(x, p['conv1']['kernel'])
  gn1 = None if train else _gn(x, p['gn1'])     # shared by conv_proj and conv1
  proj = x.shape[-1] != nout or stride != 1
  if train and proj and SHARE_PROJ_GN and x.shape[-1] % 4 == 0 and x.shape[-1] == p['conv1']['kernel'].shape[2]:
    # conv1 and conv_proj read the same GroupNorm -> ReLU of x: one autograd node (one statistics pass,
    # one GroupNorm VJP over x; the second data gradient accumulates onto the first inside the conv engine)
    y, residual = ag.conv2d_shared_gn(x, _std(ctx, p['conv1']['kernel']), _std(ctx, p['conv_proj']['kernel']),
                                      (p['gn1']['scale'], p['gn1']['bias']), stride2=stride, emit_gn_stats='raw')
    y = _conv_gn(ctx, y, p['conv2']['kernel'], p['gn2'], stride=stride, padding=((1, 1), (1, 1)), presplit=True)
    return _conv_gn(ctx, y, p['conv3']['kernel'], p['gn3'], residual=residual, presplit=True,
                    emit='both' if (last_of_stage and ops.GN_STATS_BOTH) else True)
  if train and FORK_SHORTCUT:
    # conv1's node also hands out the alias of x the shortcut reads: the shortcut's gradient then
    # arrives in conv1's backward and is added where dx is written (no separate add pass)
    y, xs = _conv_gn(ctx, x, p['conv1']['kernel'], p['gn1'], gn1, fork_input=True)
  else:
    y, xs = None, x
  if x.shape[-1] != nout or stride != 1:
    residual = _conv_gn(ctx, xs, p['conv_proj']['kernel'], p['gn1'], gn1, emit=False, stride=stride)
  else:
    residual = xs
  if y is None:
    y = _conv_gn(ctx, x, p['conv1']['kernel'], p['gn1'], gn1)
  y = _conv_gn(ctx, y, p['conv2']['kernel'], p['gn2'], stride=stride, padding=((1, 1), (1, 1)),
               presplit=True)
  y = _conv_gn(ctx, y, p['conv3']['kernel'], p['gn3'], residual=residual, presplit=True,
               emit='both' if (last_of_stage and ops.GN_STATS_BOTH) else True)
  return y


def _init_unit(gen, device, cin, nmid, stride):
  nout = nmid * 4
  def gn(c):
    return {'scale': torch.ones(1, 1, 1, c, device=device),
            'bias': torch.zeros(1, 1, 1, c, device=device)}
  def conv(kh, kw, ci, co):
    return {'kernel': base.lecun_normal(gen, (kh, kw, ci, co), kh * kw * ci, device)}
  p = {'gn1': gn(cin), 'conv1': conv(1, 1, cin, nmid), 'gn2': gn(nmid),
       'conv2': conv(3, 3, nmid, nmid), 'gn3': gn(nmid),
       'conv3': conv(1, 1, nmid, nout)}
  if cin != nout or stride != 1:
    p['conv_proj'] = conv(1, 1, cin, nout)
  return p


class ResNetV2(base.Module):
  """BiT variant (resnet.py:170-216); returns {stage: {unit: activation}}."""

  def __init__(self, config, dtype=torch.float32, in_channels=3):
    self.config = config
    self.in_channels = in_channels   # (Flax infers it at init; 3 for images)
    self.blocks = get_block_desc(config.depth)
    if config.limit_num_blocks is not None:
      self.blocks = self.blocks[: config.limit_num_blocks]
    self.level_names = [f'stage{i + 1}' for i in range(len(self.blocks))]
    self.width = int(64 * config.width)
    if self.width % 32 != 0:
      raise ValueError('GroupNorm(32) needs the base width to be a multiple of 32')

  def init_params(self, gen, device):
    w = self.width
    params = {}
    if self.config.skip_root_block:
      c = self.in_channels
      params['conv_root'] = {'kernel': base.lecun_normal(gen, (3, 3, c, w), 9 * c, device)}
    else:
      c = self.in_channels
      params['root_block'] = {
          'conv_root': {'kernel': base.lecun_normal(gen, (7, 7, c, w), 49 * c, device)}
      }
    cin = w
    for i, size in enumerate(self.blocks):
      nmid = w * 2**i
      stage = {}
      for u in range(size):
        stride = 2 if (u == 0 and i > 0) else 1
        stage[f'unit{u + 1:02d}'] = _init_unit(gen, device, cin, nmid, stride)
        cin = nmid * 4
      params[f'block{i + 1}'] = stage
    return params

  def __call__(self, params, image, *, train=False, ctx=None, rng=None):
    ctx = ctx or base.ForwardContext()
    out = {}
    kernels = _kernels(params, [])
    if base.needs_grad(*kernels):
      # training: every StdConv kernel of the encoder standardised by one launch (and one
      # backward launch) instead of ~53 latency-bound ones.
      pre = dict(getattr(ctx, 'pre_std', None) or {})
      std = ag.weight_standardize_multi(kernels)
      pre.update({id(k): s for k, s in zip(kernels, std)})
      ctx.pre_std = pre
      if ops.precision() in ops.HALF_MATH:
        # ... and their bf16 / fp16 images, forward and rotated (data gradient), by one launch
        ops.pack_weights_bf16_multi(list(std), math=ops.precision())
    else:   # inference: one launch for all StdConv kernels of this encoder
      ctx.standardize_all(kernels, ops.weight_standardize_multi)
      if ops.precision() in ops.SPLIT_PARTS:
        # split-bf16 engine: the weight images of all standardised kernels, one launch
        ops.pack_weights_split_multi([ctx._lookup(k) for k in kernels], ops.precision())
    # `image * 2 - 1` (resnet.py:199) is fused into the root conv's operand staging.
    # (the image may carry padding channels: image_encoder.pad_to_multiple(channel_pad=...))
    w_root = params['conv_root' if self.config.skip_root_block else 'root_block']
    w_root = w_root['kernel'] if self.config.skip_root_block else w_root['conv_root']['kernel']
    cin_kw = {'cin': w_root.shape[2]} if image.shape[-1] != w_root.shape[2] else {}
    if self.config.skip_root_block:
      w = _std(ctx, params['conv_root']['kernel'])
      conv = ag.conv2d if base.needs_grad(w) else ops.conv2d
      x = conv(image, w, padding=((1, 1), (1, 1)), prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0),
               emit_gn_stats='raw', **cin_kw)
    else:
      w = _std(ctx, params['root_block']['conv_root']['kernel'])
      conv = ag.conv2d if base.needs_grad(w) else ops.conv2d
      x = conv(image, w, stride=2, padding=((3, 3), (3, 3)), prologue=ops.PRO_AFFINE,
               in_affine=(2.0, -1.0), **cin_kw)
      pool = ag.max_pool_3x3s2 if base.needs_grad(x) else ops.max_pool_3x3s2
      x = out['stem'] = pool(x)
    for i, size in enumerate(self.blocks):
      stage = {}
      nmid = self.width * 2**i
      for u in range(size):
        name = f'unit{u + 1:02d}'
        stride = 2 if (u == 0 and i > 0) else 1
        x = stage[name] = residual_unit(ctx, params[f'block{i + 1}'][name], x, stride, nmid,
                                        last_of_stage=u + 1 == size)
      out[f'stage{i + 1}'] = stage
    return out

  default_config = staticmethod(default_configs.resnet)
