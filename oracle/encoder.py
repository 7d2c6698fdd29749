"""Oracle (test infrastructure): BiT ResNet-v2 + FPN image encoder in numpy.

Restates ``snap/models/resnet.py`` and ``snap/models/image_encoder.py``.
Tensors are channels-last (NHWC), conv kernels HWIO, Dense kernels (in, out) --
exactly the Flax parameter layout, so a Flax param tree drops in unchanged.
"""
import numpy as np

from oracle import configs


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def conv2d(x, kernel, strides=(1, 1), padding=((0, 0), (0, 0)), bias=None):
  """flax.linen.Conv (NHWC x HWIO cross-correlation), explicit padding."""
  kh, kw, cin, cout = kernel.shape
  n, h, w, c = x.shape
  assert c == cin, (c, cin)
  (pt, pb), (pl, pr) = padding
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  sh, sw = strides
  ho = (h + pt + pb - kh) // sh + 1
  wo = (w + pl + pr - kw) // sw + 1
  s = xp.strides
  patches = np.lib.stride_tricks.as_strided(
      xp,
      shape=(n, ho, wo, kh, kw, cin),
      strides=(s[0], s[1] * sh, s[2] * sw, s[1], s[2], s[3]),
      writeable=False,
  )
  out = patches.reshape(n * ho * wo, kh * kw * cin) @ kernel.reshape(-1, cout)
  out = out.reshape(n, ho, wo, cout)
  if bias is not None:
    out = out + bias
  return out.astype(x.dtype)


def fp16_round(x):
  """float32 -> nearest IEEE binary16 (ties to even; beyond 65504 -> inf, tiny values flush through
  the subnormals), returned as float32: the rounding step of the 'fp16' training engine -- the
  reference's ``dtype='float16'`` train config itself (``snap/configs/train_localization.py:93``,
  ``snap/models/resnet.py:97``)."""
  with np.errstate(over='ignore'):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def bf16_round(x):
  """float32 -> nearest bfloat16 (ties to even), returned as float32.

  The training-precision conv engine multiplies bf16-rounded operands with f32
  accumulation (the build's analogue of the reference's ``dtype='float16'`` train
  config, ``snap/configs/train_localization.py:25``); this is the rounding step of
  its restatement.
  """
  u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
  r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
  out = r.view(np.float32)
  return np.where(np.isfinite(x), out, np.asarray(x, dtype=np.float32))


def same_padding(size, k, s):
  """XLA 'SAME' padding for one spatial dim."""
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return (total // 2, total - total // 2)


def max_pool(x, window=(3, 3), strides=(2, 2), padding=((1, 1), (1, 1))):
  """flax.linen.max_pool: pads with -inf (resnet.py:99)."""
  kh, kw = window
  (pt, pb), (pl, pr) = padding
  xp = np.pad(
      x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf
  )
  n, h, w, c = xp.shape
  sh, sw = strides
  ho = (h - kh) // sh + 1
  wo = (w - kw) // sw + 1
  s = xp.strides
  patches = np.lib.stride_tricks.as_strided(
      xp,
      shape=(n, ho, wo, kh, kw, c),
      strides=(s[0], s[1] * sh, s[2] * sw, s[1], s[2], s[3]),
      writeable=False,
  )
  return patches.max(axis=(3, 4))


def standardize(x, axis, eps):
  """resnet.py:34-41 -- fp32 (here: the array's dtype, >= fp32) statistics."""
  dtype = x.dtype
  ctype = np.float64 if dtype == np.float64 else np.float32
  x = x.astype(ctype)
  x = x - np.mean(x, axis=axis, keepdims=True)
  x = x / np.sqrt(np.mean(np.square(x), axis=axis, keepdims=True) + ctype(eps))
  return x.astype(dtype)


def group_norm(x, scale, bias, ngroups=32):
  """resnet.py:46-70."""
  input_shape = x.shape
  group_shape = x.shape[:-1] + (ngroups, x.shape[-1] // ngroups)
  x = x.reshape(group_shape)
  x = standardize(x, axis=(1, 2, 4), eps=1e-5)
  x = x.reshape(input_shape)
  return x * scale.reshape(1, 1, 1, -1) + bias.reshape(1, 1, 1, -1)


def std_conv(x, kernel, strides=(1, 1), padding=((0, 0), (0, 0))):
  """resnet.py:73-79: weight-standardised conv (over H, W, I per out-channel)."""
  kernel = standardize(kernel, axis=(0, 1, 2), eps=1e-10)
  return conv2d(x, kernel, strides, padding)


def relu(x):
  return np.maximum(x, 0)


def resize_bilinear_x2(x):
  """jax.image.resize(..., 'bilinear') for an exact x2 up-sampling.

  Half-pixel centres, edge taps renormalised == edge clamp
  (image_encoder.py:90).
  """
  def up(a, axis):
    size = a.shape[axis]
    dst = np.arange(2 * size)
    src = (dst + 0.5) / 2 - 0.5
    lo = np.floor(src).astype(np.int64)
    w_hi = (src - lo).astype(a.dtype)
    w_lo = (1 - w_hi).astype(a.dtype)
    i_lo = np.clip(lo, 0, size - 1)
    i_hi = np.clip(lo + 1, 0, size - 1)
    shape = [1] * a.ndim
    shape[axis] = -1
    return (
        np.take(a, i_lo, axis) * w_lo.reshape(shape)
        + np.take(a, i_hi, axis) * w_hi.reshape(shape)
    )
  return up(up(x, 1), 2)


# ----------------------------------------------------------------------------
# ResNet-v2 (BiT)
# ----------------------------------------------------------------------------
def residual_unit(params, x, strides=(1, 1), nmid=None):
  """resnet.py:103-132."""
  nmid = nmid or x.shape[-1] // 4
  nout = nmid * 4
  residual = x
  x = relu(group_norm(x, params['gn1']['scale'], params['gn1']['bias']))
  if x.shape[-1] != nout or tuple(strides) != (1, 1):
    # 1x1 conv, Flax default padding 'SAME' == no padding for a 1x1 kernel.
    residual = std_conv(x, params['conv_proj']['kernel'], strides)
  x = std_conv(x, params['conv1']['kernel'])
  x = relu(group_norm(x, params['gn2']['scale'], params['gn2']['bias']))
  x = std_conv(x, params['conv2']['kernel'], strides, ((1, 1), (1, 1)))
  x = relu(group_norm(x, params['gn3']['scale'], params['gn3']['bias']))
  x = std_conv(x, params['conv3']['kernel'])
  return x + residual


def resnet_stage(params, x, block_size, nmid, first_stride=(1, 1)):
  """resnet.py:135-155."""
  out = {}
  x = out['unit01'] = residual_unit(params['unit01'], x, first_stride, nmid)
  for i in range(1, block_size):
    name = f'unit{i + 1:02d}'
    x = out[name] = residual_unit(params[name], x, (1, 1), nmid)
  return x, out


def resnet_v2(params, config, image):
  """resnet.py:170-216.  Returns {'stem'?, 'stage1': {'unit01': ...}, ...}."""
  blocks = configs.get_block_desc(config['depth'])
  if config.get('limit_num_blocks') is not None:
    blocks = blocks[: config['limit_num_blocks']]
  width = int(64 * config['width'])
  out = {}
  x = image * 2 - 1
  if config['skip_root_block']:
    x = std_conv(x, params['conv_root']['kernel'], (1, 1), ((1, 1), (1, 1)))
  else:
    x = std_conv(
        x, params['root_block']['conv_root']['kernel'], (2, 2),
        ((3, 3), (3, 3)),
    )
    x = out['stem'] = max_pool(x)
  x, out['stage1'] = resnet_stage(params['block1'], x, blocks[0], width)
  for i, block_size in enumerate(blocks[1:], 1):
    x, out[f'stage{i + 1}'] = resnet_stage(
        params[f'block{i + 1}'], x, block_size, width * 2**i, (2, 2)
    )
  return out


# ----------------------------------------------------------------------------
# FPN decoder + ImageEncoder
# ----------------------------------------------------------------------------
def fpn_decoder(params, input_features):
  """image_encoder.py:42-94 with activation='relu', norm='bit_resnet'."""
  out_features = []
  f_prev = None
  for level, f_skip in enumerate(input_features):
    f = relu(f_skip)
    f = group_norm(
        f,
        params[f'{level}_skip_norm']['scale'],
        params[f'{level}_skip_norm']['bias'],
    )
    f = conv2d(f, params[f'{level}_skip_conv']['kernel'])
    if f_prev is not None:
      assert f.shape[-3] == f_prev.shape[-3] * 2
      assert f.shape[-2] == f_prev.shape[-2] * 2
      f = f + resize_bilinear_x2(f_prev)
    f_prev = f
    out_features.append(f)
  return out_features


def pad_to_multiple(images, stride):
  """image_encoder.py:32-39 (quirk: a divisible size is padded by a full stride)."""
  shape = np.array(images.shape[-3:-1])
  pad = stride - shape % stride
  return np.pad(images, [(0, 0), (0, pad[0]), (0, pad[1]), (0, 0)])


def image_encoder(params, config, image):
  """image_encoder.py:97-144.  image: [N, H, W, 3].

  Returns dict(features=[coarse..fine], strides=[...]).
  """
  enc_cfg = config['encoder']
  blocks = configs.get_block_desc(enc_cfg['depth'])
  if enc_cfg.get('limit_num_blocks') is not None:
    blocks = blocks[: enc_cfg['limit_num_blocks']]
  level_names_all = [f'stage{i + 1}' for i in range(len(blocks))]
  num_pyr_levels = config.get('num_pyr_levels')
  if num_pyr_levels is None:
    num_pyr_levels = len(level_names_all)
  max_stride = (not enc_cfg['skip_root_block']) * 2 + num_pyr_levels - 1
  level_names = level_names_all[:num_pyr_levels][::-1]

  input_shape = np.array(image.shape[-3:-1])
  image_padded = pad_to_multiple(image, 2**max_stride)
  padded_shape = np.array(image_padded.shape[-3:-1])
  encoder_features = resnet_v2(params['encoder'], enc_cfg, image_padded)
  skip_features = []
  for layer_name in level_names:
    _, f = sorted(encoder_features[layer_name].items())[-1]
    skip_features.append(f)
  out_features = fpn_decoder(params['decoder'], skip_features)
  strides = [padded_shape / np.array(f.shape[-3:-1]) for f in out_features]
  crops = []
  for s, f in zip(strides, out_features):
    h, w = np.round(np.ceil(input_shape / s)).astype(int)
    crops.append(f[..., :h, :w, :])
  return dict(features=crops, strides=strides)


# ----------------------------------------------------------------------------
# layers.py
# ----------------------------------------------------------------------------
def dense(params, x):
  return x @ params['kernel'] + params['bias']


def mlp(params, config, x):
  """layers.py:55-78 (activation relu)."""
  for i, _ in enumerate(config['layers']):
    if i > 0 or config['apply_input_activation']:
      x = relu(x)
    x = dense(params[f'Dense_{i}'], x)
  return x


def normalize(x, axis=-1, eps=1e-5):
  """layers.py:45-52."""
  ctype = np.float64 if x.dtype == np.float64 else np.float32
  x_ = x.astype(ctype)
  invalid = np.linalg.norm(x_, axis=axis, keepdims=True) < eps
  y = np.where(invalid, ctype(eps), x_)
  z = x_ / np.linalg.norm(y, axis=axis, keepdims=True)
  return np.where(invalid, 0, z.astype(x.dtype)).astype(x.dtype)


def masked_softmax(x, mask, axis):
  """layers.py:37-42."""
  valid = mask.any(axis=axis, keepdims=True)
  mask = np.where(valid, mask, True)
  x = np.where(mask, x, -np.inf)
  x = x - x.max(axis=axis, keepdims=True)
  e = np.exp(x)
  return e / e.sum(axis=axis, keepdims=True)
