"""Oracle (test infrastructure): ViT image encoder in numpy.

PARITY UNPINNED: the reference has NO ViT (``snap/models/image_encoder.py:103`` accepts only
``encoder_name == 'resnet'``); BASELINE.json ``configs[4]`` asks for a ViT-B/16 encoder anyway.
This restates the published architecture (Dosovitskiy et al. 2021, "An Image is Worth 16x16
Words") with the parameter tree of the big_vision / scenic ``vit.py`` encoders (the family the
reference's BiT ResNet loader, ``resnet.py:223-233``, comes from): ``embedding`` (patch conv),
``pos_embedding``, ``Transformer/encoderblock_i/{LayerNorm_0, MultiHeadDotProductAttention_0/
{query,key,value,out}, LayerNorm_1, MlpBlock_0/{Dense_0,Dense_1}}``, ``Transformer/encoder_norm``,
plus a linear ``proj`` to the feature dimension of the lift.  No class token (dense features).
It checks the HIP kernels against plain float64 maths; there is nothing upstream to pin it to.
"""
import numpy as np

from oracle import encoder as o_enc


def gelu_tanh(x):
  """flax.linen.gelu (approximate=True)."""
  return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x**3)))


def layer_norm(x, gamma, beta, eps=1e-6):
  """flax.linen.LayerNorm over the last axis (biased variance, eps inside the sqrt)."""
  mean = x.mean(-1, keepdims=True)
  var = np.square(x - mean).mean(-1, keepdims=True)
  return (x - mean) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def attention(qkv, scale=None, bf16_operands=False):
  """qkv [B, N, 3, H, D] -> softmax(scale * q k^T) v, heads concatenated: [B, N, H*D].

  bf16_operands: round q * scale * log2(e), k and v to bf16 first, as the HIP kernel does before
  its matrix-core products (its probabilities are rounded to bf16 too, relative to a running
  maximum -- not modelled; that leaves an error of order 2^-9 of the value range)."""
  B, N, _, H, D = qkv.shape
  scale = D ** -0.5 if scale is None else scale
  q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]           # [B, N, H, D]
  if bf16_operands:
    l2e = np.float32(scale * 1.4426950408889634)
    q = o_enc.bf16_round(q.astype(np.float32) * l2e).astype(np.float64) / float(l2e)
    k = o_enc.bf16_round(k.astype(np.float32)).astype(np.float64)
    v = o_enc.bf16_round(v.astype(np.float32)).astype(np.float64)
  s = np.einsum('bqhd,bkhd->bhqk', q, k) * scale
  s = s - s.max(-1, keepdims=True)
  p = np.exp(s)
  p = p / p.sum(-1, keepdims=True)
  return np.einsum('bhqk,bkhd->bqhd', p, v).reshape(B, N, H * D)


def resize_posemb(posemb, grid_from, grid_to):
  """Bilinear resize (half-pixel centres, edge clamp) of a [1, h*w, C] position embedding."""
  if tuple(grid_from) == tuple(grid_to):
    return posemb
  h0, w0 = grid_from
  h1, w1 = grid_to
  p = posemb.reshape(h0, w0, -1)
  ys = np.clip((np.arange(h1) + 0.5) * h0 / h1 - 0.5, 0, h0 - 1)
  xs = np.clip((np.arange(w1) + 0.5) * w0 / w1 - 0.5, 0, w0 - 1)
  y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, h0 - 1); fy = (ys - y0)[:, None, None]
  x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, w0 - 1); fx = (xs - x0)[None, :, None]
  top = p[y0][:, x0] * (1 - fx) + p[y0][:, x1] * fx
  bot = p[y1][:, x0] * (1 - fx) + p[y1][:, x1] * fx
  return (top * (1 - fy) + bot * fy).reshape(1, h1 * w1, -1).astype(posemb.dtype)


def vit_encoder(params, config, image):
  """image [N, H, W, 3] in [0, 1] (H, W multiples of the patch size) -> [N, H/p, W/p, output_dim]."""
  p = config['patch_size']
  heads = config['num_heads']
  x = image * image.dtype.type(2) - image.dtype.type(1)                    # as resnet.py:199
  x = o_enc.conv2d(x, params['embedding']['kernel'], (p, p)) + params['embedding']['bias']
  N, h, w, C = x.shape
  x = x.reshape(N, h * w, C) + resize_posemb(params['pos_embedding'], config['posemb_grid'], (h, w))
  D = C // heads
  for i in range(config['num_layers']):
    blk = params['Transformer'][f'encoderblock_{i}']
    y = layer_norm(x, blk['LayerNorm_0']['scale'], blk['LayerNorm_0']['bias'])
    att = blk['MultiHeadDotProductAttention_0']
    qkv = np.stack([
        y @ att[n]['kernel'].reshape(C, C) + att[n]['bias'].reshape(C) for n in ('query', 'key', 'value')
    ], axis=2).reshape(N, h * w, 3, heads, D)
    a = attention(qkv)
    x = x + a @ att['out']['kernel'].reshape(C, C) + att['out']['bias']
    y = layer_norm(x, blk['LayerNorm_1']['scale'], blk['LayerNorm_1']['bias'])
    mlp = blk['MlpBlock_0']
    y = gelu_tanh(y @ mlp['Dense_0']['kernel'] + mlp['Dense_0']['bias'])
    x = x + y @ mlp['Dense_1']['kernel'] + mlp['Dense_1']['bias']
  norm = params['Transformer']['encoder_norm']
  x = layer_norm(x, norm['scale'], norm['bias'])
  x = x @ params['proj']['kernel'] + params['proj']['bias']
  return x.reshape(N, h, w, -1)
