"""ViT image encoder for ``encoder_name='vit'`` (BASELINE.json configs[4]: ViT-B/16).

The reference has no ViT (``snap/models/image_encoder.py:103`` raises for anything but
'resnet'); this is the published ViT block with the big_vision / scenic parameter tree (see
``oracle/vit.py``), forward and hand-written backward.  Every Dense runs on the conv engine (bf16
operands by default -- ``config.matmul_precision``), LayerNorm and the fused-softmax attention
are the kernels of ``csrc/vit_ops.hip``; bias, GELU and the residual adds ride in the GEMM
epilogues.
"""
import numpy as np
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.models import base


class ViTEncoder(base.Module):
  """[N, H, W, 3] in [0, 1] -> dense features [N, H/p, W/p, output_dim] (no class token)."""

  def __init__(self, config, output_dim, dtype=torch.float32):
    if config.hidden_size % config.num_heads or config.hidden_size // config.num_heads != 64:
      raise ValueError('ViT: head dimension must be 64 (hidden_size / num_heads)')
    self.config = config
    self.output_dim = output_dim

  def init_params(self, gen, device):
    cfg = self.config
    C, p, H = cfg.hidden_size, cfg.patch_size, cfg.num_heads
    D = C // H
    gh, gw = cfg.posemb_grid

    def dense(cin, cout, shape=None, bshape=None):
      return {'kernel': base.glorot_uniform(gen, shape or (cin, cout), cin, cout, device),
              'bias': torch.zeros(bshape or (cout,), device=device)}

    def ln():
      return {'scale': torch.ones(C, device=device), 'bias': torch.zeros(C, device=device)}

    blocks = {}
    for i in range(cfg.num_layers):
      blocks[f'encoderblock_{i}'] = {
          'LayerNorm_0': ln(),
          'MultiHeadDotProductAttention_0': {
              'query': dense(C, C, (C, H, D), (H, D)), 'key': dense(C, C, (C, H, D), (H, D)),
              'value': dense(C, C, (C, H, D), (H, D)), 'out': dense(C, C, (H, D, C), (C,)),
          },
          'LayerNorm_1': ln(),
          'MlpBlock_0': {'Dense_0': dense(C, cfg.mlp_dim), 'Dense_1': dense(cfg.mlp_dim, C)},
      }
    blocks['encoder_norm'] = ln()
    return {
        'embedding': {'kernel': base.lecun_normal(gen, (p, p, 3, C), p * p * 3, device),
                      'bias': torch.zeros(C, device=device)},
        'pos_embedding': (torch.randn((1, gh * gw, C), generator=gen) * 0.02).to(device),
        'Transformer': blocks,
        'proj': dense(C, self.output_dim),
    }

  def _posemb(self, posemb, grid):
    gh, gw = self.config.posemb_grid
    if (gh, gw) == tuple(grid):
      return posemb
    p = posemb.reshape(1, gh, gw, -1).permute(0, 3, 1, 2)
    p = torch.nn.functional.interpolate(p, size=tuple(grid), mode='bilinear', align_corners=False)
    return p.permute(0, 2, 3, 1).reshape(1, grid[0] * grid[1], -1).contiguous()

  def _forward_train(self, params, image):
    """Differentiable path (hand-written VJP kernels through ``snap_amd.autograd``).  The GEMM
    precision follows ``ops.MATMUL_PRECISION`` (``trainer.train_step(precision=...)``); GELU
    runs as its own kernel so that the pre-activation is kept for the backward."""
    cfg = self.config
    C, H = cfg.hidden_size, cfg.num_heads
    D = C // H
    p = cfg.patch_size
    emb = params['embedding']
    x = ag.conv2d(image.contiguous(), emb['kernel'], stride=p, prologue=ops.PRO_AFFINE,
                  in_affine=(2.0, -1.0), bias=emb['bias'])
    N, h, w, _ = x.shape
    x = (x.reshape(N, h * w, C) + self._posemb(params['pos_embedding'], (h, w))).contiguous()
    for i in range(cfg.num_layers):
      blk = params['Transformer'][f'encoderblock_{i}']
      att = blk['MultiHeadDotProductAttention_0']
      wqkv = torch.cat([att[n]['kernel'].reshape(C, C) for n in ('query', 'key', 'value')], dim=1)
      bqkv = torch.cat([att[n]['bias'].reshape(C) for n in ('query', 'key', 'value')])
      y = ag.layer_norm(x, blk['LayerNorm_0']['scale'], blk['LayerNorm_0']['bias'])
      qkv = ag.dense(y, wqkv.contiguous(), bqkv).reshape(N, h * w, 3, H, D)
      a = ag.attention(qkv)
      x = ag.dense(a, att['out']['kernel'].reshape(C, C), att['out']['bias'], residual=x)
      y = ag.layer_norm(x, blk['LayerNorm_1']['scale'], blk['LayerNorm_1']['bias'])
      mlp = blk['MlpBlock_0']
      y = ag.gelu(ag.dense(y, mlp['Dense_0']['kernel'], mlp['Dense_0']['bias']))
      x = ag.dense(y, mlp['Dense_1']['kernel'], mlp['Dense_1']['bias'], residual=x)
    norm = params['Transformer']['encoder_norm']
    x = ag.layer_norm(x, norm['scale'], norm['bias'])
    x = ag.dense(x, params['proj']['kernel'], params['proj']['bias'])
    return x.reshape(N, h, w, self.output_dim)

  def __call__(self, params, image, train=False, ctx=None):
    if base.needs_grad(image, params['proj']['kernel'], params['embedding']['kernel']):
      return self._forward_train(params, image)
    cfg = self.config
    math = cfg.get('matmul_precision', 'bf16')
    C, H = cfg.hidden_size, cfg.num_heads
    D = C // H
    p = cfg.patch_size
    # patch embedding: p x p stride-p conv on 2*image-1 (3 input channels: exact f32 engine)
    x = ops.conv2d(image.contiguous(), params['embedding']['kernel'], stride=p,
                   prologue=ops.PRO_AFFINE, in_affine=(2.0, -1.0), bias=params['embedding']['bias'])
    N, h, w, _ = x.shape
    x = (x.reshape(N, h * w, C) + self._posemb(params['pos_embedding'], (h, w))).contiguous()
    # 'bf16' GEMMs: every dense layer's INPUT is produced in bf16 by the kernel in front of it (LayerNorm, attention,
    # the GELU epilogue of the first MLP layer) -- the values the engine would round to anyway -- so that both GEMM
    # operands travel by LDS-DMA (conv_ps.hip, one part); qkv is bf16 too (the attention rounds K / V to it anyway: its
    # panels move at half the bytes); the residual stream stays f32
    hb = math == 'bf16' and ops.tuning().BF16_PS and C % 16 == 0 and cfg.mlp_dim % 16 == 0
    for i in range(cfg.num_layers):
      blk = params['Transformer'][f'encoderblock_{i}']
      att = blk['MultiHeadDotProductAttention_0']
      wqkv = torch.cat([att[n]['kernel'].reshape(C, C) for n in ('query', 'key', 'value')], dim=1)
      bqkv = torch.cat([att[n]['bias'].reshape(C) for n in ('query', 'key', 'value')])
      y = ops.layer_norm(x, blk['LayerNorm_0']['scale'], blk['LayerNorm_0']['bias'], out_half=hb)
      qkv = ops.dense(y, wqkv.contiguous(), bqkv, math=math, bf16_ring=hb, out_half=hb).reshape(N, h * w, 3, H, D)
      a = ops.attention(qkv, out_half=hb)
      x = ops.dense(a, att['out']['kernel'].reshape(C, C), att['out']['bias'], residual=x, math=math, bf16_ring=hb)
      y = ops.layer_norm(x, blk['LayerNorm_1']['scale'], blk['LayerNorm_1']['bias'], out_half=hb)
      mlp = blk['MlpBlock_0']
      y = ops.dense(y, mlp['Dense_0']['kernel'], mlp['Dense_0']['bias'], gelu=True, math=math, out_half=hb, bf16_ring=hb)
      x = ops.dense(y, mlp['Dense_1']['kernel'], mlp['Dense_1']['bias'], residual=x, math=math, bf16_ring=hb)
    norm = params['Transformer']['encoder_norm']
    x = ops.layer_norm(x, norm['scale'], norm['bias'], out_half=hb)
    x = ops.dense(x, params['proj']['kernel'], params['proj']['bias'], math=math, bf16_ring=hb)
    return x.reshape(N, h, w, self.output_dim)


def pad_to_patch(images, patch):
  """Zero-pad H, W up to the next multiple of the patch size (nothing if already divisible)."""
  shape = np.array(images.shape[-3:-1])
  pad = (-shape) % patch
  if not pad.any():
    return images
  return torch.nn.functional.pad(images, (0, 0, 0, int(pad[1]), 0, int(pad[0])))
