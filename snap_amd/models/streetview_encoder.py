"""StreetView encoder: per-view image features lifted into a 3-D feature volume.

Mirrors ``snap/models/streetview_encoder.py:181-287``.  The projection, top-K view
selection, bilinear gather, depth-score interpolation and multi-view pooling of
the reference (five XLA stages that materialise [B,N,K,160] tensors) are ONE HIP
kernel here (``ops.lift_pool`` -> lift.hip); the fusion MLP runs on the MFMA conv
engine with the validity mask fused into its last epilogue.
"""
import copy

import numpy as np
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import image_encoder
from snap_amd.models import layers
from snap_amd.models import types


class StreetViewEncoder(base.Module):
  """Encode a set of posed images into a 3-D feature grid."""

  def __init__(self, config, dtype=torch.float32):
    if config.pretrained_path is not None:
      raise NotImplementedError(
          'pretrained_path: Flax checkpoint loading is out of scope '
          '(and broken in the reference, streetview_encoder.py:189).'
      )
    self.config = config
    self.image_encoder = image_encoder.ImageEncoder(config.image_encoder, dtype)
    fd = config.feature_dim
    self.weighted = bool(config.do_weighted_fusion)
    self.default_fusion = (self.weighted and bool(config.fusion_use_variance)
                           and not config.fusion_add_minmax)
    self.fusion_mlp = layers.MLP(config.fusion, in_dim=ops.pooled_channels(
        fd, self.weighted, bool(config.fusion_use_variance), bool(config.fusion_add_minmax)))
    self.proj_mlp = None
    if self.weighted:
      # fusion features and depth scores from one linear layer (streetview_encoder.py:207-213)
      proj_config = copy.deepcopy(config.proj_mlp)
      proj_config.layers = (fd + config.num_scale_bins,)
      self.proj_mlp = layers.MLP(proj_config, in_dim=config.image_encoder.output_dim)
    elif config.image_encoder.output_dim != fd:
      raise ValueError('do_weighted_fusion=False: the image features are pooled as they are, '
                       'image_encoder.output_dim must equal feature_dim')
    # per-observation correction MLP on [features, log10 depth, viewing ray]
    # (streetview_encoder.py:214-216, 263-267; only without the weighted fusion)
    self.depth_mlp = None
    if not self.weighted and config.depth_mlp is not None:
      if tuple(config.depth_mlp.layers)[-1] != fd:
        raise ValueError('depth_mlp must map back to feature_dim (its output is added to the features)')
      self.depth_mlp = layers.MLP(config.depth_mlp, in_dim=fd + 4)

  def init_params(self, gen, device):
    params = {'image_encoder': self.image_encoder.init_params(gen, device)}
    if self.proj_mlp is not None:
      params['proj_mlp'] = self.proj_mlp.init_params(gen, device)
    if self.depth_mlp is not None:
      params['depth_mlp'] = self.depth_mlp.init_params(gen, device)
    params['fusion_mlp'] = self.fusion_mlp.init_params(gen, device)
    return params

  def _project(self, p, f, train):
    """``proj_mlp`` over the image features [B, V, h, w, C].  The encoder hands them over as a CROP of its
    padded output (``pad_to_multiple`` pads 512 -> 544, image_encoder.py:32-39,139-143): instead of
    copying the crop (268 MB at C2) the single Dense of the projection gathers its rows from the padded
    tensor through the engine's row list (``rows_in``: a constant index table, uploaded once)."""
    cfgp = self.proj_mlp.config
    self.last_projection_path = 'copy'         # (tests assert 'rows' at the bench geometry: no silent 268 MB copy)
    if (train or f.is_contiguous() or not f.is_cuda or len(cfgp.layers) != 1 or not ops.NATIVE_GLUE
        or base.needs_grad(f, p['Dense_0']['kernel'], p['Dense_0']['bias'])):
      return self.proj_mlp(p, f.contiguous(), train)
    src = f._base
    B, V, h, w, C = f.shape
    st = f.stride()
    # image pitch of the flattened (scene, view) index; the stride of a size-1 axis carries no information
    # (the query slice has V = 1: its st[1] is arbitrary)
    pitch = st[1] if V > 1 else (st[0] if B > 1 else h * st[2])
    ok = (src is not None and src.is_contiguous() and src.dtype == torch.float32 and st[4] == 1 and st[3] == C
          and st[2] % C == 0 and pitch % st[2] == 0 and (B == 1 or V == 1 or st[0] == V * pitch)
          and (f.storage_offset() - src.storage_offset()) % C == 0)
    if not ok:
      return self.proj_mlp(p, f.contiguous(), train)
    Wp, HpWp = st[2] // C, pitch // C
    off = (f.storage_offset() - src.storage_offset()) // C
    M = B * V * h * w
    total = src.numel() // C
    if off + (B * V - 1) * HpWp + (h - 1) * Wp + w > total or total >= 2 ** 31:
      return self.proj_mlp(p, f.contiguous(), train)

    def make():
      n = np.arange(B * V, dtype=np.int64)[:, None, None] * HpWp
      rows = off + n + np.arange(h, dtype=np.int64)[None, :, None] * Wp + np.arange(w, dtype=np.int64)[None, None, :]
      return torch.from_numpy(rows.reshape(-1).astype(np.int32))
    rows = base.device_const(('proj_crop_rows', B * V, h, w, Wp, HpWp, off), f.device, make)
    count = base.device_const(('proj_crop_count', M), f.device, lambda: torch.tensor([M], dtype=torch.int32))
    pro = ops.PRO_RELU if cfgp.apply_input_activation else ops.PRO_NONE
    d = p['Dense_0']
    y = ops.dense(src.reshape(total, C), d['kernel'], d['bias'], cin=d['kernel'].shape[0], prologue=pro,
                  rows_in=rows, row_count=count)
    self.last_projection_path = 'rows'
    return y[:M].reshape(B, V, h, w, y.shape[-1])     # (rows past M are never written nor read)

  def _fused_pool_ok(self, params, f_images):
    """The fusion MLP + vertical max pooling run as ONE kernel (ops.mlp2_pool_max) when nothing
    needs gradients, the conv engine in use is the one the kernel is written for and the MLP
    has the two-layer shape of the reference's configs."""
    layers_ = tuple(self.config.fusion.layers)
    if len(layers_) != 2 or ops.precision() != 'bf16x3':
      return False
    p = params['fusion_mlp']
    if base.needs_grad(f_images, *(p[f'Dense_{i}'][k] for i in range(2) for k in ('kernel', 'bias'))):
      return False
    return ops.mlp2_pool_supported(self.fusion_mlp.in_dim, layers_[0], layers_[1])

  def __call__(self, params, data, train=False, ctx=None, rng=None, pool_max=False):
    """``pool_max``: the caller only needs the vertical max of the feature volume
    (BEVMapper with ``pooling='max'`` and ``materialize_volume=False``); when the fused kernel
    applies, ``pred['feature_plane']`` is returned and ``feature_volume.features`` is None."""
    cfg = self.config
    ctx = ctx or base.ForwardContext()
    f_image_pyr = data.get('image_feature_pyr')
    if f_image_pyr is None:
      images = data['images'].to(torch.float32)
      B, V = images.shape[:2]
      # nn.vmap over scenes with shared params == one batch of B*V images
      # (GroupNorm statistics are per image).
      pyr = self.image_encoder(
          params['image_encoder'], images.reshape(B * V, *images.shape[2:]),
          train, ctx=ctx,
      )
      f_image_pyr = types.FeatureImagePyramid(
          features=[f.reshape(B, V, *f.shape[1:]) for f in pyr.features],
          strides=pyr.strides,
      )
    f_images = f_image_pyr.features[-1]
    B, V = f_images.shape[:2]
    feature_stride = f_image_pyr.strides[-1]
    inv_stride = tuple(float(v) for v in (1.0 / feature_stride[::-1]))
    scale = base.device_const(('camera_scale', inv_stride), f_images.device,
                              lambda: torch.tensor(inv_stride, dtype=torch.float32))
    cameras = data['camera'].scale(scale)
    scene_t_view = data['T_view2scene']
    pred = {'image_feature_pyramid': f_image_pyr}

    if self.weighted:
      f_images = self._project(params['proj_mlp'], f_images, train)
      pred['scores_images'] = f_images[..., -cfg.num_scale_bins:]
    else:
      f_images = f_images.contiguous()

    xyz = data['xyz_query']
    xyz_flat = xyz.reshape(len(xyz), -1, 3).contiguous()
    k_vs = cfg.top_k_view_selection
    K = k_vs if (k_vs and V > k_vs) else 0
    kw = dict(K=K, fisheye=cameras.is_fisheye, feature_dim=cfg.feature_dim,
              num_bins=cfg.num_scale_bins, depth_min_max=cfg.depth_min_max,
              max_view_distance=cfg.get('max_view_distance'))
    fused = split = classed = records = False
    if base.needs_grad(f_images):
      lift = ag.lift_pool
      if xyz.dim() == 5:                     # (the traversal hint of the inference branch below)
        kw.update(grid_yz=tuple(xyz.shape[2:4]))
      # the masked fusion MLP reads the rows of valid voxels only (row list): the ~40 % of a map's voxels that
      # no view sees need no 1 KB row of zeros (with an input activation its VJP gates on every row: then kept)
      if self.depth_mlp is None and self.fusion_mlp.reads_listed_rows_only(params['fusion_mlp'], xyz_flat.shape[0] * xyz_flat.shape[1]):
        kw.update(valid_rows_only=True)
      if not self.default_fusion:
        kw.update(weighted=self.weighted, use_variance=bool(cfg.fusion_use_variance),
                  add_minmax=bool(cfg.fusion_add_minmax))
    else:
      lift = ops.lift_pool
      if xyz.dim() == 5:                     # [B, X, Y, Z, 3]: a voxel grid (traversal hint)
        kw.update(grid_yz=tuple(xyz.shape[2:4]))
      fused = pool_max and self.default_fusion and self._fused_pool_ok(params, f_images)
      # (the pre-split rows exist in the batched kernel only, which addresses its taps by 32-bit
      #  byte offsets: snap_lift_pool_f32 rejects out_split for f_images >= 4 GB)
      split = (fused and ops.POOLED_SPLIT and not cfg.fusion.apply_input_activation
               and cfg.feature_dim % 8 == 0 and (K or V) <= 4 and f_images.numel() * 4 < 2 ** 32)
      if fused:                              # the fused kernel reads the rows of valid voxels only,
        kw.update(valid_rows_only=True, out_split=split)   # pre-split: its A operand goes by LDS-DMA
        # rows classed by their number of observations: one observation (most voxels of a map,
        # every voxel of the query) = zero variance, neither written nor read nor multiplied
        classed = split and ops.CLASS_ROWS and cfg.feature_dim % 16 == 0
        kw.update(class_rows=classed)
        # ... and those rows are never written: a tap record instead, blended inside the MLP kernel
        records = classed and ops.LIFT_IN_CONSUMER
        kw.update(tap_records=records)
      if not self.default_fusion:
        kw.update(weighted=self.weighted, use_variance=bool(cfg.fusion_use_variance),
                  add_minmax=bool(cfg.fusion_add_minmax))
    cam_p = rt_p = None
    if self.depth_mlp is not None:
      pooled, valid = self._lift_with_depth_mlp(params, f_images, cameras, scene_t_view, xyz_flat, K, train)
    else:
      # (packed once: the lazy volume closure below reuses them)
      cam_p, rt_p = cameras.packed().to(torch.float32), scene_t_view.packed().to(torch.float32)
      pooled, valid, *classes = lift(f_images, cam_p, rt_p, xyz_flat, **kw)
    grid_shape = (-1, *xyz.shape[-4:-1])
    if fused:
      p = params['fusion_mlp']
      nvar = cfg.feature_dim // 16                      # (the variance slabs follow the mean's)
      plane, pvalid = ops.mlp2_pool_max(
          pooled.reshape(-1, pooled.shape[-1]), (classes[0] if classed else valid).reshape(-1),
          p['Dense_0']['kernel'], p['Dense_0']['bias'], p['Dense_1']['kernel'], p['Dense_1']['bias'],
          cin=self.fusion_mlp.in_dim, Z=grid_shape[-1],
          relu_in=bool(cfg.fusion.apply_input_activation), x_split=split,
          zero_slabs=(nvar, nvar) if classed else None,
          gather=(f_images, classes[1]) if records else None)
      # the reference's pytree entry (streetview_encoder.py:282-286), produced on first access
      # by the unfused chain on the same inputs: lift -> fusion MLP -> mask
      kw_plain = {k: v for k, v in kw.items()
                  if k not in ('valid_rows_only', 'out_split', 'class_rows', 'tap_records')}
      engine = ops.precision()                 # (the engine of THIS apply, whenever the access comes)

      def volume(f_images=f_images, xyz_flat=xyz_flat, p=p):
        with ops.engine_scope(engine):
          pooled_d, valid_d = ops.lift_pool(f_images, cam_p, rt_p, xyz_flat, **kw_plain)[:2]
          f = self.fusion_mlp(p, pooled_d, False, row_mask=valid_d)
        return f.reshape(*grid_shape, f.shape[-1])

      pred['feature_volume'] = types.LazyFeatureVolume(volume, valid=valid.reshape(grid_shape))
      pred['feature_plane'] = types.FeaturePlane(
          features=plane.reshape(*grid_shape[:-1], plane.shape[-1]),
          valid=pvalid.reshape(grid_shape[:-1]))
      return pred
    f_grid = self.fusion_mlp(params['fusion_mlp'], pooled, train, row_mask=valid)
    f_grid = f_grid.reshape(*grid_shape, f_grid.shape[-1])
    valid = valid.reshape(grid_shape)
    pred['feature_volume'] = types.FeatureVolume(features=f_grid, valid=valid)
    return pred

  def _lift_with_depth_mlp(self, params, f_images, cameras, scene_t_view, xyz_flat, K, train):
    """streetview_encoder.py:263-267: the observations leave the lift un-pooled
    (ops.lift_observations), a per-observation MLP on [features, log10 depth, ray] is added to
    them (conv engine, residual epilogue), a second pass pools them (ops.lift_pool_observations)."""
    cfg = self.config
    p = params['depth_mlp']
    n = len(cfg.depth_mlp.layers)
    cam, Rt = cameras.packed().to(torch.float32), scene_t_view.packed().to(torch.float32)
    common = dict(K=K, fisheye=cameras.is_fisheye, feature_dim=cfg.feature_dim,
                  max_view_distance=cfg.get('max_view_distance'))
    if base.needs_grad(f_images, *(p[f'Dense_{i}'][k] for i in range(n) for k in ('kernel', 'bias'))):
      # training: the same three steps as autograd nodes (the lift passes' VJPs are the
      # deterministic records / sort / gather kernels; the MLP is the conv engine's)
      obs, feat, _ = ag.lift_observations(f_images, cam, Rt, xyz_flat, **common)
      h = obs
      for i in range(n):
        d = p[f'Dense_{i}']
        pro = ops.PRO_RELU if (i == 0 and cfg.depth_mlp.apply_input_activation) else ops.PRO_NONE
        last = i + 1 == n
        h = ag.dense(h, d['kernel'], d['bias'], cin=d['kernel'].shape[0], prologue=pro,
                     relu=not last, residual=feat if last else None)
      return ag.lift_pool_observations(
          h, tuple(f_images.shape), cam, Rt, xyz_flat, use_variance=bool(cfg.fusion_use_variance),
          add_minmax=bool(cfg.fusion_add_minmax), **common)
    obs, feat, _ = ops.lift_observations(f_images, cam, Rt, xyz_flat, **common)
    h = obs
    for i in range(n):
      d = p[f'Dense_{i}']
      pro = ops.PRO_RELU if (i == 0 and cfg.depth_mlp.apply_input_activation) else ops.PRO_NONE
      last = i + 1 == n
      h = ops.dense(h, d['kernel'], d['bias'], cin=d['kernel'].shape[0], prologue=pro,
                    relu=not last, residual=feat if last else None)
    return ops.lift_pool_observations(
        h, tuple(f_images.shape), cam, Rt, xyz_flat, use_variance=bool(cfg.fusion_use_variance),
        add_minmax=bool(cfg.fusion_add_minmax), **common)

  default_config = staticmethod(default_configs.streetview_encoder)
