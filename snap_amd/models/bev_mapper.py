"""BEV mapper: StreetView (+ aerial) -> fused 2-D neural map + matching features.

Mirrors ``snap/models/bev_mapper.py:40-296`` (``VerticalPooling``, ``BEVMapper``).
"""
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import image_encoder
from snap_amd.models import layers
from snap_amd.models import semantic_raster_encoder
from snap_amd.models import streetview_encoder
from snap_amd.models import types


class VerticalPooling(base.Module):
  """Masked reduce over the vertical (or modality) axis (bev_mapper.py:40-88).

  'max' / 'sum' / 'mean': one masked-reduce kernel.  'softmax' / 'weighted': a Dense(1)
  confidence head + masked softmax over the levels + weighted sum, ONE kernel that reads the
  volume once (bev_conf_pool.hip).  'mlp': masked volume flattened over (Z, D) -> MLP on the
  conv engine; needs ``num_levels`` and ``feature_dim`` (Flax infers them at init).
  """

  def __init__(self, config, dtype=torch.float32, num_levels=None, feature_dim=None):
    if config.pooling not in ('max', 'sum', 'mean', 'softmax', 'weighted', 'mlp'):
      raise NotImplementedError(config.pooling)
    self.config = config
    self.feature_dim = feature_dim
    self.fusion_mlp = None
    if config.pooling == 'mlp':
      if num_levels is None or feature_dim is None:
        raise ValueError("pooling='mlp' needs num_levels and feature_dim")
      self.fusion_mlp = layers.MLP(config.mlp, in_dim=num_levels * feature_dim)
    elif config.pooling in ('softmax', 'weighted') and feature_dim is None:
      raise ValueError(f"pooling='{config.pooling}' needs feature_dim")

  def init_params(self, gen, device):
    if self.config.pooling in ('softmax', 'weighted'):
      d = self.feature_dim
      return {'confidence_head': {
          'kernel': base.lecun_normal(gen, (d, 1), d, device),
          'bias': torch.zeros(1, device=device),
      }}
    if self.config.pooling == 'mlp':
      return {'fusion_mlp': self.fusion_mlp.init_params(gen, device)}
    return {}

  def __call__(self, params, feature_volume):
    pooling = self.config.pooling
    feats = feature_volume.features.contiguous()
    valid = feature_volume.valid.contiguous()
    if pooling in ('softmax', 'weighted'):
      head = params['confidence_head']
      fn = (ag.vertical_pool_conf if base.needs_grad(feats, head['kernel'], head['bias'])
            else ops.vertical_pool_conf)
      w = head['kernel'] if fn is ag.vertical_pool_conf else head['kernel'].reshape(-1).contiguous()
      plane, pvalid, scores, weights = fn(feats, valid, w, head['bias'], pooling == 'weighted')
      return {'scores': scores, 'weights': weights,
              'plane': types.FeaturePlane(features=plane, valid=pvalid)}
    if pooling == 'mlp':
      Z, D = feats.shape[-2:]
      lead = feats.shape[:-2]
      masked = feats * valid[..., None].to(feats.dtype)                  # where(valid, f, 0)
      flat = masked.reshape(*lead, Z * D)
      pvalid = valid.any(-1)
      out = self.fusion_mlp(params['fusion_mlp'], flat, row_mask=pvalid.contiguous())
      return {'plane': types.FeaturePlane(features=out, valid=pvalid)}
    vp = ag.vertical_pool if base.needs_grad(feats) else ops.vertical_pool
    plane, pvalid = vp(feats, valid, pooling)
    return {'plane': types.FeaturePlane(features=plane, valid=pvalid)}


def _median_lower_upper_mean(x):
  """``jnp.median`` over the last axis (mean of the two middle values if even)."""
  s, _ = torch.sort(x, dim=-1)
  n = x.shape[-1]
  return 0.5 * (s[..., (n - 1) // 2] + s[..., n // 2])


class BEVMapper(base.Module):
  """Encode a set of images (and an aerial tile) into a 2-D feature plane."""

  def __init__(self, config, grid, semantic_map_classes=None, dtype=torch.float32):
    if config.pretrained_path is not None:
      raise NotImplementedError('pretrained_path (broken in the reference, bev_mapper.py:101)')
    self.config = config
    self.grid = grid
    self.semantic_map_classes = semantic_map_classes
    feature_dimensions = []
    self.streetview_encoder = self.aerial_encoder = self.semantic_encoder = None
    self.modality_fusion = None
    if config.streetview_encoder is not None:
      self.streetview_encoder = streetview_encoder.StreetViewEncoder(
          config.streetview_encoder, dtype
      )
      cell = grid.cell_size if hasattr(grid, 'cell_size') else None
      num_levels = None if cell is None else len(torch.arange(0, config.scene_z_height, cell))
      self.vertical_pooling = VerticalPooling(
          config.pooling, dtype, num_levels=num_levels,
          feature_dim=config.streetview_encoder.feature_dim,
      )
      feature_dimensions.append(config.streetview_encoder.feature_dim)
    if config.aerial_encoder is not None:
      self.aerial_encoder = image_encoder.ImageEncoder(config.aerial_encoder, dtype)
      feature_dimensions.append(config.aerial_encoder.output_dim)
    if config.semantic_encoder is not None:
      self.semantic_encoder = semantic_raster_encoder.SemanticRasterEncoder(
          config.semantic_encoder, semantic_map_classes, dtype)
      feature_dimensions.append(config.semantic_encoder.encoder.output_dim)
    if not feature_dimensions:
      raise ValueError('Need to create at least one input encoder.')
    elif len(feature_dimensions) > 1:
      if not all(d == feature_dimensions[0] for d in feature_dimensions):
        raise ValueError(f'Encoder have different output dimensions: {feature_dimensions}')
      # max / sum / mean are fused with the matching head in one kernel (plane_fuse_match); the
      # learned modes ('softmax' / 'weighted' / 'mlp', bev_mapper.py:63-78 reused at :250-252)
      # pool the stacked planes with the same kernels as the vertical pooling
      self.modality_fusion = VerticalPooling(
          config.modality_fusion, dtype, num_levels=len(feature_dimensions),
          feature_dim=feature_dimensions[0])
    self.feature_dim = feature_dimensions[0]
    if config.bev_net is not None:
      raise NotImplementedError('BEV network not yet implemented')

  def init_params(self, gen, device):
    params = {}
    if self.streetview_encoder is not None:
      params['streetview_encoder'] = self.streetview_encoder.init_params(gen, device)
      params['vertical_pooling'] = self.vertical_pooling.init_params(gen, device)
    if self.aerial_encoder is not None:
      params['aerial_encoder'] = self.aerial_encoder.init_params(gen, device)
    if self.semantic_encoder is not None:
      params['semantic_encoder'] = self.semantic_encoder.init_params(gen, device)
    if self.modality_fusion is not None:
      params['modality_fusion'] = self.modality_fusion.init_params(gen, device)
    if self.config.matching_dim is not None:
      dm = self.config.matching_dim
      # variance_scaling(1/sqrt(dm), 'fan_in', 'truncated_normal') (bev_mapper.py:145-153)
      std = ((1.0 / dm**0.5) / self.feature_dim) ** 0.5
      params['matching_proj'] = {
          'kernel': base.truncated_normal(gen, (self.feature_dim, dm), std, device),
          'bias': torch.zeros(dm, device=device),
      }
    if self.config.add_confidence:
      # nn.Sequential([nn.Dense(1)]) (bev_mapper.py:154-157): Flax names the layer 'layers_0'
      params['confidence_head'] = {'layers_0': {
          'kernel': base.lecun_normal(gen, (self.feature_dim, 1), self.feature_dim, device),
          'bias': torch.zeros(1, device=device),
      }}
    return params

  def build_xyz_query(self, data, train, is_query, rng=None):
    """bev_mapper.py:162-196: voxel-centre query points [B, X, Y, Z, 3]."""
    cfg = self.config
    scene_t_view = data['T_view2scene']
    t = scene_t_view.t
    xy = data.get('xy_bev')
    if xy is None:
      xy = self.grid.index_to_xyz(self.grid.grid_index(device=t.device).to(t.dtype))
    if xy.dim() != 4:
      xy = xy[None].expand(len(t), *xy.shape)
    z_offset = data.get('z_offset')
    if z_offset is None:
      camera_heights = _median_lower_upper_mean(t[..., -1])
      z_offset = camera_heights - cfg.get('scene_z_offset', 4.0)
      if train and is_query and cfg.get('scene_z_offset_range') is not None:
        z_min, z_max = cfg.get('scene_z_offset_range')
        gen = torch.Generator(device='cpu')
        gen.manual_seed(0 if rng is None else int(rng) + 7919)
        u = torch.rand(z_offset.shape, generator=gen)
        if z_offset.is_cuda and z_offset.dtype == torch.float32:
          # (through the pinned staging ring: a pageable-source copy stalls the host until the stream has drained --
          #  2.3 ms in the middle of a C3 training step's forward, and an empty launch queue behind it)
          u = ops.upload_table(u.numpy(), z_offset.device).view(torch.float32).reshape(z_offset.shape)
        else:
          u = u.to(z_offset)
        z_offset = z_offset + (z_min + (z_max - z_min) * u)
    scene_z_height = cfg.get('scene_z_height', 12.0)
    cell = self.grid.cell_size
    nz = len(torch.arange(0, scene_z_height, cell))
    z = (
        torch.arange(nz, device=t.device, dtype=t.dtype)[None] * cell
        + z_offset[:, None]
        + cell / 2
    )
    B, X, Y = xy.shape[:3]
    if t.is_cuda and t.dtype == torch.float32 and ops.NATIVE_GLUE and not base.needs_grad(z, xy):
      # (xy: an expanded view when the grid is shared by the scenes -- hand over its base)
      xy_c = xy[0].contiguous() if xy.stride(0) == 0 else xy.contiguous()
      return ops.voxel_points(xy_c, z.contiguous())                              # one pass
    xyz = torch.empty(B, X, Y, nz, 3, dtype=t.dtype, device=t.device)
    xyz[..., :2] = xy[:, :, :, None, :]
    xyz[..., 2] = z[:, None, None, :]
    return xyz

  def encode_streetview(self, params, data, train, is_query, ctx=None, rng=None):
    if 'xyz_query' not in data:
      data['xyz_query'] = self.build_xyz_query(data, train, is_query, rng)
    # materialize_volume=False (an option of this implementation; under jit the reference's
    # unused feature volume is dead code too): with max pooling the fusion MLP and the pooling
    # run as one kernel and only the plane is written
    pool_max = (not self.config.get('materialize_volume', True)
                and self.vertical_pooling.config.pooling == 'max')
    pred = self.streetview_encoder(params['streetview_encoder'], data, train=train, ctx=ctx,
                                   pool_max=pool_max)
    if 'feature_plane' in pred:
      pred['vertical_pooling'] = {}
      return pred
    pred['vertical_pooling'] = self.vertical_pooling(
        params.get('vertical_pooling', {}), pred['feature_volume'])
    pred['feature_plane'] = pred['vertical_pooling'].pop('plane')
    return pred

  def start_aerial(self, params, data, train=False, ctx=None):
    """Inference: launch the aerial encoder on a second HIP stream NOW, so that it runs next to
    the StreetView chain it does not depend on (``__call__`` joins it before the fusion).  The
    encoder's deep stages are launches of a few dozen workgroups; on their own stream they fill
    compute units the StreetView kernels leave idle instead of serialising behind them."""
    if self.aerial_encoder is None or 'rasters' not in data or train or not ops.OVERLAP_AERIAL:
      return
    rgb = data['rasters']['rgb']
    if not rgb.is_cuda:
      return

    def leaves(t):
      if isinstance(t, dict):
        for v in t.values():
          yield from leaves(v)
      else:
        yield t
    if base.needs_grad(rgb, *leaves(params['aerial_encoder'])):
      return
    # the side stream of the input's OWN device; every tensor the encoder reads (image, parameters)
    # was produced on the main stream before this point, so one wait orders them -- no tensor is
    # freed while the side stream still reads it because the caller keeps `data` / `params` alive
    # until the join
    main, side = torch.cuda.current_stream(rgb.device), ops.side_stream(rgb.device)
    side.wait_stream(main)                      # the inputs were produced on the main stream
    with torch.cuda.stream(side):
      pred = self.encode_aerial(params, rgb, train=train, ctx=ctx)
      done = side.record_event()
    data['_aerial_async'] = (pred, done)

  def encode_aerial(self, params, aerial_rgb, train=False, ctx=None):
    pyramid = self.aerial_encoder(params['aerial_encoder'], aerial_rgb, train=train, ctx=ctx)
    features = pyramid.features[-1].contiguous()
    plane = types.FeaturePlane(
        features=features,
        valid=torch.ones(features.shape[:-1], dtype=torch.bool, device=features.device),
    )
    return {'feature_plane': plane}

  def encode_semantics(self, params, semantic_raster, train=False, ctx=None):
    """bev_mapper.py:214-223."""
    pyramid = self.semantic_encoder(params['semantic_encoder'], semantic_raster, train=train, ctx=ctx)
    features = pyramid.features[-1].contiguous()      # highest-resolution level
    plane = types.FeaturePlane(
        features=features,
        valid=torch.ones(features.shape[:-1], dtype=torch.bool, device=features.device),
    )
    return {'feature_plane': plane}

  def __call__(self, params, data, train=False, debug=False, is_query=False,
               ctx=None, rng=None):
    cfg = self.config
    ctx = ctx or base.ForwardContext()
    pred = {}
    feature_planes = []
    if self.streetview_encoder is not None:
      pred['streetview'] = self.encode_streetview(
          params, data, train=train, is_query=is_query, ctx=ctx, rng=rng
      )
      feature_planes.append(pred['streetview']['feature_plane'])
    if self.aerial_encoder is not None and 'rasters' in data:
      pending = data.pop('_aerial_async', None)
      if pending is not None:                   # started by start_aerial on the side stream
        pred['aerial'], done = pending
        main = torch.cuda.current_stream()
        main.wait_event(done)
        plane = pred['aerial']['feature_plane']
        plane.features.record_stream(main)
        plane.valid.record_stream(main)
      else:
        pred['aerial'] = self.encode_aerial(params, data['rasters']['rgb'], train=train, ctx=ctx)
      feature_planes.append(pred['aerial']['feature_plane'])
    if self.semantic_encoder is not None and 'rasters' in data:
      # (there are no semantic rasters for query images, bev_mapper.py:273-278)
      pred['semantic'] = self.encode_semantics(params, data['rasters']['semantics'], train=train, ctx=ctx)
      feature_planes.append(pred['semantic']['feature_plane'])
    if not feature_planes:
      raise ValueError('No map encoder given.')

    # fuse_neural_maps (bev_mapper.py:225-252; `train` is not forwarded there, so
    # modality dropout never fires) + matching head (:284-291), one kernel.
    has_match = cfg.matching_dim is not None
    mp = params.get('matching_proj') if has_match else None
    pooling = (
        self.modality_fusion.config.pooling if self.modality_fusion is not None else 'max'
    )
    if len(feature_planes) > 1 and pooling not in ('max', 'sum', 'mean'):
      # learned modality fusion: VerticalPooling over the stacked planes (bev_mapper.py:246-252)
      stacked = types.FeatureVolume(
          features=torch.stack([p.features for p in feature_planes], dim=-2).contiguous(),
          valid=torch.stack([p.valid for p in feature_planes], dim=-1).contiguous())
      feature_planes = [self.modality_fusion(params['modality_fusion'], stacked)['plane']]
      pooling = 'max'                 # (one plane left: the fused kernel only adds the matching head)
    single = len(feature_planes) == 1
    if single:
      pooling = 'max'                   # nothing to fuse (fuse_neural_maps returns the plane, :230-231)
    feats = [p.features for p in feature_planes]
    pfm = ops.plane_fuse_match
    if base.needs_grad(*feats, *( [mp['kernel'], mp['bias']] if has_match else [])):
      pfm = ag.plane_fuse_match
    fused, fvalid, matching = pfm(
        feats,
        [p.valid for p in feature_planes],
        pooling,
        mp['kernel'] if has_match else None,
        mp['bias'] if has_match else None,
        normalize=bool(cfg.normalize_matching_features),
        want_fused=not single,
    )
    if single:
      plane = feature_planes[0]
    else:
      plane = types.FeaturePlane(features=fused, valid=fvalid)
    pred['bev_features'] = plane
    if has_match:
      pred['bev_matching'] = types.FeaturePlane(features=matching, valid=plane.valid)
    if cfg.add_confidence:
      # bev_mapper.py:292-295: where(valid, log_sigmoid(Dense(1)(features)), 0)
      head = params['confidence_head']['layers_0']
      if base.needs_grad(plane.features, head['kernel'], head['bias']):
        pred['bev_confidence'] = ag.confidence_head(plane.features, plane.valid.contiguous(),
                                                    head['kernel'], head['bias'])
      else:
        pred['bev_confidence'] = ops.confidence_head(
            plane.features.contiguous(), plane.valid.contiguous(),
            head['kernel'].reshape(-1).contiguous(), head['bias'])       # (bias: a device scalar)
    return pred

  default_config = staticmethod(default_configs.bev_mapper)
