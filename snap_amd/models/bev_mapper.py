"""BEV mapper: StreetView (+ aerial) -> fused 2-D neural map + matching features.

Mirrors ``snap/models/bev_mapper.py:40-296`` (``VerticalPooling``, ``BEVMapper``).
"""
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import image_encoder
from snap_amd.models import streetview_encoder
from snap_amd.models import types


class VerticalPooling(base.Module):
  """Masked reduce over the vertical (or modality) axis (bev_mapper.py:40-88)."""

  def __init__(self, config, dtype=torch.float32):
    if config.pooling not in ('max', 'sum', 'mean'):
      # 'weighted' / 'softmax' / 'mlp' are non-default variants.
      raise NotImplementedError(config.pooling)
    self.config = config

  def init_params(self, gen, device):
    return {}

  def __call__(self, params, feature_volume):
    vp = ag.vertical_pool if base.needs_grad(feature_volume.features) else ops.vertical_pool
    plane, valid = vp(
        feature_volume.features.contiguous(), feature_volume.valid.contiguous(),
        self.config.pooling,
    )
    return {'plane': types.FeaturePlane(features=plane, valid=valid)}


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
      self.vertical_pooling = VerticalPooling(config.pooling, dtype)
      feature_dimensions.append(config.streetview_encoder.feature_dim)
    if config.aerial_encoder is not None:
      self.aerial_encoder = image_encoder.ImageEncoder(config.aerial_encoder, dtype)
      feature_dimensions.append(config.aerial_encoder.output_dim)
    if config.semantic_encoder is not None:
      raise NotImplementedError('semantic modality is out of scope (SURVEY 2.1 #20)')
    if not feature_dimensions:
      raise ValueError('Need to create at least one input encoder.')
    elif len(feature_dimensions) > 1:
      if not all(d == feature_dimensions[0] for d in feature_dimensions):
        raise ValueError(f'Encoder have different output dimensions: {feature_dimensions}')
      self.modality_fusion = VerticalPooling(config.modality_fusion, dtype)
    self.feature_dim = feature_dimensions[0]
    if config.bev_net is not None:
      raise NotImplementedError('BEV network not yet implemented')
    if config.add_confidence:
      raise NotImplementedError('add_confidence (non-default confidence head)')

  def init_params(self, gen, device):
    params = {}
    if self.streetview_encoder is not None:
      params['streetview_encoder'] = self.streetview_encoder.init_params(gen, device)
      params['vertical_pooling'] = {}
    if self.aerial_encoder is not None:
      params['aerial_encoder'] = self.aerial_encoder.init_params(gen, device)
    if self.modality_fusion is not None:
      params['modality_fusion'] = {}
    if self.config.matching_dim is not None:
      dm = self.config.matching_dim
      # variance_scaling(1/sqrt(dm), 'fan_in', 'truncated_normal') (bev_mapper.py:145-153)
      std = ((1.0 / dm**0.5) / self.feature_dim) ** 0.5
      params['matching_proj'] = {
          'kernel': base.truncated_normal(gen, (self.feature_dim, dm), std, device),
          'bias': torch.zeros(dm, device=device),
      }
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
        u = torch.rand(z_offset.shape, generator=gen).to(z_offset)
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
    xyz = torch.empty(B, X, Y, nz, 3, dtype=t.dtype, device=t.device)
    xyz[..., :2] = xy[:, :, :, None, :]
    xyz[..., 2] = z[:, None, None, :]
    return xyz

  def encode_streetview(self, params, data, train, is_query, ctx=None, rng=None):
    if 'xyz_query' not in data:
      data['xyz_query'] = self.build_xyz_query(data, train, is_query, rng)
    pred = self.streetview_encoder(params['streetview_encoder'], data, train=train, ctx=ctx)
    pred['vertical_pooling'] = self.vertical_pooling({}, pred['feature_volume'])
    pred['feature_plane'] = pred['vertical_pooling'].pop('plane')
    return pred

  def encode_aerial(self, params, aerial_rgb, train=False, ctx=None):
    pyramid = self.aerial_encoder(params['aerial_encoder'], aerial_rgb, train=train, ctx=ctx)
    features = pyramid.features[-1].contiguous()
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
      pred['aerial'] = self.encode_aerial(params, data['rasters']['rgb'], train=train, ctx=ctx)
      feature_planes.append(pred['aerial']['feature_plane'])
    if not feature_planes:
      raise ValueError('No map encoder given.')

    # fuse_neural_maps (bev_mapper.py:225-252; `train` is not forwarded there, so
    # modality dropout never fires) + matching head (:284-291), one kernel.
    has_match = cfg.matching_dim is not None
    mp = params.get('matching_proj') if has_match else None
    pooling = (
        self.modality_fusion.config.pooling if self.modality_fusion is not None else 'max'
    )
    single = len(feature_planes) == 1
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
    return pred

  default_config = staticmethod(default_configs.bev_mapper)
