"""Default configuration dictionaries of the hot path.

Same factory names, keys and default values as ``snap/configs/defaults.py``
(:122-270, :343-361) so that configs written for the reference are accepted
unchanged.  Built on :mod:`snap_amd.utils.config_dict` because ml_collections is
not installable offline; real ``ml_collections.ConfigDict`` objects work too.
"""
import enum
from typing import Any, Iterable

from snap_amd.utils import config_dict
from snap_amd.utils.config_dict import ConfigDict


class MapModalities(str, enum.Enum):
  STREETVIEW = 'streetview'
  AERIAL = 'aerial'
  SEMANTIC = 'semantic'


def parse_argument_string(args_str: None | str) -> dict[str, Any]:
  """'k=v,k=v' -> dict with defaults (defaults.py:51-59)."""
  defaults = {'image_encoder': 'R50', 'modalities': 'streetview+aerial'}
  given = {}
  for item in (args_str or '').split(','):
    if item:
      key, value = item.split('=')
      given[key] = value
  unknown = set(given) - set(defaults)
  if unknown:
    raise ValueError(f'Unknown args: {unknown}')
  return {**defaults, **given}


def mlp() -> ConfigDict:
  return ConfigDict(
      activation='relu', layers=config_dict.placeholder(tuple),
      apply_input_activation=False,
  ).lock()


_RESNET_VARIANTS = {
    'R50': {},
    'R101': dict(depth=101, limit_num_blocks=4, checkpoint_blocks=True,
                 checkpoint_units=True),
    'R152x2': dict(width=2, depth=152, limit_num_blocks=3,
                   checkpoint_blocks=True, checkpoint_units=True),
}


def resnet(name: str = 'R50') -> ConfigDict:
  if name not in _RESNET_VARIANTS:
    raise ValueError(f'Unknown ResNet name: {name}')
  cfg = ConfigDict(
      width=1, depth=50, limit_num_blocks=4, skip_root_block=False,
      checkpoint_blocks=False, checkpoint_units=False,
      pretrained_path='path_to/checkpoint.npz',
  ).lock()
  cfg.update(_RESNET_VARIANTS[name])
  return cfg


def vit(name: str = 'B/16') -> ConfigDict:
  """ViT encoder (``encoder_name='vit'``): not in the reference (image_encoder.py:103);
  BASELINE.json configs[4].  Sizes of the published variants."""
  variants = {
      'Ti/16': dict(hidden_size=192, num_layers=12, num_heads=3, mlp_dim=768),
      'S/16': dict(hidden_size=384, num_layers=12, num_heads=6, mlp_dim=1536),
      'B/16': dict(hidden_size=768, num_layers=12, num_heads=12, mlp_dim=3072),
  }
  if name not in variants:
    raise ValueError(f'Unknown ViT name: {name}')
  return ConfigDict(
      patch_size=16, posemb_grid=(32, 32), matmul_precision='bf16', **variants[name],
  ).lock()


def image_encoder(encoder_name: str = 'resnet') -> ConfigDict:
  if encoder_name == 'vit':
    return ConfigDict(
        encoder_name='vit', encoder=vit(), output_dim=128,
        num_pyr_levels=config_dict.placeholder(int),
    ).lock()
  return ConfigDict(
      encoder_name='resnet', encoder=resnet(), output_dim=128,
      num_pyr_levels=config_dict.placeholder(int),
  ).lock()


def aerial_encoder() -> ConfigDict:
  cfg = image_encoder()
  cfg.encoder.skip_root_block = True
  return cfg


def semantic_raster_encoder() -> ConfigDict:
  """defaults.py:191-198."""
  encoder = image_encoder()
  encoder.encoder.skip_root_block = True
  encoder.encoder.depth = 26
  encoder.encoder.width = 2
  encoder.encoder.pretrained_path = None
  encoder.encoder.limit_num_blocks = 4
  return ConfigDict(encoder=encoder, embedding_dim=8).lock()


def streetview_encoder() -> ConfigDict:
  dim = 128
  fusion = mlp()
  fusion.layers = (2 * dim, dim)
  proj = mlp()
  proj.apply_input_activation = True
  return ConfigDict(
      image_encoder=image_encoder(), feature_dim=dim, fusion=fusion,
      proj_mlp=proj, depth_mlp=config_dict.placeholder(ConfigDict),
      do_weighted_fusion=True, num_scale_bins=32, top_k_view_selection=4,
      depth_min_max=(1.0, 32.0), fusion_add_minmax=False,
      fusion_use_variance=True, max_view_distance=config_dict.placeholder(float),
      pretrained_path=config_dict.placeholder(str),
  ).lock()


def vertical_pooling() -> ConfigDict:
  dim = 128
  fusion = mlp()
  fusion.layers = (2 * dim, dim)
  return ConfigDict(pooling='max', mlp=fusion).lock()


def bev_mapper(
    modalities: Iterable[str] = (MapModalities.STREETVIEW, MapModalities.AERIAL)
) -> ConfigDict:
  cfg = ConfigDict(
      streetview_encoder=config_dict.placeholder(ConfigDict),
      scene_z_offset=4.0, scene_z_offset_range=(-2, 2), scene_z_height=12.0,
      pooling=vertical_pooling(),
      aerial_encoder=config_dict.placeholder(ConfigDict),
      semantic_encoder=config_dict.placeholder(ConfigDict),
      modality_fusion=vertical_pooling(),
      bev_net=config_dict.placeholder(ConfigDict),
      matching_dim=32, normalize_matching_features=True, add_confidence=False,
      apply_modality_dropout=True, pretrained_path=config_dict.placeholder(str),
      materialize_volume=True,      # (not a reference key) False: only the pooled plane is needed
  )
  for m in modalities:
    m = MapModalities(m)
    if m == MapModalities.STREETVIEW:
      cfg.streetview_encoder = streetview_encoder()
    elif m == MapModalities.AERIAL:
      cfg.aerial_encoder = aerial_encoder()
    else:
      cfg.semantic_encoder = semantic_raster_encoder()
  return cfg.lock()


def semantic_net() -> ConfigDict:
  """defaults.py:286-342."""
  return ConfigDict(
      bev_mapper=bev_mapper(), decoder_type='mlp', decoder_dim=128, mlp_num_layers=2,
      resnet_num_units=8, apply_random_flip=False,
      area_classes=('crosswalk', 'sidewalk', 'road', 'terrain', 'building'),
      area_frequencies=(
          ('crosswalk', 0.036434), ('sidewalk', 0.226553), ('road', 0.446990),
          ('terrain', 0.085374), ('building', 0.204649),
      ),
      object_classes_exclusive=('fence', 'pole', 'tree'),
      object_classes_independent=('traffic_sign', 'traffic_light', 'street_light'),
      object_frequencies=(
          ('fence', 0.006257), ('pole', 0.001172), ('tree', 0.001924),
          ('traffic_sign', 0.000960), ('traffic_light', 0.000559),
          ('street_light', 0.000738), ('void', 0.988391),
      ),
  ).lock()


def bev_localizer() -> ConfigDict:
  return ConfigDict(
      bev_mapper=bev_mapper(),
      bev_mapper_query=config_dict.placeholder(ConfigDict),
      add_confidence_query=False, add_confidence_map=False,
      mask_score_out_of_bounds=False, clip_negative_scores=True,
      add_temperature=True, init_temperature=2.0,
      num_pose_samples=config_dict.placeholder(int),
      num_pose_sampling_retries=1, query_frustum_depth=16.0,
      filter_points_in_fov=False,
      threshold_remove_accurate_poses=config_dict.placeholder(tuple),
      do_grid_refinement=False,
  ).lock()
