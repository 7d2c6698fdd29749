"""Encode a 2-D semantic map, given as multi-channel boolean rasters, into a neural map
(``snap/models/semantic_raster_encoder.py:27-84``)."""
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import image_encoder

# snap/data/types.py:35-42
SURFEL_ROAD_CLASSES = ('crosswalk', 'sidewalk', 'pavedroad', 'stopline', 'line', 'otherlanemarking')


class SemanticRasterEncoder(base.Module):
  """Rasters [B, H, W, N] bool -> FeatureImagePyramid.

  Surfel-road classes are mutually exclusive: one multi-class label (argmax) and one embedding;
  every other class is an independent binary label with its own embedding slot (:33-46, :63-79).
  The lookups are one kernel (``csrc/semantic.hip``); the encoder is the shared ImageEncoder.
  """

  def __init__(self, config, raster_classes, dtype=torch.float32):
    self.config = config
    self.raster_classes = tuple(raster_classes)
    self.indices_surfel_road = [i for i, c in enumerate(self.raster_classes) if c in SURFEL_ROAD_CLASSES]
    self.indices_other_classes = [i for i, c in enumerate(self.raster_classes) if c not in SURFEL_ROAD_CLASSES]
    if not self.indices_surfel_road:
      raise ValueError('SemanticRasterEncoder: no surfel-road class among the raster classes')
    if config.embedding_dim % 4:
      raise NotImplementedError('embedding_dim must be a multiple of 4')
    in_channels = config.embedding_dim * (1 + len(self.indices_other_classes))
    self.encoder = image_encoder.ImageEncoder(config.encoder, dtype, in_channels=in_channels)

  def init_params(self, gen, device):
    E = self.config.embedding_dim
    nr, no = len(self.indices_surfel_road), len(self.indices_other_classes)
    std = 1.0 / E**0.5     # flax default_embed_init: variance_scaling(1, 'fan_in', 'normal', out_axis=0)
    return {
        'encoder': self.encoder.init_params(gen, device),
        'embeddings_surfel_road': {'embedding': (torch.randn((nr, E), generator=gen) * std).to(device)},
        'embeddings_other_classes': {'embedding': (torch.randn((2 * no, E), generator=gen) * std).to(device)},
    }

  def __call__(self, params, rasters, train=False, ctx=None):
    if rasters.shape[-1] != len(self.raster_classes):
      raise ValueError('rasters: last axis must list every raster class')
    t_road = params['embeddings_surfel_road']['embedding']
    t_other = params['embeddings_other_classes']['embedding']
    embed = ag.semantic_embed if base.needs_grad(t_road, t_other) else ops.semantic_embed
    f_rasters = embed(rasters.contiguous(), self.indices_surfel_road, self.indices_other_classes,
                      t_road, t_other)
    return self.encoder(params['encoder'], f_rasters, train=train, ctx=ctx)

  default_config = staticmethod(default_configs.semantic_raster_encoder)
