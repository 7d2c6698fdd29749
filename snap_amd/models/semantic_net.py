"""Semantic-mapping head on the neural map (``snap/models/semantic_net.py``).

``SemanticNet`` = ``BEVMapper`` + a per-cell decoder ('mlp' or Dense + ResNetStage + MLP, all on
the conv engine) -> area / object logits; ``SemanticNetModel`` is the trainer-facing wrapper with
the label construction and the (class-balanced) cross-entropy losses and recalls, host-side
torch on [B, H, W, n_classes] logits like ``BEVLocalizerModel.loss_metrics_function``.
"""
import numpy as np
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.configs import defaults as default_configs
from snap_amd.models import base
from snap_amd.models import bev_mapper
from snap_amd.models import layers
from snap_amd.models import resnet
from snap_amd.models import types


def masked_mean(x, mask, axis):
  """layers.py:31-34: mean over ``axis`` where ``mask``; 0 for an empty mask."""
  x, mask = torch.broadcast_tensors(x, mask)
  any_ = mask.any(axis, keepdim=True)
  div = torch.where(any_, mask, torch.ones_like(mask)).sum(axis)
  return (x * mask).sum(axis) / div


def balancing_weights(frequencies, classes, binary=False, eps=1e-3):
  """semantic_net.py:38-53."""
  freq = np.array([frequencies[c] for c in classes], dtype=np.float64)
  if not binary:
    freq = freq / freq.sum()
  freq = freq.clip(min=eps)
  weights = 1 / (freq * len(classes))
  if binary:
    weights_neg = 1 / ((1 - freq).clip(min=eps) * len(classes))
    return weights, weights_neg
  return weights


def multiclass_crossentropy_metrics(logits, labels, valid, classes, frequencies, namespace=None):
  """semantic_net.py:56-85."""
  nll = -torch.log_softmax(logits, dim=-1).gather(-1, labels[..., None])[..., 0]
  if frequencies:
    w = torch.as_tensor(balancing_weights(dict(frequencies), classes), dtype=nll.dtype, device=nll.device)
    nll = nll * w[labels]
  nll = masked_mean(nll, valid, (1, 2))
  mask = labels[..., None] == torch.arange(logits.shape[-1], device=logits.device)
  correct = torch.argmax(logits, dim=-1) == labels
  acc = masked_mean(correct.to(logits.dtype), valid, (1, 2))
  recall = masked_mean(correct[..., None].to(logits.dtype), valid[..., None] & mask, (1, 2))
  suffix = f'/{namespace}' if namespace else ''
  metrics = {f'accuracy{suffix}': acc, f'recall/average{suffix}': recall.mean(-1)}
  for i, c in enumerate(classes):
    metrics[f'recall/{c}'] = recall[..., i]
  return nll, metrics


def binary_crossentropy_metrics(logits, gt_mask, valid, classes, frequencies, namespace=None):
  """semantic_net.py:88-111."""
  ls = torch.nn.functional.logsigmoid
  gm = gt_mask.to(logits.dtype)
  nll = -gm * ls(logits) - (1 - gm) * ls(-logits)        # optax.sigmoid_binary_cross_entropy
  if frequencies:
    w_pos, w_neg = balancing_weights(dict(frequencies), classes, binary=True)
    w_pos = torch.as_tensor(w_pos, dtype=nll.dtype, device=nll.device)
    w_neg = torch.as_tensor(w_neg, dtype=nll.dtype, device=nll.device)
    nll = nll * torch.where(gt_mask, w_pos, w_neg)
  nll = masked_mean(nll.mean(-1), valid, (1, 2))
  correct = (torch.sigmoid(logits) > 0.5) == gt_mask
  recall = masked_mean(correct.to(logits.dtype), valid[..., None] & gt_mask, (1, 2))
  suffix = f'/{namespace}' if namespace else ''
  metrics = {f'recall/average{suffix}': recall.mean(-1)}
  for i, c in enumerate(classes):
    metrics[f'recall/{c}'] = recall[..., i]
  return nll, metrics


def batched_raster_flip(raster, flip_mask):
  """semantic_net.py:114-120: per-sample flips of the two spatial axes."""
  out = raster
  for i in range(2):
    flipped = torch.flip(out, dims=(i + 1,))
    sel = flip_mask[:, i].reshape(-1, *([1] * (out.dim() - 1)))
    out = torch.where(sel, flipped, out)
  return out


class SemanticNet(base.Module):
  """semantic_net.py:123-199."""

  def __init__(self, config, grid, dtype=torch.float32, semantic_map_classes=None):
    self.config = config
    self.grid = grid
    self.bev_mapper = bev_mapper.BEVMapper(config.bev_mapper, grid, semantic_map_classes, dtype)
    num_classes = len(config.area_classes)
    self.object_classes = tuple(config.object_classes_exclusive) + tuple(config.object_classes_independent)
    if self.object_classes:
      num_classes += len(self.object_classes) + 1            # void
    self.num_classes = num_classes
    dim = config.decoder_dim
    in_dim = self.bev_mapper.feature_dim
    mlp_config = default_configs.mlp()
    if config.decoder_type == 'mlp':
      mlp_config.layers = (dim,) * config.mlp_num_layers + (num_classes,)
      self.decoder_mlp = layers.MLP(mlp_config, in_dim=in_dim)
    elif config.decoder_type == 'resnet_stage':
      if dim % 128:
        raise NotImplementedError('resnet_stage decoder: GroupNorm(32) needs decoder_dim % 128 == 0')
      mlp_config.layers = (dim, num_classes)
      self.decoder_mlp = layers.MLP(mlp_config, in_dim=dim)
      self.in_dim = in_dim
    else:
      raise ValueError(f'Unknown {config.decoder_type}')

  def init_params(self, gen, device):
    cfg = self.config
    params = {'bev_mapper': self.bev_mapper.init_params(gen, device)}
    if cfg.decoder_type == 'mlp':
      params['decoder'] = self.decoder_mlp.init_params(gen, device)
    else:
      dim = cfg.decoder_dim
      stage = {}
      for u in range(cfg.resnet_num_units):
        stage[f'unit{u + 1:02d}'] = resnet._init_unit(gen, device, dim, dim // 4, 1)
      params['decoder'] = {      # nn.Sequential naming: layers_i
          'layers_0': {'kernel': base.glorot_uniform(gen, (self.in_dim, dim), self.in_dim, dim, device),
                       'bias': torch.zeros(dim, device=device)},
          'layers_1': stage,
          'layers_3': self.decoder_mlp.init_params(gen, device),
      }
    return params

  def _decode(self, params, feats, ctx):
    cfg = self.config
    if cfg.decoder_type == 'mlp':
      return self.decoder_mlp(params, feats)
    p0 = params['layers_0']
    dense = ag.dense if base.needs_grad(feats, p0['kernel'], p0['bias']) else ops.dense
    x = dense(feats, p0['kernel'], p0['bias'])
    for u in range(cfg.resnet_num_units):
      x = resnet.residual_unit(ctx, params['layers_1'][f'unit{u + 1:02d}'], x, 1, None)
    return self.decoder_mlp(params['layers_3'], x)

  def __call__(self, params, data, train=False, debug=False, rng=None, ctx=None):
    cfg = self.config
    ctx = ctx or base.ForwardContext()
    if 'map' in data:
      data = data['map']
    pred = self.bev_mapper(params['bev_mapper'], data, train, ctx=ctx, rng=rng)
    neural_map = pred['bev_features']
    flips = None
    feats, valid = neural_map.features, neural_map.valid
    if train and cfg.apply_random_flip:
      gen = torch.Generator(device='cpu')
      gen.manual_seed(0 if rng is None else int(rng) + 104729)
      flips = (torch.rand((len(feats), 2), generator=gen) < 0.5).to(feats.device)
      feats = batched_raster_flip(feats, flips)
      valid = batched_raster_flip(valid, flips)
    logits = self._decode(params['decoder'], feats.contiguous(), ctx).to(torch.float32)
    logits = torch.where(valid[..., None], logits, torch.zeros_like(logits))
    if flips is not None:
      logits = batched_raster_flip(logits, flips)
    na = len(cfg.area_classes)
    pred['logits_areas'] = logits[..., :na]
    if self.object_classes:
      ne = len(cfg.object_classes_exclusive) + 1
      pred['logits_objects_exclusive'] = logits[..., na:na + ne]
      pred['logits_objects_independent'] = logits[..., na + ne:]
    return pred

  # Flax-style functional API (as BEVLocalizer)
  def init(self, seed, device='cpu'):
    gen = torch.Generator().manual_seed(int(seed))
    return {'params': self.init_params(gen, device)}

  def apply(self, variables, data, train=False, debug=False, rngs=None, mutable=False):
    rng = None if not rngs else rngs.get('sampling')
    return self(variables['params'], data, train=train, debug=debug, rng=rng)

  default_config = staticmethod(default_configs.semantic_net)


class SemanticNetModel(base.BaseModel):
  """Trainer-facing wrapper (semantic_net.py:206-360)."""

  def build_flax_model(self):
    meta = self.dataset_meta_data
    return SemanticNet(self.config, meta['grid'].bev(), self.dtype, meta.get('semantic_map_classes'))

  @classmethod
  def default_flax_model_config(cls):
    return default_configs.semantic_net()

  @property
  def gt_indices(self):
    return {c: i for i, c in enumerate(self.dataset_meta_data['semantic_classes_gt'])}

  def transfer_labels_from_pcm(self, masks, masks_pcm):
    indices_gt = self.gt_indices
    indices_pcm = {c: i for i, c in enumerate(self.dataset_meta_data['semantic_map_classes'] or ())}
    masks = masks.clone()
    for name_gt, name_pcm in (('building', 'buildings_raw'), ('tree', 'tree')):
      if name_gt in indices_gt and name_pcm in indices_pcm:
        masks[..., indices_gt[name_gt]] = masks_pcm[..., indices_pcm[name_pcm]]
    return masks

  def _create_exclusive_labels(self, masks_all, classes, add_void=False):
    gt = self.gt_indices
    masks = masks_all[..., [gt[c] for c in classes]].clone()
    if 'line' in classes:                                  # group all line labels
      mask_line = masks_all[..., gt['line']]
      for c in ('stopline', 'otherlanemarking'):
        if c in gt and c not in classes:
          mask_line = mask_line | masks_all[..., gt[c]]
      masks[..., list(classes).index('line')] = mask_line
    valid = masks.any(-1)
    labels = torch.argmax(masks.to(torch.int32), dim=-1)
    if add_void:
      labels = torch.where(valid, labels, torch.full_like(labels, len(classes)))
    return labels, valid

  def create_area_labels(self, masks_all):
    return self._create_exclusive_labels(masks_all, self.config.area_classes)

  def create_object_labels(self, masks):
    labels_excl, _ = self._create_exclusive_labels(masks, self.config.object_classes_exclusive, add_void=True)
    gt = self.gt_indices
    masks_indep = masks[..., [gt[c] for c in self.config.object_classes_independent]]
    return labels_excl, masks_indep

  def _loss_metrics_areas(self, pred, masks):
    labels, valid = self.create_area_labels(masks)
    valid = pred['bev_features'].valid & valid
    return multiclass_crossentropy_metrics(
        pred['logits_areas'], labels, valid, self.config.area_classes,
        dict(self.config.area_frequencies or []))

  def _loss_metrics_objects(self, pred, masks):
    labels_excl, masks_indep = self.create_object_labels(masks)
    valid = pred['bev_features'].valid
    nll_excl, m_excl = multiclass_crossentropy_metrics(
        pred['logits_objects_exclusive'], labels_excl, valid,
        (*self.config.object_classes_exclusive, 'void'),
        dict(self.config.object_frequencies or []), namespace='excl')
    nll_indep, m_indep = binary_crossentropy_metrics(
        pred['logits_objects_independent'], masks_indep, valid,
        self.config.object_classes_independent,
        dict(self.config.object_frequencies or []), namespace='indep')
    return nll_excl, nll_indep, {**m_excl, **m_indep}

  def loss_metrics_function(self, pred, data, model_params=None):
    if 'map' in data:
      data = data['map']
    masks = data['rasters']['gt_semantics']
    masks = self.transfer_labels_from_pcm(masks, data['rasters'].get('semantics'))
    nll_areas, metrics = self._loss_metrics_areas(pred, masks)
    losses = {'nll_areas': nll_areas}
    total = nll_areas
    if 'logits_objects_exclusive' in pred:
      nll_excl, nll_indep, m_obj = self._loss_metrics_objects(pred, masks)
      total = (total + (nll_excl + nll_indep) / 2) / 2
      losses['nll_objects_exclusive'] = nll_excl
      losses['nll_objects_indep'] = nll_indep
      metrics = {**metrics, **m_obj}
    losses['total'] = total
    return losses, {f'semantics/{k}': v for k, v in metrics.items()}

  def pack_evaluation_metrics(self, training_metrics, losses, data, pred):
    if 'map' in data:
      data = data['map']
    gt_classes = self.dataset_meta_data['semantic_classes_gt']
    counts = data['rasters']['gt_semantics'].sum(dim=(-3, -2))
    gt_counts = {f'gt_counts/{c}': counts[..., i] for i, c in enumerate(gt_classes)}
    return {**training_metrics, 'loss': losses['total'], **gt_counts}
