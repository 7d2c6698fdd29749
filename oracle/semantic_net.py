"""Oracle (test infrastructure): SemanticNet forward, labels and losses in numpy.

Restates ``snap/models/semantic_net.py:38-120`` (class-balanced cross-entropies, recalls,
raster flips), ``:123-199`` (BEVMapper + decoder -> logits) and ``:225-343`` (label
construction, loss / metric assembly).  Parity unpinned by the reference (it ships no tests);
the cross-entropies are pinned to torch's implementations in ``tests/test_oracle_pins.py``.
"""
import numpy as np

from oracle import bev
from oracle import encoder


def masked_mean(x, mask, axis):
  """layers.py:31-34."""
  mask = np.broadcast_to(mask, x.shape)
  div = np.sum(np.where(mask.any(axis, keepdims=True), mask, True), axis)
  return np.sum(x * mask, axis) / div


def balancing_weights(frequencies, classes, binary=False, eps=1e-3):
  """semantic_net.py:38-53."""
  freq = np.array([frequencies[c] for c in classes], dtype=np.float64)
  if not binary:
    freq = freq / freq.sum()
  freq = freq.clip(min=eps)
  weights = 1 / (freq * len(classes))
  if binary:
    return weights, 1 / ((1 - freq).clip(min=eps) * len(classes))
  return weights


def _log_softmax(x):
  x = x - x.max(-1, keepdims=True)
  return x - np.log(np.exp(x).sum(-1, keepdims=True))


def multiclass_crossentropy_metrics(logits, labels, valid, classes, frequencies, namespace=None):
  """semantic_net.py:56-85."""
  nll = -np.take_along_axis(_log_softmax(logits), labels[..., None], -1)[..., 0]
  if frequencies:
    nll = nll * balancing_weights(dict(frequencies), classes)[labels]
  nll = masked_mean(nll, valid, (1, 2))
  mask = labels[..., None] == np.arange(logits.shape[-1])
  correct = np.argmax(logits, -1) == labels
  acc = masked_mean(correct.astype(logits.dtype), valid, (1, 2))
  recall = masked_mean(np.broadcast_to(correct[..., None], mask.shape).astype(logits.dtype),
                       valid[..., None] & mask, (1, 2))
  suffix = f'/{namespace}' if namespace else ''
  metrics = {f'accuracy{suffix}': acc, f'recall/average{suffix}': recall.mean(-1)}
  for i, c in enumerate(classes):
    metrics[f'recall/{c}'] = recall[..., i]
  return nll, metrics


def _log_sigmoid(x):
  return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def binary_crossentropy_metrics(logits, gt_mask, valid, classes, frequencies, namespace=None):
  """semantic_net.py:88-111."""
  gm = gt_mask.astype(logits.dtype)
  nll = -gm * _log_sigmoid(logits) - (1 - gm) * _log_sigmoid(-logits)
  if frequencies:
    w_pos, w_neg = balancing_weights(dict(frequencies), classes, binary=True)
    nll = nll * np.where(gt_mask, w_pos, w_neg)
  nll = masked_mean(nll.mean(-1), valid, (1, 2))
  correct = ((1 / (1 + np.exp(-logits))) > 0.5) == gt_mask
  recall = masked_mean(correct.astype(logits.dtype), valid[..., None] & gt_mask, (1, 2))
  suffix = f'/{namespace}' if namespace else ''
  metrics = {f'recall/average{suffix}': recall.mean(-1)}
  for i, c in enumerate(classes):
    metrics[f'recall/{c}'] = recall[..., i]
  return nll, metrics


def semantic_net(params, config, grid, data):
  """semantic_net.py:167-199 (eval: no flips)."""
  if 'map' in data:
    data = data['map']
  pred = bev.bev_mapper(params['bev_mapper'], config['bev_mapper'], grid, data)
  plane = pred['bev_features']
  x = plane['features']
  dec = params['decoder']
  object_classes = tuple(config['object_classes_exclusive']) + tuple(config['object_classes_independent'])
  num_classes = len(config['area_classes']) + (len(object_classes) + 1 if object_classes else 0)
  if config['decoder_type'] == 'mlp':
    layers = (config['decoder_dim'],) * config['mlp_num_layers'] + (num_classes,)
    x = encoder.mlp(dec, dict(layers=layers, apply_input_activation=False), x)
  else:
    x = encoder.dense(dec['layers_0'], x)
    x, _ = encoder.resnet_stage(dec['layers_1'], x, config['resnet_num_units'], None)
    x = encoder.mlp(dec['layers_3'], dict(layers=(config['decoder_dim'], num_classes),
                                           apply_input_activation=False), x)
  logits = np.where(plane['valid'][..., None], x.astype(np.float32), 0)
  na = len(config['area_classes'])
  pred['logits_areas'] = logits[..., :na]
  if object_classes:
    ne = len(config['object_classes_exclusive']) + 1
    pred['logits_objects_exclusive'] = logits[..., na:na + ne]
    pred['logits_objects_independent'] = logits[..., na + ne:]
  return pred


def create_exclusive_labels(masks_all, classes, gt_indices, add_void=False):
  """semantic_net.py:243-264."""
  masks = masks_all[..., [gt_indices[c] for c in classes]].copy()
  if 'line' in classes:
    mask_line = masks_all[..., gt_indices['line']]
    for c in ('stopline', 'otherlanemarking'):
      if c in gt_indices and c not in classes:
        mask_line = mask_line | masks_all[..., gt_indices[c]]
    masks[..., list(classes).index('line')] = mask_line
  valid = masks.any(-1)
  labels = np.argmax(masks, -1)
  if add_void:
    labels = np.where(valid, labels, len(classes))
  return labels, valid


def loss_metrics(pred, config, gt_classes, map_classes, rasters):
  """semantic_net.py:225-241,286-343."""
  gt = {c: i for i, c in enumerate(gt_classes)}
  pcm = {c: i for i, c in enumerate(map_classes or ())}
  masks = rasters['gt_semantics'].copy()
  for name_gt, name_pcm in (('building', 'buildings_raw'), ('tree', 'tree')):
    if name_gt in gt and name_pcm in pcm:
      masks[..., gt[name_gt]] = rasters['semantics'][..., pcm[name_pcm]]
  bev_valid = pred['bev_features']['valid']
  labels, valid = create_exclusive_labels(masks, config['area_classes'], gt)
  nll_areas, metrics = multiclass_crossentropy_metrics(
      pred['logits_areas'], labels, bev_valid & valid, config['area_classes'],
      dict(config['area_frequencies'] or []))
  losses = {'nll_areas': nll_areas}
  total = nll_areas
  if 'logits_objects_exclusive' in pred:
    labels_excl, _ = create_exclusive_labels(masks, config['object_classes_exclusive'], gt, add_void=True)
    masks_indep = masks[..., [gt[c] for c in config['object_classes_independent']]]
    nll_excl, m_excl = multiclass_crossentropy_metrics(
        pred['logits_objects_exclusive'], labels_excl, bev_valid,
        (*config['object_classes_exclusive'], 'void'), dict(config['object_frequencies'] or []),
        namespace='excl')
    nll_indep, m_indep = binary_crossentropy_metrics(
        pred['logits_objects_independent'], masks_indep, bev_valid,
        config['object_classes_independent'], dict(config['object_frequencies'] or []),
        namespace='indep')
    total = (total + (nll_excl + nll_indep) / 2) / 2
    losses['nll_objects_exclusive'] = nll_excl
    losses['nll_objects_indep'] = nll_indep
    metrics = {**metrics, **m_excl, **m_indep}
  losses['total'] = total
  return losses, {f'semantics/{k}': v for k, v in metrics.items()}
