"""Evaluation contract of the localiser (``snap/evaluator.py:46-109,205-238``).

``eval_step`` is what the reference jits per batch: forward (train=False), the model's
loss/metric function, then ``pack_localization_metrics``.  ``eval_on_batches`` collects
the per-example rows of the valid entries of every batch (the role of
``eval_on_dataset``; the TFDS iterator is out of scope -- any iterable of batch dicts
works), ``write_eval_dump`` / ``read_eval_dump`` keep the ``results.npz`` (+ config)
layout and ``compute_recall`` is the cumulative recall curve of the paper's plots.
"""
import io
import json
import os
from typing import Any, Dict, Iterable

import numpy as np
import torch

from snap_amd.utils import geometry


def compute_distance_view_to_map(m_t_vq, m_t_vm):
  """Rotation / translation distance from the query view to its CLOSEST (in translation)
  map view.  m_t_vq: Transform3D [...]; m_t_vm: Transform3D [..., V]."""
  Rq_inv = m_t_vq.R.transpose(-1, -2)[..., None, :, :]            # broadcast over the V views
  rel = geometry.Transform3D(
      Rq_inv @ m_t_vm.R,
      torch.einsum('...ij,...j->...i', Rq_inv, m_t_vm.t - m_t_vq.t[..., None, :]),
  )
  dr, dt = rel.magnitude()
  dt_closest, idx = dt.min(-1)
  dr_closest = torch.gather(dr, -1, idx[..., None])[..., 0]
  return dr_closest, dt_closest


def pack_localization_metrics(training_metrics, losses, data, pred) -> Dict[str, torch.Tensor]:
  """Per-example evaluation record ([B] tensors)."""
  m_t_vq = data['T_query2map'] @ data['query']['T_view2scene'][..., 0]
  dr_closest, dt_closest = compute_distance_view_to_map(m_t_vq, data['map']['T_view2scene'])
  B = losses['total'].shape[0]
  dev = losses['total'].device

  def opt(key):  # dataset fields that only real batches carry
    v = data.get(key)
    return v if v is not None else torch.full((B,), float('nan'), device=dev)

  return dict(
      error_max_meter=training_metrics['loc/err_max_position'],
      error_max_deg=training_metrics['loc/err_max_rotation'],
      recall_top1=training_metrics['loc/recall_top1'],
      pose_score_max=pred['scores_poses'][..., 1:].max(-1).values,
      overlap=opt('overlap'),
      time_delta_days=opt('time_delta_days'),
      closest_map_view_meter=dt_closest,
      closest_map_view_deg=dr_closest,
      loss=losses['total'],
  )


@torch.no_grad()
def eval_step(params, batch, *, rng, model) -> Dict[str, torch.Tensor]:
  pred = model.flax_model.apply({'params': params}, batch, train=False, mutable=False,
                                debug=False, rngs={'sampling': rng})
  losses, metrics = model.loss_metrics_function(pred, batch, params)
  # evaluator.py:100-108: the localizer packs here, other models pack themselves.
  from snap_amd.models import bev_localizer
  if isinstance(model, bev_localizer.BEVLocalizerModel):
    return pack_localization_metrics(metrics, losses, batch, pred)
  if hasattr(model, 'pack_evaluation_metrics'):
    return model.pack_evaluation_metrics(metrics, losses, batch, pred)
  raise ValueError(f'No packing function for model {type(model).__name__}.')


def eval_on_batches(model, params, batches: Iterable[Dict[str, Any]], rng: int = 0,
                    in_flight: int = 2) -> Dict[str, np.ndarray]:
  """Rows of every example with batch_mask set, stacked per metric.

  ``in_flight`` batches are enqueued on alternating HIP streams before the oldest one's metrics are
  fetched (``snap_amd.pipeline``): the batches are independent, the rows are the same bits in the
  same order whatever the value; 1 = one batch at a time."""
  import collections
  from snap_amd import pipeline
  rows: Dict[str, list] = {}
  ring = None
  pending = collections.deque()

  def fetch(entry):
    i, batch, metrics = entry
    with ring.slot(i):                   # (the copies queue on the batch's own stream and wait for it only)
      keep = batch['batch_mask'].to(torch.bool).cpu().numpy()
      for k, v in metrics.items():
        rows.setdefault(k, []).append(v.detach().to(torch.float64).cpu().numpy()[keep])

  for i, batch in enumerate(batches):
    if ring is None:
      ring = pipeline.BatchesInFlight(in_flight, batch['batch_mask'].device)
    with ring.slot(i):
      metrics = eval_step(params, batch, rng=rng, model=model)
    pending.append((i, batch, metrics))  # (the batch stays alive until its stream has consumed it)
    if len(pending) >= ring.n:
      fetch(pending.popleft())
  while pending:
    fetch(pending.popleft())
  if ring is not None:
    ring.join()
  return {k: np.concatenate(v) for k, v in rows.items()}


def write_eval_dump(eval_dir, results: Dict[str, np.ndarray], config, compressed: bool = False):
  os.makedirs(eval_dir, exist_ok=True)
  buf = io.BytesIO()
  (np.savez_compressed if compressed else np.savez)(buf, **results)
  with open(os.path.join(eval_dir, 'results.npz'), 'wb') as f:
    f.write(buf.getvalue())
  cfg = config.to_dict() if hasattr(config, 'to_dict') else dict(config)
  with open(os.path.join(eval_dir, 'config.json'), 'w') as f:
    json.dump(cfg, f, indent=1, default=str)


def read_eval_dump(eval_dir):
  with open(os.path.join(eval_dir, 'results.npz'), 'rb') as f:
    results = dict(np.load(io.BytesIO(f.read()), allow_pickle=False))
  with open(os.path.join(eval_dir, 'config.json')) as f:
    config = json.load(f)
  return results, config


def compute_recall(errors, max_error: float):
  """Cumulative recall (percent) at 100 thresholds in [0, max_error]."""
  thresholds = np.linspace(0, max_error, 100)
  recall = np.mean(np.asarray(errors)[None] < thresholds[:, None], axis=1)
  return thresholds, recall * 100
