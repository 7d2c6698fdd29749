"""Common building blocks (``snap/models/layers.py``): the Dense-stack MLP.

``normalize`` (layers.py:45-52) is fused into bev_ops.hip's matching head;
``masked_softmax`` / ``masked_mean`` are only used by non-default confidence
paths of the reference and have no counterpart here.
"""
import torch

from snap_amd import autograd as ag
from snap_amd import ops
from snap_amd.models import base


class MLP(base.Module):
  """ReLU MLP (layers.py:55-78); every Dense runs on the MFMA conv engine.

  The ReLU between layers is fused into the previous layer's epilogue; an input
  activation (``apply_input_activation``) is fused into the first layer's
  operand staging.  ``in_dim`` is the logical input width (the tensor may carry a
  wider, padded row stride).
  """

  def __init__(self, config, in_dim, dtype=None):
    if config.activation != 'relu':
      raise NotImplementedError(config.activation)
    self.config = config
    self.in_dim = in_dim

  def init_params(self, gen, device):
    params = {}
    d_in = self.in_dim
    for i, d in enumerate(self.config.layers):
      params[f'Dense_{i}'] = {
          'kernel': base.glorot_uniform(gen, (d_in, d), d_in, d, device),
          'bias': torch.zeros(d, device=device),
      }
      d_in = d
    return params

  def __call__(self, params, x, train=False, row_mask=None):
    n = len(self.config.layers)
    grad = any(base.needs_grad(x, params[f'Dense_{i}']['kernel'], params[f'Dense_{i}']['bias'])
               for i in range(n))
    if row_mask is not None and row_mask.numel() >= self.COMPACT_MIN_ROWS:
      if not grad:
        return self._masked_rows(params, x, row_mask)
      if all(params[f'Dense_{i}']['kernel'].shape[1] % 4 == 0 for i in range(n)):
        wb = []
        for i in range(n):
          wb += [params[f'Dense_{i}']['kernel'], params[f'Dense_{i}']['bias']]
        return ag.masked_rows_mlp(x, row_mask, bool(self.config.apply_input_activation), wb)
    for i in range(n):
      p = params[f'Dense_{i}']
      pro = ops.PRO_RELU if (i == 0 and self.config.apply_input_activation) else ops.PRO_NONE
      fn = ag.dense if base.needs_grad(x, p['kernel'], p['bias']) else ops.dense
      x = fn(
          x, p['kernel'], p['bias'], cin=p['kernel'].shape[0], prologue=pro,
          relu=(i + 1 < n), row_mask=row_mask if i + 1 == n else None,
      )
    return x

  # Below this many rows the three tiny compaction launches cost more than they save.
  COMPACT_MIN_ROWS = 1 << 16

  def reads_listed_rows_only(self, params, n_rows):
    """True when ``__call__(params, x, row_mask=mask)`` with ``n_rows`` rows never touches a row of ``x`` whose mask
    is False -- forward or backward (the compacted-row paths above; the dense fallback multiplies every row, and
    its kernel gradient would pick up whatever sits in the masked ones)."""
    n = len(self.config.layers)
    return (n_rows >= self.COMPACT_MIN_ROWS and not self.config.apply_input_activation
            and all(params[f'Dense_{i}']['kernel'].shape[1] % 4 == 0 for i in range(n)))

  def _masked_rows(self, params, x, row_mask):
    """Inference path for a row-masked MLP: rows with mask == 0 come out as zeros whatever
    the MLP computes (streetview_encoder.py:281-283), so only the listed rows are
    multiplied.  Bitwise the same values as the dense path on the kept rows."""
    n = len(self.config.layers)
    lead = x.shape[:-1]
    M = row_mask.numel()
    index, count = ops.compact_rows(row_mask)
    h = x.reshape(M, x.shape[-1])
    for i in range(n):
      p = params[f'Dense_{i}']
      pro = ops.PRO_RELU if (i == 0 and self.config.apply_input_activation) else ops.PRO_NONE
      last = i + 1 == n
      h = ops.dense(
          h, p['kernel'], p['bias'], cin=p['kernel'].shape[0], prologue=pro, relu=not last,
          rows_in=index if i == 0 else None, rows_out=index if last else None, row_count=count,
      )
    ops.fill_masked_rows_(h, row_mask)
    return h.reshape(*lead, h.shape[-1])
