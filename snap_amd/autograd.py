"""Differentiable wrappers (torch.autograd.Function) around the HIP kernels.

The reference gets gradients from ``jax.grad`` over the whole model
(snap/trainer.py:223-234).  Here every forward kernel has a hand-written VJP kernel
(``snap_amd/ops_bwd.py``); the wrappers only record what the backward needs.  Data
gradients of convolutions reuse the forward MFMA engine with the rotated /
transposed kernel; kernel gradients use the transpose-A engine (wgrad.hip).

With ``torch.no_grad()`` (inference) the wrappers reduce to the plain ``ops`` calls.
"""
import contextlib

import torch
import torch.nn.functional as F

from snap_amd import ops
from snap_amd import ops_bwd

_GN_MODES = (ops.PRO_GN_RELU, ops.PRO_RELU_GN)


def _engine_scoped(cls):
  """Class decorator of every Function below: the node records the engine in force at its forward
  (``ops.precision()``: the applying model's, see ops.engine_scope) and re-enters it in its backward
  -- autograd runs backward passes on its own threads, where the forward thread's scope is not
  visible.  A model's backward therefore runs on the model's engine whatever else the process is
  doing, and two models of different precision can be trained / evaluated side by side."""
  fwd, bwd = cls.forward, cls.backward

  def forward(ctx, *args):
    ctx._snap_engine = ops.precision()
    ctx._snap_tuning = ops.tuning()          # (the Tuning object travels with the engine)
    return fwd(ctx, *args)

  def backward(ctx, *grads):
    with ops.engine_scope(ctx._snap_engine, tuning=ctx._snap_tuning):
      return bwd(ctx, *grads)

  cls.forward = staticmethod(forward)
  cls.backward = staticmethod(backward)
  return cls


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def conv_dgrad(dy, w, x_shape, stride, padding, accumulate=None, accumulate_inplace=False, gn_bwd_stats=None):
  """d(prologue output) of a conv: transposed convolution through the forward engine.
  accumulate [N,H,W,roundup(Cin,4)]: added in the engine's epilogue (the data gradient of another
  conv reading the same activation); accumulate_inplace: the caller owns that tensor and it may be
  overwritten with the sum.  gn_bwd_stats = (x, mu, rstd, gamma, beta, mode) of the GroupNorm prologue whose
  VJP consumes the result: its statistics pass rides in the epilogue of the (half-input, unsplit) launch
  (``ops.conv2d(gn_bwd_stats=)``); ignored where the launch cannot carry it.

  dy [N,Ho,Wo,Cout]; w [KH,KW,Cin,Cout]; returns dz [N,H,W,roundup(Cin,4)] (channels
  past Cin are zero).
  Stride > 1, 1 x 1 unpadded (the projection shortcuts): the GEMM runs over the Ho x Wo pixels that
  have a gradient at all and is added into every stride-th pixel of the result.  Other strided
  kernels: dy is zero-dilated first (the three strided 3 x 3 layers of a ResNet).
  """
  N, H, W, Cin = x_shape
  KH, KW, _, Cout = w.shape
  (pt, pb), (pl, pr) = padding
  if stride > 1 and STRIDED_1X1_DGRAD and KH == 1 and KW == 1 and pt == pb == pl == pr == 0:
    Ho, Wo = dy.shape[1:3]
    t = conv_dgrad(dy, w, (N, Ho, Wo, Cin), 1, padding)             # [N, Ho, Wo, C4]: 1 / stride^2 of the rows
    if accumulate is None:
      out = torch.zeros((N, H, W, t.shape[-1]), dtype=torch.float32, device=dy.device)
      out[:, :(Ho - 1) * stride + 1:stride, :(Wo - 1) * stride + 1:stride] = t
    else:
      out = accumulate if accumulate_inplace else accumulate.clone()
      out[:, :(Ho - 1) * stride + 1:stride, :(Wo - 1) * stride + 1:stride] += t
    return out
  if stride > 1:
    Ho, Wo = dy.shape[1:3]
    Hd, Wd = (Ho - 1) * stride + 1, (Wo - 1) * stride + 1
    dyd = torch.zeros((N, Hd, Wd, Cout), dtype=dy.dtype, device=dy.device)
    dyd[:, ::stride, ::stride] = dy
    dy = dyd
  Ho, Wo = dy.shape[1:3]
  half_math = ops.precision() if ops.precision() in ops.HALF_MATH else None
  rot_img = ops.packed_rot_image(w, half_math) if half_math else None
  # (the image-only weight below is read only by the bf16 / fp16 engine, which takes the
  #  launch when the rotated kernel's input channels -- the original Cout -- form aligned quads;
  #  anything else must rebuild the f32 rotated kernel, or the f32 engine would read garbage)
  if rot_img is not None and not (Cout >= 4 and Cout % 4 == 0 and dy.data_ptr() % 16 == 0):
    rot_img = None
  if rot_img is not None:
    # training precision: the rotated kernel's bf16 image was prepared with every other image of the
    # step (ops.pack_weights_bf16_multi); the engine reads only the image, so the f32 tensor is a
    # shape carrier (no flip / permute / pad / pack launches here)
    w_rot = torch.empty((KH, KW, Cout, (Cin + 3) // 4 * 4), dtype=torch.float32, device=dy.device)
    w_rot._snap_packed = {half_math: (ops.PACK_EPOCH, w_rot._version, rot_img)}
  else:
    w_rot = w.flip(0, 1).permute(0, 1, 3, 2)                    # [KH,KW,Cout,Cin]
    if Cin % 4:                                                  # engine writes 4-channel groups:
      w_rot = F.pad(w_rot, (0, 4 - Cin % 4))                     # extra channels come out as exact zeros
    w_rot = w_rot.contiguous()
  pt2, pl2 = KH - 1 - pt, KW - 1 - pl
  pb2 = H - Ho - pt2 + KH - 1
  pr2 = W - Wo - pl2 + KW - 1
  dy = dy.contiguous()
  if half_math and stride == 1 and Cout % 8 == 0:
    # the GroupNorm VJP that wrote dy also wrote it in the engine's element type: read that twin
    # (both operands by LDS-DMA, half the bytes; same bits as rounding the f32 tensor in the loop)
    twin = ops_bwd.half_twin(dy, half_math)
    if twin is not None:
      return ops.conv2d(twin, w_rot, padding=((pt2, pb2), (pl2, pr2)), residual=accumulate,
                        gn_bwd_stats=gn_bwd_stats if GN_BWD_STATS_IN_DGRAD else None)
  return ops.conv2d(dy, w_rot, padding=((pt2, pb2), (pl2, pr2)), residual=accumulate)


STRIDED_1X1_DGRAD = True     # (tests: False keeps the zero-dilated formulation)
GN_BWD_STATS_IN_DGRAD = True  # (tests: False = the GroupNorm VJP always takes its own statistics pass)


def _own(grad):
  """A contiguous gradient buffer this node may overwrite: the incoming one itself when the op that
  produced it handed it over (ops_bwd.mark_scratch: the pose-score VJP's 1.2 GB plane stack at C3 --
  the clone was 0.5 ms per step), else a copy."""
  g = grad.contiguous()
  return g if ops_bwd.take_scratch(g) else g.clone()


def similarity_bwd(dsim, sim, fq, fm, scale, clip, num_valid, row_weight=None):
  """VJP of sim = relu(fq . fm) * scale / num_valid -- or * scale * row_weight[b, n]
  (add_confidence_query).  `dsim` is overwritten.

  Returns dfq [B,Nq,Dm], dfm [B,X,Y,Dm], dtemperature (scalar tensor, d/dT with
  scale = exp(T)), d row_weight [B,Nq] or None."""
  B, Nq, X, Y = sim.shape
  Dm = fq.shape[-1]
  XY = X * Y
  dweight = None
  if row_weight is not None:
    rowdot = ops_bwd.sim_bwd_prepare_rows_(dsim, sim, clip, (row_weight * scale).contiguous())
    dtemp = rowdot.sum()
    # d sim / d w = sim / w; a zero weight (a masked point) receives none: the masked softmax
    # hands nothing back to it either
    dweight = torch.where(row_weight > 0, rowdot / row_weight.clamp(min=1e-38), torch.zeros_like(rowdot))
  else:
    coef = (scale / num_valid).to(torch.float32).contiguous()
    dtemp = ops_bwd.sim_bwd_prepare_(dsim, sim, clip, coef).sum()
  dfq = torch.empty_like(fq)
  dfm = torch.empty_like(fm)
  for b in range(B):
    g = dsim[b].reshape(1, 1, Nq, XY)
    dfq[b] = ops.conv2d(g, fm[b].reshape(1, 1, XY, Dm)).reshape(Nq, Dm)
    dfm[b] = ops_bwd.conv2d_wgrad(g, fq[b].reshape(1, 1, Nq, Dm).contiguous(),
                                  (1, 1, XY, Dm)).reshape(X, Y, Dm)
  if row_weight is not None:
    return dfq, dfm, dtemp.to(torch.float32), dweight
  return dfq, dfm, dtemp.to(torch.float32)


# ----------------------------------------------------------------------------
# conv / dense
# ----------------------------------------------------------------------------
# A node's KERNEL gradient (dw = im2col(x)^T dy) has no consumer inside the backward chain: it leaves
# on the device's second HIP stream, next to the node's data gradient + GroupNorm VJP on the main
# stream, and the node joins it before it returns (so every later reader -- the gradient accumulation,
# the bucketed reducer's hooks, an in-place reuse of dy by the next node -- sees it complete, and
# nothing it reads is freed or overwritten while it runs).  The backward pass is a chain of 60-200 us
# launches; the second queue fills their dispatch gaps and tail waves as a second batch does in
# inference (snap_amd/pipeline.py).  Same kernels, same bits (ops.Tuning.WGRAD_SIDE_STREAM).
@contextlib.contextmanager
def _kernel_grad_stream(ref, forks):
  if not (ops.tuning().WGRAD_SIDE_STREAM and ref.is_cuda):
    yield
    return
  main = torch.cuda.current_stream(ref.device)
  side = ops.side_stream(ref.device)
  if side.cuda_stream == main.cuda_stream:
    yield
    return
  side.wait_stream(main)            # x, dy and the statistics were produced on the main stream before here
  with torch.cuda.stream(side):
    yield
  forks.append((main, side))


def _join_kernel_grads(forks):
  for main, side in forks:
    main.wait_stream(side)
  forks.clear()


@_engine_scoped
class _FusedConv(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w, gamma, beta, bias, residual, up_prev, cfg):
    stride, padding, prologue, in_affine, relu, row_mask, cin, emit, fork = cfg
    gn = None
    mu = sc = rstd = None
    if prologue in _GN_MODES:
      mu, sc, rstd = ops.group_norm_stats(
          x, gamma.reshape(-1), relu_first=prologue == ops.PRO_RELU_GN, want_rstd=True
      )
      gn = (mu, sc, beta.reshape(-1))
    y = ops.conv2d(
        x, w, stride=stride, padding=padding, cin=cin, prologue=prologue, gn=gn,
        in_affine=in_affine, bias=bias, relu=relu, residual=residual, up_prev=up_prev,
        row_mask=row_mask, emit_gn_stats=emit,
    )
    ctx.cfg = cfg
    ctx.has = (bias is not None, residual is not None, up_prev is not None)
    ctx.save_for_backward(x, w, gamma, beta, mu, sc, rstd, y if relu else None, row_mask)
    if fork:
      # a second output that IS the input: the caller routes the other consumer of x (a residual
      # unit's shortcut) through it, so that both gradients of x arrive in this node's backward
      # and the sum is formed by the kernel that writes dx instead of by a separate add pass
      return y, x.view_as(x)
    return y

  @staticmethod
  def backward(ctx, dy, dalias=None):
    stride, padding, prologue, in_affine, relu, _, cin, _, fork = ctx.cfg
    x, w, gamma, beta, mu, sc, rstd, y, row_mask = ctx.saved_tensors
    has_bias, has_res, has_up = ctx.has
    dy = dy.contiguous()
    need = ctx.needs_input_grad
    dbias = None
    if relu or row_mask is not None:
      if has_bias and need[4]:      # the gate and the bias gradient (column sums of the gated dy) in one pass
        dy, dbias = ops_bwd.epilogue_bwd_colsum(dy, y, row_mask, relu=relu)
      else:
        dy = ops_bwd.epilogue_bwd(dy, y, row_mask, relu=relu)
    gn = (mu, sc, beta.reshape(-1)) if prologue in _GN_MODES else None
    dw = None
    forks = []
    if need[1]:
      with _kernel_grad_stream(dy, forks):
        dw = ops_bwd.conv2d_wgrad(x, dy, tuple(w.shape), stride=stride, padding=padding,
                                  prologue=prologue, gn=gn, in_affine=in_affine)
    if dbias is None and has_bias and need[4]:
      dbias = ops_bwd.colsum(dy)
    dres = dy if (has_res and need[5]) else None
    dup = ops_bwd.upsample2x_bwd(dy) if (has_up and need[6]) else None
    dx = dgamma = dbeta = None
    if need[0] or (prologue in _GN_MODES and (need[2] or need[3])):
      N, H, W, Cs = x.shape
      Cin = w.shape[2]
      gnb = None
      if prologue in _GN_MODES and stride == 1 and Cs == Cin and Cin % 4 == 0:
        gnb = (x, mu, rstd, gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous(), prologue)
      dz = conv_dgrad(dy, w, (N, H, W, Cin), stride, padding, gn_bwd_stats=gnb)
      if dz.shape[-1] > Cs:
        dz = dz[..., :Cs].contiguous()
      Cin = dz.shape[-1]                                       # zero-padded channel count
      if prologue in _GN_MODES:
        fused_add = dalias is not None and Cs == Cin
        dx, dgamma, dbeta = ops_bwd.group_norm_bwd(
            x, dz, mu, rstd, gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous(), prologue,
            add=dalias.contiguous() if fused_add else None,
            half=ops.precision() if ops.precision() in ops.HALF_MATH else None,
        )
        if fused_add:
          dalias = None
        dgamma = dgamma.reshape(gamma.shape)
        dbeta = dbeta.reshape(beta.shape)
      elif prologue == ops.PRO_RELU:
        xs = x if Cs == Cin else x[..., :Cin].contiguous()
        dx = ops_bwd.epilogue_bwd(dz, xs, None, relu=True)
      elif prologue == ops.PRO_AFFINE:
        dx = dz * in_affine[0]
      else:
        dx = dz
      if Cs != Cin:
        dx = F.pad(dx, (0, Cs - Cin))
    if dalias is not None:            # (no GroupNorm prologue to fold it into, or nothing else to add it to)
      dx = dalias if dx is None else dx + dalias
    _join_kernel_grads(forks)
    return dx, dw, dgamma, dbeta, dbias, dres, dup, None


@_engine_scoped
class _SharedPrologueConvPair(torch.autograd.Function):
  """Two convolutions behind ONE GroupNorm -> ReLU of the same tensor (a projection unit's conv1 and
  conv_proj, resnet.py:117-124 of the reference): the statistics are taken once, and in the backward
  the second data gradient is accumulated onto the first in the conv engine's epilogue, so that the
  GroupNorm VJP (linear in its incoming gradient) runs once over x instead of twice plus an add."""

  @staticmethod
  def forward(ctx, x, w1, w2, gamma, beta, cfg):
    stride2, emit1 = cfg
    mu, sc, rstd = ops.group_norm_stats(x, gamma.reshape(-1), want_rstd=True)
    gn = (mu, sc, beta.reshape(-1))
    y1 = ops.conv2d(x, w1, prologue=ops.PRO_GN_RELU, gn=gn, emit_gn_stats=emit1)
    y2 = ops.conv2d(x, w2, stride=stride2, prologue=ops.PRO_GN_RELU, gn=gn)
    ctx.stride2 = stride2
    ctx.save_for_backward(x, w1, w2, gamma, beta, mu, sc, rstd)
    return y1, y2

  @staticmethod
  def backward(ctx, dy1, dy2):
    x, w1, w2, gamma, beta, mu, sc, rstd = ctx.saved_tensors
    need = ctx.needs_input_grad
    pad0 = ((0, 0), (0, 0))
    gn = (mu, sc, beta.reshape(-1))
    dy1, dy2 = dy1.contiguous(), dy2.contiguous()
    dw1 = dw2 = None
    forks = []
    with _kernel_grad_stream(dy1, forks):
      if need[1]:
        dw1 = ops_bwd.conv2d_wgrad(x, dy1, tuple(w1.shape), prologue=ops.PRO_GN_RELU, gn=gn)
      if need[2]:
        dw2 = ops_bwd.conv2d_wgrad(x, dy2, tuple(w2.shape), stride=ctx.stride2, prologue=ops.PRO_GN_RELU, gn=gn)
    dx = dgamma = dbeta = None
    if need[0] or need[3] or need[4]:
      N, H, W, C = x.shape
      dz = conv_dgrad(dy1, w1, (N, H, W, C), 1, pad0)
      gnb = None
      if ctx.stride2 == 1 and C % 4 == 0:      # (the sum of both data gradients leaves the second launch's epilogue)
        gnb = (x, mu, rstd, gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous(), ops.PRO_GN_RELU)
      dz = conv_dgrad(dy2, w2, (N, H, W, C), ctx.stride2, pad0, accumulate=dz, accumulate_inplace=True,
                      gn_bwd_stats=gnb)
      dx, dgamma, dbeta = ops_bwd.group_norm_bwd(
          x, dz, mu, rstd, gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous(), ops.PRO_GN_RELU,
          half=ops.precision() if ops.precision() in ops.HALF_MATH else None)
      dgamma = dgamma.reshape(gamma.shape)
      dbeta = dbeta.reshape(beta.shape)
    _join_kernel_grads(forks)
    return dx, dw1, dw2, dgamma, dbeta, None


def conv2d_shared_gn(x, w1, w2, gn_params, *, stride2=1, emit_gn_stats=None):
  """``(conv(gn_relu(x), w1), conv(gn_relu(x), w2, stride2))`` for two 1 x 1 kernels behind one
  GroupNorm -> ReLU (whole 4-channel groups), differentiable as one node."""
  gamma, beta = gn_params
  return _SharedPrologueConvPair.apply(x, w1, w2, gamma, beta, (int(stride2), emit_gn_stats))


def conv2d(x, w, *, stride=1, padding=((0, 0), (0, 0)), cin=None, prologue=ops.PRO_NONE,
           gn_params=None, in_affine=(1.0, 0.0), bias=None, relu=False, residual=None,
           up_prev=None, row_mask=None, emit_gn_stats=None, fork_input=False):
  """Differentiable ``ops.conv2d``.  ``gn_params = (gamma, beta)`` for GN prologues
  (statistics are computed inside, so the VJP covers them).  ``fork_input``: returns ``(y, x')`` with
  ``x'`` an alias of ``x`` for its other consumer -- the gradient that comes back through ``x'`` is
  added by the kernel that writes this node's ``dx`` (one pass less over the tensor)."""
  gamma, beta = gn_params if gn_params is not None else (None, None)
  cfg = (stride, padding, prologue, tuple(in_affine), relu, row_mask, cin, emit_gn_stats, bool(fork_input))
  return _FusedConv.apply(x, w, gamma, beta, bias, residual, up_prev, cfg)


def dense(x, kernel, bias=None, *, cin=None, prologue=ops.PRO_NONE, relu=False, row_mask=None,
          residual=None):
  lead = x.shape[:-1]
  M = 1
  for s in lead:
    M *= int(s)
  y = conv2d(
      x.reshape(1, 1, M, x.shape[-1]), kernel.reshape(1, 1, *kernel.shape),
      cin=cin if cin is not None else kernel.shape[0], prologue=prologue, bias=bias, relu=relu,
      row_mask=row_mask,
      residual=None if residual is None else residual.reshape(1, 1, M, kernel.shape[1]),
  )
  return y.reshape(*lead, kernel.shape[1])


@_engine_scoped
class _SemanticEmbed(torch.autograd.Function):
  """Embedding lookups of the semantic rasters; table gradients = one-hot^T @ dy on the
  deterministic wgrad engine (exact f32), then the few rows are regrouped on the host."""

  @staticmethod
  def forward(ctx, rasters, table_road, table_other, idx_road, idx_other):
    ctx.idx = (tuple(idx_road), tuple(idx_other))
    ctx.save_for_backward(rasters)
    ctx.shapes = (tuple(table_road.shape), tuple(table_other.shape))
    return ops.semantic_embed(rasters, idx_road, idx_other, table_road, table_other)

  @staticmethod
  def backward(ctx, dy):
    (rasters,) = ctx.saved_tensors
    idx_road, idx_other = ctx.idx
    nr, no = len(idx_road), len(idx_other)
    E = ctx.shapes[0][1]
    onehot = ops.semantic_onehot(rasters, idx_road, idx_other)
    M, KP = onehot.shape
    W = dy.shape[-1]
    G = ops_bwd.conv2d_wgrad(onehot.reshape(1, 1, M, KP), dy.contiguous().reshape(1, 1, M, W),
                             (1, 1, KP, W), math='f32').reshape(KP, W)
    d_road = G[:nr, :E].contiguous()
    d_other = torch.zeros(ctx.shapes[1], dtype=G.dtype, device=G.device)
    for j in range(no):
      for b in (0, 1):      # class j with bit b read row j + b (semantic_raster_encoder.py:72-75)
        d_other[j + b] += G[nr + 2 * j + b, E * (1 + j):E * (2 + j)]
    return None, d_road, d_other, None, None


def semantic_embed(rasters, idx_road, idx_other, table_road, table_other):
  return _SemanticEmbed.apply(rasters, table_road, table_other, tuple(idx_road), tuple(idx_other))


# ----------------------------------------------------------------------------
# ViT pieces (vit_ops.hip / vit_bwd.hip)
# ----------------------------------------------------------------------------
@_engine_scoped
class _LayerNorm(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, gamma, beta, eps):
    ctx.eps = eps
    ctx.save_for_backward(x, gamma)
    return ops.layer_norm(x, gamma, beta, eps)

  @staticmethod
  def backward(ctx, dy):
    x, gamma = ctx.saved_tensors
    dx, dgamma, dbeta = ops_bwd.layer_norm_bwd(x, dy.contiguous(), gamma, ctx.eps)
    return dx, dgamma, dbeta, None


def layer_norm(x, gamma, beta, eps=1e-6):
  return _LayerNorm.apply(x.contiguous(), gamma, beta, eps)


@_engine_scoped
class _Gelu(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return ops.gelu(x)

  @staticmethod
  def backward(ctx, dy):
    (x,) = ctx.saved_tensors
    return ops_bwd.gelu_bwd(x, dy.contiguous())


def gelu(x):
  return _Gelu.apply(x.contiguous())


@_engine_scoped
class _Attention(torch.autograd.Function):

  @staticmethod
  def forward(ctx, qkv, scale):
    out, lse = ops.attention(qkv, scale, want_lse=True)
    ctx.scale = scale
    ctx.save_for_backward(qkv, out, lse)
    return out

  @staticmethod
  def backward(ctx, dout):
    qkv, out, lse = ctx.saved_tensors
    return ops_bwd.attention_bwd(qkv, out, dout.contiguous(), lse, ctx.scale), None


def attention(qkv, scale=None):
  return _Attention.apply(qkv.contiguous(), scale)


@_engine_scoped
class _MaskedRowsMLP(torch.autograd.Function):
  """ReLU MLP over the rows with mask != 0 only; masked rows of the output are zero.

  Same function as the dense MLP with ``row_mask`` on its last layer (the reference,
  streetview_encoder.py:281-283) -- the masked rows carry no gradient to the parameters
  or the input -- but every GEMM (forward, data gradient, kernel gradient) walks the
  device-side row list, so unobserved voxels cost nothing.
  """

  @staticmethod
  def forward(ctx, x, mask, relu_input, *wb):
    n = len(wb) // 2
    M = mask.numel()
    index, count = ops.compact_rows(mask)
    pro0 = ops.PRO_RELU if relu_input else ops.PRO_NONE
    acts = []
    h = x.reshape(M, x.shape[-1])
    # training-precision engines, two layers (the fusion / projection MLPs): the hidden activation --
    # and, in the backward pass, the gradient w.r.t. it -- exist ONLY in the engine's element type
    # (compact rows).  Every consumer rounds them to that type anyway, so nothing changes in the
    # arithmetic; they move at half the bytes and their GEMMs read them by LDS-DMA.
    half = (MASKED_MLP_HALF and ops.precision() in ops.HALF_MATH and n == 2 and x.shape[-1] % 4 == 0
            and wb[0].shape[0] >= 4 and wb[0].shape[1] % 8 == 0 and wb[2].shape[1] % 4 == 0)
    for i in range(n):
      W, b = wb[2 * i], wb[2 * i + 1]
      last = i + 1 == n
      h = ops.dense(h, W, b, cin=W.shape[0], prologue=pro0 if i == 0 else ops.PRO_NONE,
                    relu=not last, rows_in=index if i == 0 else None,
                    rows_out=index if last else None, row_count=count,
                    out_half=half and not last)
      if not last:
        acts.append(h)
    ops.fill_masked_rows_(h, mask)
    ctx.relu_input = relu_input
    ctx.half = half
    ctx.n = n
    ctx.xshape = x.shape
    ctx.save_for_backward(x, mask, index, count, *acts, *[wb[2 * i] for i in range(n)])
    return h.reshape(*x.shape[:-1], h.shape[-1])

  @staticmethod
  def backward(ctx, dy):
    n = ctx.n
    saved = ctx.saved_tensors
    x, mask, index, count = saved[:4]
    acts = saved[4:4 + n - 1]
    Ws = saved[4 + n - 1:]
    M = mask.numel()
    Cs = x.shape[-1]
    x2 = x.reshape(M, Cs)
    g = dy.contiguous().reshape(M, dy.shape[-1])
    grads = [None] * (2 * n)
    dx = None
    if ctx.half:
      return _MaskedRowsMLP._backward_half(ctx, x2, g, mask, index, count, acts[0], Ws)
    for i in reversed(range(n)):
      last = i + 1 == n
      W = Ws[i]
      cin, cout = W.shape
      db_fused = None
      if not last:                      # ReLU gate of this layer's output (compact rows)
        if ctx.needs_input_grad[4 + 2 * i]:
          g, db_fused = ops_bwd.epilogue_bwd_colsum(g, acts[i], None, relu=True, row_count=count)
        else:
          g = ops_bwd.epilogue_bwd(g, acts[i], None, relu=True)
      rows_dy = index if last else None
      inp = x2 if i == 0 else acts[i - 1]
      pro = ops.PRO_RELU if (i == 0 and ctx.relu_input) else ops.PRO_NONE
      if ctx.needs_input_grad[3 + 2 * i]:
        grads[2 * i] = ops_bwd.dense_wgrad_rows(inp, g.reshape(M, cout), cin, cout, prologue=pro,
                                                rows_z=index if i == 0 else None, rows_dy=rows_dy,
                                                row_count=count)
      if ctx.needs_input_grad[4 + 2 * i]:
        grads[2 * i + 1] = (db_fused if db_fused is not None
                            else ops_bwd.colsum(g, rows=rows_dy, row_count=count))
      if i > 0 or ctx.needs_input_grad[0]:
        width = Cs if i == 0 else cin               # d input row width (x keeps its padded stride)
        Wt = W.t()
        if width != cin:
          Wt = F.pad(Wt, (0, width - cin))
        Wt = Wt.contiguous()
        gi = ops.dense(g, Wt, None, cin=cout, rows_in=rows_dy,
                       rows_out=index if i == 0 else None, row_count=count)
        if i == 0:
          ops.fill_masked_rows_(gi, mask)
          if ctx.relu_input:
            gi = ops_bwd.epilogue_bwd(gi, x2, None, relu=True)
          dx = gi.reshape(ctx.xshape)
        g = gi
    return (dx, None, None, *grads)


def _masked_rows_mlp_backward_half(ctx, x2, g, mask, index, count, h0, Ws):
  """The two-layer backward with the hidden activation h0 and the gradient w.r.t. it in the engine's
  element type (see ``_MaskedRowsMLP.forward``): same GEMMs, same rounded operands."""
  M = mask.numel()
  Cs = x2.shape[-1]
  W0, W1 = Ws
  cin0, H = W0.shape
  D1 = W1.shape[1]
  need = ctx.needs_input_grad
  grads = [None] * 4
  forks = []
  if need[5]:      # dW1 = h0^T g  (Z: half, compact; dY: f32 through the row list)
    with _kernel_grad_stream(g, forks):
      grads[2] = ops_bwd.conv2d_wgrad(h0.reshape(1, 1, M, H), g.reshape(1, 1, M, D1), (1, 1, H, D1),
                                      rows_dy=index, row_count=count).reshape(H, D1)
  if need[6]:
    grads[3] = ops_bwd.colsum(g, rows=index, row_count=count)
  # d h0 (compact, half only), then the ReLU gate + the bias gradient of layer 0 in one pass
  g1 = ops.dense(g, W1.t().contiguous(), None, cin=D1, rows_in=index, row_count=count, out_half=True)
  split = ops_bwd.dense_wgrad_tail(cin0, Cs) if need[3] else None
  tail_row = None
  gi = None
  if split is not None and split[1] == 1:
    # a single channel above the 128-channel tiles (the fusion MLP's 257th input: the view score):
    # its kernel-gradient row leaves the gate pass as a weighted column sum -- and, where the input
    # gradient is wanted and the channel sits in the last (zero-padded) quad of the rows, so does its
    # data gradient: the GEMM below then writes 256 columns (two 128-column tiles) instead of 260 (three)
    dtail = None
    if need[0] and MASKED_MLP_DX_TAIL and H == 256 and split[0] + 4 == Cs:
      gi = torch.empty((M, Cs), dtype=torch.float32, device=x2.device)
      dtail = (W0[split[0]].contiguous(), gi)
    g1, db0, tail_row = ops_bwd.epilogue_bwd_colsum(g1, h0, None, relu=True, row_count=count,
                                                    wsum=(x2, split[0], index, ctx.relu_input), dtail=dtail)
  else:
    g1, db0 = ops_bwd.epilogue_bwd_colsum(g1, h0, None, relu=True, row_count=count)
  if need[4]:
    grads[1] = db0
  pro = ops.PRO_RELU if ctx.relu_input else ops.PRO_NONE
  if need[3]:      # dW0 = x^T g1  (Z: f32 through the row list; dY: half, compact)
    with _kernel_grad_stream(g, forks):
      grads[0] = ops_bwd.dense_wgrad_rows(x2, g1, cin0, H, prologue=pro, rows_z=index, rows_dy=None,
                                          row_count=count, tail_row=tail_row)
  dx = None
  if need[0]:
    if gi is not None:     # (columns split[0] .. Cs of the listed rows: written by the gate pass above)
      ops.dense(g1, W0[:split[0]].t().contiguous(), None, cin=H, rows_out=index, row_count=count, out=gi,
                out_stride=Cs)
    else:
      Wt = W0.t()
      if Cs != cin0:
        Wt = F.pad(Wt, (0, Cs - cin0))
      gi = ops.dense(g1, Wt.contiguous(), None, cin=H, rows_out=index, row_count=count)   # half in, f32 rows out
    ops.fill_masked_rows_(gi, mask)
    if ctx.relu_input:
      gi = ops_bwd.epilogue_bwd(gi, x2, None, relu=True)
    dx = gi.reshape(ctx.xshape)
  _join_kernel_grads(forks)
  return (dx, None, None, *grads)


_MaskedRowsMLP._backward_half = staticmethod(_masked_rows_mlp_backward_half)
MASKED_MLP_HALF = True      # (tests: False pins the f32-tensor formulation; same arithmetic)
MASKED_MLP_DX_TAIL = True   # (tests: False keeps the 257th channel's data gradient inside the GEMM)


def masked_rows_mlp(x, mask, relu_input, weights_and_biases):
  return _MaskedRowsMLP.apply(x, mask, relu_input, *weights_and_biases)


@_engine_scoped
class _WeightStd(torch.autograd.Function):

  @staticmethod
  def forward(ctx, w):
    ctx.save_for_backward(w)
    return ops.weight_standardize(w)

  @staticmethod
  def backward(ctx, dws):
    (w,) = ctx.saved_tensors
    return ops_bwd.weight_standardize_bwd(w, dws.contiguous())


def weight_standardize(w):
  return _WeightStd.apply(w)


@_engine_scoped
class _WeightStdMulti(torch.autograd.Function):
  """All StdConv kernels of an encoder: one launch forward, one backward."""

  @staticmethod
  def forward(ctx, *ws):
    ctx.save_for_backward(*ws)
    return tuple(ops.weight_standardize_multi(list(ws)))

  @staticmethod
  def backward(ctx, *dwss):
    ws = ctx.saved_tensors
    live = [i for i, g in enumerate(dwss) if g is not None]
    out = [None] * len(ws)
    if live:
      dws = ops.weight_standardize_bwd_multi([ws[i] for i in live],
                                             [dwss[i].contiguous() for i in live])
      for i, g in zip(live, dws):
        out[i] = g
    return tuple(out)


def weight_standardize_multi(ws):
  return _WeightStdMulti.apply(*ws)


@_engine_scoped
class _MaxPool(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return ops.max_pool_3x3s2(x)

  @staticmethod
  def backward(ctx, dy):
    (x,) = ctx.saved_tensors
    return ops_bwd.max_pool_3x3s2_bwd(x, dy.contiguous())


def max_pool_3x3s2(x):
  return _MaxPool.apply(x)


# ----------------------------------------------------------------------------
# lift / BEV
# ----------------------------------------------------------------------------
@_engine_scoped
class _LiftPool(torch.autograd.Function):

  @staticmethod
  def forward(ctx, f_images, cam, Rt, points, cfg):
    K, fisheye, fd, nb, dmm, mvd, opts, fwd_only = cfg
    pooled, valid = ops.lift_pool(f_images, cam, Rt, points, K=K, fisheye=fisheye, feature_dim=fd,
                                  num_bins=nb, depth_min_max=dmm, max_view_distance=mvd, **opts, **fwd_only)
    ctx.cfg = cfg
    ctx.save_for_backward(f_images, cam, Rt, points)
    ctx.mark_non_differentiable(valid)
    return pooled, valid

  @staticmethod
  def backward(ctx, dpooled, _dvalid):
    K, fisheye, fd, nb, dmm, mvd, opts, _ = ctx.cfg
    f_images, cam, Rt, points = ctx.saved_tensors
    df = ops_bwd.lift_pool_bwd(f_images, cam, Rt, points, dpooled.contiguous(), K=K,
                               fisheye=fisheye, feature_dim=fd, num_bins=nb, depth_min_max=dmm,
                               max_view_distance=mvd, **opts)
    return df, None, None, None, None


def lift_pool(f_images, cam, Rt, points, *, K, fisheye, feature_dim, num_bins, depth_min_max,
              max_view_distance=None, weighted=True, use_variance=True, add_minmax=False,
              grid_yz=None, valid_rows_only=False):
  """Differentiable ``ops.lift_pool`` (every option of pool_multiview_features).  ``grid_yz`` (the forward's
  traversal hint) and ``valid_rows_only`` (rows of voxels no view sees stay unwritten: for a consumer that
  reads the rows of valid voxels only, such as the masked fusion MLP) as in ``ops.lift_pool``; the VJP treats
  the gradient rows of unobserved voxels as zero either way."""
  opts = dict(weighted=bool(weighted), use_variance=bool(use_variance), add_minmax=bool(add_minmax))
  fwd_only = dict(grid_yz=None if grid_yz is None else tuple(grid_yz), valid_rows_only=bool(valid_rows_only))
  cfg = (K, fisheye, feature_dim, num_bins, tuple(depth_min_max), max_view_distance, opts, fwd_only)
  return _LiftPool.apply(f_images, cam, Rt, points, cfg)


@_engine_scoped
class _LiftObservations(torch.autograd.Function):
  """First pass of the depth_mlp fusion (streetview_encoder.py:263-267): the un-pooled
  observations.  Depth and viewing ray are geometry: only the feature channels carry gradient."""

  @staticmethod
  def forward(ctx, f_images, cam, Rt, points, cfg):
    K, fisheye, fd, mvd = cfg
    obs, feat, valid = ops.lift_observations(f_images, cam, Rt, points, K=K, fisheye=fisheye,
                                             feature_dim=fd, max_view_distance=mvd)
    ctx.cfg = cfg
    ctx.f_shape = tuple(f_images.shape)
    ctx.save_for_backward(cam, Rt, points)
    ctx.mark_non_differentiable(valid)
    return obs, feat, valid

  @staticmethod
  def backward(ctx, dobs, dfeat, _dvalid):
    K, fisheye, fd, mvd = ctx.cfg
    cam, Rt, points = ctx.saved_tensors
    d = (dobs[..., :fd] + dfeat).contiguous()
    df = ops_bwd.lift_observations_bwd(d, ctx.f_shape, cam, Rt, points, K=K, fisheye=fisheye,
                                       feature_dim=fd, max_view_distance=mvd)
    return df, None, None, None, None


def lift_observations(f_images, cam, Rt, points, *, K, fisheye, feature_dim, max_view_distance=None):
  return _LiftObservations.apply(f_images, cam, Rt, points, (K, fisheye, feature_dim, max_view_distance))


@_engine_scoped
class _LiftPoolObservations(torch.autograd.Function):
  """Second pass: pool_multiview_features of the corrected observations."""

  @staticmethod
  def forward(ctx, obs_feat, cam, Rt, points, cfg):
    f_shape, K, fisheye, fd, mvd, uv, mm = cfg
    pooled, valid = ops.lift_pool_observations(obs_feat, f_shape, cam, Rt, points, K=K, fisheye=fisheye,
                                               feature_dim=fd, max_view_distance=mvd, use_variance=uv,
                                               add_minmax=mm)
    ctx.cfg = cfg
    ctx.save_for_backward(obs_feat, cam, Rt, points)
    ctx.mark_non_differentiable(valid)
    return pooled, valid

  @staticmethod
  def backward(ctx, dpooled, _dvalid):
    f_shape, K, fisheye, fd, mvd, uv, mm = ctx.cfg
    obs_feat, cam, Rt, points = ctx.saved_tensors
    dobs = ops_bwd.lift_pool_observations_bwd(obs_feat, f_shape, cam, Rt, points, dpooled.contiguous(),
                                              K=K, fisheye=fisheye, feature_dim=fd, max_view_distance=mvd,
                                              use_variance=uv, add_minmax=mm)
    return dobs, None, None, None, None


def lift_pool_observations(obs_feat, f_shape, cam, Rt, points, *, K, fisheye, feature_dim,
                           max_view_distance=None, use_variance=True, add_minmax=False):
  cfg = (tuple(f_shape), K, fisheye, feature_dim, max_view_distance, bool(use_variance), bool(add_minmax))
  return _LiftPoolObservations.apply(obs_feat.contiguous(), cam, Rt, points, cfg)


@_engine_scoped
class _VerticalPool(torch.autograd.Function):

  @staticmethod
  def forward(ctx, vol, valid, pooling):
    # (max: the kernel also records where every maximum sits; the VJP then writes dvol without reading vol)
    plane, pvalid, arg = ops.vertical_pool(vol, valid, pooling, want_arg=True)
    ctx.pooling = pooling
    ctx.has_arg = arg is not None
    ctx.save_for_backward(vol, valid, *(arg or ()))
    ctx.mark_non_differentiable(pvalid)
    return plane, pvalid

  @staticmethod
  def backward(ctx, dplane, _dv):
    vol, valid, *arg = ctx.saved_tensors
    return ops_bwd.vertical_pool_bwd(vol, valid, dplane.contiguous(), ctx.pooling,
                                     arg=tuple(arg) if ctx.has_arg else None), None, None


def vertical_pool(vol, valid, pooling='max'):
  return _VerticalPool.apply(vol, valid, pooling)


@_engine_scoped
class _VerticalPoolConf(torch.autograd.Function):
  """'softmax' / 'weighted' VerticalPooling.  The scores / weights outputs are diagnostic
  (pred['scores'], pred['weights']; nothing in the localisation loss reads them) and are
  returned without gradient."""

  @staticmethod
  def forward(ctx, vol, valid, w, bias, log_sig):
    plane, pvalid, scores, weights = ops.vertical_pool_conf(vol, valid, w.reshape(-1).contiguous(),
                                                            bias, log_sig)
    ctx.log_sig = log_sig
    ctx.wshape = w.shape
    ctx.save_for_backward(vol, valid, w, bias, weights)
    ctx.mark_non_differentiable(pvalid, scores, weights)
    return plane, pvalid, scores, weights

  @staticmethod
  def backward(ctx, dplane, _dv, _ds, _dw):
    vol, valid, w, bias, weights = ctx.saved_tensors
    dvol, dw, db = ops_bwd.vertical_pool_conf_bwd(
        vol, valid, w.reshape(-1).contiguous(), bias, weights, dplane.contiguous(), ctx.log_sig)
    return dvol, None, dw.reshape(ctx.wshape), db, None


def vertical_pool_conf(vol, valid, w, bias, log_sigmoid_scores):
  return _VerticalPoolConf.apply(vol, valid, w, bias, log_sigmoid_scores)


@_engine_scoped
class _PlaneFuseMatch(torch.autograd.Function):

  @staticmethod
  def forward(ctx, Wm, bm, cfg, *planes):
    valids, pooling, normalize, eps = cfg
    fused, fvalid, matching = ops.plane_fuse_match(
        list(planes), list(valids), pooling, Wm, bm, normalize=normalize, eps=eps, want_fused=True
    )
    ctx.cfg = cfg
    ctx.n = len(planes)
    ctx.save_for_backward(Wm, bm, fused, *planes)
    ctx.mark_non_differentiable(fvalid)
    if matching is None:
      matching = fused.new_zeros(())
    return fused, fvalid, matching

  @staticmethod
  def backward(ctx, dfused, _dvalid, dmatching):
    valids, pooling, normalize, eps = ctx.cfg
    Wm, bm, fused = ctx.saved_tensors[:3]
    planes = list(ctx.saved_tensors[3:])
    has_match = Wm is not None
    dplanes, dy = ops_bwd.plane_fuse_match_bwd(
        planes, list(valids), pooling, Wm, bm, normalize, eps,
        dmatching.contiguous() if has_match else None,
        dfused.contiguous() if dfused is not None else None,
    )
    dW = db = None
    if has_match:
      D, Dm = Wm.shape
      M = fused.numel() // D
      dW = ops_bwd.conv2d_wgrad(fused.reshape(1, 1, M, D), dy.reshape(1, 1, M, Dm), (1, 1, D, Dm))
      dW = dW.reshape(D, Dm)
      db = ops_bwd.colsum(dy)
    return (dW, db, None, *dplanes)


def plane_fuse_match(planes, valids, pooling='max', Wm=None, bm=None, normalize=True, eps=1e-5,
                     want_fused=True):
  cfg = (tuple(valids), pooling, normalize, eps)
  fused, fvalid, matching = _PlaneFuseMatch.apply(Wm, bm, cfg, *planes)
  return fused, fvalid, (matching if Wm is not None else None)


# ----------------------------------------------------------------------------
# pose head
# ----------------------------------------------------------------------------
@_engine_scoped
class _ConfidenceHead(torch.autograd.Function):
  """bev_confidence = where(valid, log_sigmoid(Dense(1)(features)), 0)  (bev_mapper.py:154-157,292-295)."""

  @staticmethod
  def forward(ctx, features, kernel, bias, valid):
    w = kernel.reshape(-1).contiguous()
    b = bias.reshape(-1)[:1].contiguous()
    ctx.save_for_backward(features, w, b, valid)
    ctx.kshape, ctx.bshape = tuple(kernel.shape), tuple(bias.shape)
    return ops.confidence_head(features, valid, w, b)

  @staticmethod
  def backward(ctx, dconf):
    features, w, b, valid = ctx.saved_tensors
    df, dw, db = ops_bwd.confidence_head_bwd(features, valid, w, b, dconf.contiguous())
    return df, dw.reshape(ctx.kshape), db.reshape(ctx.bshape), None


def confidence_head(features, valid, kernel, bias):
  return _ConfidenceHead.apply(features.contiguous(), kernel, bias, valid)


@_engine_scoped
class _MaskedSoftmaxRows(torch.autograd.Function):
  """layers.masked_softmax over the last axis (layers.py:38-43) of x [B, N] (+ its CDF, no gradient)."""

  @staticmethod
  def forward(ctx, x, mask):
    w, cdf = ops.masked_softmax_rows(x, mask)
    ctx.save_for_backward(w)
    ctx.mark_non_differentiable(cdf)
    return w, cdf

  @staticmethod
  def backward(ctx, dw, _dcdf):
    (w,) = ctx.saved_tensors
    return ops_bwd.masked_softmax_rows_bwd(w, dw.contiguous()), None


def masked_softmax_rows(x, mask):
  return _MaskedSoftmaxRows.apply(x.contiguous(), mask.contiguous())


@_engine_scoped
class _SimSoftmaxWeighted(torch.autograd.Function):
  """sim = relu(fq . fm) * exp(T) * weights[b, n]  (add_confidence_query, bev_localizer.py:165-168)."""

  @staticmethod
  def forward(ctx, fq, fm, temperature, weights, num_valid, clip, want_prob):
    scale = 1.0 if temperature is None else ops.host_exp(temperature)
    sim, stats, prob, _ = ops.sim_softmax(fq, fm, scale, clip, num_valid, want_prob=want_prob,
                                          row_weight=weights)
    ctx.scale, ctx.clip = scale, clip
    ctx.save_for_backward(fq, fm, sim, num_valid, weights)
    ctx.has_t = temperature is not None
    ctx.mark_non_differentiable(stats)
    if prob is None:
      prob = sim.new_zeros(())
    ctx.mark_non_differentiable(prob)
    return sim, stats, prob

  @staticmethod
  def backward(ctx, dsim, _ds, _dp):
    fq, fm, sim, num_valid, weights = ctx.saved_tensors
    dfq, dfm, dtemp, dw = similarity_bwd(_own(dsim), sim, fq, fm, ctx.scale, ctx.clip,
                                         num_valid, row_weight=weights)
    return dfq, dfm, (dtemp if ctx.has_t else None), dw, None, None, None


def sim_softmax_weighted(fq, fm, temperature, weights, clip_negative, num_valid, want_prob=False):
  """Returns (sim, chunk_stats, prob or None, scale) with per-point weights (differentiable)."""
  sim, stats, prob = _SimSoftmaxWeighted.apply(fq, fm, temperature, weights.contiguous(), num_valid,
                                               clip_negative, want_prob)
  scale = 1.0 if temperature is None else ops.host_exp(temperature)
  return sim, stats, (prob if want_prob else None), scale


@_engine_scoped
class _SimSoftmax(torch.autograd.Function):

  @staticmethod
  def forward(ctx, fq, fm, temperature, num_valid, clip, want_prob):
    scale = 1.0 if temperature is None else ops.host_exp(temperature)
    sim, stats, prob, _ = ops.sim_softmax(fq, fm, scale, clip, num_valid, want_prob=want_prob)
    ctx.scale, ctx.clip = scale, clip
    ctx.save_for_backward(fq, fm, sim, num_valid)
    ctx.has_t = temperature is not None
    ctx.mark_non_differentiable(stats)
    if prob is None:
      prob = sim.new_zeros(())
    ctx.mark_non_differentiable(prob)
    return sim, stats, prob

  @staticmethod
  def backward(ctx, dsim, _ds, _dp):
    fq, fm, sim, num_valid = ctx.saved_tensors
    dfq, dfm, dtemp = similarity_bwd(_own(dsim), sim, fq, fm, ctx.scale, ctx.clip,
                                     num_valid)
    return dfq, dfm, (dtemp if ctx.has_t else None), None, None, None


def sim_softmax(fq, fm, temperature, clip_negative, num_valid, want_prob=False):
  """Returns (sim, chunk_stats, prob or None, scale)."""
  sim, stats, prob = _SimSoftmax.apply(fq, fm, temperature, num_valid, clip_negative, want_prob)
  scale = 1.0 if temperature is None else ops.host_exp(temperature)
  return sim, stats, (prob if want_prob else None), scale


@_engine_scoped
class _PoseScore(torch.autograd.Function):

  @staticmethod
  def forward(ctx, sim, poses, q_xy, valid_q, map_valid, cell, mask_oob):
    ctx.cfg = (cell, mask_oob, tuple(sim.shape))
    ctx.save_for_backward(poses, q_xy, valid_q, map_valid)
    return ops.pose_score(sim, poses, q_xy, valid_q, map_valid, cell, mask_oob=mask_oob)

  @staticmethod
  def backward(ctx, dscores):
    cell, mask_oob, shape = ctx.cfg
    poses, q_xy, valid_q, map_valid = ctx.saved_tensors
    dsim = ops_bwd.pose_score_bwd(dscores.contiguous(), poses, q_xy, valid_q, map_valid, shape,
                                  cell, mask_oob=mask_oob)
    return dsim, None, None, None, None, None, None


def pose_score(sim, poses, q_xy, valid_q, map_valid, cell_size, mask_oob=False):
  return _PoseScore.apply(sim, poses, q_xy, valid_q, map_valid, cell_size, mask_oob)
