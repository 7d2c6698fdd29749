"""Host-side operator wrappers over the C ABI of libsnap_hip.so.

Every function validates dtype / device / contiguity, allocates the outputs with
torch (device memory plumbing only) and launches the HIP kernel on torch's
current stream.  There is no CPU or PyTorch fallback: tensors must live on a
ROCm device and the shared library must be built, otherwise these raise.
"""
import contextlib
import sys
import ctypes
import threading
import weakref
import os

import numpy as np
import torch

from snap_amd import _lib

PRO_NONE, PRO_AFFINE, PRO_GN_RELU, PRO_RELU_GN, PRO_RELU = 0, 1, 2, 3, 4
EPI_BIAS, EPI_RELU, EPI_RESIDUAL, EPI_UPSAMPLE2X_ADD, EPI_ROWMASK, EPI_GELU = 1, 2, 4, 8, 16, 32
POOLING = {'max': 0, 'sum': 1, 'mean': 2}
SIM_CHUNK = 64



# ----------------------------------------------------------------------------------------------------
# Tuning: every tuning / test switch of the kernel wrappers in ONE object.  None of them is needed to
# run the models and none changes a result beyond summation order; tools and tests flip them to A/B
# kernels.  The object in force is per thread and scoped (``tuning_scope(...)`` /
# ``engine_scope(engine, tuning=...)``); outside any scope it is the process default, of which the
# module attributes of the same names (``ops.CONV_TILE = '64x64'``) are thin aliases -- reading
# ``ops.CONV_TILE`` anywhere gives the value in force in the calling thread.
#   OVERLAP_AERIAL     the aerial encoder on a second HIP stream next to the StreetView encoder
#   POOLED_SPLIT       the lift hands `pooled` to the fused MLP / pool kernel pre-split (LDS-DMA A operand)
#   CLASS_ROWS         ... classed by observation count (single-observation rows carry no variance slabs)
#   LIFT_IN_CONSUMER   ... and class-1 rows are never written: a 32-byte tap record, the four image taps
#                      blended inside the fused MLP / pool kernel (the lift inside the consumer)
#   MLP_GATHER_XCD_GROUP  runs of 128-row tiles of the gathered class per XCD (its L2 keeps their taps)
#   NATIVE_GLUE        image padding / voxel-centre grid as native passes instead of torch fill + copies
#   USE_PRESPLIT       bottleneck 3x3 / closing 1x1 convs read their input through gn_norm_split (off:
#                      the pre-split convs are 10-25 percent faster, the extra pass costs what they gain)
#   CONV_TILE / CONV_BK / CONV_NO_HALO / CONV_NO_RS / CONV_RS_NSPLIT / CONV_RS_FORCE / CONV_NO_WS /
#   CONV_NO_PLAIN / CONV_RAW_RING   forced tile ('128x128' | '128x64' | '64x128' | '64x64'), f32 K-slab
#                      depth, and which body a conv launch takes (tests pin every body against the others)
#   SPLITK_STATS / USE_SPLITK / USE_FUSED_GN_STATS / GN_STATS_BOTH   GroupNorm statistics from the conv
#                      epilogues / split-K reduce passes vs the stand-alone kernels
#   MLP_POOL_NO_RING / MLP_POOL_WIDE   fused MLP / pool kernel variants (x_split = 5 / 3)
#   USE_PRESPLIT_VOTING / FUSED_TEMPLATE_PACK / PS_RES_INIT / PS_TILE   pre-split GEMM engine (direct-form voting)
#   SIM_GENERAL_KERNEL  tests: pin the general similarity kernel (the full-chunk kernel is bit-identical)
#   LATTICE_WINDOW     the refinement lattice scored from one window per point (pose_score_window; same bits)
#   WGRAD_SIDE_STREAM  backward: a node's kernel-gradient launches run on the second HIP stream next to its
#                      data-gradient / GroupNorm-VJP chain and are joined before the node returns (same bits)
# ----------------------------------------------------------------------------------------------------
_TUNING_DEFAULTS = {
    'OVERLAP_AERIAL': True,
    'POOLED_SPLIT': True,
    'CLASS_ROWS': True,
    'LIFT_IN_CONSUMER': True,
    'MLP_GATHER_XCD_GROUP': 8,
    'NATIVE_GLUE': True,
    'USE_PRESPLIT': False,
    'CONV_TILE': None,
    'CONV_BK': None,
    'CONV_NO_HALO': False,
    'CONV_NO_RS': False,
    'CONV_RS_NSPLIT': 0,
    'CONV_RS_FORCE': False,
    'CONV_NO_WS': False,
    'SPLITK_STATS': True,
    'MLP_POOL_NO_RING': False,
    'MLP_POOL_WIDE': False,
    'CONV_NO_PLAIN': False,
    'CONV_RAW_RING': False,
    'USE_PRESPLIT_VOTING': True,
    'PS_RES_INIT': True,
    'PS_TILE': 0,
    'BF16_PS': True,          # conv2d(bf16_ring=True) launches take the one-part pre-split engine (conv_ps.hip NS = 1); False: the x_half engine
    'USE_FUSED_GN_STATS': True,
    'GN_STATS_BOTH': True,
    'USE_SPLITK': True,
    'FUSED_TEMPLATE_PACK': True,
    'SIM_GENERAL_KERNEL': False,
    'LATTICE_WINDOW': True,
    'WGRAD_SIDE_STREAM': True,
}


class Tuning:
  """The switches listed above as one value object (``Tuning(CONV_TILE='64x64')``; unknown names raise)."""
  __slots__ = tuple(_TUNING_DEFAULTS)

  def __init__(self, **kw):
    for k, v in _TUNING_DEFAULTS.items():
      object.__setattr__(self, k, kw.pop(k, v))
    if kw:
      raise TypeError(f'Tuning: unknown switch(es) {sorted(kw)}')

  def replace(self, **kw):
    vals = {k: getattr(self, k) for k in _TUNING_DEFAULTS}
    for k in kw:
      if k not in vals:
        raise TypeError(f'Tuning: unknown switch {k!r}')
    vals.update(kw)
    return Tuning(**vals)

  def __repr__(self):
    diff = {k: getattr(self, k) for k, v in _TUNING_DEFAULTS.items() if getattr(self, k) != v}
    return 'Tuning(' + ', '.join(f'{k}={v!r}' for k, v in diff.items()) + ')'


_DEFAULT_TUNING = Tuning()
_TUNING_TLS = threading.local()


def tuning():
  """The ``Tuning`` in force in this thread: the innermost scope's, else the process default."""
  t = getattr(_TUNING_TLS, 'tuning', None)
  return _DEFAULT_TUNING if t is None else t


@contextlib.contextmanager
def tuning_scope(base=None, **overrides):
  """Run the enclosed ops under ``base`` (default: the tuning in force) with ``overrides`` applied.
  Per thread, re-entrant; autograd nodes carry it into the backward thread with the engine."""
  prev = getattr(_TUNING_TLS, 'tuning', None)
  _TUNING_TLS.tuning = (base if base is not None else tuning()).replace(**overrides)
  try:
    yield _TUNING_TLS.tuning
  finally:
    _TUNING_TLS.tuning = prev


# the raw handle of torch's current stream on the current device: one C call (torch.cuda.current_stream() builds a
# Stream object through three layers of device-index resolution: ~9 us per launch, 3-13 ms of a step's host time --
# a fifth of the C5 step's, which the host bounds)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
  if _raw_stream is not None and _raw_device is not None:
    return ctypes.c_void_p(_raw_stream(_raw_device()))
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# A second HIP stream for work that does not depend on the main chain (the aerial encoder runs
# next to the StreetView encoder: its deep stages are launches of 19-73 workgroups on a 256-CU
# part).  SNAP_OVERLAP_AERIAL=0 keeps everything on one stream.
# the lift hands `pooled` to the fused MLP / pool kernel pre-split (LDS-DMA A operand); 0: as f32 rows
# ... and classed by their number of observations (single-observation rows carry no variance slabs); 0: off
# ... and the class-1 rows (one observation) are never written: the lift leaves a 32-byte tap record and the
# fused MLP / pool kernel blends the four image taps itself (the lift inside the consumer); 0: rows through HBM
# image padding and the voxel-centre grid as one native pass each instead of torch fill + strided copies; 0: torch
_SIDE_STREAMS = {}


def side_stream(device=None):
  """The second stream of ``device`` (one per GPU of the process; default: the current device)."""
  idx = torch.cuda.current_device() if device is None else torch.device(device).index
  if idx is None:
    idx = torch.cuda.current_device()
  s = _SIDE_STREAMS.get(idx)
  if s is None:
    s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
  return s


class KernelProfiler:
  """HIP-event timing of individual kernel launches on torch's current stream.

  ``bench.py`` installs one for a single step of the timed region to obtain the
  per-kernel durations behind the ``roofline`` object (events are recorded on the
  stream the kernels are launched on).  ``flops`` / ``nbytes`` are the ALGORITHMIC
  figures of DESIGN.md, not measured traffic.
  """

  def __init__(self):
    self.records = {}

  def region(self, name, flops=0.0, nbytes=0.0, tag=None):
    return _Region(self, name, flops, nbytes, tag)

  def launches(self, name):
    """Per-launch (tag, ms, flops, bytes) of one kernel family (after a sync)."""
    torch.cuda.synchronize()
    return [
        (r[4], r[0].elapsed_time(r[1]), _num(r[2]), _num(r[3]))
        for r in self.records.get(name, [])
    ]

  def summary(self):
    torch.cuda.synchronize()
    out = {}
    for name, recs in self.records.items():
      ms = sum(r[0].elapsed_time(r[1]) for r in recs)
      out[name] = dict(
          launches=len(recs), ms=ms, flops=sum(_num(r[2]) for r in recs),
          bytes=sum(_num(r[3]) for r in recs),
      )
    return out


def _num(v):
  """flops / bytes may be callables resolved after the sync (device-side row counts)."""
  return float(v()) if callable(v) else v


class _Region:

  def __init__(self, prof, name, flops, nbytes, tag=None):
    self.prof, self.name, self.flops, self.nbytes, self.tag = prof, name, flops, nbytes, tag

  def __enter__(self):
    self.start = torch.cuda.Event(enable_timing=True)
    self.start.record()

  def __exit__(self, *exc):
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    self.prof.records.setdefault(self.name, []).append(
        (self.start, end, self.flops, self.nbytes, self.tag)
    )


class _NoRegion:

  def __enter__(self):
    pass

  def __exit__(self, *exc):
    pass


_PROFILER = None
_NO_REGION = _NoRegion()


def set_profiler(prof):
  """Install (or remove with None) a KernelProfiler; returns the previous one."""
  global _PROFILER
  old, _PROFILER = _PROFILER, prof
  return old


def _region(name, flops=0.0, nbytes=0.0, tag=None):
  if _PROFILER is None:
    return _NO_REGION
  return _PROFILER.region(name, flops, nbytes, tag() if callable(tag) else tag)


def _p(t):
  if t is None:
    return None
  return ctypes.c_void_p(t.data_ptr())


def _pv(t):
  return None if t is None else t.data_ptr()


# Small host tables (per-launch item lists of the multi-tensor kernels) go to the device through a
# ring of PINNED staging buffers with a truly asynchronous copy: a pageable-source copy stalls the
# host until the runtime has staged it and leaves the GPU idle in front of the copy (~4 ms per
# training step over its 5-6 tables, rocprofv3 kernel trace).  A slot is reused only after the
# event recorded behind its last copy has completed.
_PIN_RING = []
_PIN_NEXT = 0


_PIN_LOCK = threading.Lock()


def upload_table(items, device):
  """numpy structured / plain array -> uint8 device tensor holding its bytes."""
  raw = np.ascontiguousarray(items).view(np.uint8).reshape(-1)
  device = torch.device(device)
  if device.type != 'cuda':
    return torch.from_numpy(raw.copy()).to(device)
  global _PIN_NEXT
  n = raw.size
  # one lock around slot selection, fill and the copy's issue: autograd hook / backward threads
  # (the overlapped gradient reducer) call this next to the main thread
  with _PIN_LOCK:
    if not _PIN_RING:          # the whole ring at once (a pinned allocation takes ~0.2 s: never inside a step)
      _PIN_RING.extend([torch.empty(1 << 17, dtype=torch.uint8, pin_memory=True), None] for _ in range(16))
    slot = _PIN_RING[_PIN_NEXT % 16]
    _PIN_NEXT += 1
    if slot[1] is not None:
      slot[1].synchronize()
    if slot[0].numel() < n:
      slot[0] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    slot[0][:n].numpy()[:] = raw
    out = slot[0][:n].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slot[1] = ev
  return out


# exp(temperature) is a KERNEL ARGUMENT of the similarity / sampling kernels (a host scalar).  Reading
# it back where it is needed drains the stream in the middle of a step (everything after it is then
# dispatched into an empty queue: ~3 ms of launch-bound gaps per training step).  ``prefetch_exp``
# queues the 4-byte read at the START of the apply; ``host_exp`` then only waits for that copy.
_HOST_SCALARS = {}


def prefetch_exp(t):
  if t is None or not t.is_cuda:
    return
  # one pinned scalar per temperature tensor, reused apply after apply: a read-back still in
  # flight into it was consumed by ``host_exp`` -- or is superseded by this one -- once its event has passed
  prev = _HOST_SCALARS.get(id(t))
  if prev is not None and prev[0]() is t:
    host = prev[2]
    prev[3].synchronize()
  else:
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
  host.copy_(torch.exp(t.detach().to(torch.float32)).reshape(1), non_blocking=True)
  ev = torch.cuda.Event()
  ev.record()
  if len(_HOST_SCALARS) > 64:
    _HOST_SCALARS.clear()
  _HOST_SCALARS[id(t)] = (weakref.ref(t), t._version, host, ev)


def host_exp(t):
  """float(exp(t)) -- the prefetched value when ``prefetch_exp(t)`` ran for this version of ``t``
  (same torch kernel, same bits), else a blocking read."""
  hit = _HOST_SCALARS.get(id(t))
  if hit is not None and hit[0]() is t and hit[1] == t._version:
    hit[3].synchronize()
    return float(hit[2][0])
  return float(torch.exp(t.detach().to(torch.float32)))


def _chk(t, dtype, name):
  if not isinstance(t, torch.Tensor):
    raise TypeError(f'{name}: expected a torch.Tensor, got {type(t)}')
  if not t.is_cuda:
    raise RuntimeError(
        f'{name}: tensor is on {t.device}; snap_amd ops run only on a ROCm GPU '
        '(no CPU fallback).'
    )
  if t.dtype != dtype:
    raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
  if not t.is_contiguous():
    raise ValueError(f'{name}: tensor must be contiguous')
  return t


def _f32(t, name):
  return _chk(t, torch.float32, name)


def _mask(t, name):
  if t.dtype == torch.bool:
    return _chk(t, torch.bool, name)
  return _chk(t, torch.uint8, name)


# ----------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------
class PreSplit:
  """An activation [N, H, W, C] (C % 16 == 0) already normalised and split into two bf16 parts:
  ``data`` = [N*H*W][C/16][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] (the bytes of the f32 tensor,
  re-arranged per 16-channel block).  ``conv2d`` multiplies it on the pre-split engine
  (conv_ps.hip: both operands by LDS-DMA, no normalise / split work in the K loop)."""

  def __init__(self, data, shape):
    self.data = data
    self.shape = tuple(shape)
    self.device = data.device

  def numel(self):
    return int(np.prod(self.shape))


# USE_PRESPLIT: the 3x3 and the closing 1x1 convolution of a bottleneck unit read their input
# through ``gn_norm_split`` (one pass that also replaces the statistics finalize launch) on the
# 'bf16x3' engine.  Measured at C2 (tools/unit_bench.py, profiles/r03_unit_bench.json): the
# pre-split convolutions are 10-25 % faster than the fused-prologue ones, the extra pass over the
# activation costs what they gain -- off by default; the engine's own user is the exhaustive voting.
# Tests / tuning tools: force the conv engines' output tile ('128x128' | '128x64' | '64x128' |
# '64x64'; travels as SnapConvDesc.tile_hint), the f32 engine's K-slab depth (16 | 32) and the
# im2col body for every 3x3 convolution of the split engine.  None / False = the engines' choice.


def gn_norm_split(y, gamma, beta, *, groups=32, eps=1e-5, want_stats=False):
  """relu(GroupNorm(y)) of a conv output that carries its fused partial sums, written once in the
  pre-split format -> ``PreSplit`` (or None where the pre-split engine does not apply: another
  engine, no fused statistics, C % 16 != 0)."""
  fused = getattr(y, '_snap_gn_partial', None)
  N, H, W, C = y.shape
  if (precision() != 'bf16x3' or fused is None or fused[2] or groups != 32
      or C % 16 or C > 2048 or not tuning().USE_FUSED_GN_STATS):
    return None
  lib = _lib.load()
  _f32(y, 'y'); _f32(gamma, 'gamma'); _f32(beta, 'beta')
  partial, tile_rows, _ = fused
  out = torch.empty(y.numel() * 2, dtype=torch.bfloat16, device=y.device)
  mu = sc = None
  if want_stats:
    mu = torch.empty((N, C), dtype=torch.float32, device=y.device)
    sc = torch.empty((N, C), dtype=torch.float32, device=y.device)
  with _region('gn_norm_split', 0.0, 8.0 * y.numel()):
    st = lib.snap_gn_norm_split_f32(_p(y), _p(partial), N, H * W, C, groups, eps, tile_rows,
                                    _p(gamma), _p(beta), _p(out), _p(mu), _p(sc), _stream())
  _lib.check(st, 'snap_gn_norm_split_f32')
  ps = PreSplit(out, (N, H, W, C))
  if want_stats:
    ps.stats = (mu, sc)
  return ps


def presplit(x):
  """The plain two-part split of x [..., C] f32 (C % 16 == 0) -> ``PreSplit``."""
  lib = _lib.load()
  _f32(x, 'x')
  C = x.shape[-1]
  rows = x.numel() // C
  out = torch.empty(x.numel() * 2, dtype=torch.bfloat16, device=x.device)
  with _region('presplit', 0.0, 8.0 * x.numel()):
    st = lib.snap_presplit_f32(_p(x), rows, C, _p(out), _stream())
  _lib.check(st, 'snap_presplit_f32')
  shape = tuple(x.shape) if x.dim() == 4 else (1,) * (4 - x.dim()) + tuple(x.shape)
  return PreSplit(out, shape)


def _stationary_mode():
  """SnapConvDesc.tile_hint // 1 000 000: which 1x1 bodies of the split engine may run."""
  if tuning().CONV_NO_RS:
    return 1
  if tuning().CONV_RS_FORCE:
    return 4 if tuning().CONV_NO_WS else 2
  return 3 if tuning().CONV_NO_WS else 0


class PackedWeights:
  """A conv kernel that exists ONLY as the split engine's two-part weight image (``data``: bf16,
  the layout of snap_conv2d_pack_weights_split_bf16 for ``shape`` = (KH, KW, Cin, Cout)) -- e.g. the
  shift-stacked template bank of the exhaustive voting, written directly in that form by
  ``pack_stacked_templates_split``.  ``conv2d`` takes it with a ``PreSplit`` input."""

  def __init__(self, data, shape):
    self.data = data
    self.shape = tuple(int(v) for v in shape)

  def numel(self):
    return int(np.prod(self.shape))


def pack_stacked_templates_split(templates, S):
  """templates [R, H, W, D] -> ``PackedWeights`` of the shift-stacked bank [H+S-1, W+S-1, D, R S^2]
  (``stack_templates`` + ``pack_weights_split_bf16(., 2)`` in one pass; the f32 bank is never built)."""
  lib = _lib.load()
  _f32(templates, 'templates')
  R, H, W, D = templates.shape
  KH, KW, RS = H + S - 1, W + S - 1, R * S * S
  nbytes = lib.snap_conv2d_packed_weights_split_bytes(KH * KW, D, RS, 2)
  if nbytes == 0:
    return None
  out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=templates.device)
  with _region('pack_stacked_templates', 0.0, 4.0 * templates.numel() + float(nbytes)):
    st = lib.snap_pack_stacked_templates_split_bf16(_p(templates), H, W, D, R, S, _p(out), nbytes, _stream())
  _lib.check(st, 'snap_pack_stacked_templates_split_bf16')
  return PackedWeights(out, (KH, KW, D, RS))


def conv2d_presplit_supported(x_shape, w_shape, stride=1, padding=((0, 0), (0, 0))):
  """True when ``conv2d(presplit(x), w, ...)`` has an engine for the shape: the pre-split engine's
  own limits (``snap_conv2d_presplit_supported``) and a two-part weight image within the split
  engine's 32-bit offsets.  Callers with a plain-input alternative ask before they pre-split."""
  lib = _lib.load()
  N, H, W, Cs = x_shape
  KH, KW, Cin, Cout = w_shape
  (pt, pb), (pl, pr) = padding
  if Cs != Cin or Cin % 16:
    return False
  Ho = (H + pt + pb - KH) // stride + 1
  Wo = (W + pl + pr - KW) // stride + 1
  if Ho <= 0 or Wo <= 0:
    return False
  if lib.snap_conv2d_packed_weights_split_bytes(KH * KW, Cin, Cout, 2) == 0:
    return False
  d = _lib.SnapConvDesc(N, H, W, Cin, Cs, KH, KW, stride, pt, pl, Ho, Wo, Cout, Cout, PRO_NONE, 0, 1.0, 0.0)
  return bool(lib.snap_conv2d_presplit_supported(ctypes.byref(d)))


def conv2d(
    x, w, *, stride=1, padding=((0, 0), (0, 0)), cin=None, prologue=PRO_NONE,
    gn=None, in_affine=(1.0, 0.0), bias=None, relu=False, residual=None,
    up_prev=None, row_mask=None, rows_in=None, rows_out=None, row_count=None, out=None,
    emit_gn_stats=None, math=None, gelu=False, res_init=None, ps_tile=None, out_half=False, out_stride=None,
    gn_bwd_stats=None, bf16_ring=False,
):
  """NHWC implicit-GEMM conv on the matrix cores.  x [N,H,W,Cs]; w [KH,KW,Cin,Cout] (HWIO).

  x may be a ``PreSplit`` (``gn_norm_split`` / ``presplit``): the launch then runs on the
  pre-split engine -- 'bf16x3' arithmetic, prologue NONE, no row lists; ``res_init`` (default
  ``PS_RES_INIT``) loads ``residual`` into the accumulators before the K loop, ``ps_tile``
  (default ``PS_TILE``) forces a row tile.

  math = 'f32' (exact f32 MFMA) | 'bf16x6' / 'bf16x3' (f32-grade split-bf16 engine: every
  operand split into 3 / 2 bf16 parts after the f32 prologue, 6 / 3 part products accumulated in
  f32 -- ~2^-24 / ~2^-17 relative error per product) | 'bf16' (operands rounded to bf16, f32
  accumulate: the training-precision engine); None = ``MATMUL_PRECISION``.

  gn = (mu [N,Cin], sc [N,Cin], beta [Cin]) for PRO_GN_RELU / PRO_RELU_GN.
  rows_in / rows_out (int32 [M]) + row_count (int32 [1], device): row-indexed launch
  over a compacted row list (see ``compact_rows``); ``out`` supplies the destination.
  emit_gn_stats = 'raw' | 'relu' | 'both': the epilogue also emits the partial sums from which
  ``group_norm_stats(y, ...)`` (same ``relu_first``) builds its result without re-reading
  y; they travel as ``y._snap_gn_partial`` ('both' = 'raw' plus, where the engine has the kernel
  variant -- split-bf16, GroupNorm -> ReLU prologue, 128 x 128 tiles --, those of relu(y) as
  ``y._snap_gn_partial_relu``).  Ignored where the shape does not allow it.
  out_half (training-precision engines 'bf16' / 'fp16' only): the result is written ONLY rounded to the
  engine's element type and returned as a bf16 / f16 tensor (the hidden activations and inter-layer
  gradients of the masked MLP: every consumer rounds them to that type anyway).
  gn_bwd_stats = (x_gn, mu, rstd, gamma, beta, mode) (a half-precision input -- the data-gradient launch that
  reads a GroupNorm VJP's twin): the epilogue also emits the statistics of the GroupNorm VJP that consumes THIS
  result as its incoming gradient (x_gn [N,Ho,Wo,Cout] is that GroupNorm's input, mode its prologue);
  ``ops_bwd.group_norm_bwd`` finds them on the result (``y._snap_gnb_partial``) and skips its first pass.
  Silently not done where the launch splits K or the shape has no statistics layout.
  bf16_ring (a bf16 input, math 'bf16', Cin % 16 == 0, no row lists / statistics / split-K): the launch runs on the
  ONE-PART pre-split engine (conv_ps.hip, NS = 1: both operands by LDS-DMA through the three-stage ring, 256 x 128
  tiles) -- the arithmetic of the training-precision engine (1 x 1: the same bits), 500-730 instead of 350-560
  TFLOP/s on large dense layers; it never splits K, so small-M / deep-K launches (the training step's) stay where
  they are: opt-in per call (the ViT encoder's inference path).  ``Tuning.BF16_PS = False`` turns it off.
  out_stride (with ``out``, a multiple of 4 >= Cout): out's rows hold out_stride floats and the result goes
  to their first Cout (the other columns are not touched); no statistics, residual or up-sampling epilogue.
  Returns y [N,Ho,Wo,Cout] (``out`` itself, [.., out_stride], with out_stride).
  """
  lib = _lib.load()
  ps = isinstance(x, PreSplit)
  xh = (not ps) and x.dtype in (torch.bfloat16, torch.float16)
  if xh:
    # the input already in the training-precision engine's element type (the half twin a GroupNorm
    # VJP wrote next to its f32 gradient): both operands by LDS-DMA (conv_bf16.hip)
    want = 'fp16' if x.dtype == torch.float16 else 'bf16'
    math = want if math is None else math
    if (math != want or prologue != PRO_NONE or rows_in is not None or x.shape[-1] % 8 or w.shape[2] % 8
        or emit_gn_stats is not None):
      raise ValueError('conv2d: a half-precision input takes the matching engine, prologue NONE, whole '
                       'channel octets, no input row list / statistics')
    _chk(x, x.dtype, 'x')
    N, H, W, Cs = x.shape
  if ps:
    if (prologue != PRO_NONE or rows_in is not None or rows_out is not None or row_count is not None
        or math not in (None, 'bf16x3')):
      raise ValueError('conv2d: a PreSplit input takes prologue NONE, no row lists, math bf16x3')
    math = 'bf16x3'
    xs, x = x, x.data
    _chk(x, torch.bfloat16, 'x')
    N, H, W, Cs = xs.shape
  elif not xh:
    _f32(x, 'x')
    N, H, W, Cs = x.shape
  pw = isinstance(w, PackedWeights)
  if pw:
    if not ps:
      raise ValueError('conv2d: PackedWeights (a two-part split image) go with a PreSplit input')
    w_img, w = w, w.data          # (the engine reads only the image; `w` passes a non-NULL pointer)
    KH, KW, Cin, Cout = w_img.shape
  else:
    _f32(w, 'w')
    KH, KW, Cin, Cout = w.shape
  if cin is None:
    cin = Cs
  if cin != Cin:
    raise ValueError(f'conv2d: kernel expects Cin={Cin}, input has {cin}')
  (pt, pb), (pl, pr) = padding
  Ho = (H + pt + pb - KH) // stride + 1
  Wo = (W + pl + pr - KW) // stride + 1
  yh = None
  if out_half:
    hm = precision() if math is None else math
    if (hm not in HALF_MATH or out is not None or emit_gn_stats is not None or up_prev is not None
        or prologue not in (PRO_NONE, PRO_RELU) or Cs % 4 or Cin < 4 or Cout % 4):
      raise ValueError('conv2d: out_half needs a training-precision engine launch (prologue NONE / RELU, '
                       'no statistics / up-sampling epilogue, channel quads)')
    yh = torch.empty((N, Ho, Wo, Cout), dtype=torch.float16 if hm == 'fp16' else torch.bfloat16, device=x.device)
    y = yh                      # (shape carrier for the checks below; the f32 pointer passed is NULL)
  elif out is None:
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
  else:
    y = _f32(out, 'out')
    if out_stride is not None:
      if (out_stride % 4 or out_stride < Cout or y.numel() != N * Ho * Wo * out_stride or emit_gn_stats is not None
          or residual is not None or up_prev is not None or isinstance(x, PreSplit)):
        raise ValueError('conv2d: out_stride needs out [rows, out_stride], out_stride % 4 == 0, a plain epilogue')
    elif y.numel() != N * Ho * Wo * Cout:
      raise ValueError('conv2d: out has the wrong size')
  if out_stride is not None and out is None:
    raise ValueError('conv2d: out_stride goes with out')
  for t, nm in ((rows_in, 'rows_in'), (rows_out, 'rows_out'), (row_count, 'row_count')):
    if t is not None:
      _chk(t, torch.int32, nm)
  epi = 0
  mu = sc = beta = None
  if prologue in (PRO_GN_RELU, PRO_RELU_GN):
    mu, sc, beta = gn
    _f32(mu, 'gn_mu'); _f32(sc, 'gn_sc'); _f32(beta, 'gn_beta')
    if mu.numel() != N * Cin or sc.numel() != N * Cin or beta.numel() != Cin:
      raise ValueError('conv2d: GroupNorm statistics have the wrong size')
  if bias is not None:
    _f32(bias, 'bias'); epi |= EPI_BIAS
    if bias.numel() != Cout:
      raise ValueError('conv2d: bias size')
  if relu:
    epi |= EPI_RELU
  if gelu:                                   # tanh-approximated GELU (ViT MLP)
    epi |= EPI_GELU
  if residual is not None:
    _f32(residual, 'residual'); epi |= EPI_RESIDUAL
    if residual.shape != y.shape:
      raise ValueError(f'conv2d: residual {tuple(residual.shape)} vs {tuple(y.shape)}')
  if up_prev is not None:
    _f32(up_prev, 'up_prev'); epi |= EPI_UPSAMPLE2X_ADD
    if tuple(up_prev.shape) != (N, Ho // 2, Wo // 2, Cout):
      raise ValueError('conv2d: up_prev shape')
  if row_mask is not None:
    _mask(row_mask, 'row_mask'); epi |= EPI_ROWMASK
    if row_mask.numel() != N * Ho * Wo:
      raise ValueError('conv2d: row_mask size')
  d = _lib.SnapConvDesc(
      N, H, W, Cin, Cs, KH, KW, stride, pt, pl, Ho, Wo, Cout, Cout if out_stride is None else int(out_stride), prologue,
      epi, float(in_affine[0]), float(in_affine[1]),
  )
  if tuning().CONV_TILE:
    bm, bn = (int(v) for v in tuning().CONV_TILE.split('x'))
    d.tile_hint = bm * 1000 + bn
  d.tile_hint += 1000000 * _stationary_mode()
  M = N * Ho * Wo
  # the engine the launch will take (the statistics layout depends on it)
  math = precision() if math is None else math
  if math not in ('f32', 'bf16', 'fp16', 'bf16x3', 'bf16x6'):
    raise ValueError(f'conv2d: math={math!r}')
  parts = SPLIT_PARTS.get(math, 0)
  if parts and lib.snap_conv2d_packed_weights_split_bytes(KH * KW, Cin, Cout, parts) == 0:
    math, parts = 'f32', 0     # weight image beyond the split engine's 32-bit offsets (template banks)
  qparts = parts if (Cs % 4 == 0 and Cin >= 4 and not ps) else 0   # (what SnapConvExtras.w_split_parts will say)
  # a bf16 input whose launch needs nothing but the GEMM + a plain epilogue runs on the ONE-PART pre-split engine
  # (conv_ps.hip, NS = 1: both operands by LDS-DMA through the three-stage ring, 256 x 128 tiles) -- the same
  # arithmetic as the training-precision engine (operands rounded to bf16, f32 accumulate)
  ps1 = bool(bf16_ring and xh and x.dtype == torch.bfloat16 and math == 'bf16' and tuning().BF16_PS and Cs == Cin and Cin % 16 == 0
             and rows_out is None and row_count is None and up_prev is None and gn_bwd_stats is None
             and out is None and Cout % 4 == 0 and lib.snap_conv2d_presplit_supported(ctypes.byref(d)))
  if xh and out_half and not ps1:
    raise ValueError('conv2d: a half-precision input AND output need the one-part pre-split engine (bf16, Cin % 16 == 0, '
                     'no row lists)')
  ex = None
  partial = partial2 = None
  kws = None
  rows32 = False
  gnb_done = False
  if rows_in is not None or rows_out is not None or row_count is not None:
    ex = _lib.SnapConvExtras(_pv(rows_in), _pv(rows_out), _pv(row_count), None, 0, 0, None, 0,
                             None, 0)
  else:
    pst = (tuning().PS_TILE if ps_tile is None else int(ps_tile)) if ps else 0
    if ps:
      wbytes = lib.snap_conv2d_presplit_workspace_bytes(ctypes.byref(d), pst) if tuning().USE_SPLITK else 0
    elif ps1 or (qparts == 2 and lib.snap_conv2d_stationary_kind(ctypes.byref(d), qparts)):
      wbytes = 0        # a stationary-operand kernel / the one-part pre-split engine takes the launch: it never splits K
    else:
      wbytes = lib.snap_conv2d_workspace_bytes(ctypes.byref(d)) if (tuning().USE_SPLITK and yh is None) else 0
    if wbytes:   # small-M / deep-K layer: split K
      kws = torch.empty(wbytes // 4, dtype=torch.float32, device=x.device)
      ex = _lib.SnapConvExtras(None, None, None, None, 0, 0, kws.data_ptr(), wbytes, None, 0)
      half_engine = math in ('bf16', 'fp16') and Cs % 4 == 0 and Cin % 4 == 0 and Cin >= 4 and not ps
      if emit_gn_stats is not None and (qparts >= 2 or half_engine) and tuning().SPLITK_STATS:
        # the reduce pass of the split / bf16 / fp16 engine emits the partial sums (per 32-row slab)
        pbytes = lib.snap_conv2d_splitk_gn_partial_bytes(ctypes.byref(d))
        if pbytes:
          partial = torch.empty(pbytes // 4, dtype=torch.float32, device=x.device)
          ex.gn_partial = partial.data_ptr()
          ex.gn_partial_bytes = pbytes
          ex.gn_partial_relu = int(emit_gn_stats == 'relu')
          ex.gn_partial_rows = 32
          rows32 = True
    elif (gn_bwd_stats is not None and xh and emit_gn_stats is None and out is None and up_prev is None
          and not relu and not gelu and row_mask is None):
      pbytes = lib.snap_conv2d_gn_partial_bytes_ex(ctypes.byref(d), 0)
      gx, gmu, grs, gga, gbe, gmode = gn_bwd_stats
      if (pbytes and tuple(gx.shape) == (N, Ho, Wo, Cout) and gmu.numel() == N * Cout and grs.numel() == N * Cout
          and gga.numel() == Cout and gbe.numel() == Cout and gmode in (PRO_GN_RELU, PRO_RELU_GN)):
        for t, nm in ((gx, 'gn_bwd x'), (gmu, 'gn_bwd mu'), (grs, 'gn_bwd rstd'), (gga, 'gn_bwd gamma'), (gbe, 'gn_bwd beta')):
          _f32(t, nm)
        partial = torch.empty(pbytes // 4, dtype=torch.float32, device=x.device)
        ex = _lib.SnapConvExtras(None, None, None, partial.data_ptr(), pbytes, 0, None, 0, None, 0)
        ex.gnb_x, ex.gnb_mu, ex.gnb_rstd = gx.data_ptr(), gmu.data_ptr(), grs.data_ptr()
        ex.gnb_gamma, ex.gnb_beta, ex.gnb_mode = gga.data_ptr(), gbe.data_ptr(), int(gmode)
        gnb_done = True
    elif emit_gn_stats is not None:
      pbytes = (lib.snap_conv2d_presplit_gn_partial_bytes(ctypes.byref(d), pst) if ps
                else lib.snap_conv2d_gn_partial_bytes_ex(ctypes.byref(d), qparts))
      if pbytes:
        partial = torch.empty(pbytes // 4, dtype=torch.float32, device=x.device)
        ex = _lib.SnapConvExtras(None, None, None, partial.data_ptr(), pbytes,
                                 int(emit_gn_stats == 'relu'), None, 0, None, 0)
        if emit_gn_stats == 'both' and math in SPLIT_PARTS:
          # the statistics of y AND of relu(y): a request (SnapConvExtras.gn_partial2_done)
          partial2 = torch.empty(pbytes // 4, dtype=torch.float32, device=x.device)
          ex.gn_partial2 = partial2.data_ptr()
          ex.gn_partial2_bytes = pbytes
  family = 'conv_igemm'
  wpk = None
  # the RGB root convolution (7 x 7 / stride 2 / pad 3) of an image stored with 4 floats per pixel
  # runs on the split engine with its own weight image (a K slab = 4 pixels of a kernel row)
  if ps and (parts != 2 or Cin % 16):
    raise ValueError('conv2d: the pre-split engine needs Cin % 16 == 0 and a two-part weight image')
  root = (parts and (KH, KW, stride, Cin, Cs) == (7, 7, 2, 3, 4) and (pt, pl) == (3, 3)
          and prologue in (PRO_NONE, PRO_AFFINE) and rows_in is None and rows_out is None
          and row_count is None and partial is None and not ps)
  if root:
    wpk = _packed_weights(w, math + '/root', parts)
    if ex is None:
      ex = _lib.SnapConvExtras(None, None, None, None, 0, 0, None, 0, None, 0)
    ex.w_bf16 = wpk.data_ptr()
    ex.w_bf16_bytes = wpk.numel() * 2
    ex.w_split_parts = parts
    ex.w_split_root = 1
    family = f'conv_split_{math}'
  elif ps1:
    wpk = _packed_weights(w, 'bf16/ps1', 1)
    if ex is None:
      ex = _lib.SnapConvExtras(None, None, None, None, 0, 0, None, 0, None, 0)
    ex.w_bf16 = wpk.data_ptr()
    ex.w_bf16_bytes = wpk.numel() * 2
    ex.w_split_parts = 1
    ex.x_presplit = 1
    ex.ps_tile = int(tuning().PS_TILE if ps_tile is None else ps_tile)
    if yh is not None:
      ex.y_half = yh.data_ptr()
    family = 'conv_bf16'
  elif math != 'f32' and Cs % 4 == 0 and Cin >= 4:
    wpk = w_img.data if pw else _packed_weights(w, math, parts)
    if ex is None:
      ex = _lib.SnapConvExtras(None, None, None, None, 0, 0, None, 0, None, 0)
    ex.w_bf16 = wpk.data_ptr()
    ex.w_bf16_bytes = wpk.numel() * 2
    ex.w_split_parts = parts
    ex.w_half = int(math == 'fp16')
    ex.x_half = int(xh)
    if yh is not None:
      ex.y_half = yh.data_ptr()
    family = f'conv_split_{math}' if parts else ('conv_fp16' if math == 'fp16' else 'conv_bf16')
    if ps:
      ex.x_presplit = 1
      ex.ps_tile = pst
      ex.ps_res_init = int(tuning().PS_RES_INIT if res_init is None else bool(res_init))
  if tuning().CONV_BK or tuning().CONV_NO_HALO or tuning().CONV_RS_NSPLIT or tuning().CONV_NO_PLAIN or tuning().CONV_RAW_RING:
    if ex is None:
      ex = _lib.SnapConvExtras(None, None, None, None, 0, 0, None, 0, None, 0)
    ex.bk_hint = int(tuning().CONV_BK or 0)
    ex.tune_flags = (int(bool(tuning().CONV_NO_HALO)) | 2 * int(bool(tuning().CONV_RAW_RING)) | 8 * int(bool(tuning().CONV_NO_PLAIN))
                     | ((int(tuning().CONV_RS_NSPLIT) & 15) << 4))
  kflops = 2.0 * KH * KW * Cin * Cout
  if row_count is None:
    flops = kflops * M
    nbytes = 4.0 * ((xs.numel() if ps else x.numel() * (0.5 if xh else 1.0)) + (w_img.numel() if pw else w.numel()) + y.numel()
                    + (residual.numel() if residual is not None else 0))
  else:  # resolved after the sync: only the listed rows are multiplied / moved
    flops = lambda: kflops * int(row_count.item())
    nbytes = lambda: 4.0 * (int(row_count.item()) * (Cin + Cout) + w.numel())
  kind = (lib.snap_conv2d_stationary_kind(ctypes.byref(d), qparts)
          if (qparts == 2 and rows_in is None and rows_out is None and row_count is None) else 0)
  with _region(
      family, flops, nbytes,
      lambda: f'{"PS_" if ps else "PS1_" if ps1 else ("", "RS_", "WS_", "WS_")[kind]}M{M}{"r" if row_count is not None else ""}_K{KH}x{KW}x{Cin}_N{Cout}'
              f'_s{stride}_p{prologue}_e{epi}',
  ):
    st = lib.snap_conv2d_nhwc_ex_f32(
        ctypes.byref(d), _p(x), _p(w), None if yh is not None else _p(y), _p(mu), _p(sc), _p(beta), _p(bias),
        _p(residual), _p(up_prev), _p(row_mask), None if ex is None else ctypes.byref(ex),
        _stream(),
    )
  _lib.check(st, 'snap_conv2d_nhwc_ex_f32')
  if gnb_done:
    # (the GroupNorm VJP checks that it is handed the same x before it trusts these sums)
    y._snap_gnb_partial = (partial, lib.snap_conv2d_tile_rows_ex(ctypes.byref(d), 0), gn_bwd_stats[0].data_ptr(),
                           gn_bwd_stats[0]._version, int(gn_bwd_stats[5]))
  elif partial is not None:
    tile_rows = (32 if rows32 else lib.snap_conv2d_presplit_tile_rows(ctypes.byref(d), pst) if ps
                 else lib.snap_conv2d_tile_rows_ex(ctypes.byref(d), qparts))
    y._snap_gn_partial = (partial, tile_rows, emit_gn_stats == 'relu')
    if partial2 is not None and ex.gn_partial2_done:
      y._snap_gn_partial_relu = (partial2, y._snap_gn_partial[1], True)
  return y


def pack_weights_bf16(w, half=False):
  """w [KH,KW,Cin,Cout] f32 -> the training-precision engine's weight image
  [Cout][KH*KW][roundup(Cin,8)] in bf16, or (half) IEEE float16 -- math 'fp16'."""
  lib = _lib.load()
  _f32(w, 'w')
  KH, KW, Cin, Cout = w.shape
  nbytes = lib.snap_conv2d_packed_weights_bytes(KH * KW, Cin, Cout)
  out = torch.empty(nbytes // 2, dtype=torch.float16 if half else torch.bfloat16, device=w.device)
  fn = lib.snap_conv2d_pack_weights_f16 if half else lib.snap_conv2d_pack_weights_bf16
  st = fn(_p(w), KH * KW, Cin, Cout, _p(out), nbytes, _stream())
  _lib.check(st, 'snap_conv2d_pack_weights_f16' if half else 'snap_conv2d_pack_weights_bf16')
  return out


# the training-precision engines: operands rounded to bf16 / IEEE half, f32 accumulate
HALF_MATH = ('bf16', 'fp16')


SPLIT_PARTS = {'bf16x3': 2, 'bf16x6': 3}
# Bumped by every ``ForwardContext`` (= every ``apply``): the bf16 weight images are prepared
# once per apply and tensor (the map and query passes of one apply share them), never carried
# from one apply / timed step to the next -- like the StdConv standardisation.
PACK_EPOCH = 0


def _packed_weights(w, math, parts):
  """The engine's weight image of ``w`` for this apply: an explicit ``w._snap_packed[math]``
  (tools) wins; otherwise packed at first use and remembered on the tensor for the current
  ``PACK_EPOCH`` while the tensor is not modified in place."""
  slot = getattr(w, '_snap_packed', None)
  if slot is not None:
    hit = slot.get(math)
    if hit is not None:
      if not isinstance(hit, tuple):
        return hit
      if hit[0] == PACK_EPOCH and hit[1] == w._version:
        return hit[2]
  if math.endswith('/root'):
    wpk = pack_weights_split_root_bf16(w, parts)
  else:
    wpk = pack_weights_split_bf16(w, parts) if parts else pack_weights_bf16(w, half=(math == 'fp16'))
  if slot is None:
    slot = {}
    try:
      w._snap_packed = slot
    except AttributeError:
      return wpk
  slot[math] = (PACK_EPOCH, w._version, wpk)
  return wpk


def pack_weights_split_bf16(w, parts):
  """w [KH,KW,Cin,Cout] f32 -> the split engine's image [parts][Cout][KH*KW][roundup(Cin,8)]
  (part 0 = bf16(w), part p = bf16 of the exact f32 residual of parts < p)."""
  lib = _lib.load()
  _f32(w, 'w')
  KH, KW, Cin, Cout = w.shape
  nbytes = lib.snap_conv2d_packed_weights_split_bytes(KH * KW, Cin, Cout, parts)
  out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
  st = lib.snap_conv2d_pack_weights_split_bf16(_p(w), KH * KW, Cin, Cout, parts, _p(out), nbytes,
                                               _stream())
  _lib.check(st, 'snap_conv2d_pack_weights_split_bf16')
  return out


def pack_weights_split_root_bf16(w, parts):
  """w [7,7,3,Cout] f32 -> the split engine's ROOT image ([Cout/128][14 slabs = (kh, group of 4
  kw)][parts][128][16 k = 4 kw x (RGB + 0)])."""
  lib = _lib.load()
  _f32(w, 'w')
  if tuple(w.shape[:3]) != (7, 7, 3):
    raise ValueError('pack_weights_split_root_bf16: w must be [7, 7, 3, Cout]')
  Cout = w.shape[3]
  nbytes = lib.snap_conv2d_packed_weights_split_root_bytes(Cout, parts)
  out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
  st = lib.snap_conv2d_pack_weights_split_root_bf16(_p(w), Cout, parts, _p(out), nbytes, _stream())
  _lib.check(st, 'snap_conv2d_pack_weights_split_root_bf16')
  return out


_PACK_ITEM = np.dtype([('w', np.uint64), ('out', np.uint64), ('taps', np.int32), ('Cin', np.int32),
                       ('Cout', np.int32), ('block_begin', np.int32)])


def pack_weights_split_multi(ws, math):
  """Prepare the split engine's weight image of every kernel in ``ws`` (HWIO tensors) with ONE
  launch and remember it on the tensors for the current apply (see ``_packed_weights``)."""
  parts = SPLIT_PARTS[math]
  lib = _lib.load()
  todo = []
  for w in ws:
    slot = getattr(w, '_snap_packed', None)
    hit = None if slot is None else slot.get(math)
    if isinstance(hit, tuple) and hit[0] == PACK_EPOCH and hit[1] == w._version:
      continue
    KH, KW, Cin, Cout = w.shape
    if Cin < 4 or lib.snap_conv2d_packed_weights_split_bytes(KH * KW, Cin, Cout, parts) == 0:
      continue                       # runs on the f32 engine (conv2d decides the same way)
    todo.append(w)
  if not todo:
    return
  items = np.zeros(len(todo), dtype=_PACK_ITEM)
  outs = []
  blk = 0
  for i, w in enumerate(todo):
    _f32(w, 'w')
    KH, KW, Cin, Cout = w.shape
    nbytes = lib.snap_conv2d_packed_weights_split_bytes(KH * KW, Cin, Cout, parts)
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    outs.append(out)
    items[i] = (w.data_ptr(), out.data_ptr(), KH * KW, Cin, Cout, blk)
    blk += lib.snap_conv2d_pack_weights_split_blocks(KH * KW, Cin, Cout)
  table = upload_table(items, todo[0].device)
  st = lib.snap_conv2d_pack_weights_split_multi_bf16(_p(table), len(todo), blk, parts, _stream())
  _lib.check(st, 'snap_conv2d_pack_weights_split_multi_bf16')
  for w, out in zip(todo, outs):
    slot = getattr(w, '_snap_packed', None)
    if slot is None:
      slot = {}
      w._snap_packed = slot
    slot[math] = (PACK_EPOCH, w._version, out)


def pack_weights_bf16_multi(ws, with_rotated=True, math='bf16'):
  """Training precision (math 'bf16' | 'fp16'): the engine's image of every kernel in ``ws`` (HWIO
  tensors) -- and, for the data-gradient convolutions, of its rotated transpose -- with ONE launch;
  remembered on the tensors for the current apply (``_packed_weights`` / ``packed_rot_image``)."""
  lib = _lib.load()
  if math not in HALF_MATH:
    raise ValueError(f'pack_weights_bf16_multi: math={math!r}')
  half = math == 'fp16'
  todo = []
  for w in ws:
    slot = getattr(w, '_snap_packed', None)
    hit = None if slot is None else slot.get(math)
    if isinstance(hit, tuple) and hit[0] == PACK_EPOCH and hit[1] == w._version:
      continue
    if w.shape[2] < 4:
      continue                       # (runs on the f32 engine)
    todo.append(w)
  if not todo:
    return
  n = len(todo) * (2 if with_rotated else 1)
  items = np.zeros(n, dtype=_PACK_ITEM)
  outs = []
  blk = 0
  i = 0
  for w in todo:
    _f32(w, 'w')
    KH, KW, Cin, Cout = w.shape
    taps = KH * KW
    for rot in ((False, True) if with_rotated else (False,)):
      nbytes = (lib.snap_conv2d_packed_weights_bytes(taps, Cout, (Cin + 3) // 4 * 4) if rot
                else lib.snap_conv2d_packed_weights_bytes(taps, Cin, Cout))
      out = torch.empty(nbytes // 2, dtype=torch.float16 if half else torch.bfloat16, device=w.device)
      outs.append(out)
      items[i] = (w.data_ptr(), out.data_ptr(), -taps if rot else taps, Cin, Cout, blk)
      blk += lib.snap_conv2d_pack_weights_blocks(-taps if rot else taps, Cin, Cout)
      i += 1
  table = upload_table(items, todo[0].device)
  fn = lib.snap_conv2d_pack_weights_multi_f16 if half else lib.snap_conv2d_pack_weights_multi_bf16
  st = fn(_p(table), n, blk, _stream())
  _lib.check(st, 'snap_conv2d_pack_weights_multi_f16' if half else 'snap_conv2d_pack_weights_multi_bf16')
  k = 0
  for w in todo:
    slot = getattr(w, '_snap_packed', None)
    if slot is None:
      slot = {}
      w._snap_packed = slot
    slot[math] = (PACK_EPOCH, w._version, outs[k]); k += 1
    if with_rotated:
      slot[math + '/rot'] = (PACK_EPOCH, w._version, outs[k]); k += 1


def packed_rot_image(w, math='bf16'):
  """The rotated image ``pack_weights_bf16_multi`` prepared for ``w`` in this apply, or None."""
  slot = getattr(w, '_snap_packed', None)
  hit = None if slot is None else slot.get(math + '/rot')
  if isinstance(hit, tuple) and hit[0] == PACK_EPOCH and hit[1] == w._version:
    return hit[2]
  return None


def dense(x, kernel, bias=None, *, cin=None, prologue=PRO_NONE, relu=False,
          row_mask=None, rows_in=None, rows_out=None, row_count=None, out=None, math=None,
          gelu=False, residual=None, out_half=False, out_stride=None, bf16_ring=False):
  """x [..., Cs] @ kernel [Cin, Cout] (+bias) through the conv engine (1x1).  out_stride: ``conv2d``."""
  lead = x.shape[:-1]
  M = int(np.prod(lead)) if len(lead) else 1
  if out_stride is not None:
    conv2d(x.reshape(1, 1, M, x.shape[-1]), kernel.reshape(1, 1, *kernel.shape),
           cin=cin if cin is not None else kernel.shape[0], prologue=prologue, bias=bias, relu=relu, row_mask=row_mask,
           rows_in=rows_in, rows_out=rows_out, row_count=row_count, out=out, math=math, gelu=gelu,
           out_stride=out_stride)
    return out
  y = conv2d(
      x.reshape(1, 1, M, x.shape[-1]), kernel.reshape(1, 1, *kernel.shape),
      cin=cin if cin is not None else kernel.shape[0], prologue=prologue,
      bias=bias, relu=relu, row_mask=row_mask, rows_in=rows_in, rows_out=rows_out,
      row_count=row_count, out=None if out is None else out.reshape(1, 1, M, kernel.shape[1]),
      math=math, gelu=gelu,
      residual=None if residual is None else residual.reshape(1, 1, M, kernel.shape[1]), out_half=out_half,
      bf16_ring=bf16_ring,
  )
  return y.reshape(*lead, kernel.shape[1])


def semantic_embed(rasters, idx_road, idx_other, table_road, table_other):
  """rasters [..., N] bool -> [..., (1 + len(idx_other)) * E] (semantic_raster_encoder.py:63-79)."""
  lib = _lib.load()
  _mask(rasters, 'rasters'); _f32(table_road, 'table_road'); _f32(table_other, 'table_other')
  N = rasters.shape[-1]
  M = rasters.numel() // N
  E = table_road.shape[1]
  nr, no = len(idx_road), len(idx_other)
  if tuple(table_road.shape) != (nr, E) or tuple(table_other.shape) != (2 * no, E):
    raise ValueError('semantic_embed: table shapes')
  ir = (ctypes.c_int32 * max(nr, 1))(*idx_road)
  io = (ctypes.c_int32 * max(no, 1))(*idx_other)
  out = torch.empty((*rasters.shape[:-1], (1 + no) * E), dtype=torch.float32, device=rasters.device)
  st = lib.snap_semantic_embed_f32(_p(rasters), M, N, ctypes.cast(ir, ctypes.c_void_p), nr,
                                   ctypes.cast(io, ctypes.c_void_p), no, _p(table_road),
                                   _p(table_other), E, _p(out), _stream())
  _lib.check(st, 'snap_semantic_embed_f32')
  return out


def semantic_onehot(rasters, idx_road, idx_other):
  """[M, KP] one-hot matrix of the embedding rows each pixel reads (KP = roundup(nr + 2 no, 4))."""
  lib = _lib.load()
  _mask(rasters, 'rasters')
  N = rasters.shape[-1]
  M = rasters.numel() // N
  nr, no = len(idx_road), len(idx_other)
  KP = (nr + 2 * no + 3) // 4 * 4
  ir = (ctypes.c_int32 * max(nr, 1))(*idx_road)
  io = (ctypes.c_int32 * max(no, 1))(*idx_other)
  onehot = torch.empty((M, KP), dtype=torch.float32, device=rasters.device)
  st = lib.snap_semantic_onehot_f32(_p(rasters), M, N, ctypes.cast(ir, ctypes.c_void_p), nr,
                                    ctypes.cast(io, ctypes.c_void_p), no, _p(onehot), KP, _stream())
  _lib.check(st, 'snap_semantic_onehot_f32')
  return onehot


def stack_templates(tw, S, layout='hwdr'):
  """tw [H, W, D, R] (layout 'hwdr') or templates [R, H, W, D] ('rhwd') -> shift-stacked filter
  bank [H+S-1, W+S-1, D, R*S*S] (see snap_hip.h)."""
  lib = _lib.load()
  _f32(tw, 'tw')
  if layout == 'hwdr':
    H, W, D, R = tw.shape
    fn, name = lib.snap_stack_templates_f32, 'snap_stack_templates_f32'
  else:
    R, H, W, D = tw.shape
    fn, name = lib.snap_stack_templates_rhwd_f32, 'snap_stack_templates_rhwd_f32'
  tws = torch.empty((H + S - 1, W + S - 1, D, R * S * S), dtype=torch.float32, device=tw.device)
  st = fn(_p(tw), _p(tws), H, W, D, R, S, _stream())
  _lib.check(st, name)
  return tws


def layer_norm(x, gamma, beta, eps=1e-6, out_half=False):
  """LayerNorm over the last axis (flax.linen.LayerNorm: biased variance, eps inside the sqrt).
  out_half: the result ONLY rounded to bf16 (the operand of a 'bf16' dense layer, which rounds it anyway)."""
  lib = _lib.load()
  _f32(x, 'x'); _f32(gamma, 'gamma'); _f32(beta, 'beta')
  C = x.shape[-1]
  M = x.numel() // C
  if out_half:
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with _region('layer_norm', 0.0, 6.0 * x.numel()):
      st = lib.snap_layer_norm_bf16out_f32(_p(x), _p(gamma), _p(beta), _p(y), M, C, float(eps), _stream())
    _lib.check(st, 'snap_layer_norm_bf16out_f32')
    return y
  y = torch.empty_like(x)
  with _region('layer_norm', 0.0, 8.0 * x.numel()):
    st = lib.snap_layer_norm_f32(_p(x), _p(gamma), _p(beta), _p(y), M, C, float(eps), _stream())
  _lib.check(st, 'snap_layer_norm_f32')
  return y


def attention(qkv, scale=None, want_lse=False, out_half=False):
  """Multi-head self-attention on the bf16 matrix cores.  qkv [B, N, 3, H, 64] (fused QKV
  projection output, f32) -> [B, N, H*64] f32 = softmax(scale * Q K^T) V per head.
  want_lse: also return the base-2 log-sum-exp of the scaled scores [B, H, N] (for the VJP).
  out_half (inference): the result ONLY rounded to bf16 (the operand of the 'bf16' output projection)."""
  lib = _lib.load()
  B, N, three, H, D = qkv.shape
  if three != 3:
    raise ValueError('attention: qkv must be [B, N, 3, H, D]')
  scale = D ** -0.5 if scale is None else float(scale)
  if qkv.dtype == torch.bfloat16:
    # the QKV projection's bf16-only output: K / V move at half the bytes (inference; the result in bf16 too)
    if want_lse or not out_half:
      raise ValueError('attention: a bf16 qkv is the inference form (out_half=True, no lse)')
    _chk(qkv, torch.bfloat16, 'qkv')
    out = torch.empty((B, N, H * D), dtype=torch.bfloat16, device=qkv.device)
    with _region('attention', 4.0 * B * H * N * N * D, 2.0 * (qkv.numel() + out.numel())):
      st = lib.snap_attention_bf16io(_p(qkv), _p(out), B, N, H, D, scale, _stream())
    _lib.check(st, 'snap_attention_bf16io')
    return out
  _f32(qkv, 'qkv')
  if out_half:
    if want_lse:
      raise ValueError('attention: out_half is the inference form (no lse)')
    out = torch.empty((B, N, H * D), dtype=torch.bfloat16, device=qkv.device)
    with _region('attention', 4.0 * B * H * N * N * D, 4.0 * qkv.numel() + 2.0 * out.numel()):
      st = lib.snap_attention_bf16out_f32(_p(qkv), _p(out), B, N, H, D, scale, _stream())
    _lib.check(st, 'snap_attention_bf16out_f32')
    return out
  out = torch.empty((B, N, H * D), dtype=torch.float32, device=qkv.device)
  lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if want_lse else None
  with _region('attention', 4.0 * B * H * N * N * D, 4.0 * (qkv.numel() + out.numel())):
    st = lib.snap_attention_lse_bf16_f32(_p(qkv), _p(out), _p(lse), B, N, H, D, scale, _stream())
  _lib.check(st, 'snap_attention_lse_bf16_f32')
  return (out, lse) if want_lse else out


def gelu(x):
  """tanh-form GELU as a stand-alone kernel (training path: the pre-activation is kept)."""
  lib = _lib.load()
  _f32(x, 'x')
  y = torch.empty_like(x)
  with _region('gelu', 0.0, 8.0 * x.numel()):
    st = lib.snap_gelu_f32(_p(x), _p(y), x.numel(), _stream())
  _lib.check(st, 'snap_gelu_f32')
  return y


def compact_rows(mask, lo=1, hi=255):
  """mask [...] (bool/uint8) -> (index int32 [M] -- first `count` entries valid, ascending --
  and count int32 [1]) of the rows with lo <= mask <= hi (default: mask != 0); everything stays
  on the device."""
  lib = _lib.load()
  _mask(mask, 'mask')
  M = mask.numel()
  wsb = lib.snap_compact_rows_workspace_bytes(M)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.int32, device=mask.device)
  index = torch.empty(M, dtype=torch.int32, device=mask.device)
  count = torch.empty(1, dtype=torch.int32, device=mask.device)
  st = lib.snap_compact_rows_range_u8(_p(mask), M, int(lo), int(hi), _p(index), _p(count), _p(ws),
                                      ws.numel() * 4, _stream())
  _lib.check(st, 'snap_compact_rows_range_u8')
  return index, count


def mlp2_pool_supported(cin, hidden, out_dim):
  """Shapes the fused fusion-MLP + vertical-max-pool kernel takes (mlp_pool.hip)."""
  return hidden % 32 == 0 and hidden <= 256 and out_dim % 4 == 0 and out_dim <= 128 and cin >= 4


def mlp2_pool_max(x, row_mask, w0, b0, w1, b1, *, cin, Z, relu_in=False, x_split=False,
                  zero_slabs=None, gather=None):
  """Fusion MLP (Dense -> relu -> Dense) over the rows with row_mask != 0 + max over the Z
  levels of every column, in one kernel on the bf16x3 engine (streetview_encoder.py:279-286 +
  bev_mapper.py:78-88).  x [M, Cs] (M = columns * Z, level fastest), row_mask [M];
  w0 [cin, H], w1 [H, D] -> plane [M / Z, D] f32, pvalid [M / Z] bool.  The hidden activations
  and the [M, D] volume are never written.  ``x_split``: x holds the rows pre-split
  (``lift_pool(out_split=True)``: [16-channel slab][hi | lo][16] bf16 in an f32 container); same
  bits, the kernel's A operand then travels by LDS-DMA.  ``zero_slabs`` = (first, count):
  row_mask is a CLASS per row (uint8, ``lift_pool(class_rows=True)``): rows of class 1 are exactly
  zero over those 16-channel slabs (and did not write them), rows of class >= 2 are complete --
  two row lists into one plane, the zero slabs of the first neither read nor multiplied; the same
  plane bit for bit.  ``gather`` = (f_images [B,V,h,w,C], records [B,N,8]) from
  ``lift_pool(tap_records=True)``: the class-1 rows do not exist in x -- the kernel blends their four
  image taps per GEMM0 slab from the records (the lift inside the consumer); the same plane bit for bit."""
  lib = _lib.load()
  _f32(x, 'x'); _mask(row_mask, 'row_mask')
  for t, n in ((w0, 'w0'), (b0, 'b0'), (w1, 'w1'), (b1, 'b1')):
    _f32(t, n)
  M, Cs = x.shape
  H, D = w1.shape
  if M % Z != 0 or row_mask.numel() != M or tuple(w0.shape) != (cin, H):
    raise ValueError('mlp2_pool_max: shapes')
  if not mlp2_pool_supported(cin, H, D):
    raise ValueError(f'mlp2_pool_max: unsupported widths {cin} -> {H} -> {D}')
  if zero_slabs is None:
    index, count = compact_rows(row_mask)
    index_z = count_z = None
    zlo = zn = 0
  else:
    if row_mask.dtype != torch.uint8:
      raise ValueError('mlp2_pool_max: zero_slabs needs the uint8 row classes')
    zlo, zn = int(zero_slabs[0]), int(zero_slabs[1])
    index, count = compact_rows(row_mask, 2, 255)
    index_z, count_z = compact_rows(row_mask, 1, 1)
  w0p = pack_weights_split_bf16(w0.reshape(1, 1, cin, H), 2)
  w1p = pack_weights_split_bf16(w1.reshape(1, 1, H, D), 2)
  ncols = M // Z
  plane = torch.empty(ncols, D, dtype=torch.float32, device=x.device)
  pvalid = torch.empty(ncols, dtype=torch.bool, device=x.device)
  kflops = 2.0 * (cin * H + H * D)
  kflops_z = 2.0 * ((cin - 16 * zn) * H + H * D)

  def rows_():
    return int(count.item()), (int(count_z.item()) if count_z is not None else 0)

  if gather is not None:
    f_img, recs = gather
    _f32(f_img, 'f_images')
    fd = (cin - 1) // 2
    if (zero_slabs is None or not x_split or relu_in or recs.dtype != torch.int32 or recs.numel() != M * 8
        or (zlo, zn) != (fd // 16, fd // 16) or fd % 16 or cin != 2 * fd + 1):
      raise ValueError('mlp2_pool_max: gather needs classed pre-split rows of mean | var | score and [M, 8] records')
    with _region('mlp2_pool_bf16x3', lambda: kflops * rows_()[0] + kflops_z * rows_()[1],
                 lambda: 4.0 * (rows_()[0] * cin + rows_()[1] * 8 + plane.numel()),
                 lambda: f'M{M}r_K{cin}_H{H}_N{D}_Z{Z}_gather'):
      st = lib.snap_mlp2_pool_max_gather_f32(
          _p(x), M, cin, Cs, _p(index), _p(count), _p(index_z), _p(count_z), _p(f_img), f_img.numel() * 4,
          f_img.shape[-2], f_img.shape[-1], fd, _p(recs), int(tuning().MLP_GATHER_XCD_GROUP), _p(w0p), w0p.numel() * 2,
          _p(b0), H, _p(w1p), w1p.numel() * 2, _p(b1), D, Z, ncols, _p(plane), _p(pvalid), _stream())
    _lib.check(st, 'snap_mlp2_pool_max_gather_f32')
    return plane, pvalid
  with _region('mlp2_pool_bf16x3', lambda: kflops * rows_()[0] + kflops_z * rows_()[1],
               lambda: 4.0 * (rows_()[0] * cin + rows_()[1] * (cin - 16 * zn) + plane.numel()),
               lambda: f'M{M}r_K{cin}_H{H}_N{D}_Z{Z}'):
    st = lib.snap_mlp2_pool_max_classes_f32(
        _p(x), M, cin, Cs, _p(index), _p(count), _p(index_z) if index_z is not None else None,
        _p(count_z) if count_z is not None else None, zlo, zn, _p(w0p), w0p.numel() * 2, _p(b0), H,
        _p(w1p), w1p.numel() * 2, _p(b1), D, int(relu_in), (3 if tuning().MLP_POOL_WIDE else 5 if tuning().MLP_POOL_NO_RING else 1) if x_split else 0, Z, ncols, _p(plane),
        _p(pvalid), _stream())
  _lib.check(st, 'snap_mlp2_pool_max_classes_f32')
  return plane, pvalid


def fill_masked_rows_(y, mask, value=0.0):
  """In place: y[m, :] = value where mask[m] == 0.  y [..., C]."""
  lib = _lib.load()
  _f32(y, 'y'); _mask(mask, 'mask')
  C = y.shape[-1]
  M = y.numel() // C
  if mask.numel() != M:
    raise ValueError('fill_masked_rows_: mask size')
  st = lib.snap_fill_masked_rows_f32(_p(y), _p(mask), M, C, float(value), _stream())
  _lib.check(st, 'snap_fill_masked_rows_f32')
  return y


def weight_standardize(w, eps=1e-10):
  """StdConv kernel standardisation.  w [KH,KW,Cin,Cout] -> same shape."""
  lib = _lib.load()
  _f32(w, 'w')
  out = torch.empty_like(w)
  K = w.shape[0] * w.shape[1] * w.shape[2]
  with _region('weight_standardize', 0.0, 16.0 * w.numel()):
    st = lib.snap_weight_standardize_f32(_p(w), _p(out), K, w.shape[3], eps, _stream())
  _lib.check(st, 'snap_weight_standardize_f32')
  return out


_WSTD_ITEM = np.dtype([('w', '<u8'), ('dws', '<u8'), ('out', '<u8'), ('K', '<i4'), ('Cout', '<i4'),
                       ('block_begin', '<i4'), ('reserved', '<i4')])   # == SnapWstdItem


def _wstd_table(ws, dwss, outs, cols=32):
  """cols: output columns per workgroup of the kernel the table is for."""
  items = np.zeros(len(ws), dtype=_WSTD_ITEM)
  blk = 0
  for i, w in enumerate(ws):
    _f32(w, 'w'); _f32(outs[i], 'out')
    K = w.shape[0] * w.shape[1] * w.shape[2]
    items[i] = (w.data_ptr(), 0 if dwss is None else _f32(dwss[i], 'dws').data_ptr(),
                outs[i].data_ptr(), K, w.shape[3], blk, 0)
    blk += (w.shape[3] + cols - 1) // cols
  table = upload_table(items, ws[0].device)
  return table, blk


def weight_standardize_multi(ws, eps=1e-10):
  """``weight_standardize`` of a list of HWIO kernels in ONE launch."""
  lib = _lib.load()
  outs = [torch.empty_like(w) for w in ws]
  table, blocks = _wstd_table(ws, None, outs, cols=32)
  with _region('weight_standardize', 0.0, 16.0 * sum(w.numel() for w in ws)):
    st = lib.snap_weight_standardize_multi_f32(_p(table), len(ws), blocks, eps, _stream())
  _lib.check(st, 'snap_weight_standardize_multi_f32')
  return outs


def weight_standardize_bwd_multi(ws, dwss, eps=1e-10):
  """d w for a list of kernels given d standardise(w), ONE launch."""
  lib = _lib.load()
  outs = [torch.empty_like(w) for w in ws]
  table, blocks = _wstd_table(ws, dwss, outs)
  st = lib.snap_weight_standardize_bwd_multi_f32(_p(table), len(ws), blocks, eps, _stream())
  _lib.check(st, 'snap_weight_standardize_bwd_multi_f32')
  return outs


# the last unit of a ResNet stage emits the statistics of relu(y) too (its FPN level reads them)
# The arithmetic of every conv / dense ("engine"):
#   'f32'    the exact f32 matrix-core path (inference, parity);
#   'bf16x3' / 'bf16x6'  f32-grade: 2 / 3 bf16 parts per operand, f32 accumulate (conv_split.hip);
#   'bf16' / 'fp16'      operands rounded to the half type, f32 accumulate -- the reference's
#                        dtype='float16' train config (train_localization.py:93, trainer.py:387-397).
# The engine is a property of the MODEL (``BaseModel(config, meta, dtype, engine=...)``: dtype selects
# it exactly as the reference's ``model_cls(config.model, meta, dtype)`` does): every apply and the
# train step enter ``engine_scope(model.engine)``, a per-thread scope that the autograd nodes carry
# into the backward thread (autograd._engine_scoped).  MATMUL_PRECISION is only the PROCESS DEFAULT
# for calls outside any scope (direct ``ops.conv2d`` calls of tools and kernel tests).
MATMUL_PRECISION = 'f32'
ENGINES = ('f32', 'bf16', 'fp16', 'bf16x3', 'bf16x6')
_ENGINE_TLS = threading.local()


def precision():
  """The engine in force: the innermost ``engine_scope`` of this thread, else the process default."""
  e = getattr(_ENGINE_TLS, 'engine', None)
  return MATMUL_PRECISION if e is None else e


@contextlib.contextmanager
def engine_scope(engine, tuning=None):
  """Run the enclosed ops on ``engine`` (None: no change) and, if given, under the ``Tuning`` object
  ``tuning``.  Per thread, re-entrant; two models of different precision interleave freely in one process."""
  if tuning is not None:
    with tuning_scope(tuning):
      with engine_scope(engine):
        yield
    return
  if engine is None:
    yield
    return
  if engine not in ENGINES:
    raise ValueError(f'engine {engine!r}: expected one of {ENGINES}')
  prev = getattr(_ENGINE_TLS, 'engine', None)
  _ENGINE_TLS.engine = engine
  try:
    yield
  finally:
    _ENGINE_TLS.engine = prev


def engine_of_dtype(dtype):
  """reference ``dtype`` -> engine: float32 -> the process default's f32-class engine ('f32' unless
  the default is itself 'bf16x3' / 'bf16x6'), float16 -> 'fp16', bfloat16 -> 'bf16'."""
  if dtype in (torch.float16, 'float16', 'fp16'):
    return 'fp16'
  if dtype in (torch.bfloat16, 'bfloat16', 'bf16'):
    return 'bf16'
  if dtype in (torch.float32, 'float32', 'f32', None):
    return None                      # f32 class: the process default (exact f32 unless configured)
  raise ValueError(f'dtype {dtype!r}: expected float32 | float16 | bfloat16')


def group_norm_stats(x, gamma, *, groups=32, eps=1e-5, relu_first=False, want_rstd=False):
  """x [N,H,W,C] -> (mu [N,C], sc [N,C]) with sc = rstd * gamma (+ rstd [N,C])."""
  lib = _lib.load()
  _f32(x, 'x'); _f32(gamma, 'gamma')
  N, H, W, C = x.shape
  HW = H * W
  wsb = lib.snap_group_norm_stats_workspace_bytes(N, HW, C, groups)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=x.device)
  mu = torch.empty((N, C), dtype=torch.float32, device=x.device)
  sc = torch.empty((N, C), dtype=torch.float32, device=x.device)
  rstd = torch.empty((N, C), dtype=torch.float32, device=x.device) if want_rstd else None
  fused = getattr(x, '_snap_gn_partial', None)
  if fused is not None and fused[2] != bool(relu_first):
    fused = getattr(x, '_snap_gn_partial_relu', None) if relu_first else None
  if fused is not None and fused[2] == bool(relu_first) and groups == 32 and tuning().USE_FUSED_GN_STATS:
    partial, tile_rows, _ = fused        # emitted by the conv that produced x
    with _region('group_norm_stats', 0.0, 4.0 * partial.numel()):
      st = lib.snap_group_norm_stats_from_partial_f32(
          _p(partial), N, HW, C, groups, eps, tile_rows, _p(gamma), _p(mu), _p(sc), _p(rstd),
          _stream(),
      )
    _lib.check(st, 'snap_group_norm_stats_from_partial_f32')
    return (mu, sc, rstd) if want_rstd else (mu, sc)
  with _region('group_norm_stats', 0.0, 8.0 * x.numel()):
    st = lib.snap_group_norm_stats_f32(
        _p(x), N, HW, C, C, groups, eps, int(relu_first), _p(gamma), _p(mu),
        _p(sc), _p(rstd), _p(ws), ws.numel() * 4, _stream(),
    )
  _lib.check(st, 'snap_group_norm_stats_f32')
  if want_rstd:
    return mu, sc, rstd
  return mu, sc


def group_norm_apply(x, mu, sc, beta, mode):
  lib = _lib.load()
  _f32(x, 'x'); _f32(mu, 'mu'); _f32(sc, 'sc'); _f32(beta, 'beta')
  N, H, W, C = x.shape
  y = torch.empty_like(x)
  st = lib.snap_group_norm_apply_f32(
      _p(x), _p(y), N, H * W, C, _p(mu), _p(sc), _p(beta), mode, _stream()
  )
  _lib.check(st, 'snap_group_norm_apply_f32')
  return y


def pad_image(x, pad_h, pad_w, pad_c=0, out=None):
  """x [..., H, W, C] -> [..., H + pad_h, W + pad_w, C + pad_c], zeros at the bottom / right / in
  the extra channels, one pass (pad_to_multiple, image_encoder.py:32-39).  ``out``: a contiguous tensor
  of that shape to write into (a slice of a joint batch: several image sets padded side by side without
  concatenating them first)."""
  lib = _lib.load()
  _f32(x, 'x')
  *lead, H, W, C = x.shape
  N = int(np.prod(lead)) if lead else 1
  shape = (*lead, H + pad_h, W + pad_w, C + pad_c)
  if out is None:
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
  else:
    y = _f32(out, 'out')
    if tuple(y.shape) != tuple(shape):
      raise ValueError(f'pad_image: out {tuple(y.shape)} != {tuple(shape)}')
  st = lib.snap_pad_image_f32(_p(x), N, H, W, C, int(pad_h), int(pad_w), int(pad_c), _p(y), _stream())
  _lib.check(st, 'snap_pad_image_f32')
  return y


def voxel_points(xy, z):
  """xy [X, Y, 2] or [B, X, Y, 2] (BEV cell centres), z [B, Z] (level heights per scene) ->
  [B, X, Y, Z, 3] voxel centres (bev_mapper.py:162-196), one pass."""
  lib = _lib.load()
  _f32(xy, 'xy'); _f32(z, 'z')
  B, Z = z.shape
  X, Y = xy.shape[-3:-1]
  batched = xy.dim() == 4
  if xy.shape[-1] != 2 or (batched and xy.shape[0] != B):
    raise ValueError('voxel_points: shapes')
  out = torch.empty((B, X, Y, Z, 3), dtype=torch.float32, device=z.device)
  st = lib.snap_voxel_points_f32(_p(xy), int(batched), _p(z), B, X * Y, Z, _p(out), _stream())
  _lib.check(st, 'snap_voxel_points_f32')
  return out


def max_pool_3x3s2(x):
  lib = _lib.load()
  _f32(x, 'x')
  N, H, W, C = x.shape
  Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
  y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
  st = lib.snap_max_pool_3x3s2_f32(_p(x), _p(y), N, H, W, C, _stream())
  _lib.check(st, 'snap_max_pool_3x3s2_f32')
  return y


# ----------------------------------------------------------------------------
# lift
# ----------------------------------------------------------------------------
def pooled_channels(feature_dim, weighted=True, use_variance=True, add_minmax=False):
  """Channels of the pooled statistics: mean | var? | max, min? | score_max?
  (pool_multiview_features, streetview_encoder.py:141-178)."""
  return feature_dim * (1 + int(use_variance) + 2 * int(add_minmax)) + int(weighted)


def pooled_stride(feature_dim, weighted=True, use_variance=True, add_minmax=False):
  return (pooled_channels(feature_dim, weighted, use_variance, add_minmax) + 3) // 4 * 4


def lift_pool(f_images, cam, Rt, points, *, K, fisheye, feature_dim, num_bins,
              depth_min_max, max_view_distance=None, weighted=True, use_variance=True,
              add_minmax=False, grid_yz=None, valid_rows_only=False, out_split=False,
              class_rows=False, tap_records=False):
  """Fused k1-k5.  f_images [B,V,h,w,C]; cam [B,V,11]; Rt [B,V,12]; points [B,N,3].

  K = 0 selects all views.  Returns pooled [B,N,stride] (mean|var|score_max|pad by default;
  see ``pooled_channels`` for the other fusion options), valid [B,N] bool.
  ``grid_yz`` = (Y, Z): the points are the voxel centres of an [X, Y, Z] grid, level fastest --
  a traversal hint (8 x 8 column blocks per XCD), the results do not depend on it.
  ``valid_rows_only``: rows of voxels no view sees are left unwritten (uninitialised memory) --
  only for consumers that read the rows of valid voxels.  ``out_split``: rows are written
  pre-split for the split-bf16 engines ([16-channel slab][hi | lo][16] bf16, returned in an f32
  container of 16 * slabs floats per row) -- ``mlp2_pool_max(x_split=True)`` takes them.
  ``class_rows`` (with out_split): a third result, classes [B,N] uint8 = 0 invalid / 1 one visible
  observation / 2 several, and rows of class 1 do not write their (all-zero) variance slabs --
  ``mlp2_pool_max(zero_slabs=(fd / 16, fd / 16))`` takes the classes as its row mask.
  ``tap_records`` (with class_rows, out_split, valid_rows_only): a fourth result, records [B,N,8] int32;
  a class-1 voxel writes NO row but its four-tap record (byte offset | flags | wi1 | wj1 | depth score) --
  ``mlp2_pool_max(gather=(f_images, records))`` blends the taps inside the MLP kernel (the lift inside
  the consumer): those rows never exist in memory.
  """
  lib = _lib.load()
  _f32(f_images, 'f_images'); _f32(cam, 'cam'); _f32(Rt, 'Rt'); _f32(points, 'points')
  B, V, h, w, C = f_images.shape
  N = points.shape[1]
  stride = pooled_stride(feature_dim, weighted, use_variance, add_minmax)
  if out_split:
    stride = (pooled_channels(feature_dim, weighted, use_variance, add_minmax) + 15) // 16 * 16
  pooled = torch.empty((B, N, stride), dtype=torch.float32, device=f_images.device)
  valid = torch.empty((B, N), dtype=torch.uint8 if class_rows else torch.bool, device=f_images.device)
  d = _lib.SnapLiftDesc(
      B, V, h, w, C, feature_dim, num_bins if weighted else 0, N, K, int(fisheye), stride,
      float(depth_min_max[0]), float(depth_min_max[1]),
      -1.0 if max_view_distance is None else float(max_view_distance),
      int(weighted), int(use_variance), int(add_minmax),
  )
  if grid_yz is not None and N % (int(grid_yz[0]) * int(grid_yz[1])) == 0:
    d.grid_y, d.grid_z = int(grid_yz[0]), int(grid_yz[1])
  d.valid_rows_only = int(bool(valid_rows_only))
  d.out_split = int(bool(out_split))
  d.class_rows = int(bool(class_rows))
  recs = None
  if tap_records:
    if not (class_rows and out_split and valid_rows_only):
      raise ValueError('lift_pool: tap_records needs class_rows, out_split and valid_rows_only')
    recs = torch.empty((B, N, 8), dtype=torch.int32, device=f_images.device)
  with _region(
      'lift_pool', 0.0, 4.0 * (f_images.numel() + points.numel() + pooled.numel())
  ):
    if recs is not None:
      st = lib.snap_lift_pool_records_f32(
          ctypes.byref(d), _p(f_images), _p(cam), _p(Rt), _p(points), _p(pooled),
          _p(valid), _p(recs), _stream(),
      )
    else:
      st = lib.snap_lift_pool_f32(
          ctypes.byref(d), _p(f_images), _p(cam), _p(Rt), _p(points), _p(pooled),
          _p(valid), _stream(),
      )
  _lib.check(st, 'snap_lift_pool_records_f32' if recs is not None else 'snap_lift_pool_f32')
  if recs is not None:
    return pooled, valid != 0, valid, recs
  if class_rows:
    return pooled, valid != 0, valid
  return pooled, valid


def _obs_desc(f_shape, N, K, fisheye, feature_dim, max_view_distance, use_variance, add_minmax, stride):
  B, V, h, w, C = f_shape
  return _lib.SnapLiftDesc(
      B, V, h, w, C, feature_dim, 0, N, K, int(fisheye), stride, 1.0, 2.0,
      -1.0 if max_view_distance is None else float(max_view_distance),
      0, int(use_variance), int(add_minmax))


def lift_observations(f_images, cam, Rt, points, *, K, fisheye, feature_dim, max_view_distance=None):
  """First pass of the depth_mlp fusion (streetview_encoder.py:263-267): per-observation features,
  un-pooled.  f_images [B,V,h,w,fd] -> obs [B,N,S,fd+4] (features | log10 depth | unit ray),
  obs_feat [B,N,S,fd], valid [B,N]; S = K (top-K order) or V."""
  lib = _lib.load()
  _f32(f_images, 'f_images'); _f32(cam, 'cam'); _f32(Rt, 'Rt'); _f32(points, 'points')
  B, V, h, w, C = f_images.shape
  if C != feature_dim:
    raise ValueError('lift_observations: f_images must carry feature_dim channels')
  N = points.shape[1]
  S = K if K else V
  obs = torch.empty((B, N, S, feature_dim + 4), dtype=torch.float32, device=f_images.device)
  feat = torch.empty((B, N, S, feature_dim), dtype=torch.float32, device=f_images.device)
  valid = torch.empty((B, N), dtype=torch.bool, device=f_images.device)
  d = _obs_desc(f_images.shape, N, K, fisheye, feature_dim, max_view_distance, True, False, 0)
  with _region('lift_pool', 0.0, 4.0 * (f_images.numel() + obs.numel() + feat.numel())):
    st = lib.snap_lift_observations_f32(ctypes.byref(d), _p(f_images), _p(cam), _p(Rt), _p(points),
                                        _p(obs), _p(feat), _p(valid), _stream())
  _lib.check(st, 'snap_lift_observations_f32')
  return obs, feat, valid


def lift_pool_observations(obs_feat, f_shape, cam, Rt, points, *, K, fisheye, feature_dim,
                           max_view_distance=None, use_variance=True, add_minmax=False):
  """Second pass: pool_multiview_features(obs_feat, visible, None, add_minmax, use_variance)
  (streetview_encoder.py:141-178) -> pooled [B,N,stride], valid [B,N]."""
  lib = _lib.load()
  _f32(obs_feat, 'obs_feat'); _f32(cam, 'cam'); _f32(Rt, 'Rt'); _f32(points, 'points')
  B, N = points.shape[:2]
  stride = pooled_stride(feature_dim, False, use_variance, add_minmax)
  pooled = torch.empty((B, N, stride), dtype=torch.float32, device=obs_feat.device)
  valid = torch.empty((B, N), dtype=torch.bool, device=obs_feat.device)
  d = _obs_desc(f_shape, N, K, fisheye, feature_dim, max_view_distance, use_variance, add_minmax, stride)
  with _region('lift_pool', 0.0, 4.0 * (obs_feat.numel() + pooled.numel())):
    st = lib.snap_lift_pool_observations_f32(ctypes.byref(d), _p(cam), _p(Rt), _p(points),
                                             _p(obs_feat), _p(pooled), _p(valid), _stream())
  _lib.check(st, 'snap_lift_pool_observations_f32')
  return pooled, valid


def project_points(cam, Rt, points, fisheye):
  lib = _lib.load()
  _f32(cam, 'cam'); _f32(Rt, 'Rt'); _f32(points, 'points')
  B, V = cam.shape[:2]
  N = points.shape[1]
  dev = cam.device
  p2d = torch.empty((B, N, V, 2), dtype=torch.float32, device=dev)
  vis = torch.empty((B, N, V), dtype=torch.bool, device=dev)
  depth = torch.empty((B, N, V), dtype=torch.float32, device=dev)
  st = lib.snap_project_points_f32(
      B, V, N, int(fisheye), _p(cam), _p(Rt), _p(points), _p(p2d), _p(vis),
      _p(depth), _stream(),
  )
  _lib.check(st, 'snap_project_points_f32')
  return p2d, vis, depth


# ----------------------------------------------------------------------------
# BEV
# ----------------------------------------------------------------------------
def vertical_pool(vol, valid, pooling='max', want_arg=False):
  """vol [..., Z, D], valid [..., Z] -> plane [..., D], pvalid [...].  want_arg (max pooling, Z <= 64, D <= 128):
  also (argz, ties) [..., D] uint8 -- the level of every maximum and how many levels hold it, which
  ``ops_bwd.vertical_pool_bwd(arg=...)`` reads instead of the volume; None where not available."""
  lib = _lib.load()
  _f32(vol, 'vol'); _mask(valid, 'valid')
  lead = vol.shape[:-2]
  Z, D = vol.shape[-2:]
  M = int(np.prod(lead))
  plane = torch.empty((*lead, D), dtype=torch.float32, device=vol.device)
  pvalid = torch.empty(lead, dtype=torch.bool, device=vol.device)
  if want_arg:
    if pooling != 'max' or Z > 64 or D > 128:
      return (*vertical_pool(vol, valid, pooling), None)
    argz = torch.empty((*lead, D), dtype=torch.uint8, device=vol.device)
    ties = torch.empty((*lead, D), dtype=torch.uint8, device=vol.device)
    with _region('vertical_pool', 0.0, 4.0 * (vol.numel() + plane.numel()) + valid.numel()):
      st = lib.snap_vertical_pool_max_arg_f32(_p(vol), _p(valid), _p(plane), _p(pvalid), _p(argz), _p(ties),
                                              M, Z, D, _stream())
    _lib.check(st, 'snap_vertical_pool_max_arg_f32')
    return plane, pvalid, (argz, ties)
  with _region('vertical_pool', 0.0, 4.0 * (vol.numel() + plane.numel()) + valid.numel()):
    st = lib.snap_vertical_pool_f32(
        _p(vol), _p(valid), _p(plane), _p(pvalid), M, Z, D, POOLING[pooling], _stream()
    )
  _lib.check(st, 'snap_vertical_pool_f32')
  return plane, pvalid


def vertical_pool_conf(vol, valid, w, bias, log_sigmoid_scores):
  """'softmax' / 'weighted' VerticalPooling.  vol [..., Z, D]; valid [..., Z]; w [D] (or
  [D,1]); bias [1] -> plane [..., D], valid_any [...], scores [..., Z], weights [..., Z]."""
  lib = _lib.load()
  _f32(vol, 'vol'); _mask(valid, 'valid'); _f32(w, 'w'); _f32(bias, 'bias')
  Z, D = vol.shape[-2:]
  lead = vol.shape[:-2]
  M = vol.numel() // (Z * D)
  plane = torch.empty((*lead, D), dtype=torch.float32, device=vol.device)
  pvalid = torch.empty(lead, dtype=torch.bool, device=vol.device)
  scores = torch.empty((*lead, Z), dtype=torch.float32, device=vol.device)
  weights = torch.empty((*lead, Z), dtype=torch.float32, device=vol.device)
  with _region('vertical_pool', 0.0, 4.0 * (vol.numel() + plane.numel())):
    st = lib.snap_vertical_pool_conf_f32(
        _p(vol), _p(valid), _p(w), _p(bias), M, Z, D, int(log_sigmoid_scores), _p(plane),
        _p(pvalid), _p(scores), _p(weights), _stream(),
    )
  _lib.check(st, 'snap_vertical_pool_conf_f32')
  return plane, pvalid, scores, weights


def plane_fuse_match(planes, valids, pooling='max', Wm=None, bm=None,
                     normalize=True, eps=1e-5, want_fused=True):
  """Fuse modality planes and apply the matching head.

  planes: list of [..., D]; valids: list of [...] bool or None (all valid).
  Returns fused [..., D], fvalid [...], matching [..., Dm] (None if Wm is None).
  """
  lib = _lib.load()
  n = len(planes)
  lead = planes[0].shape[:-1]
  D = planes[0].shape[-1]
  M = int(np.prod(lead))
  dev = planes[0].device
  for i, p in enumerate(planes):
    _f32(p, f'planes[{i}]')
    if p.shape != planes[0].shape:
      raise ValueError('plane_fuse_match: plane shapes differ')
  pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in planes])
  vv = (ctypes.c_void_p * n)(
      *[None if v is None else _mask(v, 'valid').data_ptr() for v in valids]
  )
  fused = torch.empty((*lead, D), dtype=torch.float32, device=dev) if want_fused else None
  fvalid = torch.empty(lead, dtype=torch.bool, device=dev)
  matching = None
  Dm = 0
  if Wm is not None:
    _f32(Wm, 'Wm'); _f32(bm, 'bm')
    Dm = Wm.shape[1]
    matching = torch.empty((*lead, Dm), dtype=torch.float32, device=dev)
  st = lib.snap_plane_fuse_match_f32(
      ctypes.cast(pp, ctypes.c_void_p), ctypes.cast(vv, ctypes.c_void_p), n, M,
      D, POOLING[pooling], _p(fused), _p(fvalid), _p(Wm), _p(bm), Dm,
      int(normalize), eps, _p(matching), _stream(),
  )
  _lib.check(st, 'snap_plane_fuse_match_f32')
  return fused, fvalid, matching


# ----------------------------------------------------------------------------
# pose
# ----------------------------------------------------------------------------


def sim_softmax(fq, fm, scale, clip_negative, num_valid, want_prob=False,
                want_rowstats=False, row_weight=None, math=None):
  """fq [B,Nq,Dm], fm [B,X,Y,Dm], num_valid [B] float ->
  sim [B,Nq,X,Y], chunk_stats [B,Nq,NC,2], (prob), (rowstats).  row_weight [B,Nq]: per-point
  confidence weights that replace the 1 / num_valid normalisation (add_confidence_query).
  math: 'f32' -- the contraction as the exact k-ordered fmaf chain (f32 MFMA); 'bf16x6' / 'bf16x3'
  -- on the bf16 matrix cores with 3 / 2-part split operands (f32 grade / ~2^-17 per product).
  None: 'f32' while ``MATMUL_PRECISION`` is 'f32', else 'bf16x6'."""
  lib = _lib.load()
  _f32(fq, 'fq'); _f32(fm, 'fm'); _f32(num_valid, 'num_valid')
  if row_weight is not None:
    _f32(row_weight, 'row_weight')
    if tuple(row_weight.shape) != tuple(fq.shape[:2]):
      raise ValueError('sim_softmax: row_weight must be [B, Nq]')
  B, Nq, Dm = fq.shape
  X, Y = fm.shape[1:3]
  XY = X * Y
  NC = (XY + SIM_CHUNK - 1) // SIM_CHUNK
  dev = fq.device
  sim = torch.empty((B, Nq, X, Y), dtype=torch.float32, device=dev)
  stats = torch.empty((B, Nq, NC, 2), dtype=torch.float32, device=dev)
  prob = torch.empty_like(sim) if want_prob else None
  rowstats = (
      torch.empty(lib.snap_sim_rowstats_bytes(B, Nq) // 4, dtype=torch.float32, device=dev).view(B, Nq, 2)
      if (want_prob or want_rowstats) else None
  )
  if math is None:
    math = 'f32' if precision() == 'f32' else 'bf16x6'
  parts = SPLIT_PARTS.get(math, 0)
  if parts and not (want_prob or want_rowstats) and Dm in (16, 32, 64):
    wsb = lib.snap_sim_split_workspace_bytes(B, Nq, XY, Dm, parts)
    ws = torch.empty(wsb // 2, dtype=torch.bfloat16, device=dev)
    with _region('sim_softmax', 2.0 * B * Nq * XY * Dm,
                 4.0 * (fq.numel() + fm.numel() + sim.numel() + stats.numel())):
      st = lib.snap_sim_softmax_split_f32(
          _p(fq), _p(fm), B, Nq, XY, Dm, float(scale), int(bool(clip_negative)) | (2 if tuning().SIM_GENERAL_KERNEL else 0),
          _p(num_valid), _p(row_weight), parts, _p(sim), _p(stats), _p(ws), wsb, _stream())
    _lib.check(st, 'snap_sim_softmax_split_f32')
    return sim, stats, None, None
  with _region(
      'sim_softmax', 2.0 * B * Nq * XY * Dm,
      4.0 * (fq.numel() + fm.numel() + sim.numel() + stats.numel()),
  ):
    st = lib.snap_sim_softmax_weighted_f32(
        _p(fq), _p(fm), B, Nq, XY, Dm, float(scale), int(clip_negative),
        _p(num_valid), _p(row_weight), _p(sim), _p(stats), _p(prob), _p(rowstats), _stream(),
    )
  _lib.check(st, 'snap_sim_softmax_weighted_f32')
  return sim, stats, prob, rowstats


def masked_softmax_rows(x, mask):
  """layers.masked_softmax over the last axis of x [B, N] (layers.py:38-43) -> (weights, inclusive
  CDF), both [B, N]."""
  lib = _lib.load()
  _f32(x, 'x'); _mask(mask, 'mask')
  B, N = x.shape
  w = torch.empty_like(x)
  cdf = torch.empty_like(x)
  st = lib.snap_masked_softmax_rows_f32(_p(x), _p(mask), B, N, _p(w), _p(cdf), _stream())
  _lib.check(st, 'snap_masked_softmax_rows_f32')
  return w, cdf


def confidence_head(features, valid, kernel, bias):
  """where(valid, log_sigmoid(features @ kernel + bias), 0): features [..., D], kernel [D] ->
  [...] (bev_mapper.py:292-295)."""
  lib = _lib.load()
  _f32(features, 'features'); _f32(kernel, 'kernel')
  if not torch.is_tensor(bias):
    bias = torch.tensor([float(bias)], dtype=torch.float32, device=features.device)
  bias = _f32(bias.reshape(-1)[:1].contiguous(), 'bias')      # a device scalar: no host sync
  if valid is not None:
    _mask(valid, 'valid')
  D = features.shape[-1]
  M = features.numel() // D
  out = torch.empty(features.shape[:-1], dtype=torch.float32, device=features.device)
  st = lib.snap_confidence_head_f32(_p(features), _p(valid), _p(kernel), _p(bias), M, D, _p(out),
                                    _stream())
  _lib.check(st, 'snap_confidence_head_f32')
  return out


def ransac_sample(fq, fm, chunk_stats, scale, clip_negative, S, seed=0,
                  uniforms=None, row_table=True, row_cdf=None, sim=None, row_unscale=None):
  """Draw S correspondences per scene ~ prob_points.  Returns int32 [B,S,3].
  row_table=False takes the table-free path (same samples; tests compare the two)."""
  lib = _lib.load()
  _f32(fq, 'fq'); _f32(fm, 'fm'); _f32(chunk_stats, 'chunk_stats')
  B, Nq, Dm = fq.shape
  X, Y = fm.shape[1:3]
  if uniforms is not None:
    _f32(uniforms, 'uniforms')
    if tuple(uniforms.shape) != (B, S, 2):
      raise ValueError('ransac_sample: uniforms must be [B,S,2]')
  corr = torch.empty((B, S, 3), dtype=torch.int32, device=fq.device)
  ws = None
  if row_table:
    ws = torch.empty(lib.snap_ransac_sample_workspace_bytes(B, Nq) // 4, dtype=torch.float32,
                     device=fq.device)
  with _region('ransac_sample', 0.0, 12.0 * B * S):
    if sim is not None:
      _f32(sim, 'sim'); _f32(row_unscale, 'row_unscale')
      if sim.numel() != B * Nq * X * Y or tuple(row_unscale.shape) != (B, Nq):
        raise ValueError('ransac_sample: sim must be [B,Nq,X,Y], row_unscale [B,Nq]')
    st = lib.snap_ransac_sample_sim_f32(
        _p(fq), _p(fm), _p(chunk_stats), _p(row_cdf), _p(sim), _p(row_unscale), B, Nq, X, Y, Dm,
        float(scale), int(clip_negative), S, int(seed) & 0xFFFFFFFFFFFFFFFF, _p(uniforms),
        _p(corr), _p(ws), 0 if ws is None else ws.numel() * 4, _stream(),
    )
  _lib.check(st, 'snap_ransac_sample_sim_f32')
  return corr


def poses_from_corr(corr, q_xy, P, retries, cell_size):
  """corr int32 [B,P*retries*2,3], q_xy [B,Nq,2] -> poses [B,P,3] (angle,tx,ty)."""
  lib = _lib.load()
  _chk(corr, torch.int32, 'corr'); _f32(q_xy, 'q_xy')
  B, Nq = q_xy.shape[:2]
  if corr.shape[1] != P * retries * 2:
    raise ValueError('poses_from_corr: corr size')
  poses = torch.empty((B, P, 3), dtype=torch.float32, device=corr.device)
  st = lib.snap_poses_from_corr_f32(
      _p(corr), _p(q_xy), B, Nq, P, retries, float(cell_size), _p(poses), _stream()
  )
  _lib.check(st, 'snap_poses_from_corr_f32')
  return poses


def pose_score(sim, poses, q_xy, valid_q, map_valid, cell_size, mask_oob=False):
  """sim [B,Nq,X,Y]; poses [B,P,3]; q_xy [B,Nq,2]; valid_q [B,Nq] -> scores [B,P]."""
  lib = _lib.load()
  _f32(sim, 'sim'); _f32(poses, 'poses'); _f32(q_xy, 'q_xy'); _mask(valid_q, 'valid_q')
  if map_valid is not None:
    _mask(map_valid, 'map_valid')
  B, Nq, X, Y = sim.shape
  P = poses.shape[1]
  wsb = lib.snap_pose_score_workspace_bytes(B, Nq, P, X, Y)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=sim.device)
  scores = torch.empty((B, P), dtype=torch.float32, device=sim.device)
  # algorithmic bytes: every VALID query point's score plane read once + poses + scores.
  with _region(
      'pose_score', 0.0, 4.0 * (sim.numel() + poses.numel() + scores.numel())
  ):
    st = lib.snap_pose_score_f32(
        _p(sim), _p(poses), _p(q_xy), _p(valid_q), _p(map_valid), B, Nq, X, Y, P,
        float(cell_size), int(mask_oob), _p(scores), _p(ws), ws.numel() * 4,
        _stream(),
    )
  _lib.check(st, 'snap_pose_score_f32')
  return scores


def pose_score_window_supported(X, Y, radius_cells):
  """True where ``pose_score_window`` takes a window of that radius on an [X, Y] score plane."""
  return bool(_lib.load().snap_pose_score_window_supported(int(X), int(Y), int(radius_cells)))


def pose_score_window(sim, poses, centers, radius_cells, q_xy, valid_q, cell_size):
  """``pose_score`` (mask_oob=False) for poses clustered around ``centers`` [B,3]: every pose of scene b maps
  every query point to within ``radius_cells`` cells of where centers[b] maps it (the refinement lattice of
  grid_refinement).  Same bits as ``pose_score``; a point's score plane is read as one small window."""
  lib = _lib.load()
  _f32(sim, 'sim'); _f32(poses, 'poses'); _f32(centers, 'centers'); _f32(q_xy, 'q_xy'); _mask(valid_q, 'valid_q')
  B, Nq, X, Y = sim.shape
  P = poses.shape[1]
  if tuple(centers.shape) != (B, 3):
    raise ValueError('pose_score_window: centers [B, 3]')
  wsb = lib.snap_pose_score_window_workspace_bytes(B, Nq, P, X, Y)
  ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=sim.device)
  scores = torch.empty((B, P), dtype=torch.float32, device=sim.device)
  WR, WC = min(2 * radius_cells + 3, X), min((2 * radius_cells + 9) & ~3, Y)
  with _region('pose_score', 0.0, 4.0 * (B * Nq * WR * WC + poses.numel() + scores.numel())):
    st = lib.snap_pose_score_window_f32(
        _p(sim), _p(poses), _p(centers), int(radius_cells), _p(q_xy), _p(valid_q), B, Nq, X, Y, P,
        float(cell_size), _p(scores), _p(ws), ws.numel() * 4, _stream())
  _lib.check(st, 'snap_pose_score_window_f32')
  return scores


def refine_lattice(init, offs_r, offs_p):
  """init [B,3]; offs_r [nr] (rad); offs_p [np] (m) -> [B, nr*np*np, 3]."""
  lib = _lib.load()
  _f32(init, 'init'); _f32(offs_r, 'offs_r'); _f32(offs_p, 'offs_p')
  B = init.shape[0]
  nr, np_ = offs_r.numel(), offs_p.numel()
  out = torch.empty((B, nr * np_ * np_, 3), dtype=torch.float32, device=init.device)
  st = lib.snap_refine_lattice_f32(
      _p(init), _p(offs_r), _p(offs_p), B, nr, np_, _p(out), _stream()
  )
  _lib.check(st, 'snap_refine_lattice_f32')
  return out


def argmax_rows(scores, start=0):
  """First-index argmax of scores[:, start:] -> int32 [B] (relative to start)."""
  lib = _lib.load()
  _f32(scores, 'scores')
  B, P = scores.shape
  idx = torch.empty((B,), dtype=torch.int32, device=scores.device)
  st = lib.snap_argmax_rows_f32(_p(scores), B, P, start, _p(idx), _stream())
  _lib.check(st, 'snap_argmax_rows_f32')
  return idx


# ----------------------------------------------------------------------------
# generic grid operators (snap/utils/grids.py:116-153)
# ----------------------------------------------------------------------------
def interpolate_nd(array, points, valid_array=None):
  """array [s_0..s_{n-1}, D], points [K, n] (corner-origin coordinates) -> values [K, D],
  valid [K] (grids.py:116-137: linear, 'nearest' extension, zero-weight-invalid-tap rule)."""
  lib = _lib.load()
  _f32(array, 'array'); _f32(points, 'points')
  K, n = points.shape
  if array.dim() != n + 1 or not 1 <= n <= 3:
    raise ValueError(f'interpolate_nd: array {tuple(array.shape)} vs points [K, {n}]')
  D = array.shape[-1]
  size = (ctypes.c_int32 * n)(*array.shape[:n])
  if valid_array is not None:
    _mask(valid_array, 'valid_array')
    if tuple(valid_array.shape) != tuple(array.shape[:n]):
      raise ValueError('interpolate_nd: valid_array shape')
  values = torch.empty((K, D), dtype=torch.float32, device=array.device)
  valid = torch.empty((K,), dtype=torch.bool, device=array.device)
  st = lib.snap_interpolate_nd_f32(_p(array), size, n, D, _p(valid_array), _p(points), K,
                                   _p(values), _p(valid), _stream())
  _lib.check(st, 'snap_interpolate_nd_f32')
  return values, valid


def expectation_nd(pdf, extent):
  """pdf [..., *extent] -> expected (fractional) index [..., n] (grids.py:148-153)."""
  lib = _lib.load()
  _f32(pdf, 'pdf')
  n = len(extent)
  if tuple(pdf.shape[-n:]) != tuple(extent) or not 1 <= n <= 3:
    raise ValueError(f'expectation_nd: pdf {tuple(pdf.shape)} vs extent {tuple(extent)}')
  lead = pdf.shape[:-n]
  rows = int(np.prod(lead)) if len(lead) else 1
  size = (ctypes.c_int32 * n)(*extent)
  out = torch.empty((rows, n), dtype=torch.float32, device=pdf.device)
  st = lib.snap_expectation_nd_f32(_p(pdf), rows, size, n, _p(out), _stream())
  _lib.check(st, 'snap_expectation_nd_f32')
  return out.reshape(*lead, n)


# ----------------------------------------------------------------------------
# exhaustive voting
# ----------------------------------------------------------------------------
def rotate_templates(feat, valid, tfm, num_rotations, cell_size, want_tw=True):
  """feat [H,W,D], valid [H,W], tfm [R/4,4] -> templates [R,H,W,D], tvalid [R,H,W],
  tw [H,W,D,R] (None unless want_tw), cw [H,W,1,R], tcount [R]."""
  lib = _lib.load()
  _f32(feat, 'feat'); _mask(valid, 'valid'); _f32(tfm, 'tfm')
  H, W, D = feat.shape
  R = num_rotations
  dev = feat.device
  templates = torch.empty((R, H, W, D), dtype=torch.float32, device=dev)
  tvalid = torch.empty((R, H, W), dtype=torch.bool, device=dev)
  tw = torch.empty((H, W, D, R), dtype=torch.float32, device=dev) if want_tw else None
  cw = torch.empty((H, W, 1, R), dtype=torch.float32, device=dev)
  tcount = torch.empty((R,), dtype=torch.float32, device=dev)
  st = lib.snap_rotate_templates_f32(
      _p(feat), _p(valid), _p(tfm), H, W, D, R, float(cell_size), _p(templates),
      _p(tvalid), _p(tw), _p(cw), _p(tcount), _stream(),
  )
  _lib.check(st, 'snap_rotate_templates_f32')
  return templates, tvalid, tw, cw, tcount


def pad_map(m, mvalid):
  lib = _lib.load()
  _f32(m, 'map'); _mask(mvalid, 'mvalid')
  H, W, D = m.shape
  dev = m.device
  mp = torch.empty((3 * H - 2, 3 * W - 2, D), dtype=torch.float32, device=dev)
  mvp = torch.empty((3 * H - 2, 3 * W - 2), dtype=torch.float32, device=dev)
  st = lib.snap_pad_map_f32(_p(m), _p(mvalid), H, W, D, _p(mp), _p(mvp), _stream())
  _lib.check(st, 'snap_pad_map_f32')
  return mp, mvp


def voting_fft_supported(R, H, W, D, Hm, Wm):
  """True where ``voting_fft`` takes the geometry (<= 1024 transform points per axis)."""
  return _lib.load().snap_voting_fft_workspace_bytes(R, H, W, D, Hm, Wm) > 0


def voting_fft_workspace_bytes(R, H, W, D, Hm, Wm):
  """Bytes of device workspace one ``voting_fft`` call allocates (0: unsupported geometry)."""
  return int(_lib.load().snap_voting_fft_workspace_bytes(R, H, W, D, Hm, Wm))


def voting_fft(templates, tvalid, m, mvalid, tcount, threshold, use_overlap=True):
  """Frequency-domain template matching: templates [R,H,W,D], tvalid [R,H,W], m [Hm,Wm,D],
  mvalid [Hm,Wm], tcount [R] -> finalised scores [R, 3Hm-1-H, 3Wm-1-W]."""
  lib = _lib.load()
  _f32(templates, 'templates'); _f32(m, 'map'); _f32(tcount, 'tcount')
  if use_overlap:
    _mask(tvalid, 'tvalid'); _mask(mvalid, 'mvalid')
  R, H, W, D = templates.shape
  Hm, Wm = m.shape[:2]
  if m.shape[2] != D:
    raise ValueError(f'voting_fft: map has {m.shape[2]} channels, templates {D}')
  nbytes = lib.snap_voting_fft_workspace_bytes(R, H, W, D, Hm, Wm)
  if nbytes == 0:
    raise ValueError(f'voting_fft: unsupported geometry R={R} H={H} W={W} D={D} map {Hm}x{Wm}')
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=templates.device)
  scores = torch.empty((R, 3 * Hm - 1 - H, 3 * Wm - 1 - W), dtype=torch.float32, device=templates.device)
  with _region('voting_fft', flops=0.0, nbytes=4.0 * (templates.numel() + m.numel() + scores.numel())):
    st = lib.snap_voting_fft_f32(
        _p(templates), _p(tvalid) if use_overlap else None, _p(m), _p(mvalid) if use_overlap else None,
        _p(tcount), R, H, W, D, Hm, Wm, float(threshold), int(bool(use_overlap)), _p(ws), nbytes, _p(scores),
        _stream(),
    )
  _lib.check(st, 'snap_voting_fft_f32')
  return scores


def voting_fft_rotated(feat, valid, tfm, cell_size, m, mvalid, num_rotations, min_overlap=0.05):
  """exhaustive_pose_voting in the frequency domain with the templates sampled on the fly: feat [H,H,D],
  valid [H,H], tfm [R/4,4] (first-quadrant template transforms), m [Hm,Wm,D], mvalid [Hm,Wm] ->
  scores [R, 3Hm-1-H, 3Wm-1-H]."""
  lib = _lib.load()
  _f32(feat, 'feat'); _mask(valid, 'valid'); _f32(tfm, 'tfm'); _f32(m, 'map'); _mask(mvalid, 'mvalid')
  H, W, D = feat.shape
  R = int(num_rotations)
  Hm, Wm = m.shape[:2]
  if H != W or R % 4 or tuple(tfm.shape) != (R // 4, 4) or m.shape[2] != D:
    raise ValueError('voting_fft_rotated: square query plane, R % 4 == 0, tfm [R/4, 4], equal channel counts')
  nbytes = lib.snap_voting_fft_workspace_bytes(R, H, H, D, Hm, Wm)
  if nbytes == 0:
    raise ValueError(f'voting_fft_rotated: unsupported geometry R={R} H={H} D={D} map {Hm}x{Wm}')
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=feat.device)
  scores = torch.empty((R, 3 * Hm - 1 - H, 3 * Wm - 1 - H), dtype=torch.float32, device=feat.device)
  with _region('voting_fft', flops=0.0, nbytes=4.0 * (R * feat.numel() + m.numel() + scores.numel())):
    st = lib.snap_voting_fft_rotated_f32(
        _p(feat), _p(valid), _p(tfm), float(cell_size), _p(m), _p(mvalid), R, H, D, Hm, Wm, float(min_overlap),
        _p(ws), nbytes, _p(scores), _stream())
  _lib.check(st, 'snap_voting_fft_rotated_f32')
  return scores


def template_finalize(raw, cnt, tcount, R, threshold, use_overlap=True):
  """raw, cnt [Ho,Wo,Rp] -> scores [R,Ho,Wo]."""
  lib = _lib.load()
  _f32(raw, 'raw'); _f32(tcount, 'tcount')
  if cnt is not None:
    _f32(cnt, 'cnt')
  Ho, Wo, Rp = raw.shape
  scores = torch.empty((R, Ho, Wo), dtype=torch.float32, device=raw.device)
  st = lib.snap_template_finalize_f32(
      _p(raw), _p(cnt), _p(tcount), Ho, Wo, R, Rp, float(threshold),
      int(use_overlap), _p(scores), _stream(),
  )
  _lib.check(st, 'snap_template_finalize_f32')
  return scores


# Module attributes of the switches: aliases of the tuning in force (read) / the process default (write).
class _OpsModule(type(sys)):
  pass


for _name in _TUNING_DEFAULTS:
  setattr(_OpsModule, _name, property(
      lambda self, _n=_name: getattr(tuning(), _n),
      lambda self, v, _n=_name: object.__setattr__(_DEFAULT_TUNING, _n, v)))
sys.modules[__name__].__class__ = _OpsModule
