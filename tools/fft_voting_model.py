"""Index-exact numpy model of the frequency-domain exhaustive voting (snap_amd/csrc/voting_fft.hip).

Test infrastructure / design aid, NOT a product path: it restates, stage by stage and index by
index, what the HIP kernels do -- in-place mixed-radix decimation-in-frequency forward transforms
(natural order in, digit-reversed order out), pointwise products in the permuted order, in-place
decimation-in-time inverse transforms (the exact conjugate transposes of the forward stages:
digit-reversed in, natural order out) -- so that the algebra (channel pairs packed as complex
numbers, rotation pairs packed for the overlap count, the reference's un-flipped q_valid quirk,
pose_exhaustive_voting.py:72-104) is checked against oracle/voting.py on the CPU before the kernels
are trusted with it (tests/test_host_logic.py::test_fft_voting_model_equals_the_direct_form).
"""
import numpy as np


def fft_size(n):
  """Smallest N >= n of the form 2^a or 3 * 2^a (a >= 2): the kernel's supported lengths."""
  best = None
  for base in (1, 3):
    N = base * 4
    while N < n:
      N *= 2
    best = N if best is None or N < best else best
  return best


def radices(N):
  """Stage radices of the forward transform, in order: an optional 3, an optional 2, then 4s."""
  out = []
  n = N
  if n % 3 == 0:
    out.append(3); n //= 3
  k = 0
  while n > 1:
    assert n % 2 == 0, N
    n //= 2; k += 1
  if k % 2:
    out.append(2)
  out += [4] * (k // 2)
  return out


def _dft_small(x, R, sign):
  """R-point DFT of the R rows of x (exp(sign * 2 pi i q m / R))."""
  q = np.arange(R)
  w = np.exp(sign * 2j * np.pi * np.outer(q, q) / R)
  return np.tensordot(w, x, axes=(1, 0))


def dif_forward(x, axis=0):
  """In-place DIF stages along `axis`: out[perm(k)] = sum_n x[n] exp(-2 pi i n k / N)."""
  x = np.moveaxis(np.array(x, dtype=np.complex128), axis, 0).copy()
  N = x.shape[0]
  L = N
  for R in radices(N):
    M = L // R
    for b0 in range(0, N, L):
      for k in range(M):
        idx = b0 + k + M * np.arange(R)
        y = _dft_small(x[idx], R, -1)
        tw = np.exp(-2j * np.pi * k * np.arange(R) / L)
        x[idx] = y * tw.reshape((R,) + (1,) * (x.ndim - 1))
    L = M
  return np.moveaxis(x, 0, axis)


def dit_inverse(x, axis=0):
  """The conjugate transpose of dif_forward, stage by stage in reverse: N * ifft of the natural-order
  spectrum whose digit-reversed image is x; output in natural order."""
  x = np.moveaxis(np.array(x, dtype=np.complex128), axis, 0).copy()
  N = x.shape[0]
  rs = radices(N)
  Ls = []
  L = N
  for R in rs:
    Ls.append(L); L //= R
  for R, L in zip(reversed(rs), reversed(Ls)):
    M = L // R
    for b0 in range(0, N, L):
      for k in range(M):
        idx = b0 + k + M * np.arange(R)
        tw = np.exp(+2j * np.pi * k * np.arange(R) / L)
        y = x[idx] * tw.reshape((R,) + (1,) * (x.ndim - 1))
        x[idx] = _dft_small(y, R, +1)
  return np.moveaxis(x, 0, axis)


def template_matching_fft(q, q_valid, m, m_valid, min_overlap=0.05, dtype=np.float32):
  """template_matching (padded mode) as the kernels compute it.  q [R,H,W,D], m [Hm,Wm,D]."""
  R, H, W, D = q.shape
  Hm, Wm = m.shape[:2]
  Hp, Wp = 3 * Hm - 2, 3 * Wm - 2
  Ho, Wo = Hp - H + 1, Wp - W + 1
  N1, N2 = fft_size(Hp), fft_size(Wp)
  P = (D + 1) // 2
  qq = np.zeros((R, H, W, 2 * P)); qq[..., :D] = q
  mm = np.zeros((Hm, Wm, 2 * P)); mm[..., :D] = m
  # edge padding folded into the load: clamp(a - (Hm - 1), 0, Hm - 1)
  ia = np.clip(np.arange(Hp) - (Hm - 1), 0, Hm - 1)
  ib = np.clip(np.arange(Wp) - (Wm - 1), 0, Wm - 1)
  zm = np.zeros((N1, N2, P), np.complex128)
  zm[:Hp, :Wp] = mm[ia][:, ib][..., 0::2] + 1j * mm[ia][:, ib][..., 1::2]
  Zm = dif_forward(dif_forward(zm, 0), 1)             # [k1', k2', p] (both axes digit-reversed)
  zq = np.zeros((R, N1, N2, P), np.complex128)
  zq[:, :H, :W] = qq[..., 0::2] + 1j * qq[..., 1::2]
  X1 = dif_forward(zq, 1)                              # K1: along i
  X2 = dif_forward(X1, 2)                              # K2: along j
  S = (np.conj(X2) * Zm[None]).sum(-1)                 # [R, k1', k2']
  Y = dit_inverse(S, 2)                                # K2 tail: along k2 -> b
  out = dit_inverse(Y, 1).real / (N1 * N2)             # K3: along k1 -> a
  scores = out[:, :Ho, :Wo]
  if min_overlap is not None:
    # overlap count: rotations r and r + R2 packed as one complex template; the count filter is the
    # 180-degree rotated mask (the reference passes q_valid un-flipped to a true convolution)
    R2 = (R + 1) // 2
    cw = q_valid[:, ::-1, ::-1].astype(np.float64)
    zc = np.zeros((R2, N1, N2), np.complex128)
    zc[:, :H, :W] = cw[:R2]
    zc[: R - R2, :H, :W] += 1j * cw[R2:]
    mv = np.zeros((N1, N2), np.complex128)
    mv[Hm - 1:2 * Hm - 1, Wm - 1:2 * Wm - 1] = m_valid
    Zv = dif_forward(dif_forward(mv, 0), 1)
    C = np.conj(dif_forward(dif_forward(zc, 1), 2)) * Zv[None]
    c = dit_inverse(dit_inverse(C, 2), 1) / (N1 * N2)
    cnt = np.concatenate([c.real, -c.imag[: R - R2]], 0)[:, :Ho, :Wo]
    cnt = np.rint(cnt)
    scores = np.where(cnt > min_overlap * H * W, scores, -np.inf)
  with np.errstate(divide='ignore', invalid='ignore'):
    scores = scores / q_valid.sum((-1, -2), keepdims=True)
  return scores.astype(dtype)
