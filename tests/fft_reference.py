"""Test infrastructure: float64 ``numpy.fft`` restatement of ``template_matching``
(``/root/reference/snap/models/pose_exhaustive_voting.py:72-104``), independent of every HIP
formulation -- the checker of the frequency-domain voting at sizes the direct-form oracle
(``oracle/voting.py``, a sliding-window sum) cannot reach in reasonable time (256 x 256 cells, N = 768
transform points per axis).  Pinned to ``oracle/voting.template_matching`` at the small geometries in
the CPU suite (tests/test_oracle.py::test_fft64_template_matching_equals_the_oracle); nothing in
``snap_amd/`` imports it.

    scores[r, a, b] = sum_{i,j,d} q[r,i,j,d] m_pad[a+i, b+j, d]
                    = IFFT2( sum_d conj(FFT2(q[r,..,d])) FFT2(m_pad[..,d]) )[a, b]      (zero-padded to F >= 3 Hm - 2)

The overlap count is the same correlation of the 0 / 1 masks (the reference's un-flipped q_valid
quirk, :97-99: the map mask against the 180-degree rotated template mask); in float64 its error is
~1e-10, so rounding to the nearest integer is exact.
"""
import math

import numpy as np
import scipy.fft


def template_matching_fft64(q, q_valid, m, m_valid, min_overlap=0.05):
  """q [R,H,W,D], q_valid [R,H,W], m [Hm,Wm,D], m_valid [Hm,Wm] -> float64 [R, 2Hm-2+Hm-H+1.., ..]
  (the padded mode, do_padding=True: [R, 3Hm-2-H+1, 3Wm-2-W+1])."""
  R, H, W, D = q.shape
  Hm, Wm = m.shape[:2]
  m_pad = np.pad(m.astype(np.float64), ((Hm - 1,) * 2, (Wm - 1,) * 2, (0, 0)), mode='edge')
  P0, P1 = m_pad.shape[:2]
  Ho, Wo = P0 - H + 1, P1 - W + 1
  F0, F1 = scipy.fft.next_fast_len(P0, real=True), scipy.fft.next_fast_len(P1, real=True)
  Fm = scipy.fft.rfft2(m_pad, s=(F0, F1), axes=(0, 1))                      # [F0, F1/2+1, D]
  scores = np.empty((R, Ho, Wo), np.float64)
  for r in range(R):
    Fq = scipy.fft.rfft2(q[r].astype(np.float64), s=(F0, F1), axes=(0, 1))
    scores[r] = scipy.fft.irfft2((np.conj(Fq) * Fm).sum(-1), s=(F0, F1), axes=(0, 1))[:Ho, :Wo]
  if min_overlap is not None:
    mv = np.pad(m_valid.astype(np.float64), ((Hm - 1,) * 2, (Wm - 1,) * 2), mode='constant')
    Fv = scipy.fft.rfft2(mv, s=(F0, F1))
    qv_flip = q_valid[:, ::-1, ::-1].astype(np.float64)
    thr = min_overlap * math.prod(q_valid.shape[-2:])
    for r in range(R):
      Fqv = scipy.fft.rfft2(qv_flip[r], s=(F0, F1))
      cnt = np.rint(scipy.fft.irfft2(np.conj(Fqv) * Fv, s=(F0, F1))[:Ho, :Wo])
      scores[r] = np.where(cnt > thr, scores[r], -np.inf)
  with np.errstate(divide='ignore', invalid='ignore'):
    scores = scores / q_valid.sum((-1, -2), keepdims=True).astype(np.float64)
  return scores
