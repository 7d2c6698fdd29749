// The bilinear sample of one rotated-template cell (snap/models/pose_exhaustive_voting.py:44-60:
// templates_t_grid @ grid_xy / cell_size, grids.interpolate_nd with the NaN-mask validity rule),
// shared by rotate_templates_kernel (voting.hip), the validity-only pass and the first transform of
// the frequency-domain voting (voting_fft_body.h), so that all three produce the same bits.
// Plain C++ (also compiled by g++ into the tests' CPU emulation of voting_fft_body.h).
#ifndef SNAP_CSRC_ROTATE_SAMPLE_H_
#define SNAP_CSRC_ROTATE_SAMPLE_H_

#include <math.h>
#include <stdint.h>

#ifndef SNAP_ROT_DEV
#ifdef __HIPCC__
#define SNAP_ROT_DEV __device__ __forceinline__
#else
#define SNAP_ROT_DEV static inline
#endif
#endif

struct SnapRotSample {
  bool ok;                  // inside the grid AND all four taps valid (even those with zero weight)
  int i0, i1, j0, j1;       // clamped tap rows / columns
  float w00, w01, w10, w11; // bilinear weights of (i0,j0), (i0,j1), (i1,j0), (i1,j1)
};

// tfm = (cos, sin, tx, ty) of templates_t_grid for this rotation; (si, sj) = source cell.
// The geometry alone (ok = inside the grid; the taps' validity not looked at): the taps are clamped into the
// grid, so a caller may fetch them -- features and validity -- before it knows whether the sample counts.
SNAP_ROT_DEV SnapRotSample snap_rot_geom(const float* tfm, int si, int sj, int H, int W, float cell) {
  SnapRotSample r;
  const float c = tfm[0], s = tfm[1], tx = tfm[2], ty = tfm[3];
  // cell centre in metres, transformed, back to cell units.
  const float gx = ((float)si + 0.5f) * cell, gy = ((float)sj + 0.5f) * cell;
  const float xm = (c * gx - s * gy) + tx;
  const float ym = (s * gx + c * gy) + ty;
  const float u = xm / cell, v = ym / cell;
  r.ok = (u >= 0.f) && (u < (float)H) && (v >= 0.f) && (v < (float)W);
  const float cu = u - 0.5f, cv = v - 0.5f;
  const float fu = floorf(cu), fv = floorf(cv);
  const float wu1 = cu - fu, wu0 = 1.f - wu1, wv1 = cv - fv, wv0 = 1.f - wv1;
  r.i0 = (int)fminf(fmaxf(fu, 0.f), (float)(H - 1));
  r.i1 = (int)fminf(fmaxf(fu + 1.f, 0.f), (float)(H - 1));
  r.j0 = (int)fminf(fmaxf(fv, 0.f), (float)(W - 1));
  r.j1 = (int)fminf(fmaxf(fv + 1.f, 0.f), (float)(W - 1));
  r.w00 = wu0 * wv0; r.w01 = wu0 * wv1; r.w10 = wu1 * wv0; r.w11 = wu1 * wv1;
  return r;
}

SNAP_ROT_DEV SnapRotSample snap_rot_sample(const float* tfm, int si, int sj, int H, int W, float cell,
                                           const uint8_t* valid) {
  SnapRotSample r = snap_rot_geom(tfm, si, sj, H, W, cell);
  // NaN-mask validity: every tap must be valid, even with zero weight.
  r.ok = r.ok && valid[r.i0 * W + r.j0] && valid[r.i0 * W + r.j1] && valid[r.i1 * W + r.j0] && valid[r.i1 * W + r.j1];
  return r;
}

// the interpolated value, in the summation order of rotate_templates_kernel
SNAP_ROT_DEV float snap_rot_mix(const SnapRotSample& r, float a00, float a01, float a10, float a11) {
  return ((r.w00 * a00 + r.w01 * a01) + r.w10 * a10) + r.w11 * a11;
}

// source cell (si, sj) of destination (di, dj) under jnp.rot90(quarter, k, axes=(2, 1)) (H == W)
SNAP_ROT_DEV void snap_rot90_source(int k, int di, int dj, int H, int W, int* si, int* sj) {
  if (k == 0) { *si = di; *sj = dj; }
  else if (k == 1) { *si = H - 1 - dj; *sj = di; }
  else if (k == 2) { *si = H - 1 - di; *sj = W - 1 - dj; }
  else { *si = dj; *sj = H - 1 - di; }
}

#endif  // SNAP_CSRC_ROTATE_SAMPLE_H_
