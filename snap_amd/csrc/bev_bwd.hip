// Backward kernels of the lift and BEV stages (training path).
//   * lift_pool backward: d pooled[B,N,257] -> d f_images[B,V,h,w,C]  (VJP of
//     streetview_encoder.py:69-178; geometry carries no gradient).  Two forms:
//     - deterministic (default of the training step): every (voxel, selected view) observation
//       becomes a RECORD (its gradient vector + the four bilinear taps); the records are sorted by
//       the image pixel of their first tap (stable radix sort: ties stay in voxel order) and one
//       half-wave per pixel GATHERS the records of the <= 4 lists that can touch it, in that
//       order -- no atomics, bitwise reproducible, and the image gradient is written once;
//     - scatter: the taps are scatter-added with hardware fp32 atomics (order-dependent sums).
//   * vertical pooling backward (VJP of bev_mapper.py:56-88; max: equal split among ties,
//     as jnp.max's VJP does)
//   * modality fusion + matching head backward (VJP of bev_mapper.py:225-252,284-291,
//     layers.py:45-52)
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace {

// ------------------------------- lift ----------------------------------------
struct LiftBwdArgs {
  SnapLiftDesc d;
  const float* f;
  const float* cam;
  const float* Rt;
  const float* pts;
  const float* dpooled;
  float* df;
  // deterministic form: record slot = voxel * nsel + selection rank
  float* rec_vec;        // [slots][feature_dim]: d f of the observation (tap weights not applied)
  float* rec_hdr;        // [slots][12]: w00 w01 w10 w11 | g(bin0) g(bin1) | bin0|bin1<<16 | i0|i1<<16 | j0|j1<<16
  unsigned* keys;        // [slots]: pixel id of tap (i0, j0), or npix for an empty slot
  unsigned* count;       // [npix + 1]: records per key
  unsigned npix;
  const float* obs_in;   // MODE 2: the (corrected) observations [B, N, S, fd]
  float* dobs;           // MODE 2: their gradient
};

struct ProjB {
  float pi, pj, depth, dist;
  bool vis;
};

__device__ __forceinline__ ProjB project_b(const float* __restrict__ cam,
                                           const float* __restrict__ Rt, float px, float py,
                                           float pz, int fisheye) {
  const float eps = 1e-3f;
  float pv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r0 = Rt[0 * 3 + i], r1 = Rt[1 * 3 + i], r2 = Rt[2 * 3 + i];
    const float tinv = -((r0 * Rt[9] + r1 * Rt[10]) + r2 * Rt[11]);
    pv[i] = tinv + ((r0 * px + r1 * py) + r2 * pz);
  }
  ProjB o;
  o.depth = pv[2];
  bool valid = pv[2] >= eps;
  const float z = fmaxf(pv[2], eps);
  float x = pv[0] / z, y = pv[1] / z;
  if (fisheye) {
    const float radius2 = x * x + y * y;
    const bool in_center = radius2 < eps * eps;
    const float radius = sqrtf(in_center ? eps * eps : radius2);
    const float theta = atanf(radius);
    const float t2 = theta * theta;
    const float offset = (cam[6] * t2 + cam[7] * (t2 * t2)) + cam[8] * (t2 * t2 * t2);
    float dist = (offset + 1.f) * theta / radius;
    dist = in_center ? 1.f : dist;
    x *= dist;
    y *= dist;
    valid = valid && (in_center || ((radius < cam[10]) && (dist > 0.f)));
  }
  x = x * cam[2] + cam[4];
  y = y * cam[3] + cam[5];
  valid = valid && (x >= 0.f) && (x < cam[0]) && (y >= 0.f) && (y < cam[1]);
  o.pi = y;
  o.pj = x;
  o.vis = valid;
  const float dx = px - Rt[9], dy = py - Rt[10], dz = pz - Rt[11];
  o.dist = sqrtf((dx * dx + dy * dy) + dz * dz);
  return o;
}

struct TapsB {
  int i0, i1, j0, j1;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ TapsB taps_b(float pi, float pj, int h, int w, int selective) {
  TapsB t;
  float ci = pi - 0.5f, cj = pj - 0.5f;
  if (selective) {
    ci = fmaxf(fminf(ci, (float)(h - 1)), 0.f);
    cj = fmaxf(fminf(cj, (float)(w - 1)), 0.f);
  }
  const float fi = floorf(ci), fj = floorf(cj);
  const float wi1 = ci - fi, wj1 = cj - fj;
  const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;
  t.i0 = (int)fminf(fmaxf(fi, 0.f), (float)(h - 1));
  t.i1 = (int)fminf(fmaxf(fi + 1.f, 0.f), (float)(h - 1));
  t.j0 = (int)fminf(fmaxf(fj, 0.f), (float)(w - 1));
  t.j1 = (int)fminf(fmaxf(fj + 1.f, 0.f), (float)(w - 1));
  t.w00 = wi0 * wj0;
  t.w01 = wi0 * wj1;
  t.w10 = wi1 * wj0;
  t.w11 = wi1 * wj1;
  return t;
}

__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}

// MODE 0: scatter with float atomics (default fusion options only)
// MODE 1: records of the deterministic form, every fusion option (desc->weighted / use_variance /
//         add_minmax: mean | var? | max, min? | score_max? -- pool_multiview_features :141-178)
// MODE 2: VJP of the POOLING of given observations (depth_mlp fusion, second pass): a.obs_in
//         [B, N, S, fd] -> a.dobs [B, N, S, fd]; no image taps
// MODE 3: records of given observation gradients (depth_mlp fusion, first pass): only the taps,
//         keys and counts are produced -- the record vectors ARE the caller's d obs [slots][fd]
template <int KMAX, int MODE>
__global__ __launch_bounds__(256) void lift_pool_bwd_kernel(const LiftBwdArgs a) {
  const SnapLiftDesc& d = a.d;
  const int hl = threadIdx.x & 31;
  const int64_t gv = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int64_t total = (int64_t)d.B * d.N;
  if (gv >= total) return;
  const int b = (int)(gv / d.N);
  const int fd = d.feature_dim;
  const bool all_views = d.K == 0;
  const int nsel = all_views ? d.V : d.K;
  constexpr bool RECORDS = MODE == 1 || MODE == 3;
  if constexpr (RECORDS) {       // every slot of this voxel starts empty (key = npix sorts last)
    if (hl < nsel) a.keys[gv * nsel + hl] = a.npix;
  }
  const bool weighted = MODE == 0 || (MODE == 1 && d.weighted);
  const bool use_var = MODE == 0 || d.use_variance;
  const bool minmax = MODE != 0 && d.add_minmax;
  const float* p = a.pts + gv * 3;
  const float px = p[0], py = p[1], pz = p[2];

  ProjB pr;
  pr.pi = pr.pj = pr.depth = 0.f;
  pr.dist = INFINITY;
  pr.vis = false;
  if (hl < d.V)
    pr = project_b(a.cam + ((int64_t)b * d.V + hl) * 11, a.Rt + ((int64_t)b * d.V + hl) * 12, px, py,
                   pz, d.fisheye);
  float key_d = (hl < d.V && pr.vis) ? pr.dist : INFINITY;
  int key_i = (hl < d.V) ? hl : 1000 + hl;
  int sel[KMAX];
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    if (r >= nsel) { sel[r] = 0; continue; }
    if (all_views) { sel[r] = r; continue; }
    float bd = key_d;
    int bi = key_i;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 32);
      const int oi = __shfl_xor(bi, o, 32);
      if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    sel[r] = bi;
    if (hl == bi) { key_d = INFINITY; key_i = 1000 + hl; }
  }

  // ---- recompute the forward quantities -----------------------------------
  f32x4 feat[KMAX];
  float score[KMAX], wb1[KMAX];
  int bin0[KMAX], bin1[KMAX], view[KMAX];
  TapsB tp[KMAX];
  bool ok[KMAX];
  bool any = false;
  const float log_range = logf(d.depth_max / d.depth_min);
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    feat[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    score[r] = 0.f; wb1[r] = 0.f; bin0[r] = bin1[r] = 0; view[r] = 0;
    ok[r] = false;
    tp[r] = TapsB{0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
    if (r >= nsel) continue;
    const int v = sel[r];
    view[r] = v;
    const float pi = __shfl(pr.pi, v, 32);
    const float pj = __shfl(pr.pj, v, 32);
    const float depth = __shfl(pr.depth, v, 32);
    const bool vis = __shfl((int)pr.vis, v, 32) != 0;
    ok[r] = vis;
    if (!vis) continue;
    any = true;
    if constexpr (MODE == 2) {                    // the observation is given
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = hl + 32 * e;
        if (c < fd) feat[r][e] = a.obs_in[(gv * nsel + r) * (int64_t)fd + c];
      }
      continue;
    }
    tp[r] = taps_b(pi, pj, d.h, d.w, all_views ? 0 : 1);
    if constexpr (MODE == 3) continue;            // taps only
    const TapsB& t = tp[r];
    const float* img = a.f + ((int64_t)b * d.V + v) * d.h * d.w * d.C;
    const float* r00 = img + ((int64_t)t.i0 * d.w + t.j0) * d.C;
    const float* r01 = img + ((int64_t)t.i0 * d.w + t.j1) * d.C;
    const float* r10 = img + ((int64_t)t.i1 * d.w + t.j0) * d.C;
    const float* r11 = img + ((int64_t)t.i1 * d.w + t.j1) * d.C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = hl + 32 * e;  // lane owns channels hl, hl+32, ..: every access below is
      if (c < fd)                 // one contiguous 128-byte line per half-wave
        feat[r][e] = ((t.w00 * r00[c] + t.w01 * r01[c]) + t.w10 * r10[c]) + t.w11 * r11[c];
    }
    if (!weighted) continue;      // (scores = None: no depth-score bins in f_images)
    const float dc = fminf(fmaxf(depth, d.depth_min), d.depth_max);
    const float tt = logf(dc / d.depth_min) / log_range;
    const float index = 0.5f + tt * (float)(d.num_bins - 1);
    const float c = index - 0.5f;
    const float fl = floorf(c);
    wb1[r] = c - fl;
    bin0[r] = (int)fminf(fmaxf(fl, 0.f), (float)(d.num_bins - 1));
    bin1[r] = (int)fminf(fmaxf(fl + 1.f, 0.f), (float)(d.num_bins - 1));
    const int c0 = fd + bin0[r], c1 = fd + bin1[r];
    const float s0 = ((t.w00 * r00[c0] + t.w01 * r01[c0]) + t.w10 * r10[c0]) + t.w11 * r11[c0];
    const float s1 = ((t.w00 * r00[c1] + t.w01 * r01[c1]) + t.w10 * r10[c1]) + t.w11 * r11[c1];
    score[r] = (1.f - wb1[r]) * s0 + wb1[r] * s1;
  }
  if constexpr (MODE == 2) {
    if (!any) {                                   // pooled == 0 (masked): no gradient
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (r < nsel)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = hl + 32 * e;
            if (c < fd) a.dobs[(gv * nsel + r) * (int64_t)fd + c] = 0.f;
          }
      return;
    }
  } else {
    if (!any) return;  // pooled == 0 (masked): no gradient
  }
  if constexpr (MODE == 3) {
    // the record vector of slot (voxel, r) is the caller's d obs row: header, key and count only
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (!ok[r] || hl != 0) continue;
      const TapsB& t = tp[r];
      const int64_t rid = gv * nsel + r;
      float* h = a.rec_hdr + rid * 12;
      reinterpret_cast<f32x4*>(h)[0] = f32x4{t.w00, t.w01, t.w10, t.w11};
      reinterpret_cast<f32x4*>(h)[1] = f32x4{0.f, 0.f, __int_as_float(0), __int_as_float(t.i0 | (t.i1 << 16))};
      h[8] = __int_as_float(t.j0 | (t.j1 << 16));
      const unsigned key = (unsigned)((((int64_t)b * d.V + view[r]) * d.h + t.i0) * d.w + t.j0);
      a.keys[rid] = key;
      atomicAdd(a.count + key, 1u);
    }
    return;
  }

  // ---- pooling weights: softmax(where = visible, initial = 0) of the depth scores, or 1 / count
  float m = 0.f, smax = -INFINITY;
  float wgt[KMAX], den = 0.f;
  if (weighted) {
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (ok[r]) { m = fmaxf(m, score[r]); smax = fmaxf(smax, score[r]); }
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      wgt[r] = ok[r] ? expf(score[r] - m) : 0.f;
      den += wgt[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      wgt[r] = ok[r] ? 1.f : 0.f;
      den += wgt[r];
    }
  }
  f32x4 mean = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    wgt[r] = wgt[r] / den;
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] += wgt[r] * feat[r][e];
  }

  // ---- upstream gradients: mean | var? | max, min? | score_max? ------------------------------
  const float* g = a.dpooled + gv * d.out_stride;
  const int o_var = fd, o_mm = fd * (1 + (use_var ? 1 : 0));
  const int o_smax = fd * (1 + (use_var ? 1 : 0) + (minmax ? 2 : 0));
  f32x4 dmean = {0.f, 0.f, 0.f, 0.f}, dvar = {0.f, 0.f, 0.f, 0.f};
  f32x4 dmx = {0.f, 0.f, 0.f, 0.f}, dmn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = hl + 32 * e;
    if (c < fd) {
      dmean[e] = g[c];
      if (use_var) dvar[e] = g[o_var + c];
      if (minmax) { dmx[e] = g[o_mm + c]; dmn[e] = g[o_mm + fd + c]; }
    }
  }
  // max / min over the visible views: the cotangent is split equally among ties (jnp.max's VJP)
  f32x4 vmx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, vmn = {INFINITY, INFINITY, INFINITY, INFINITY};
  f32x4 nmx = {0.f, 0.f, 0.f, 0.f}, nmn = {0.f, 0.f, 0.f, 0.f};
  if (minmax) {
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (ok[r])
#pragma unroll
        for (int e = 0; e < 4; ++e) { vmx[e] = fmaxf(vmx[e], feat[r][e]); vmn[e] = fminf(vmn[e], feat[r][e]); }
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (ok[r])
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nmx[e] += feat[r][e] == vmx[e] ? 1.f : 0.f;
          nmn[e] += feat[r][e] == vmn[e] ? 1.f : 0.f;
        }
  }
  const float dsmax = weighted ? g[o_smax] : 0.f;
  int nmax = 0;
#pragma unroll
  for (int r = 0; r < KMAX; ++r) nmax += (ok[r] && score[r] == smax) ? 1 : 0;

  // d w_k = sum_c f_kc dmean_c + (f_kc - mean_c)^2 dvar_c   (half-wave reduction; weighted only)
  float dw[KMAX], dwbar = 0.f;
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    float t = 0.f;
    if (ok[r] && weighted) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dl = feat[r][e] - mean[e];
        t += feat[r][e] * dmean[e] + (dl * dl) * dvar[e];
      }
    }
    dw[r] = weighted ? half_sum(t) : 0.f;
    dwbar += wgt[r] * dw[r];
  }
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    if (!ok[r]) {
      if constexpr (MODE == 2) {
        if (r < nsel)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = hl + 32 * e;
            if (c < fd) a.dobs[(gv * nsel + r) * (int64_t)fd + c] = 0.f;
          }
      }
      continue;
    }
    const float ds = weighted ? wgt[r] * (dw[r] - dwbar) + ((score[r] == smax) ? dsmax / (float)nmax : 0.f) : 0.f;
    float dfe[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = wgt[r] * dmean[e] + 2.f * wgt[r] * (feat[r][e] - mean[e]) * dvar[e];
      if (minmax) {
        if (feat[r][e] == vmx[e]) t += dmx[e] / nmx[e];
        if (feat[r][e] == vmn[e]) t += dmn[e] / nmn[e];
      }
      dfe[e] = t;
    }
    const TapsB& t = tp[r];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = hl + 32 * e;
        if (c < fd) a.dobs[(gv * nsel + r) * (int64_t)fd + c] = dfe[e];
      }
      continue;
    }
    if constexpr (MODE == 1) {
      const int64_t rid = gv * nsel + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = hl + 32 * e;
        if (c < fd) a.rec_vec[rid * fd + c] = dfe[e];
      }
      if (hl == 0) {
        float* h = a.rec_hdr + rid * 12;
        reinterpret_cast<f32x4*>(h)[0] = f32x4{t.w00, t.w01, t.w10, t.w11};
        reinterpret_cast<f32x4*>(h)[1] =
            f32x4{(1.f - wb1[r]) * ds, wb1[r] * ds, __int_as_float(bin0[r] | (bin1[r] << 16)),
                  __int_as_float(t.i0 | (t.i1 << 16))};
        h[8] = __int_as_float(t.j0 | (t.j1 << 16));
        const unsigned key = (unsigned)((((int64_t)b * d.V + view[r]) * d.h + t.i0) * d.w + t.j0);
        a.keys[rid] = key;
        atomicAdd(a.count + key, 1u);            // (integer: order-independent)
      }
      continue;
    }
    float* img = a.df + ((int64_t)b * d.V + view[r]) * d.h * d.w * d.C;
    float* r00 = img + ((int64_t)t.i0 * d.w + t.j0) * d.C;
    float* r01 = img + ((int64_t)t.i0 * d.w + t.j1) * d.C;
    float* r10 = img + ((int64_t)t.i1 * d.w + t.j0) * d.C;
    float* r11 = img + ((int64_t)t.i1 * d.w + t.j1) * d.C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = hl + 32 * e;
      if (c >= fd) continue;
      const float df = dfe[e];
      unsafeAtomicAdd(r00 + c, t.w00 * df);
      unsafeAtomicAdd(r01 + c, t.w01 * df);
      unsafeAtomicAdd(r10 + c, t.w10 * df);
      unsafeAtomicAdd(r11 + c, t.w11 * df);
    }
    // score taps: lanes 0..7 = (tap, bin)
    if (hl < 8) {
      const int st = hl >> 1, sb = hl & 1;
      float* rt = st == 0 ? r00 : (st == 1 ? r01 : (st == 2 ? r10 : r11));
      const float wt = st == 0 ? t.w00 : (st == 1 ? t.w01 : (st == 2 ? t.w10 : t.w11));
      const float wbin = sb ? wb1[r] : (1.f - wb1[r]);
      unsafeAtomicAdd(rt + fd + (sb ? bin1[r] : bin0[r]), wt * wbin * ds);
    }
  }
}

// ---------------------------------------------------------------------------
// Batched record producer for the default fusion options (weighted, variance, no min / max; <= 4
// selected views): the structure of the forward's lift_pool_batched_kernel (lift.hip).
//   phase A  lane = voxel: projection into the views, selection, tap geometry and depth bins run
//            once per voxel on a full lane set (in the half-wave-per-voxel kernel above 4 of 32
//            lanes do that work while 28 wait); records go to LDS.
//   phase B  lane = channel quad: the workgroup's 256 voxels, ordered by their number of
//            observations so that the two half-waves of a wave walk voxels of one class, are taken
//            one per half-wave: float4 taps (one 512-byte row per tap), pooling weights, the VJP of
//            mean / variance / score_max, one float4 of the record vector per lane.
// Same record format, same keys and counts as lift_pool_bwd_kernel<KMAX, 1>: the sort and the
// gather pass do not change.  Deterministic like it (integer counts, fixed-order sums); the
// half-wave reduction of d w_k sums the channels in another order than the strided kernel, so
// the two producers agree to rounding, not bit for bit.
template <int KMAX>
__global__ __launch_bounds__(256) void lift_pool_bwd_batched_kernel(const LiftBwdArgs a) {
  __shared__ __attribute__((aligned(16))) int recs[256][KMAX][4];   // pixel key | packed | wi1 | wj1
  __shared__ float wbs[256][KMAX];
  __shared__ int ijs[256][KMAX];                                    // i0 | j0 << 16
  __shared__ __attribute__((aligned(16))) float vhdr[256][4];       // (-, -, voxel or -1, observations)
  __shared__ int cls_cnt[4][KMAX + 2];
  __shared__ uint8_t order[256];
  const SnapLiftDesc& d = a.d;
  const int hl = threadIdx.x & 31;
  const int hw = threadIdx.x >> 5;
  const int fd = d.feature_dim;
  const int nq = fd >> 2;
  const bool all_views = d.K == 0;
  const int nsel = all_views ? d.V : d.K;
  const int64_t total = (int64_t)d.B * d.N;
  const float log_range = logf(d.depth_max / d.depth_min);

  // ---------------- phase A: lane = voxel ----------------
  {
    const int64_t gv = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = gv < total;
    const int b = live ? (int)(gv / d.N) : 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) {
      const float* p = a.pts + gv * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    float kd[KMAX], kpi[KMAX], kpj[KMAX], kdep[KMAX];
    int kv[KMAX];
#pragma unroll
    for (int r = 0; r < KMAX; ++r) { kd[r] = INFINITY; kpi[r] = kpj[r] = kdep[r] = 0.f; kv[r] = -1; }
    for (int v = 0; v < d.V; ++v) {
      const ProjB pr = project_b(a.cam + ((int64_t)b * d.V + v) * 11, a.Rt + ((int64_t)b * d.V + v) * 12, px, py,
                                 pz, d.fisheye);
      const bool vis = live && pr.vis;
      if (all_views) {
#pragma unroll
        for (int r = 0; r < KMAX; ++r)
          if (r == v) { kv[r] = vis ? v : -1; kpi[r] = pr.pi; kpj[r] = pr.pj; kdep[r] = pr.depth; }
      } else if (vis) {
        // stable insertion (strict <): equal distances keep the lower view index first
        float cd = pr.dist, cpi = pr.pi, cpj = pr.pj, cdep = pr.depth;
        int cv = v;
#pragma unroll
        for (int r = 0; r < KMAX; ++r) {
          if (r < nsel && cd < kd[r]) {
            const float td = kd[r], tpi = kpi[r], tpj = kpj[r], tdep = kdep[r];
            const int tv = kv[r];
            kd[r] = cd; kpi[r] = cpi; kpj[r] = cpj; kdep[r] = cdep; kv[r] = cv;
            cd = td; cpi = tpi; cpj = tpj; cdep = tdep; cv = tv;
          }
        }
      }
    }
    int nvis = 0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nsel || kv[r] < 0) continue;
      const TapsB t = taps_b(kpi[r], kpj[r], d.h, d.w, all_views ? 0 : 1);
      // the 1-D weights the products of taps_b are built from (phase B rebuilds the same products)
      float ci = kpi[r] - 0.5f, cj = kpj[r] - 0.5f;
      if (!all_views) {
        ci = fmaxf(fminf(ci, (float)(d.h - 1)), 0.f);
        cj = fmaxf(fminf(cj, (float)(d.w - 1)), 0.f);
      }
      const float wi1 = ci - floorf(ci), wj1 = cj - floorf(cj);
      const float dc = fminf(fmaxf(kdep[r], d.depth_min), d.depth_max);
      const float tt = logf(dc / d.depth_min) / log_range;
      const float index = 0.5f + tt * (float)(d.num_bins - 1);
      const float c = index - 0.5f;
      const float fl = floorf(c);
      const int b0 = (int)fminf(fmaxf(fl, 0.f), (float)(d.num_bins - 1));
      const int b1 = (int)fminf(fmaxf(fl + 1.f, 0.f), (float)(d.num_bins - 1));
      int* rec = recs[threadIdx.x][nvis];
      rec[0] = (int)(unsigned)((((int64_t)b * d.V + kv[r]) * d.h + t.i0) * d.w + t.j0);
      rec[1] = kv[r] | ((t.i1 != t.i0) << 8) | ((t.j1 != t.j0) << 9) | (b0 << 10) | (b1 << 18);
      rec[2] = __float_as_int(wi1);
      rec[3] = __float_as_int(wj1);
      wbs[threadIdx.x][nvis] = c - fl;
      ijs[threadIdx.x][nvis] = t.i0 | (t.j0 << 16);
      ++nvis;
    }
    if (live) {                    // slots without an observation: key = npix sorts last
      for (int r = nvis; r < nsel; ++r) a.keys[gv * nsel + r] = a.npix;
    }
    vhdr[threadIdx.x][2] = __int_as_float(live ? (int)gv : -1);      // (B * N < 2^31: checked by the launcher)
    vhdr[threadIdx.x][3] = __int_as_float(nvis);
    const int key = live ? nvis : KMAX + 1;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < KMAX + 2; ++c) {
      const unsigned long long m = __ballot(key == c);
      if (key == c) rank = __popcll(m & ((1ull << ln) - 1ull));
      if (ln == 0) cls_cnt[wv][c] = __popcll(m);
    }
    __syncthreads();
    int pos = rank;
#pragma unroll
    for (int c = 0; c < KMAX + 2; ++c)
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (c < key || (c == key && w < wv)) pos += cls_cnt[w][c];
    order[pos] = (uint8_t)threadIdx.x;
  }
  __syncthreads();

  // ---------------- phase B: lane = channel quad ----------------
  const char* fb = reinterpret_cast<const char*>(a.f);
  const uint32_t Cb = (uint32_t)d.C * 4u, Wb = (uint32_t)d.w * Cb, fdb = (uint32_t)fd * 4u;
  const uint32_t lane_off = 16u * hl;
  const bool lane_on = hl < nq;
  for (int j = 0; j < 32; ++j) {
    const int v = order[8 * j + hw];                    // (half-wave uniform)
    const int gvi = __float_as_int(vhdr[v][2]);
    const int nvis = __float_as_int(vhdr[v][3]);
    if (gvi < 0 || nvis == 0) continue;                 // pooled == 0 (masked): no gradient, no record
    const int64_t gv = gvi;
    if (nvis == 1) {
      // ONE observation (72 % of the observed voxels of a four-view map, every voxel of a query): its pooling
      // weight is e / e = 1 exactly (the forward takes the same shortcut), so mean = f, the variance term
      // 2 w (f - mean) dvar vanishes and d f = d mean -- the record's vector IS the row of dpooled: nothing is
      // gathered from the image and nothing written but the header; bit 31 of the sort key (above the bits the
      // sort looks at) tells the sum pass to read the vector there -- and d score = w (dw - w dw) + d score_max
      // = d score_max.
      if (hl == 0) {
        const float ds = (a.dpooled + gv * d.out_stride)[2 * fd];
        const int64_t rid = gv * nsel;
        const i32x4 q4 = *reinterpret_cast<const i32x4*>(recs[v][0]);
        const int pk = q4[1];
        const float wi1 = __int_as_float(q4[2]), wj1 = __int_as_float(q4[3]);
        const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;
        const int ij = ijs[v][0];
        const int i0 = ij & 0xffff, j0 = ij >> 16;
        const int i1 = i0 + ((pk >> 8) & 1), j1 = j0 + ((pk >> 9) & 1);
        const float wb1 = wbs[v][0];
        float* h = a.rec_hdr + rid * 12;
        reinterpret_cast<f32x4*>(h)[0] = f32x4{wi0 * wj0, wi0 * wj1, wi1 * wj0, wi1 * wj1};
        reinterpret_cast<f32x4*>(h)[1] =
            f32x4{(1.f - wb1) * ds, wb1 * ds, __int_as_float(((pk >> 10) & 0xff) | (((pk >> 18) & 0xff) << 16)),
                  __int_as_float(i0 | (i1 << 16))};
        h[8] = __int_as_float(j0 | (j1 << 16));
        const unsigned key = (unsigned)q4[0];
        a.keys[rid] = key | 0x80000000u;
        atomicAdd(a.count + key, 1u);            // (integer: order-independent)
      }
      continue;
    }
    f32x4 feat[KMAX];
    float score[KMAX], w4[KMAX][4];
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nvis) continue;
      const i32x4 q4 = *reinterpret_cast<const i32x4*>(recs[v][r]);
      const int pk = q4[1];
      const float wi1 = __int_as_float(q4[2]), wj1 = __int_as_float(q4[3]);
      const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;
      const float w00 = wi0 * wj0, w01 = wi0 * wj1, w10 = wi1 * wj0, w11 = wi1 * wj1;
      w4[r][0] = w00; w4[r][1] = w01; w4[r][2] = w10; w4[r][3] = w11;
      const uint32_t o00 = (uint32_t)q4[0] * Cb;
      const uint32_t o01 = o00 + ((pk >> 9) & 1 ? Cb : 0u);
      const uint32_t o10 = o00 + ((pk >> 8) & 1 ? Wb : 0u);
      const uint32_t o11 = o10 + (o01 - o00);
      const uint32_t c0 = fdb + ((pk >> 10) & 0xff) * 4u, c1 = fdb + ((pk >> 18) & 0xff) * 4u;
      const float t00 = *reinterpret_cast<const float*>(fb + (o00 + c0));
      const float t01 = *reinterpret_cast<const float*>(fb + (o01 + c0));
      const float t10 = *reinterpret_cast<const float*>(fb + (o10 + c0));
      const float t11 = *reinterpret_cast<const float*>(fb + (o11 + c0));
      const float u00 = *reinterpret_cast<const float*>(fb + (o00 + c1));
      const float u01 = *reinterpret_cast<const float*>(fb + (o01 + c1));
      const float u10 = *reinterpret_cast<const float*>(fb + (o10 + c1));
      const float u11 = *reinterpret_cast<const float*>(fb + (o11 + c1));
      feat[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lane_on) {
        const f32x4 a00 = *reinterpret_cast<const f32x4*>(fb + (o00 + lane_off));
        const f32x4 a01 = *reinterpret_cast<const f32x4*>(fb + (o01 + lane_off));
        const f32x4 a10 = *reinterpret_cast<const f32x4*>(fb + (o10 + lane_off));
        const f32x4 a11 = *reinterpret_cast<const f32x4*>(fb + (o11 + lane_off));
#pragma unroll
        for (int e = 0; e < 4; ++e) feat[r][e] = ((w00 * a00[e] + w01 * a01[e]) + w10 * a10[e]) + w11 * a11[e];
      }
      const float wb1 = wbs[v][r];
      const float s0 = ((w00 * t00 + w01 * t01) + w10 * t10) + w11 * t11;
      const float s1 = ((w00 * u00 + w01 * u01) + w10 * u10) + w11 * u11;
      score[r] = (1.f - wb1) * s0 + wb1 * s1;
    }
    // pooling weights: softmax(where = visible, initial = 0) of the depth scores
    float m = 0.f, smax = -INFINITY;
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (r < nvis) { m = fmaxf(m, score[r]); smax = fmaxf(smax, score[r]); }
    float wgt[KMAX], den = 0.f;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      wgt[r] = r < nvis ? expf(score[r] - m) : 0.f;
      den += wgt[r];
    }
    f32x4 mean = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      wgt[r] = wgt[r] / den;
      if (r < nvis) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mean[e] += wgt[r] * feat[r][e];
      }
    }
    // upstream gradients: mean | var | score_max
    const float* g = a.dpooled + gv * d.out_stride;
    f32x4 dmean = {0.f, 0.f, 0.f, 0.f}, dvar = {0.f, 0.f, 0.f, 0.f};
    if (lane_on) {
      dmean = *reinterpret_cast<const f32x4*>(g + 4 * hl);
      dvar = *reinterpret_cast<const f32x4*>(g + fd + 4 * hl);
    }
    const float dsmax = g[2 * fd];
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) nmax += (r < nvis && score[r] == smax) ? 1 : 0;
    // d w_k = sum_c f_kc dmean_c + (f_kc - mean_c)^2 dvar_c   (half-wave reduction)
    float dw[KMAX], dwbar = 0.f;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      float t = 0.f;
      if (r < nvis) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = feat[r][e] - mean[e];
          t += feat[r][e] * dmean[e] + (dl * dl) * dvar[e];
        }
      }
      dw[r] = r < nvis ? half_sum(t) : 0.f;            // (half-wave uniform condition)
      dwbar += wgt[r] * dw[r];
    }
    const int b = (int)(gv / d.N);
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nvis) continue;
      const float ds = wgt[r] * (dw[r] - dwbar) + ((score[r] == smax) ? dsmax / (float)nmax : 0.f);
      const int64_t rid = gv * nsel + r;
      if (lane_on) {
        f32x4 dfe;
#pragma unroll
        for (int e = 0; e < 4; ++e) dfe[e] = wgt[r] * dmean[e] + 2.f * wgt[r] * (feat[r][e] - mean[e]) * dvar[e];
        *reinterpret_cast<f32x4*>(a.rec_vec + rid * fd + 4 * hl) = dfe;
      }
      if (hl == 0) {
        const int pk = recs[v][r][1];
        const int ij = ijs[v][r];
        const int i0 = ij & 0xffff, j0 = ij >> 16;
        const int i1 = i0 + ((pk >> 8) & 1), j1 = j0 + ((pk >> 9) & 1);
        const float wb1 = wbs[v][r];
        float* h = a.rec_hdr + rid * 12;
        reinterpret_cast<f32x4*>(h)[0] = f32x4{w4[r][0], w4[r][1], w4[r][2], w4[r][3]};
        reinterpret_cast<f32x4*>(h)[1] =
            f32x4{(1.f - wb1) * ds, wb1 * ds, __int_as_float(((pk >> 10) & 0xff) | (((pk >> 18) & 0xff) << 16)),
                  __int_as_float(i0 | (i1 << 16))};
        h[8] = __int_as_float(j0 | (j1 << 16));
        const unsigned key = (unsigned)recs[v][r][0];
        a.keys[rid] = key;
        atomicAdd(a.count + key, 1u);            // (integer: order-independent)
      }
    }
    (void)b;
  }
}

// The image gradient from the sorted records, every record read ONCE.  A record whose first tap is pixel
// (i0, j0) touches (i0 + di, j0 + dj), di, dj in {0, 1} (a tap clamped at the border falls back on the first
// row / column).  Pass 1 (one half-wave per KEY pixel) walks that pixel's list in sorted order and keeps four
// sums, one per offset (di, dj); pass 2 adds, for every pixel, the four sums that land on it -- in the fixed
// order (i-1, j-1), (i-1, j), (i, j-1), (i, j): deterministic.  (The earlier one-pass gather walked the lists
// of the four neighbours from every pixel: each 560-byte record read four times, 7 GB per C3 step.)
// Lane hl owns channels 4 hl .. 4 hl + 3 and depth bin hl.
struct LiftGatherArgs {
  SnapLiftDesc d;
  const unsigned* vals;      // record slot of every sorted entry
  const unsigned* start;     // [npix + 1] exclusive prefix of the per-key counts
  const unsigned* keys;      // sorted keys (bit 31: the record's vector is the row of dpooled)
  const float* rec_vec;
  const float* rec_hdr;
  const float* dpooled;
  int nsel;
  float* taps;               // [npix][4 offsets][C]
  float* df;
  unsigned npix;
};

__global__ __launch_bounds__(256) void lift_pool_bwd_taps_kernel(const LiftGatherArgs a) {
  const SnapLiftDesc& d = a.d;
  const int hl = threadIdx.x & 31;
  const int64_t p = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (p >= (int64_t)a.npix) return;
  const int fd = d.feature_dim;
  const bool lane_on = 4 * hl < fd;
  f32x4 acc[4];
  float accb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { acc[s] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[s] = 0.f; }
  // the record slots of up to 32 entries by one coalesced load (lane = entry, handed round by shuffles),
  // U records in flight
  constexpr int U = 8;
  const unsigned k0 = a.start[p], k1 = a.start[p + 1];
  for (unsigned base = k0; base < k1; base += 32) {
    const unsigned slot = a.vals[min(base + (unsigned)hl, k1 - 1u)];
    const unsigned kflag = a.keys[min(base + (unsigned)hl, k1 - 1u)];
    const unsigned nb = min(32u, k1 - base);
    for (unsigned b0 = 0; b0 < nb; b0 += U) {
      f32x4 w[U], g[U], v[U];
      float jp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned rid = (unsigned)__shfl((int)slot, (int)min(b0 + u, nb - 1u), 32);
        const bool pooled_row = __shfl((int)kflag, (int)min(b0 + u, nb - 1u), 32) < 0;
        const float* h = a.rec_hdr + (int64_t)rid * 12;
        w[u] = reinterpret_cast<const f32x4*>(h)[0];
        g[u] = reinterpret_cast<const f32x4*>(h)[1];
        jp[u] = h[8];
        const float* vp = pooled_row ? a.dpooled + (int64_t)(rid / (unsigned)a.nsel) * d.out_stride
                                     : a.rec_vec + (int64_t)rid * fd;
        v[u] = lane_on ? *reinterpret_cast<const f32x4*>(vp + 4 * hl) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {              // accumulated in list order: deterministic
        if (b0 + u >= nb) break;
        const int ip = __float_as_int(g[u][3]), jq = __float_as_int(jp[u]);
        const bool ei = (ip >> 16) != (ip & 0xffff), ej = (jq >> 16) != (jq & 0xffff);
        // taps 00, 01, 10, 11 -> offset (di & ei, dj & ej); weights of taps that share an offset add up
        float ws[4];
        ws[0] = ((w[u][0] + (ej ? 0.f : w[u][1])) + (ei ? 0.f : w[u][2])) + ((ei || ej) ? 0.f : w[u][3]);
        ws[1] = (ej ? w[u][1] : 0.f) + ((ej && !ei) ? w[u][3] : 0.f);
        ws[2] = (ei ? w[u][2] : 0.f) + ((ei && !ej) ? w[u][3] : 0.f);
        ws[3] = (ei && ej) ? w[u][3] : 0.f;
        const int bins = __float_as_int(g[u][2]);
        const float gb = (hl == (bins & 0xffff) ? g[u][0] : 0.f) + (hl == (bins >> 16) ? g[u][1] : 0.f);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[s][e] += ws[s] * v[u][e];
          accb[s] += ws[s] * gb;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float* o = a.taps + (p * 4 + s) * d.C;
    if (lane_on) *reinterpret_cast<f32x4*>(o + 4 * hl) = acc[s];
    if (hl < d.num_bins) o[fd + hl] = accb[s];
  }
}

__global__ __launch_bounds__(256) void lift_pool_bwd_combine_kernel(const LiftGatherArgs a) {
  const SnapLiftDesc& d = a.d;
  const int C4 = d.C >> 2;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)a.npix * C4) return;
  const int64_t p = t / C4;
  const int q = (int)(t - p * C4);
  const int hw = d.h * d.w;
  const int rem = (int)(p % hw);
  const int i = rem / d.w, j = rem - i * d.w;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 3; s >= 0; --s) {
    const int di = s >> 1, dj = s & 1;
    if (i - di < 0 || j - dj < 0) continue;
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.taps + ((p - (int64_t)di * d.w - dj) * 4 + s) * d.C + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += v[e];
  }
  *reinterpret_cast<f32x4*>(a.df + p * d.C + 4 * q) = acc;
}

// ------------------------------- vertical pool --------------------------------
__global__ __launch_bounds__(256) void vertical_pool_bwd_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid,
    const float* __restrict__ dplane, float* __restrict__ dvol, int64_t M, int Z, int D,
    int pooling) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  const int nq = D >> 2;
  const uint8_t* vv = vvalid + m * Z;
  const float* base = vol + m * Z * D;
  float* dbase = dvol + m * Z * D;
  for (int q = hl; q < nq; q += 32) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dplane + m * D + 4 * q);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 cnt = {0.f, 0.f, 0.f, 0.f};
    int nvalid = 0;
    for (int z = 0; z < Z; ++z) {
      if (!vv[z]) continue;
      ++nvalid;
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (v[e] > best[e]) { best[e] = v[e]; cnt[e] = 1.f; }
        else if (v[e] == best[e]) cnt[e] += 1.f;
      }
    }
    for (int z = 0; z < Z; ++z) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (vv[z]) {
        if (pooling == SNAP_POOL_MAX) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[e] == best[e]) ? g[e] / cnt[e] : 0.f;
        } else if (pooling == SNAP_POOL_SUM) {
          o = g;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = g[e] / (float)nvalid;
        }
      }
      *reinterpret_cast<f32x4*>(dbase + (int64_t)z * D + 4 * q) = o;
    }
  }
}

// Max pooling with the forward's record of where the maximum sits (snap_vertical_pool_max_arg_f32): a level
// receives the gradient iff it is the recorded one -- the volume is not read at all (2 x 2 GB per C3 step in
// the kernel above).  Channels whose maximum is held by several levels (ties > 1: the gradient is shared, as
// jnp.max's VJP does) take the two passes of the kernel above for their quad: the same values, bit for bit.
__global__ __launch_bounds__(256) void vertical_pool_max_bwd_arg_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, const uint8_t* __restrict__ argz,
    const uint8_t* __restrict__ ties, const float* __restrict__ dplane, float* __restrict__ dvol, int64_t M,
    int Z, int D) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  const int nq = D >> 2;
  const uint8_t* vv = vvalid + m * Z;
  const float* base = vol + m * Z * D;
  float* dbase = dvol + m * Z * D;
  for (int q = hl; q < nq; q += 32) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dplane + m * D + 4 * q);
    const unsigned pa = *reinterpret_cast<const unsigned*>(argz + m * D + 4 * q);
    const unsigned pc = *reinterpret_cast<const unsigned*>(ties + m * D + 4 * q);
    const bool shared = ((pc >> 0) & 255) > 1 || ((pc >> 8) & 255) > 1 || ((pc >> 16) & 255) > 1 || (pc >> 24) > 1;
    if (!shared) {
      for (int z = 0; z < Z; ++z) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (vv[z]) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (((pc >> (8 * e)) & 255) == 1 && (int)((pa >> (8 * e)) & 255) == z) ? g[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dbase + (int64_t)z * D + 4 * q) = o;
      }
      continue;
    }
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 cnt = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < Z; ++z) {
      if (!vv[z]) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (v[e] > best[e]) { best[e] = v[e]; cnt[e] = 1.f; }
        else if (v[e] == best[e]) cnt[e] += 1.f;
      }
    }
    for (int z = 0; z < Z; ++z) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (vv[z]) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] == best[e]) ? g[e] / cnt[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dbase + (int64_t)z * D + 4 * q) = o;
    }
  }
}

// ------------------------------- fuse + matching --------------------------------
struct FuseBwdArgs {
  const float* planes[4];
  const uint8_t* valids[4];
  float* dplanes[4];
  int num_planes;
  int64_t M;
  int D, pooling;
  const float* Wm;
  const float* bm;
  int Dm, normalize;
  float eps;
  const float* dmatching;  // [M,Dm] or null
  const float* dfused;     // [M,D] or null
  float* dy;               // [M,Dm] grad w.r.t. the Dense output (pre-normalisation)
};

constexpr int FB_MAXQ = 2;

__global__ __launch_bounds__(256) void plane_fuse_match_bwd_kernel(const FuseBwdArgs a) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= a.M) return;
  const int nq = a.D >> 2;
  // recompute the fused vector and the routing weights
  f32x4 fused[FB_MAXQ], x[4][FB_MAXQ];
  bool pv[4];
  int count = 0;
  const float init = a.pooling == SNAP_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int i = 0; i < FB_MAXQ; ++i) fused[i] = f32x4{init, init, init, init};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    pv[p] = false;
#pragma unroll
    for (int i = 0; i < FB_MAXQ; ++i) x[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p >= a.num_planes) continue;
    pv[p] = a.valids[p] ? (a.valids[p][m] != 0) : true;
    if (!pv[p]) continue;
    ++count;
#pragma unroll
    for (int i = 0; i < FB_MAXQ; ++i) {
      const int q = hl + 32 * i;
      if (q < nq) {
        x[p][i] = *reinterpret_cast<const f32x4*>(a.planes[p] + m * a.D + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          fused[i][e] = a.pooling == SNAP_POOL_MAX ? fmaxf(fused[i][e], x[p][i][e])
                                                   : fused[i][e] + x[p][i][e];
      }
    }
  }
  const bool any = count > 0;
#pragma unroll
  for (int i = 0; i < FB_MAXQ; ++i) {
    if (!any) fused[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (any && a.pooling == SNAP_POOL_MEAN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fused[i][e] = fused[i][e] / (float)count;
    }
  }
  // ---- matching head backward: dy (lane j) ----------------------------------
  const int j = hl;
  float dyj = 0.f;
  if (a.dmatching) {
    float y = (j < a.Dm) ? a.bm[j] : 0.f;
#pragma unroll
    for (int i = 0; i < FB_MAXQ; ++i)
      for (int q = 0; q < 32; ++q) {
        const int cq = q + 32 * i;
        if (cq >= nq) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = __shfl(fused[i][e], q, 32);
          if (j < a.Dm) y += xv * a.Wm[(int64_t)(4 * cq + e) * a.Dm + j];
        }
      }
    const float dz = (j < a.Dm && any) ? a.dmatching[m * a.Dm + j] : 0.f;
    if (a.normalize) {
      const float nrm = sqrtf(half_sum((j < a.Dm) ? y * y : 0.f));
      if (nrm >= a.eps) {
        const float z = y / nrm;
        const float zdz = half_sum((j < a.Dm) ? z * dz : 0.f);
        dyj = (dz - z * zdz) / nrm;
      }
    } else {
      dyj = dz;
    }
    if (j < a.Dm) a.dy[m * a.Dm + j] = dyj;
  }
  // ---- d fused = Wm dy (+ dfused) and routing to the planes -------------------
#pragma unroll
  for (int i = 0; i < FB_MAXQ; ++i) {
    const int q = hl + 32 * i;
    f32x4 df = {0.f, 0.f, 0.f, 0.f};
    if (a.dmatching) {
      for (int jj = 0; jj < a.Dm; ++jj) {
        const float dv = __shfl(dyj, jj, 32);
        if (q < nq) {
#pragma unroll
          for (int e = 0; e < 4; ++e) df[e] += a.Wm[(int64_t)(4 * q + e) * a.Dm + jj] * dv;
        }
      }
    }
    if (q >= nq) continue;
    if (a.dfused && any) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.dfused + m * a.D + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) df[e] += g[e];
    }
    f32x4 ties = {0.f, 0.f, 0.f, 0.f};
    if (a.pooling == SNAP_POOL_MAX) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (pv[p]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ties[e] += (x[p][i][e] == fused[i][e]) ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= a.num_planes || !a.dplanes[p]) continue;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (pv[p]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.pooling == SNAP_POOL_MAX) o[e] = (x[p][i][e] == fused[i][e]) ? df[e] / ties[e] : 0.f;
          else if (a.pooling == SNAP_POOL_SUM) o[e] = df[e];
          else o[e] = df[e] / (float)count;
        }
      }
      *reinterpret_cast<f32x4*>(a.dplanes[p] + m * a.D + 4 * q) = o;
    }
  }
}

// The same VJP for 128-channel planes with a matching head (the training step's planes), as a persistent
// kernel: in the kernel above each of the 2 x 128 x Dm products of a cell costs a lane exchange and an L1 load
// of the kernel entry.  Here a lane keeps ITS column of Wm in registers for the forward product, the
// transposed kernel sits in LDS for d fused = Wm dy (a conflict-free 16-byte read per output channel of dy),
// and the cell's fused vector / dy come back from LDS as broadcast reads.  Same terms in the same order:
// the same bits.
__global__ __launch_bounds__(256) void plane_fuse_match_bwd_d128_kernel(const FuseBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float wt[32][128];      // Wm^T, rows beyond Dm zero
  __shared__ __attribute__((aligned(16))) float sv[8][128];
  __shared__ __attribute__((aligned(16))) float sd[8][32];
  const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int j = hl;
  for (int i = threadIdx.x; i < 32 * 128; i += 256) {
    const int jj = i >> 7, k = i & 127;
    wt[jj][k] = jj < a.Dm ? a.Wm[(int64_t)k * a.Dm + jj] : 0.f;
  }
  float wreg[128];
#pragma unroll
  for (int k = 0; k < 128; ++k) wreg[k] = j < a.Dm ? a.Wm[(int64_t)k * a.Dm + j] : 0.f;
  const float bj = j < a.Dm ? a.bm[j] : 0.f;
  __syncthreads();
  const float init = a.pooling == SNAP_POOL_MAX ? -INFINITY : 0.f;
  for (int64_t m = (int64_t)blockIdx.x * 8 + hw; m < a.M; m += (int64_t)gridDim.x * 8) {
    f32x4 fused = {init, init, init, init};
    f32x4 x[4];
    bool pv[4];
    int count = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      pv[p] = false;
      x[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p < a.num_planes) {
        pv[p] = a.valids[p] ? (a.valids[p][m] != 0) : true;
        if (pv[p]) x[p] = *reinterpret_cast<const f32x4*>(a.planes[p] + m * 128 + 4 * hl);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!pv[p]) continue;
      ++count;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        fused[e] = a.pooling == SNAP_POOL_MAX ? fmaxf(fused[e], x[p][e]) : fused[e] + x[p][e];
    }
    const bool any = count > 0;
    if (!any) fused = f32x4{0.f, 0.f, 0.f, 0.f};
    if (any && a.pooling == SNAP_POOL_MEAN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fused[e] = fused[e] / (float)count;
    }
    // ---- matching head backward: dy (lane j) ----------------------------------
    *reinterpret_cast<f32x4*>(&sv[hw][4 * hl]) = fused;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float y = bj;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(&sv[hw][4 * q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) y += xv[e] * wreg[4 * q + e];
    }
    float dyj = 0.f;
    const float dz = (j < a.Dm && any) ? a.dmatching[m * a.Dm + j] : 0.f;
    if (a.normalize) {
      const float nrm = sqrtf(half_sum((j < a.Dm) ? y * y : 0.f));
      if (nrm >= a.eps) {
        const float z = y / nrm;
        const float zdz = half_sum((j < a.Dm) ? z * dz : 0.f);
        dyj = (dz - z * zdz) / nrm;
      }
    } else {
      dyj = dz;
    }
    if (j < a.Dm) a.dy[m * a.Dm + j] = dyj;
    // ---- d fused = Wm dy (+ dfused) and routing to the planes -------------------
    sd[hw][hl] = dyj;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 df = {0.f, 0.f, 0.f, 0.f};
    for (int j4 = 0; j4 < a.Dm; j4 += 4) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(&sd[hw][j4]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j4 + u >= a.Dm) break;
        const f32x4 w = *reinterpret_cast<const f32x4*>(&wt[j4 + u][4 * hl]);
#pragma unroll
        for (int e = 0; e < 4; ++e) df[e] += w[e] * dv[u];
      }
    }
    __builtin_amdgcn_wave_barrier();       // (sv / sd are rewritten by the next cell)
    if (a.dfused && any) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.dfused + m * 128 + 4 * hl);
#pragma unroll
      for (int e = 0; e < 4; ++e) df[e] += g[e];
    }
    f32x4 ties = {0.f, 0.f, 0.f, 0.f};
    if (a.pooling == SNAP_POOL_MAX) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (pv[p]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ties[e] += (x[p][e] == fused[e]) ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= a.num_planes || !a.dplanes[p]) continue;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (pv[p]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.pooling == SNAP_POOL_MAX) o[e] = (x[p][e] == fused[e]) ? df[e] / ties[e] : 0.f;
          else if (a.pooling == SNAP_POOL_SUM) o[e] = df[e];
          else o[e] = df[e] / (float)count;
        }
      }
      *reinterpret_cast<f32x4*>(a.dplanes[p] + m * 128 + 4 * hl) = o;
    }
  }
}

}  // namespace

extern "C" int snap_lift_pool_bwd_f32(const SnapLiftDesc* desc, const float* f_images,
                                      const float* cam, const float* Rt, const float* points,
                                      const float* dpooled, float* df_images, void* stream) {
  if (!desc || !f_images || !cam || !Rt || !points || !dpooled || !df_images) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  if (d.B <= 0 || d.V <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.V > 32 || d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.feature_dim <= 0)
    return SNAP_ERR_UNSUPPORTED;
  if (d.C != d.feature_dim + d.num_bins || d.C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (d.out_stride < 2 * d.feature_dim + 1 || d.out_stride % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;
  if (!d.weighted || !d.use_variance || d.add_minmax) return SNAP_ERR_UNSUPPORTED;   // (the det form has them)
  const int nsel = d.K == 0 ? d.V : d.K;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = (size_t)d.B * d.V * d.h * d.w * d.C * sizeof(float);
  if (hipMemsetAsync(df_images, 0, bytes, s) != hipSuccess) return SNAP_ERR_LAUNCH;
  LiftBwdArgs a{d, f_images, cam, Rt, points, dpooled, df_images};
  const dim3 grid((unsigned)snap_cdiv((int64_t)d.B * d.N, 8));
  if (nsel <= 1) hipLaunchKernelGGL((lift_pool_bwd_kernel<1, 0>), grid, dim3(256), 0, s, a);
  else if (nsel <= 4) hipLaunchKernelGGL((lift_pool_bwd_kernel<4, 0>), grid, dim3(256), 0, s, a);
  else if (nsel <= 8) hipLaunchKernelGGL((lift_pool_bwd_kernel<8, 0>), grid, dim3(256), 0, s, a);
  else return SNAP_ERR_UNSUPPORTED;
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// ---- deterministic form ------------------------------------------------------------------
namespace {
struct LiftDetLayout {
  size_t slots, npix;
  size_t off_vec, off_hdr, off_keys, off_keys_out, off_vals_out, off_count, off_start, off_taps, off_tmp;
  size_t tmp_bytes, total;
  int bits;
};
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int lift_det_layout(const SnapLiftDesc& d, LiftDetLayout* L, bool with_vec = true) {
  const int nsel = d.K == 0 ? d.V : d.K;
  L->slots = (size_t)d.B * d.N * nsel;
  L->npix = (size_t)d.B * d.V * d.h * d.w;
  if (L->slots >= 0x7fffffffULL || L->npix >= 0x7ffffffeULL || d.h > 0x7fff || d.w > 0x7fff) return SNAP_ERR_BAD_SHAPE;
  L->bits = 1;
  while (((size_t)1 << L->bits) <= L->npix) ++L->bits;          // keys 0 .. npix
  size_t sort_tmp = 0, scan_tmp = 0;
  unsigned* nul = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_tmp, nul, nul, rocprim::counting_iterator<unsigned>(0), nul,
                                L->slots, 0, L->bits, (hipStream_t)0) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  if (rocprim::exclusive_scan(nullptr, scan_tmp, nul, nul, 0u, L->npix + 2, rocprim::plus<unsigned>(),
                              (hipStream_t)0) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  L->tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  size_t o = 0;
  L->off_vec = o;       o += with_vec ? align256(L->slots * d.feature_dim * sizeof(float)) : 0;
  L->off_hdr = o;       o += align256(L->slots * 12 * sizeof(float));
  L->off_keys = o;      o += align256(L->slots * sizeof(unsigned));
  L->off_keys_out = o;  o += align256(L->slots * sizeof(unsigned));
  L->off_vals_out = o;  o += align256(L->slots * sizeof(unsigned));
  L->off_count = o;     o += align256((L->npix + 2) * sizeof(unsigned));
  L->off_start = o;     o += align256((L->npix + 2) * sizeof(unsigned));
  L->off_taps = o;      o += align256(L->npix * 4 * d.C * sizeof(float));
  L->off_tmp = o;       o += align256(L->tmp_bytes);
  L->total = o;
  return SNAP_OK;
}
}  // namespace

// the shapes the deterministic form takes: ONE validator for the workspace query and the launch, so
// that "workspace_bytes != 0" means "the launch will not refuse the shape" (callers fall back to the
// scatter kernel on 0)
static int lift_det_validate(const SnapLiftDesc& d) {
  if (d.B <= 0 || d.V <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.V > 32 || d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.feature_dim <= 0 || d.num_bins > 32)
    return SNAP_ERR_UNSUPPORTED;
  if (d.C != d.feature_dim + (d.weighted ? d.num_bins : 0) || d.C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (d.out_stride < d.feature_dim * (1 + (d.use_variance ? 1 : 0) + (d.add_minmax ? 2 : 0)) + (d.weighted ? 1 : 0) ||
      d.out_stride % 4 != 0)
    return SNAP_ERR_BAD_SHAPE;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;
  const int nsel = d.K == 0 ? d.V : d.K;
  if (nsel > 8) return SNAP_ERR_UNSUPPORTED;
  return SNAP_OK;
}

extern "C" size_t snap_lift_pool_bwd_det_workspace_bytes(const SnapLiftDesc* desc) {
  if (!desc) return 0;
  if (lift_det_validate(*desc) != SNAP_OK) return 0;
  LiftDetLayout L;
  if (lift_det_layout(*desc, &L) != SNAP_OK) return 0;
  return L.total;
}

extern "C" int snap_lift_pool_bwd_det_f32(const SnapLiftDesc* desc, const float* f_images,
                                          const float* cam, const float* Rt, const float* points,
                                          const float* dpooled, float* df_images, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (!desc || !f_images || !cam || !Rt || !points || !dpooled || !df_images || !workspace) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  const int vrc = lift_det_validate(d);
  if (vrc != SNAP_OK) return vrc;
  const int nsel = d.K == 0 ? d.V : d.K;
  LiftDetLayout L;
  const int rc = lift_det_layout(d, &L);
  if (rc != SNAP_OK) return rc;
  if (workspace_bytes < L.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return SNAP_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  LiftBwdArgs a{d, f_images, cam, Rt, points, dpooled, df_images};
  a.rec_vec = reinterpret_cast<float*>(ws + L.off_vec);
  a.rec_hdr = reinterpret_cast<float*>(ws + L.off_hdr);
  a.keys = reinterpret_cast<unsigned*>(ws + L.off_keys);
  a.count = reinterpret_cast<unsigned*>(ws + L.off_count);
  a.npix = (unsigned)L.npix;
  unsigned* keys_out = reinterpret_cast<unsigned*>(ws + L.off_keys_out);
  unsigned* vals_out = reinterpret_cast<unsigned*>(ws + L.off_vals_out);
  unsigned* start = reinterpret_cast<unsigned*>(ws + L.off_start);
  if (hipMemsetAsync(a.count, 0, (L.npix + 2) * sizeof(unsigned), s) != hipSuccess) return SNAP_ERR_LAUNCH;
  // 1. records (+ their sort keys, + the per-pixel record counts)
  const dim3 grid((unsigned)snap_cdiv((int64_t)d.B * d.N, 8));
  // default fusion options, <= 4 views, f_images addressable with 32-bit byte offsets: the batched
  // producer (lane = voxel for the geometry, lane = channel quad for the gradients)
  const bool batched = d.weighted && d.use_variance && !d.add_minmax && nsel <= 4 && d.h < 65536 && d.w < 65536 &&
                       (int64_t)d.B * d.V * d.h * d.w * d.C * 4 < 0xfff00000LL && !(d.tune_flags & 1);
  const dim3 gridb((unsigned)snap_cdiv((int64_t)d.B * d.N, 256));
  if (batched && nsel <= 1) hipLaunchKernelGGL((lift_pool_bwd_batched_kernel<1>), gridb, dim3(256), 0, s, a);
  else if (batched) hipLaunchKernelGGL((lift_pool_bwd_batched_kernel<4>), gridb, dim3(256), 0, s, a);
  else if (nsel <= 1) hipLaunchKernelGGL((lift_pool_bwd_kernel<1, 1>), grid, dim3(256), 0, s, a);
  else if (nsel <= 4) hipLaunchKernelGGL((lift_pool_bwd_kernel<4, 1>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lift_pool_bwd_kernel<8, 1>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  // 2. stable sort of the record slots by key (ties keep slot order = voxel order) and the list
  //    starts (exclusive scan of the counts)
  size_t tmp = L.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws + L.off_tmp, tmp, a.keys, keys_out, rocprim::counting_iterator<unsigned>(0),
                                vals_out, L.slots, 0, L.bits, s) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  tmp = L.tmp_bytes;
  if (rocprim::exclusive_scan(ws + L.off_tmp, tmp, a.count, start, 0u, L.npix + 2, rocprim::plus<unsigned>(), s) !=
      hipSuccess)
    return SNAP_ERR_LAUNCH;
  // 3. per-key sums by tap offset, then the four that land on every pixel: df_images written exactly once
  LiftGatherArgs g{d, vals_out, start, keys_out, a.rec_vec, a.rec_hdr, dpooled, nsel, reinterpret_cast<float*>(ws + L.off_taps), df_images,
                   (unsigned)L.npix};
  hipLaunchKernelGGL(lift_pool_bwd_taps_kernel, dim3((unsigned)snap_cdiv((int64_t)L.npix, 8)), dim3(256), 0, s, g);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(lift_pool_bwd_combine_kernel, dim3((unsigned)snap_cdiv((int64_t)L.npix * (d.C / 4), 256)),
                     dim3(256), 0, s, g);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// ---- depth_mlp fusion (streetview_encoder.py:263-267): the VJPs of its two lift passes ------------
extern "C" int snap_lift_pool_observations_bwd_f32(const SnapLiftDesc* desc, const float* cam,
                                                   const float* Rt, const float* points,
                                                   const float* obs_feat, const float* dpooled,
                                                   float* dobs, void* stream) {
  if (!desc || !cam || !Rt || !points || !obs_feat || !dpooled || !dobs) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  if (d.B <= 0 || d.V <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.V > 32 || d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.feature_dim <= 0 || d.weighted)
    return SNAP_ERR_UNSUPPORTED;
  if (d.out_stride < d.feature_dim * (1 + (d.use_variance ? 1 : 0) + (d.add_minmax ? 2 : 0)) || d.out_stride % 4 != 0)
    return SNAP_ERR_BAD_SHAPE;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;
  const int nsel = d.K == 0 ? d.V : d.K;
  if (nsel > 8) return SNAP_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  LiftBwdArgs a{d, nullptr, cam, Rt, points, dpooled, nullptr};
  a.obs_in = obs_feat;
  a.dobs = dobs;
  const dim3 grid((unsigned)snap_cdiv((int64_t)d.B * d.N, 8));
  if (nsel <= 1) hipLaunchKernelGGL((lift_pool_bwd_kernel<1, 2>), grid, dim3(256), 0, s, a);
  else if (nsel <= 4) hipLaunchKernelGGL((lift_pool_bwd_kernel<4, 2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lift_pool_bwd_kernel<8, 2>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_lift_observations_bwd_workspace_bytes(const SnapLiftDesc* desc) {
  if (!desc) return 0;
  LiftDetLayout L;
  if (lift_det_layout(*desc, &L, false) != SNAP_OK) return 0;
  return L.total;
}

extern "C" int snap_lift_observations_bwd_f32(const SnapLiftDesc* desc, const float* cam,
                                              const float* Rt, const float* points,
                                              const float* dobs, float* df_images, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  if (!desc || !cam || !Rt || !points || !dobs || !df_images || !workspace) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  if (d.B <= 0 || d.V <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.V > 32 || d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.feature_dim <= 0 || d.C != d.feature_dim)
    return SNAP_ERR_UNSUPPORTED;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;
  const int nsel = d.K == 0 ? d.V : d.K;
  if (nsel > 8) return SNAP_ERR_UNSUPPORTED;
  SnapLiftDesc dd = d;
  dd.num_bins = 0;                                   // (the gather writes feature channels only)
  LiftDetLayout L;
  const int rc = lift_det_layout(dd, &L, false);
  if (rc != SNAP_OK) return rc;
  if (workspace_bytes < L.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) return SNAP_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  LiftBwdArgs a{dd, nullptr, cam, Rt, points, nullptr, df_images};
  a.rec_hdr = reinterpret_cast<float*>(ws + L.off_hdr);
  a.keys = reinterpret_cast<unsigned*>(ws + L.off_keys);
  a.count = reinterpret_cast<unsigned*>(ws + L.off_count);
  a.npix = (unsigned)L.npix;
  unsigned* keys_out = reinterpret_cast<unsigned*>(ws + L.off_keys_out);
  unsigned* vals_out = reinterpret_cast<unsigned*>(ws + L.off_vals_out);
  unsigned* start = reinterpret_cast<unsigned*>(ws + L.off_start);
  if (hipMemsetAsync(a.count, 0, (L.npix + 2) * sizeof(unsigned), s) != hipSuccess) return SNAP_ERR_LAUNCH;
  const dim3 grid((unsigned)snap_cdiv((int64_t)dd.B * dd.N, 8));
  if (nsel <= 1) hipLaunchKernelGGL((lift_pool_bwd_kernel<1, 3>), grid, dim3(256), 0, s, a);
  else if (nsel <= 4) hipLaunchKernelGGL((lift_pool_bwd_kernel<4, 3>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lift_pool_bwd_kernel<8, 3>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  size_t tmp = L.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws + L.off_tmp, tmp, a.keys, keys_out, rocprim::counting_iterator<unsigned>(0),
                                vals_out, L.slots, 0, L.bits, s) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  tmp = L.tmp_bytes;
  if (rocprim::exclusive_scan(ws + L.off_tmp, tmp, a.count, start, 0u, L.npix + 2, rocprim::plus<unsigned>(), s) !=
      hipSuccess)
    return SNAP_ERR_LAUNCH;
  LiftGatherArgs g{dd, vals_out, start, keys_out, dobs, a.rec_hdr, nullptr, nsel, reinterpret_cast<float*>(ws + L.off_taps), df_images,
                   (unsigned)L.npix};
  hipLaunchKernelGGL(lift_pool_bwd_taps_kernel, dim3((unsigned)snap_cdiv((int64_t)L.npix, 8)), dim3(256), 0, s, g);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(lift_pool_bwd_combine_kernel, dim3((unsigned)snap_cdiv((int64_t)L.npix * (dd.C / 4), 256)),
                     dim3(256), 0, s, g);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_vertical_pool_max_bwd_arg_f32(const float* vol, const uint8_t* vvalid, const uint8_t* argz,
                                                  const uint8_t* ties, const float* dplane, float* dvol,
                                                  int64_t M, int32_t Z, int32_t D, void* stream) {
  if (!vol || !vvalid || !argz || !ties || !dplane || !dvol) return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || Z > 255 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(vertical_pool_max_bwd_arg_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), vol, vvalid, argz, ties, dplane, dvol, M, Z, D);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_vertical_pool_bwd_f32(const float* vol, const uint8_t* vvalid,
                                          const float* dplane, float* dvol, int64_t M, int32_t Z,
                                          int32_t D, int32_t pooling, void* stream) {
  if (!vol || !vvalid || !dplane || !dvol) return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (pooling < SNAP_POOL_MAX || pooling > SNAP_POOL_MEAN) return SNAP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(vertical_pool_bwd_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), vol, vvalid, dplane, dvol, M, Z, D, pooling);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_plane_fuse_match_bwd_f32(const float* const* planes,
                                             const uint8_t* const* valids, float* const* dplanes,
                                             int32_t num_planes, int64_t M, int32_t D,
                                             int32_t pooling, const float* Wm, const float* bm,
                                             int32_t Dm, int32_t normalize, float eps,
                                             const float* dmatching, const float* dfused, float* dy,
                                             void* stream) {
  if (!planes || !dplanes) return SNAP_ERR_NULL;
  if (num_planes < 1 || num_planes > 4) return SNAP_ERR_UNSUPPORTED;
  if (M <= 0 || D <= 0 || D % 4 != 0 || D > FB_MAXQ * 128) return SNAP_ERR_BAD_SHAPE;
  if (pooling < SNAP_POOL_MAX || pooling > SNAP_POOL_MEAN) return SNAP_ERR_UNSUPPORTED;
  if (dmatching && (!Wm || !bm || !dy)) return SNAP_ERR_NULL;
  if (dmatching && (Dm < 1 || Dm > 32)) return SNAP_ERR_UNSUPPORTED;
  FuseBwdArgs a;
  for (int i = 0; i < 4; ++i) {
    a.planes[i] = i < num_planes ? planes[i] : nullptr;
    a.valids[i] = (i < num_planes && valids) ? valids[i] : nullptr;
    a.dplanes[i] = i < num_planes ? dplanes[i] : nullptr;
    if (i < num_planes && !a.planes[i]) return SNAP_ERR_NULL;
  }
  a.num_planes = num_planes; a.M = M; a.D = D; a.pooling = pooling;
  a.Wm = Wm; a.bm = bm; a.Dm = Dm; a.normalize = normalize; a.eps = eps;
  a.dmatching = dmatching; a.dfused = dfused; a.dy = dy;
  if (dmatching && D == 128 && M >= 8192)
    hipLaunchKernelGGL(plane_fuse_match_bwd_d128_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(plane_fuse_match_bwd_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// VJP of snap_confidence_head_f32 (bev_mapper.py:154-157,292-295): conf = where(valid,
// log_sigmoid(f . w + bias), 0).  With s = f . w + bias and ds = g * sigmoid(-s) * [valid]:
// d f = ds w (written here), d w = sum_m ds f_m, d bias = sum_m ds -- the two sums leave as
// per-row products (prod [M, D] = ds f, dsv [M, 4] = (ds, 0, 0, 0)) for the fixed-order column
// sums of snap_colsum_f32.  One half-wave per row.
namespace {
__global__ __launch_bounds__(256) void confidence_head_bwd_kernel(
    const float* __restrict__ f, const uint8_t* __restrict__ valid, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ g, int64_t M, int D,
    float* __restrict__ df, float* __restrict__ prod, float* __restrict__ dsv) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  float acc = 0.f;
  for (int c = 4 * hl; c < D; c += 128) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(f + m * D + c);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
    acc += ((x[0] * ww[0] + x[1] * ww[1]) + x[2] * ww[2]) + x[3] * ww[3];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 32);
  const float s = acc + bias[0];
  const bool ok = valid == nullptr || valid[m];
  const float ds = ok ? g[m] / (1.f + expf(s)) : 0.f;             // sigmoid(-s)
  for (int c = 4 * hl; c < D; c += 128) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(f + m * D + c);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
    *reinterpret_cast<f32x4*>(df + m * D + c) = f32x4{ds * ww[0], ds * ww[1], ds * ww[2], ds * ww[3]};
    *reinterpret_cast<f32x4*>(prod + m * D + c) = f32x4{ds * x[0], ds * x[1], ds * x[2], ds * x[3]};
  }
  if (hl == 0) *reinterpret_cast<f32x4*>(dsv + m * 4) = f32x4{ds, 0.f, 0.f, 0.f};
}
}  // namespace

extern "C" int snap_confidence_head_bwd_f32(const float* features, const uint8_t* valid, const float* w,
                                            const float* bias, const float* dconf, int64_t M, int32_t D,
                                            float* dfeatures, float* prod, float* dsv, void* stream) {
  if (!features || !w || !bias || !dconf || !dfeatures || !prod || !dsv) return SNAP_ERR_NULL;
  if (M <= 0 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(confidence_head_bwd_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), features, valid, w, bias, dconf, M, D, dfeatures,
                     prod, dsv);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
