// Camera-ray lift (voxel-pull): for every voxel centre, project into the views,
// select the K nearest visible views, bilinearly gather the image features,
// interpolate the per-observation depth score and pool the observations with
// score-softmax weights -- ONE pass, nothing intermediate in HBM.
//
// Mapping: a 32-lane half-wave owns one voxel.
//   * projection / visibility: lane v handles view v (V <= 32);
//   * top-K selection: K argmin rounds over the half-wave with xor-shuffles,
//     ties towards the lowest view index (== jax.lax.top_k(-dist));
//   * gather: lane q loads channels 4q..4q+3 of each of the 4 taps
//     (one 512-byte contiguous row segment per tap for feature_dim = 128);
//   * softmax / mean / variance over the <= KMAX observations held in registers.
// Voxels are ordered z-fastest, so consecutive half-waves walk up an image column:
// the taps of neighbouring voxels share cache lines (L1/L2 reuse).
//
// Replaces snap/models/streetview_encoder.py:42-65 (project), :127-138 (select),
// :69-105 (gather), :109-124 (depth score), :141-178 (pool) and the camera maths
// of snap/utils/geometry.py:52-69,198-221,260-280.
#include <stdlib.h>

#include "common.h"

namespace {

struct LiftArgs {
  int xcd_group;     // batched kernels: consecutive workgroups owned by one XCD (0 = dispatch order)
  // BEV-tiled traversal (d.grid_y > 0): the voxels of an 8 x 8 block of columns (all levels) are
  // the unit an XCD works on
  int tile_cpt;      // 256-voxel chunks per tile (0 = linear order)
  int tile_log;      // log2 of the tile's extent along y, in columns (8 x 8; narrower grids: 64 / ty x ty)
  int tile_log_x;    // ... along x
  int tiles_x, tiles_y;
  int64_t tiles_total;
  SnapLiftDesc d;
  const float* f;
  const float* cam;
  const float* Rt;
  const float* pts;
  float* pooled;
  uint8_t* valid;
  // depth_mlp fusion (streetview_encoder.py:263-267), generic kernel only:
  float* obs_out;        // [B*N, nsel, fd + 4]: per-observation features | log10 depth | ray(3)
  float* obs_feat;       // [B*N, nsel, fd]: the features alone (the MLP's residual operand)
  const float* obs_in;   // [B*N, nsel, fd]: observations to pool instead of gathering them
  // tap records (batched kernel, class_rows): a voxel with ONE visible observation leaves no row in
  // `pooled` but a 32-byte record [B*N][8] = tap byte offset | clamp flags | wi1 | wj1 | score | 0 0 0;
  // the consumer (mlp2_pool_kernel<.., GATHER>) gathers and blends the four taps itself
  uint32_t* tap_recs;
};

struct Proj {
  float pi, pj;   // (row, col) coordinates in the feature map, corner origin
  float depth;
  float dist;     // distance voxel -> camera centre
  bool vis;
  float vx, vy;   // camera-frame x, y (z = depth): the viewing ray of the observation
};

// One (voxel, view) projection.  cam = wh f c k(3) max_fov pad; Rt = R(9) t(3).
__device__ __forceinline__ Proj project_one(const float* __restrict__ cam,
                                            const float* __restrict__ Rt, float px, float py,
                                            float pz, int fisheye) {
  const float eps = 1e-3f;
  // Transform3D.inv: R_inv = R^T, t_inv = -(R^T t); then t_inv + R_inv p.
  float pv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r0 = Rt[0 * 3 + i], r1 = Rt[1 * 3 + i], r2 = Rt[2 * 3 + i];
    const float tinv = -((r0 * Rt[9] + r1 * Rt[10]) + r2 * Rt[11]);
    pv[i] = tinv + ((r0 * px + r1 * py) + r2 * pz);
  }
  Proj o;
  o.vx = pv[0];
  o.vy = pv[1];
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
  o.pi = y;  // xy -> ij
  o.pj = x;
  o.vis = valid;
  const float dx = px - Rt[9], dy = py - Rt[10], dz = pz - Rt[11];
  o.dist = sqrtf((dx * dx + dy * dy) + dz * dz);
  return o;
}

struct Taps {
  int i0, i1, j0, j1;
  float w00, w01, w10, w11;
  float wi1, wj1;   // the 1-D weights the four products are built from
};

// selective != 0: streetview_encoder.py:93-105 (clip the point, floor, +1);
// selective == 0: grids.interpolate_nd / map_coordinates (clip each tap index).
__device__ __forceinline__ Taps make_taps(float pi, float pj, int h, int w, int selective) {
  Taps t;
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
  t.wi1 = wi1;
  t.wj1 = wj1;
  return t;
}

// Pooling for the NON-default fusion options of pool_multiview_features
// (streetview_encoder.py:141-178): unweighted statistics (scores = None: plain mean / variance over
// the valid views, no score channel), fusion_use_variance = False, fusion_add_minmax = True.
// Row layout: mean | var? | max, min? | score_max?  (the default kernels keep their own
// specialised code: mean | var | score_max).
template <int KMAX>
__device__ __forceinline__ void pool_generic(const SnapLiftDesc& d, const f32x4 (&feat)[KMAX],
                                             const float (&score)[KMAX], const bool (&ok)[KMAX],
                                             int nvis, int hl, float* out) {
  const int fd = d.feature_dim;
  const int nq = fd >> 2;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, var = {0.f, 0.f, 0.f, 0.f};
  f32x4 mx = {0.f, 0.f, 0.f, 0.f}, mn = {0.f, 0.f, 0.f, 0.f};
  float smax = 0.f;
  if (nvis > 0) {
    float wgt[KMAX];
    if (d.weighted) {
      float m = 0.f;
      smax = -INFINITY;
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (ok[r]) { m = fmaxf(m, score[r]); smax = fmaxf(smax, score[r]); }
      float den = 0.f;
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        wgt[r] = ok[r] ? expf(score[r] - m) : 0.f;
        den += wgt[r];
      }
#pragma unroll
      for (int r = 0; r < KMAX; ++r) wgt[r] = wgt[r] / den;
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (ok[r])
#pragma unroll
          for (int c = 0; c < 4; ++c) mean[c] += wgt[r] * feat[r][c];
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (ok[r])
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float dl = feat[r][c] - mean[c];
            var[c] += wgt[r] * (dl * dl);
          }
    } else {
      // jnp.mean / jnp.var(where=valid): sums over the valid views divided by their count
      const float cnt = (float)nvis;
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (ok[r])
#pragma unroll
          for (int c = 0; c < 4; ++c) mean[c] += feat[r][c];
#pragma unroll
      for (int c = 0; c < 4; ++c) mean[c] = mean[c] / cnt;
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (ok[r])
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float dl = feat[r][c] - mean[c];
            var[c] += dl * dl;
          }
#pragma unroll
      for (int c = 0; c < 4; ++c) var[c] = var[c] / cnt;
    }
    mx = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mn = f32x4{INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (ok[r])
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          mx[c] = fmaxf(mx[c], feat[r][c]);
          mn[c] = fminf(mn[c], feat[r][c]);
        }
  }
  int off = 0;
  if (hl < nq) *reinterpret_cast<f32x4*>(out + 4 * hl) = mean;
  off += fd;
  if (d.use_variance) {
    if (hl < nq) *reinterpret_cast<f32x4*>(out + off + 4 * hl) = var;
    off += fd;
  }
  if (d.add_minmax) {
    if (hl < nq) {
      *reinterpret_cast<f32x4*>(out + off + 4 * hl) = mx;
      *reinterpret_cast<f32x4*>(out + off + fd + 4 * hl) = mn;
    }
    off += 2 * fd;
  }
  if (hl == 0) {
    if (d.weighted) out[off++] = smax;
    for (int c = off; c < d.out_stride; ++c) out[c] = 0.f;
  }
}

template <int KMAX>
__global__ __launch_bounds__(256) void lift_pool_kernel(const LiftArgs a) {
  const SnapLiftDesc& d = a.d;
  const int hl = threadIdx.x & 31;
  // Blocks walk the voxel range in dispatch (round-robin over XCDs) order.  Measured
  // alternatives at C2 (5.7 ms): a contiguous eighth per XCD is SLOWER (6.6 ms, visibility
  // varies over the scene and unbalances the XCDs); XCD-owned chunks of 8..480 workgroups
  // (whole columns per L2) change nothing (5.7-5.8 ms) although they remove most of the
  // L2 over-fetch -- the kernel is bound by the tap gathers through L1, not by L2 misses.
  const int64_t gv = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int64_t total = (int64_t)d.B * d.N;
  if (gv >= total) return;  // whole half-wave exits together
  const int b = (int)(gv / d.N);
  const int fd = d.feature_dim;
  const int nq = fd >> 2;
  const bool all_views = d.K == 0;
  const int nsel = all_views ? d.V : d.K;

  const float* p = a.pts + gv * 3;
  const float px = p[0], py = p[1], pz = p[2];

  // ---- k1: lane v projects into view v -----------------------------------
  Proj pr;
  pr.pi = pr.pj = pr.depth = pr.vx = pr.vy = 0.f;
  pr.dist = INFINITY;
  pr.vis = false;
  if (hl < d.V) {
    pr = project_one(a.cam + ((int64_t)b * d.V + hl) * 11, a.Rt + ((int64_t)b * d.V + hl) * 12, px,
                     py, pz, d.fisheye);
  }
  float key_d = (hl < d.V && pr.vis) ? pr.dist : INFINITY;
  int key_i = (hl < d.V) ? hl : 1000 + hl;

  // ---- k2: selection ---------------------------------------------------------
  int sel[KMAX];
  float min_dist = INFINITY;
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    if (r >= nsel) { sel[r] = 0; continue; }
    if (all_views) {
      sel[r] = r;
      continue;
    }
    float bd = key_d;
    int bi = key_i;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 32);
      const int oi = __shfl_xor(bi, o, 32);
      if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (r == 0) min_dist = bd;
    sel[r] = bi;            // bi < V always while r < nsel <= V
    if (hl == bi) { key_d = INFINITY; key_i = 1000 + hl; }
  }
  if (all_views) {
    float md = key_d;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) md = fminf(md, __shfl_xor(md, o, 32));
    min_dist = md;
  }

  // ---- k3/k4: gather selected observations ----------------------------------
  f32x4 feat[KMAX];
  float score[KMAX];
  bool ok[KMAX];
  bool any = false;
  const float log_range = logf(d.depth_max / d.depth_min);
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    feat[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    score[r] = 0.f;
    ok[r] = false;
    if (r >= nsel) continue;
    const int v = sel[r];
    const float pi = __shfl(pr.pi, v, 32);
    const float pj = __shfl(pr.pj, v, 32);
    const float depth = __shfl(pr.depth, v, 32);
    const bool vis = __shfl((int)pr.vis, v, 32) != 0;
    ok[r] = vis;
    if (a.obs_out) {
      // depth_mlp input: log10(clip(depth, 0.1, 100)) and the unit viewing ray in the camera
      // frame, zero where the view does not see the point (streetview_encoder.py:264-266)
      const float vx = __shfl(pr.vx, v, 32), vy = __shfl(pr.vy, v, 32);
      if (hl == 0) {
        const float nrm = fmaxf(sqrtf((vx * vx + vy * vy) + depth * depth), 1e-5f);
        float* o = a.obs_out + (gv * nsel + r) * (int64_t)(fd + 4) + fd;
        o[0] = log10f(fminf(fmaxf(depth, 0.1f), 100.f));
        o[1] = vis ? vx / nrm : 0.f;
        o[2] = vis ? vy / nrm : 0.f;
        o[3] = vis ? depth / nrm : 0.f;
      }
    }
    if (!vis) continue;  // half-wave uniform
    any = true;
    if (a.obs_in) {        // pooling pass of the depth_mlp fusion: the observation is given
      if (hl < nq) feat[r] = *reinterpret_cast<const f32x4*>(a.obs_in + (gv * nsel + r) * (int64_t)fd + 4 * hl);
      continue;
    }
    const Taps t = make_taps(pi, pj, d.h, d.w, all_views ? 0 : 1);
    const float* img = a.f + ((int64_t)b * d.V + v) * d.h * d.w * d.C;
    const float* r00 = img + ((int64_t)t.i0 * d.w + t.j0) * d.C;
    const float* r01 = img + ((int64_t)t.i0 * d.w + t.j1) * d.C;
    const float* r10 = img + ((int64_t)t.i1 * d.w + t.j0) * d.C;
    const float* r11 = img + ((int64_t)t.i1 * d.w + t.j1) * d.C;
    if (hl < nq) {
      const f32x4 a00 = *reinterpret_cast<const f32x4*>(r00 + 4 * hl);
      const f32x4 a01 = *reinterpret_cast<const f32x4*>(r01 + 4 * hl);
      const f32x4 a10 = *reinterpret_cast<const f32x4*>(r10 + 4 * hl);
      const f32x4 a11 = *reinterpret_cast<const f32x4*>(r11 + 4 * hl);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        feat[r][e] = ((t.w00 * a00[e] + t.w01 * a01[e]) + t.w10 * a10[e]) + t.w11 * a11[e];
    }
    if (!d.weighted) continue;           // (scores = None: no depth-score bins in f_images)
    // depth score: two neighbouring log-depth bins, each bilinearly gathered.
    const float dc = fminf(fmaxf(depth, d.depth_min), d.depth_max);
    const float tt = logf(dc / d.depth_min) / log_range;
    const float index = 0.5f + tt * (float)(d.num_bins - 1);
    const float c = index - 0.5f;
    const float fl = floorf(c);
    const float wb1 = c - fl, wb0 = 1.f - wb1;
    const int b0 = (int)fminf(fmaxf(fl, 0.f), (float)(d.num_bins - 1));
    const int b1 = (int)fminf(fmaxf(fl + 1.f, 0.f), (float)(d.num_bins - 1));
    const int c0 = fd + b0, c1 = fd + b1;
    const float s0 = ((t.w00 * r00[c0] + t.w01 * r01[c0]) + t.w10 * r10[c0]) + t.w11 * r11[c0];
    const float s1 = ((t.w00 * r00[c1] + t.w01 * r01[c1]) + t.w10 * r10[c1]) + t.w11 * r11[c1];
    score[r] = wb0 * s0 + wb1 * s1;
  }

  if (a.obs_out) {         // gathering pass of the depth_mlp fusion: observations out, no pooling
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nsel || hl >= nq) continue;
      *reinterpret_cast<f32x4*>(a.obs_out + (gv * nsel + r) * (int64_t)(fd + 4) + 4 * hl) = feat[r];
      *reinterpret_cast<f32x4*>(a.obs_feat + (gv * nsel + r) * (int64_t)fd + 4 * hl) = feat[r];
    }
    if (hl == 0) {
      bool vld = any;
      if (d.max_view_distance >= 0.f && !all_views) vld = vld && (min_dist <= d.max_view_distance);
      a.valid[gv] = vld ? 1 : 0;
    }
    return;
  }
  // ---- k5: softmax-weighted mean / variance / max score ---------------------
  float* out = a.pooled + gv * d.out_stride;
  if (!(d.weighted && d.use_variance && !d.add_minmax)) {      // non-default fusion options
    int nvis = 0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) nvis += ok[r] ? 1 : 0;
    pool_generic<KMAX>(d, feat, score, ok, nvis, hl, out);
    if (hl == 0) {
      bool vld = any;
      if (d.max_view_distance >= 0.f && !all_views) vld = vld && (min_dist <= d.max_view_distance);
      a.valid[gv] = vld ? 1 : 0;
    }
    return;
  }
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, var = {0.f, 0.f, 0.f, 0.f};
  float smax = 0.f;
  if (any) {
    // jax.nn.softmax(..., where=valid, initial=0): shift = max(0, max valid score).
    float m = 0.f;
    smax = -INFINITY;
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (ok[r]) { m = fmaxf(m, score[r]); smax = fmaxf(smax, score[r]); }
    float e[KMAX], den = 0.f;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      e[r] = ok[r] ? expf(score[r] - m) : 0.f;
      den += e[r];
    }
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      const float wgt = e[r] / den;
#pragma unroll
      for (int c = 0; c < 4; ++c) mean[c] += wgt * feat[r][c];
    }
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      const float wgt = e[r] / den;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float dl = feat[r][c] - mean[c];
        var[c] += wgt * (dl * dl);
      }
    }
  }
  if (hl < nq) {
    *reinterpret_cast<f32x4*>(out + 4 * hl) = mean;
    *reinterpret_cast<f32x4*>(out + fd + 4 * hl) = var;
  }
  if (hl == 0) {
    out[2 * fd] = smax;
    for (int c = 2 * fd + 1; c < d.out_stride; ++c) out[c] = 0.f;
    bool vld = any;
    if (d.max_view_distance >= 0.f && !all_views) vld = vld && (min_dist <= d.max_view_distance);
    a.valid[gv] = vld ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
// Batched variant (nsel <= 4): a half-wave owns 32 CONSECUTIVE voxels.
//   phase A  lane j <-> voxel j: project it into the views, select, and build the tap record
//            of each selected view (tap offset, clamp flags, 1-D weights, depth bins) -- the scalar
//            per-voxel geometry now runs once per voxel on a full lane set instead of once per
//            voxel with 4 of 32 lanes doing useful work (it was ~60 % of the kernel's VALU
//            issue; the kernel is VALU-bound, PMC r01).  Records go to LDS (32 B each).
//   phase B  for each of the 32 voxels: lane q <-> channels 4q..4q+3: broadcast-read the
//            records, gather the taps, pool.  Same arithmetic as lift_pool_kernel, same bits.
// ---------------------------------------------------------------------------
constexpr int LB_REC = 4;    // dwords per (voxel, slot) record: tap byte offset | packed | wi1 | wj1  (+ wb1 apart)

// hi / lo bf16 parts of four f32, two per dword (the split of conv_split.hip / mlp_pool.hip)
__device__ __forceinline__ void split_row_quad(const f32x4& v, unsigned (&hi)[2], unsigned (&lo)[2]) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2 pr = {v[2 * h], v[2 * h + 1]};
    const bf16x2 b = __builtin_convertvector(pr, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &b, 4);
    hi[h] = u;
    const f32x2 rs = {pr[0] - __uint_as_float(u << 16), pr[1] - __uint_as_float(u & 0xffff0000u)};
    const bf16x2 bl = __builtin_convertvector(rs, bf16x2);
    __builtin_memcpy(&u, &bl, 4);
    lo[h] = u;
  }
}

#ifndef SNAP_LIFT_ABLATE
#define SNAP_LIFT_ABLATE 0     // timing experiments only (wrong results): 1 = no row stores (and what
#endif                         // feeds them: dead code), 2 = no feature loads, 4 = rows stored into a 4 MB
                               // window (every instruction stays, no HBM write traffic), 8 = phase A alone,
                               // 16 = no depth-score loads in phase A (tap records)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pair_lo(const f32x4& v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 pair_hi(const f32x4& v) { return __builtin_shufflevector(v, v, 2, 3); }

// FD128: feature_dim == 128, every lane of the half-wave owns a channel quad (no lane guards)
template <int KMAX, bool FD128>
__global__ __launch_bounds__(256, KMAX > 1 ? 6 : 8) void lift_pool_batched_kernel(const LiftArgs a) {
  // 24 KB of LDS at KMAX = 4 (records 16 + bin weights 4 + headers 4): six workgroups per CU, and
  // the launch bound keeps the registers at six waves per SIMD too (8-dword records, 36 KB: four
  // waves per SIMD whatever the register count -- measured: no difference, the kernel is bound
  // by VALU issue, not by the round trips)
  __shared__ __attribute__((aligned(16))) int recs[8][32][KMAX][LB_REC];
  __shared__ float wbs[8][32][KMAX];
  __shared__ float hdr[8][32][4];   // (-, min_dist, voxel index or -1, visible observations) per voxel
  __shared__ int cls_cnt[4][KMAX + 2];
  __shared__ uint8_t order[256];
  const SnapLiftDesc& d = a.d;
  const int hl = threadIdx.x & 31;
  const int hw = threadIdx.x >> 5;
  // Workgroups are dispatched round-robin over the 8 XCDs (one L2 each).  In dispatch order
  // every XCD sees every 8th 256-voxel chunk of the scene, so all eight L2s end up caching the
  // same image regions and most tap reads miss (PMC: 16.5 GB fetched for 0.42 GB of f_images,
  // and with its 11 GB of writes the kernel sat exactly on the achievable HBM rate).  With
  // xcd_group = G, runs of G consecutive chunks belong to ONE XCD: its taps stay in its L2.
  //
  // BEV-tiled traversal (grid dimensions known): even so, a run of consecutive voxels (x, then
  // y, then z fastest) sweeps a whole row of 128 columns every 7680 voxels, i.e. a fan that
  // covers most of every view -- far more than a 4 MB L2 holds.  Instead an XCD walks the
  // chunks of ONE 8 x 8 block of columns (1.6 m square, all levels: 15 chunks at Z = 60) before
  // it moves on; the block's footprint in a view is a narrow vertical band.  Measured at C2
  // (rocprofv3 FETCH_SIZE): 12.7 -> 8.8 GB fetched per step, kernel time unchanged (3.3 ms) AT
  // THE TIME: the kernel was bound by VALU issue.  At the end of round 2 (leaner phase B, class
  // ordering, 2.8 ms) it moves 8.8 GB of L2 misses + 5.1 GB of writes at 4.9 TB/s and IS bound
  // by them; tile sides of 4 / 16 columns, XCD-owned adjacent tile columns and fewer resident
  // workgroups were all slower.  (Giving each XCD one contiguous run of tiles
  // -- a whole scene -- was slower, 4.1 ms: the eight L2s then see very different loads.)
  int64_t lb = blockIdx.x;
  int64_t my_gv;                    // phase A: this lane's voxel (-1: none)
  if (a.tile_cpt > 0) {
    const int64_t x = lb & 7, sq = lb >> 3;
    const int64_t tile = (sq / a.tile_cpt) * 8 + x;
    if (tile >= a.tiles_total) return;
    const int chunk = (int)(sq % a.tile_cpt);
    const int tps = a.tiles_x * a.tiles_y;
    const int b = (int)(tile / tps);
    const int tr = (int)(tile - (int64_t)b * tps);
    const int tx = tr / a.tiles_y, ty = tr - tx * a.tiles_y;
    const int Z = d.grid_z, GY = d.grid_y, GX = d.N / (GY * Z);
    const int l = chunk * 256 + hw * 32 + hl;
    const int col = l / Z, z = l - col * Z;
    const int tl = a.tile_log, ts = 1 << tl, tsx = 1 << a.tile_log_x;
    const int X = tx * tsx + (col >> tl), Y = ty * ts + (col & (ts - 1));
    const bool in = col < tsx * ts && X < GX && Y < GY;
    my_gv = in ? (((int64_t)b * GX + X) * GY + Y) * Z + z : -1;
  } else {
    if (a.xcd_group > 0) {
      const int64_t G = a.xcd_group;
      const int64_t x = lb & 7, sq = lb >> 3;
      lb = ((sq / G) * 8 + x) * G + (sq % G);
    }
    my_gv = (lb * 8 + hw) * 32 + hl;
    if (my_gv >= (int64_t)d.B * d.N) my_gv = -1;
  }
  const int fd = d.feature_dim;
  const int nq = fd >> 2;
  const bool all_views = d.K == 0;
  const int nsel = all_views ? d.V : d.K;
  const float log_range = logf(d.depth_max / d.depth_min);

  // ---------------- phase A: lane = voxel ----------------
  {
    const int64_t gv = my_gv;
    const bool live = gv >= 0;
    const int b = live ? (int)(gv / d.N) : 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) {
      const float* p = a.pts + gv * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    // selected views, ascending (visible distance, view index); invisible views never
    // contribute (ok = false in the pooling), so only visible ones are kept.
    float kd[KMAX], kpi[KMAX], kpj[KMAX], kdep[KMAX];
    int kv[KMAX];
#pragma unroll
    for (int r = 0; r < KMAX; ++r) { kd[r] = INFINITY; kpi[r] = kpj[r] = kdep[r] = 0.f; kv[r] = -1; }
    float min_dist = INFINITY;
    for (int v = 0; v < d.V; ++v) {
      const Proj pr = project_one(a.cam + ((int64_t)b * d.V + v) * 11,
                                  a.Rt + ((int64_t)b * d.V + v) * 12, px, py, pz, d.fisheye);
      const bool vis = live && pr.vis;
      if (all_views) {
        // slot v <-> view v
#pragma unroll
        for (int r = 0; r < KMAX; ++r)
          if (r == v) { kv[r] = vis ? v : -1; kpi[r] = pr.pi; kpj[r] = pr.pj; kdep[r] = pr.depth; }
        if (vis) min_dist = fminf(min_dist, pr.dist);
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
    if (!all_views) min_dist = kd[0];
    hdr[hw][hl][1] = min_dist;
    hdr[hw][hl][2] = __int_as_float((int)gv);       // (B * N < 2^31, checked by the launcher)
    // The records of the visible observations are written to slots 0 .. nvis-1 in slot order
    // (the selective insertion already leaves them there; with all views, holes are closed):
    // skipped slots contribute nothing to any sum, so the pooled bits do not change, and phase B
    // needs one count per voxel instead of a flag per slot.
    int nvis = 0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nsel || kv[r] < 0) continue;
      int* rec = recs[hw][hl][nvis];
      float* wbp = &wbs[hw][hl][nvis++];
      const Taps t = make_taps(kpi[r], kpj[r], d.h, d.w, all_views ? 0 : 1);
      // depth score: two neighbouring log-depth bins
      const float dc = fminf(fmaxf(kdep[r], d.depth_min), d.depth_max);
      const float tt = logf(dc / d.depth_min) / log_range;
      const float index = 0.5f + tt * (float)(d.num_bins - 1);
      const float c = index - 0.5f;
      const float fl = floorf(c);
      const int b0 = (int)fminf(fmaxf(fl, 0.f), (float)(d.num_bins - 1));
      const int b1 = (int)fminf(fmaxf(fl + 1.f, 0.f), (float)(d.num_bins - 1));
      rec[0] = (int)(((((uint32_t)b * d.V + kv[r]) * d.h + t.i0) * d.w + t.j0) * ((uint32_t)d.C * 4u));
      rec[1] = kv[r] | ((t.i1 != t.i0) << 8) | ((t.j1 != t.j0) << 9) | (b0 << 10) | (b1 << 18);
      rec[2] = __float_as_int(t.wi1);
      rec[3] = __float_as_int(t.wj1);
      *wbp = c - fl;
    }
    hdr[hw][hl][3] = __int_as_float(nvis);
    if (a.tap_recs && live && nvis <= 1) {
      // record mode: this voxel is finished here.  Its score is the one phase B would compute (the
      // same taps, weights and expression order, lane = voxel instead of a broadcast per half-wave);
      // mean = the blended features (softmax weight e / e == 1), variance = 0: left to the consumer.
      bool vld = nvis > 0;
      if (d.max_view_distance >= 0.f && !all_views) vld = vld && (min_dist <= d.max_view_distance);
      if (nvis == 1) {
        const int* rec = recs[hw][hl][0];
        const int pk = rec[1];
        const float wi1 = __int_as_float(rec[2]), wj1 = __int_as_float(rec[3]);
        const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;
        const float w00 = wi0 * wj0, w01 = wi0 * wj1, w10 = wi1 * wj0, w11 = wi1 * wj1;
        const uint32_t Cb_ = (uint32_t)d.C * 4u, Wb_ = (uint32_t)d.w * Cb_, fdb_ = (uint32_t)d.feature_dim * 4u;
        const uint32_t o00 = (uint32_t)rec[0];
        const uint32_t o01 = o00 + ((pk >> 9) & 1 ? Cb_ : 0u);
        const uint32_t o10 = o00 + ((pk >> 8) & 1 ? Wb_ : 0u);
        const uint32_t o11 = o10 + (o01 - o00);
        const uint32_t c0 = fdb_ + ((pk >> 10) & 0xff) * 4u, c1 = fdb_ + ((pk >> 18) & 0xff) * 4u;
        const char* fb_ = reinterpret_cast<const char*>(a.f);
#if SNAP_LIFT_ABLATE & 16
        const float t00 = w00, t01 = w01, t10 = w10, t11 = w11, u00 = w00, u01 = w01, u10 = w10, u11 = w11;
        (void)fb_; (void)o11; (void)c0; (void)c1;
#else
        const float t00 = *reinterpret_cast<const float*>(fb_ + (o00 + c0));
        const float t01 = *reinterpret_cast<const float*>(fb_ + (o01 + c0));
        const float t10 = *reinterpret_cast<const float*>(fb_ + (o10 + c0));
        const float t11 = *reinterpret_cast<const float*>(fb_ + (o11 + c0));
        const float u00 = *reinterpret_cast<const float*>(fb_ + (o00 + c1));
        const float u01 = *reinterpret_cast<const float*>(fb_ + (o01 + c1));
        const float u10 = *reinterpret_cast<const float*>(fb_ + (o10 + c1));
        const float u11 = *reinterpret_cast<const float*>(fb_ + (o11 + c1));
#endif
        const float wb1 = wbs[hw][hl][0], wb0 = 1.f - wb1;
        const float s0 = ((w00 * t00 + w01 * t01) + w10 * t10) + w11 * t11;
        const float s1 = ((w00 * u00 + w01 * u01) + w10 * u10) + w11 * u11;
        const float score = wb0 * s0 + wb1 * s1;
        uint4* ro = reinterpret_cast<uint4*>(a.tap_recs + gv * 8);
        ro[0] = uint4{o00, (uint32_t)pk, (uint32_t)rec[2], (uint32_t)rec[3]};
        ro[1] = uint4{__float_as_uint(score), 0u, 0u, 0u};
      }
      a.valid[gv] = vld ? 1 : 0;
    }
    // The two half-waves of a wave walk their voxels in lock step, so a wave pays for the longer
    // of the two paths (no observation / one / several + softmax: ~30 / ~160 / ~400 instructions);
    // in voxel order 36 % of the pairs of a four-view map contain a several-observation voxel.
    // The workgroup's 256 voxels are therefore ordered by observation count (counting sort:
    // ballots per wave, wave totals through LDS) and iteration j of half-wave hw takes entry
    // 8 j + hw of that order: the two halves of a wave get NEIGHBOURS of the sorted list, i.e.
    // voxels of the same class except at the few class boundaries.  Rows are independent, so
    // the bits do not change.
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
#if SNAP_LIFT_ABLATE & 8
  return;                      // (timing: phase A alone)
#endif

  // ---------------- phase B: lane = channel quad ----------------
  // The kernel is VALU-bound (PMC r02: VALU 72 % busy), so phase B spends as few instructions
  // per voxel as the arithmetic allows: tap addresses are 32-bit byte offsets from the (scalar)
  // base of f_images (one v_add each, saddr-form loads; the launcher checks the tensor is
  // < 4 GB), the eight depth-score loads are issued BEFORE the four feature loads (one wait for
  // the twelve, not two round trips), and the four channels of a lane are blended / pooled as
  // two packed pairs (v_pk_mul_f32 / v_pk_add_f32: same IEEE operations, same bits).
  const char* fb = reinterpret_cast<const char*>(a.f);
  const uint32_t Cb = (uint32_t)d.C * 4u, Wb = (uint32_t)d.w * Cb, fdb = (uint32_t)fd * 4u;
  const uint32_t lane_off = 16u * hl;
  for (int j = 0; j < 32; ++j) {
    const int v = order[8 * j + hw];                    // (half-wave uniform, like all of its fields)
    const float* vh = &hdr[0][0][0] + 4 * v;
    const int64_t gv = __float_as_int(vh[2]);
    if (gv < 0) continue;
    const float min_dist = vh[1];
    const int nvis = __float_as_int(vh[3]);
    if (a.tap_recs && nvis <= 1) continue;      // record mode: finished in phase A
    // Per visible slot: gather + blend right away (16 live tap registers, not 64: occupancy
    // matters more here than loads in flight per wave).  Nothing is zero-initialised and every
    // use is guarded by r < nvis; voxels seen by ONE view skip the softmax (weight e/e == 1
    // exactly: mean = f, var = 0, same bits) -- the common case.
    f32x2 feat[KMAX][2];
    float score[KMAX];
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
      if (r >= nvis) continue;   // (nvis <= nsel <= KMAX)
      const int* rec = &recs[0][0][0][0] + (v * KMAX + r) * LB_REC;
      const i32x4 q4 = *reinterpret_cast<const i32x4*>(rec);   // byte offset | packed | wi1 | wj1
      const int pk = q4[1];
      const float wi1 = __int_as_float(q4[2]), wj1 = __int_as_float(q4[3]);
      const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;
      const float w00 = wi0 * wj0, w01 = wi0 * wj1, w10 = wi1 * wj0, w11 = wi1 * wj1;
      const uint32_t o00 = (uint32_t)q4[0];
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
      if (FD128 || hl < nq) {
#if SNAP_LIFT_ABLATE & 2
        const f32x4 a00 = {w00, w01, w10, w11}, a01 = a00, a10 = a00, a11 = a00;
#else
        const f32x4 a00 = *reinterpret_cast<const f32x4*>(fb + (o00 + lane_off));
        const f32x4 a01 = *reinterpret_cast<const f32x4*>(fb + (o01 + lane_off));
        const f32x4 a10 = *reinterpret_cast<const f32x4*>(fb + (o10 + lane_off));
        const f32x4 a11 = *reinterpret_cast<const f32x4*>(fb + (o11 + lane_off));
#endif
        const f32x2 p00 = {w00, w00}, p01 = {w01, w01}, p10 = {w10, w10}, p11 = {w11, w11};
        feat[r][0] = ((p00 * pair_lo(a00) + p01 * pair_lo(a01)) + p10 * pair_lo(a10)) + p11 * pair_lo(a11);
        feat[r][1] = ((p00 * pair_hi(a00) + p01 * pair_hi(a01)) + p10 * pair_hi(a10)) + p11 * pair_hi(a11);
      }
      const float wb1 = (&wbs[0][0][0])[v * KMAX + r], wb0 = 1.f - wb1;
      const float s0 = ((w00 * t00 + w01 * t01) + w10 * t10) + w11 * t11;
      const float s1 = ((w00 * u00 + w01 * u01) + w10 * u10) + w11 * u11;
      score[r] = wb0 * s0 + wb1 * s1;
    }
#if SNAP_LIFT_ABLATE & 4
    float* out = a.pooled + (gv & 4095) * d.out_stride;     // (all rows into 4 MB: no HBM writes)
#else
    float* out = a.pooled + gv * d.out_stride;
#endif
    f32x2 mean2[2] = {{0.f, 0.f}, {0.f, 0.f}}, var2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float smax = 0.f;
    if (nvis == 1) {          // half-wave uniform
      mean2[0] = feat[0][0];
      mean2[1] = feat[0][1];
      smax = score[0];
    } else if (nvis > 1) {
      // jax.nn.softmax(..., where=valid, initial=0): shift = max(0, max valid score).
      float m = 0.f;
      smax = -INFINITY;
#pragma unroll
      for (int r = 0; r < KMAX; ++r)
        if (r < nvis) { m = fmaxf(m, score[r]); smax = fmaxf(smax, score[r]); }
      float e[KMAX], den = 0.f;
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        if (r >= nvis) continue;
        e[r] = expf(score[r] - m);
        den += e[r];             // (the invisible slots of the masked form add +0: same sum)
      }
      float wgt[KMAX];
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        if (r >= nvis) continue;
        wgt[r] = e[r] / den;
        const f32x2 w2 = {wgt[r], wgt[r]};
#pragma unroll
        for (int h = 0; h < 2; ++h) mean2[h] += w2 * feat[r][h];
      }
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        if (r >= nvis) continue;
        const f32x2 w2 = {wgt[r], wgt[r]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 dl = feat[r][h] - mean2[h];
          var2[h] += w2 * (dl * dl);
        }
      }
    }
    const f32x4 mean = {mean2[0][0], mean2[0][1], mean2[1][0], mean2[1][1]};
    const f32x4 var = {var2[0][0], var2[0][1], var2[1][0], var2[1][1]};
    // (valid_rows_only: a voxel no view sees gets its validity byte, not its 1 KB row of zeros --
    // for consumers that read the rows of valid voxels only, 40 % of the map's voxels at C2)
#if SNAP_LIFT_ABLATE & 1
    const bool write_row = false;
#else
    const bool write_row = nvis > 0 || !d.valid_rows_only;
#endif
    if (d.out_split) {
      // the row as the split-bf16 engines stage it: [16-channel slab][hi | lo][16] bf16, 64 B per
      // slab (hi = bf16(v), lo = bf16(v - hi): the consumer's own split, done here once) -- the
      // fused MLP / pool kernel then moves it global -> LDS by LDS-DMA without touching a register
      char* orow = reinterpret_cast<char*>(out);
      // lanes 2i / 2i+1 hold channels k..k+3 / k+4..k+7 of one 16-byte chunk: the even lane
      // collects both hi halves, the odd lane both lo halves (one exchange), and each writes
      // ONE 16-byte chunk per statistic instead of two 8-byte halves
      {
        const bool odd = hl & 1;
        const f32x4 stat[2] = {mean, var};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint4 chunk = {0u, 0u, 0u, 0u};
          if (t == 0 || nvis > 1) {          // (one observation: the variance is exactly 0)
            unsigned hi[2], lo[2];
            split_row_quad(stat[t], hi, lo);
            // (lane ^ 1 by DPP quad_perm [1, 0, 3, 2]: one VALU move each, no LDS permute)
            const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hi[0] : lo[0]), 0xB1, 0xf, 0xf, true);
            const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hi[1] : lo[1]), 0xB1, 0xf, 0xf, true);
            chunk = odd ? uint4{g0, g1, lo[0], lo[1]} : uint4{hi[0], hi[1], g0, g1};
          }
          const int c = t * fd + 4 * (hl & ~1);          // first channel of the chunk
          // (class_rows: the variance slabs of a single-observation row are all zero, the
          //  consumer skips them -- they are not written)
          if ((FD128 || hl < nq) && write_row && !(t == 1 && d.class_rows && nvis == 1))
            *reinterpret_cast<uint4*>(orow + (c >> 4) * 64 + (odd ? 32 : 0) + (c & 15) * 2) = chunk;
        }
      }
      if (hl == 0 && write_row) {           // (2 fd % 16 == 0: score_max opens a slab of its own)
        unsigned hi[2], lo[2];
        split_row_quad(f32x4{smax, 0.f, 0.f, 0.f}, hi, lo);
        uint4* ps = reinterpret_cast<uint4*>(orow + ((2 * fd) >> 4) * 64);
        ps[0] = uint4{hi[0], 0u, 0u, 0u};
        ps[1] = uint4{0u, 0u, 0u, 0u};
        ps[2] = uint4{lo[0], 0u, 0u, 0u};
        ps[3] = uint4{0u, 0u, 0u, 0u};
      }
    } else if ((FD128 || hl < nq) && write_row) {
      *reinterpret_cast<f32x4*>(out + 4 * hl) = mean;
      *reinterpret_cast<f32x4*>(out + fd + 4 * hl) = var;
    }
    if (hl == 0) {
      if (write_row && !d.out_split) {
        out[2 * fd] = smax;
        for (int c = 2 * fd + 1; c < d.out_stride; ++c) out[c] = 0.f;
      }
      bool vld = nvis > 0;
      if (d.max_view_distance >= 0.f && !all_views) vld = vld && (min_dist <= d.max_view_distance);
      a.valid[gv] = vld ? (d.class_rows && nvis > 1 ? 2 : 1) : 0;
    }
  }
}

__global__ void project_points_kernel(int B, int V, int N, int fisheye,
                                      const float* __restrict__ cam, const float* __restrict__ Rt,
                                      const float* __restrict__ pts, float* __restrict__ p2d,
                                      uint8_t* __restrict__ vis, float* __restrict__ depth) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * N * V;
  if (i >= total) return;
  const int v = (int)(i % V);
  const int64_t bn = i / V;
  const int b = (int)(bn / N);
  const float* p = pts + bn * 3;
  const Proj pr = project_one(cam + ((int64_t)b * V + v) * 11, Rt + ((int64_t)b * V + v) * 12, p[0],
                              p[1], p[2], fisheye);
  p2d[i * 2 + 0] = pr.pi;
  p2d[i * 2 + 1] = pr.pj;
  vis[i] = pr.vis ? 1 : 0;
  depth[i] = pr.depth;
}

}  // namespace

static int lift_pool_launch(const SnapLiftDesc* desc, const float* f_images,
                            const float* cam, const float* Rt, const float* points,
                            float* pooled, uint8_t* valid, uint32_t* tap_recs, void* stream) {
  if (!desc || !f_images || !cam || !Rt || !points || !pooled || !valid) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  // tap records exist in the batched kernel with classed, pre-split, valid-only rows
  if (tap_recs && (!d.class_rows || !d.out_split || !d.valid_rows_only ||
                   (reinterpret_cast<uintptr_t>(tap_recs) & 15)))
    return SNAP_ERR_UNSUPPORTED;
  if (d.B <= 0 || d.V <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.V > 32) return SNAP_ERR_UNSUPPORTED;
  if (d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.feature_dim <= 0) return SNAP_ERR_UNSUPPORTED;
  const bool dflt = d.weighted && d.use_variance && !d.add_minmax;
  const int bins = d.weighted ? d.num_bins : 0;
  if (d.C != d.feature_dim + bins || d.C % 4 != 0 || (d.weighted && d.num_bins < 1)) return SNAP_ERR_BAD_SHAPE;
  const int chans = d.feature_dim * (1 + (d.use_variance ? 1 : 0) + (d.add_minmax ? 2 : 0)) + (d.weighted ? 1 : 0);
  if (d.out_stride < chans || d.out_stride % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (d.out_split) {      // default options' batched kernel only; whole slabs; score_max on a slab boundary
    const int nsel_ = d.K == 0 ? d.V : d.K;
    if (!dflt || nsel_ > 4 || d.feature_dim % 8 != 0 || d.out_stride < ((chans + 15) / 16) * 16 ||
        (int64_t)d.B * d.V * d.h * d.w * d.C * 4 >= (1LL << 32))
      return SNAP_ERR_UNSUPPORTED;
  }
  if (d.class_rows && (!d.out_split || d.feature_dim % 16 != 0)) return SNAP_ERR_UNSUPPORTED;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;  // K>0 means V > K
  if (!(d.depth_max > d.depth_min) || !(d.depth_min > 0.f)) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(f_images) & 15) || (reinterpret_cast<uintptr_t>(pooled) & 15))
    return SNAP_ERR_BAD_SHAPE;
  const int nsel = d.K == 0 ? d.V : d.K;
  constexpr int xcd_group = 64;     // workgroups per XCD-owned chunk (section 5h of DESIGN.md: settled)
  LiftArgs a{xcd_group, 0, 3, 3, 0, 0, 0, d, f_images, cam, Rt, points, pooled, valid, nullptr, nullptr, nullptr, tap_recs};
  const int64_t total = (int64_t)d.B * d.N;
  if (total > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  if (d.grid_y < 0 || d.grid_z < 0 || (d.grid_y > 0) != (d.grid_z > 0)) return SNAP_ERR_BAD_SHAPE;
  if (d.grid_y > 0 && d.N % (d.grid_y * d.grid_z) != 0) return SNAP_ERR_BAD_SHAPE;
  const dim3 grid((unsigned)snap_cdiv(total, 8));
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr bool batched = true;    // (the one-voxel-per-half-wave kernel remains for > 4 selected views)
  // 8 half-waves x 32 voxels per workgroup; with XCD groups the grid is padded to whole
  // (8 XCDs x G) rounds -- workgroups past the range exit at once
  const int64_t nb = snap_cdiv(total, 256);
  const int64_t round = xcd_group > 0 ? 8LL * xcd_group : 1;
  dim3 bgrid((unsigned)(snap_cdiv(nb, round) * round));
  if (d.grid_y > 0) {                // (no grid hint = linear voxel order)
    const int GX = d.N / (d.grid_y * d.grid_z);
    // 8 x 8-column tiles (4 x 4 / 16 x 16 measured slower); a grid narrower than 8 columns (the query
    // frustum is [Nq, 1, Z]) takes 64 columns as (64 / ty) x ty -- square tiles left 7 of 8 lanes dead there
    int tile_log = 3;
    while (tile_log > 0 && (1 << (tile_log - 1)) >= d.grid_y) --tile_log;
    const int ts = 1 << tile_log, tsx = 64 / ts;
    a.tile_log = tile_log;
    a.tile_log_x = 6 - tile_log;
    a.tile_cpt = (tsx * ts * d.grid_z + 255) / 256;
    a.tiles_x = (GX + tsx - 1) / tsx;
    a.tiles_y = (d.grid_y + ts - 1) / ts;
    a.tiles_total = (int64_t)d.B * a.tiles_x * a.tiles_y;
    bgrid = dim3((unsigned)(snap_cdiv(a.tiles_total, 8) * 8 * a.tile_cpt));
  }
  // (the batched kernels address the taps by 32-bit byte offsets: f_images < 4 GB)
  const bool small = (int64_t)d.B * d.V * d.h * d.w * d.C * 4 < (1LL << 32);
  const bool fd128 = d.feature_dim == 128;
  const bool use_b = (batched || d.out_split) && dflt && small;   // (out_split exists there only)
  if (use_b && nsel <= 1) {
    if (fd128) hipLaunchKernelGGL((lift_pool_batched_kernel<1, true>), bgrid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lift_pool_batched_kernel<1, false>), bgrid, dim3(256), 0, s, a);
  } else if (use_b && nsel <= 4) {
    if (fd128) hipLaunchKernelGGL((lift_pool_batched_kernel<4, true>), bgrid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lift_pool_batched_kernel<4, false>), bgrid, dim3(256), 0, s, a);
  } else if (nsel <= 1) {
    hipLaunchKernelGGL(lift_pool_kernel<1>, grid, dim3(256), 0, s, a);
  } else if (nsel <= 4) {
    hipLaunchKernelGGL(lift_pool_kernel<4>, grid, dim3(256), 0, s, a);
  } else if (nsel <= 8) {
    hipLaunchKernelGGL(lift_pool_kernel<8>, grid, dim3(256), 0, s, a);
  } else {
    return SNAP_ERR_UNSUPPORTED;
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_lift_pool_f32(const SnapLiftDesc* desc, const float* f_images,
                                  const float* cam, const float* Rt, const float* points,
                                  float* pooled, uint8_t* valid, void* stream) {
  return lift_pool_launch(desc, f_images, cam, Rt, points, pooled, valid, nullptr, stream);
}

extern "C" int snap_lift_pool_records_f32(const SnapLiftDesc* desc, const float* f_images,
                                          const float* cam, const float* Rt, const float* points,
                                          float* pooled, uint8_t* valid, uint32_t* tap_records,
                                          void* stream) {
  if (!tap_records) return SNAP_ERR_NULL;
  return lift_pool_launch(desc, f_images, cam, Rt, points, pooled, valid, tap_records, stream);
}

// depth_mlp fusion (streetview_encoder.py:263-267, do_weighted_fusion = False): the observations
// leave the lift un-pooled, a per-observation MLP corrects them, a second pass pools them.
static int launch_obs(const SnapLiftDesc& d, const float* f_images, const float* cam, const float* Rt,
                      const float* points, float* pooled, uint8_t* valid, float* obs_out,
                      float* obs_feat, const float* obs_in, hipStream_t s) {
  if (d.B <= 0 || d.V <= 0 || d.V > 32 || d.h <= 0 || d.w <= 0 || d.N <= 0) return SNAP_ERR_BAD_SHAPE;
  if (d.weighted || d.out_split || d.valid_rows_only) return SNAP_ERR_UNSUPPORTED;
  if (d.feature_dim <= 0 || d.feature_dim % 4 != 0 || d.feature_dim > 128 || d.C != d.feature_dim)
    return SNAP_ERR_BAD_SHAPE;
  if (d.K < 0 || (d.K > 0 && d.K >= d.V)) return SNAP_ERR_BAD_SHAPE;
  if ((int64_t)d.B * d.N > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  const int nsel = d.K == 0 ? d.V : d.K;
  if (nsel > 8) return SNAP_ERR_UNSUPPORTED;
  LiftArgs a{0, 0, 3, 3, 0, 0, 0, d, f_images, cam, Rt, points, pooled, valid, obs_out, obs_feat, obs_in, nullptr};
  const dim3 grid((unsigned)snap_cdiv((int64_t)d.B * d.N, 8));
  if (nsel <= 1) hipLaunchKernelGGL(lift_pool_kernel<1>, grid, dim3(256), 0, s, a);
  else if (nsel <= 4) hipLaunchKernelGGL(lift_pool_kernel<4>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(lift_pool_kernel<8>, grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_lift_observations_f32(const SnapLiftDesc* desc, const float* f_images,
                                          const float* cam, const float* Rt, const float* points,
                                          float* obs, float* obs_feat, uint8_t* valid, void* stream) {
  if (!desc || !f_images || !cam || !Rt || !points || !obs || !obs_feat || !valid) return SNAP_ERR_NULL;
  return launch_obs(*desc, f_images, cam, Rt, points, nullptr, valid, obs, obs_feat, nullptr,
                    static_cast<hipStream_t>(stream));
}

extern "C" int snap_lift_pool_observations_f32(const SnapLiftDesc* desc, const float* cam,
                                               const float* Rt, const float* points,
                                               const float* obs_feat, float* pooled, uint8_t* valid,
                                               void* stream) {
  if (!desc || !cam || !Rt || !points || !obs_feat || !pooled || !valid) return SNAP_ERR_NULL;
  const SnapLiftDesc& d = *desc;
  const int chans = d.feature_dim * (1 + (d.use_variance ? 1 : 0) + (d.add_minmax ? 2 : 0));
  if (d.out_stride < chans || d.out_stride % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  return launch_obs(d, obs_feat /* (never read) */, cam, Rt, points, pooled, valid, nullptr, nullptr,
                    obs_feat, static_cast<hipStream_t>(stream));
}

extern "C" int snap_project_points_f32(int32_t B, int32_t V, int32_t N, int32_t fisheye,
                                       const float* cam, const float* Rt, const float* points,
                                       float* p2d, uint8_t* vis, float* depth, void* stream) {
  if (!cam || !Rt || !points || !p2d || !vis || !depth) return SNAP_ERR_NULL;
  if (B <= 0 || V <= 0 || N <= 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)B * N * V;
  hipLaunchKernelGGL(project_points_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, V, N, fisheye, cam, Rt, points, p2d, vis,
                     depth);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
