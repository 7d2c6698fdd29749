// Backward kernels of the pose head (training path):
//   * pose_score backward: d scores[B,P] -> d sim_points[B,Nq,X,Y]
//       (VJP of pose_estimation.py:63-82: bilinear scatter of every pose's gradient
//        into the point's score plane, accumulated in LDS, one plane at a time)
//   * similarity backward helper: G = d sim * [sim > 0] * scale / num_valid in place
//       + the temperature gradient sum(d sim * sim)   (VJP of bev_localizer.py:157-173;
//        prob_points is behind stop_gradient in the reference, :178)
// The two GEMM-shaped contractions of the similarity VJP (d fq = G fm, d fm = G^T fq)
// run on the MFMA engines (conv_igemm.hip / wgrad.hip).
#include "common.h"

namespace {

constexpr int PB_THREADS = 1024;
constexpr int PB_PPT = 10;
constexpr int PB_LDS_FLOATS = 24 * 1024;  // 96 KiB accumulator plane

struct ScoreBwdArgs {
  const float* dscores;   // [B,P]
  const float* table;     // [B,P,4] (A, B, Cx, Cy) cell-unit affine poses
  const float* q_xy;      // [B,Nq,2]
  const uint8_t* valid_q; // [B,Nq]
  const uint8_t* map_valid;
  int B, Nq, X, Y, P;
  int mask_oob;
  int points_per_chunk;
  float* dsim;            // [B,Nq,X,Y]
};

template <bool MASK>
__global__ __launch_bounds__(PB_THREADS) void pose_score_bwd_kernel(const ScoreBwdArgs a) {
  extern __shared__ float plane[];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int XY = a.X * a.Y;
  const float Xf = (float)a.X, Yf = (float)a.Y;
  const int n_begin = blockIdx.x * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  const uint8_t* mvalid = a.map_valid ? a.map_valid + (int64_t)b * XY : nullptr;
  const int passes = (a.P + PB_THREADS * PB_PPT - 1) / (PB_THREADS * PB_PPT);
  for (int n = n_begin; n < n_end; ++n) {
    float* dst = a.dsim + ((int64_t)b * a.Nq + n) * XY;
    if (!a.valid_q[(int64_t)b * a.Nq + n]) {   // block-uniform: zero gradient plane
      for (int i = tid; i < XY; i += PB_THREADS) dst[i] = 0.f;
      continue;
    }
    for (int i = tid; i < XY; i += PB_THREADS) plane[i] = 0.f;
    __syncthreads();
    const float qx = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
    const float qy = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
    for (int ps = 0; ps < passes; ++ps) {
#pragma unroll
      for (int k = 0; k < PB_PPT; ++k) {
        const int p = (ps * PB_PPT + k) * PB_THREADS + tid;
        if (p >= a.P) continue;
        const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + p];
        const float g = a.dscores[(int64_t)b * a.P + p];
        const float cu = fmaf(t[0], qx, fmaf(-t[1], qy, t[2]));
        const float cv = fmaf(t[1], qx, fmaf(t[0], qy, t[3]));
        const float fu = floorf(cu), fv = floorf(cv);
        const int i0 = (int)fminf(fmaxf(fu, 0.f), Xf - 1.f);
        const int i1 = (int)fminf(fmaxf(fu + 1.f, 0.f), Xf - 1.f);
        const int j0 = (int)fminf(fmaxf(fv, 0.f), Yf - 1.f);
        const int j1 = (int)fminf(fmaxf(fv + 1.f, 0.f), Yf - 1.f);
        const float wu1 = cu - fu, wu0 = 1.f - wu1;
        const float wv1 = cv - fv, wv0 = 1.f - wv1;
        bool ok = true;
        if (MASK) {
          const float u = cu + 0.5f, v = cv + 0.5f;
          ok = (u >= 0.f) && (u < Xf) && (v >= 0.f) && (v < Yf);
          ok = ok && mvalid[i0 * a.Y + j0] && mvalid[i0 * a.Y + j1] && mvalid[i1 * a.Y + j0] &&
               mvalid[i1 * a.Y + j1];
        }
        if (ok && g != 0.f) {
          atomicAdd(&plane[i0 * a.Y + j0], (wu0 * wv0) * g);
          atomicAdd(&plane[i0 * a.Y + j1], (wu0 * wv1) * g);
          atomicAdd(&plane[i1 * a.Y + j0], (wu1 * wv0) * g);
          atomicAdd(&plane[i1 * a.Y + j1], (wu1 * wv1) * g);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < XY; i += PB_THREADS) dst[i] = plane[i];
    __syncthreads();
  }
}

// Fast path (P <= PB_THREADS * PB_PPT, no validity mask): every thread keeps its 10 poses and
// their cotangents in registers for the whole point loop (the general kernel re-reads the 200 KB
// pose table from L2 for every point), samples with the forward kernel's clamped-coordinate
// formulation (pose.hip: identical derivative; the zero-weight taps of clamped samples are
// skipped) and keeps the four CORNER cells -- where every sample that leaves the map in both
// directions lands, thousands per plane -- in registers instead of serialising LDS atomics on
// one address; they are reduced per wave (DPP) and added once per plane.
typedef float bf32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(PB_THREADS) void pose_score_bwd_fast_kernel(const ScoreBwdArgs a) {
  extern __shared__ float plane[];
  __shared__ float corner[4];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int Y = a.Y;
  const int XY = a.X * Y;
  const float Xf = (float)a.X, Yf = (float)Y;
  const int n_begin = blockIdx.x * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  float pc[PB_PPT], ps[PB_PPT], ptx[PB_PPT], pty[PB_PPT], g[PB_PPT];
#pragma unroll
  for (int k = 0; k < PB_PPT; ++k) {
    const int p = k * PB_THREADS + tid;
    const bool live = p < a.P;
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + (live ? p : 0)];
    pc[k] = t[0]; ps[k] = t[1]; ptx[k] = t[2]; pty[k] = t[3];
    g[k] = live ? a.dscores[(int64_t)b * a.P + p] : 0.f;
  }
  for (int n = n_begin; n < n_end; ++n) {
    float* dst = a.dsim + ((int64_t)b * a.Nq + n) * XY;
    if (!a.valid_q[(int64_t)b * a.Nq + n]) {   // block-uniform: zero gradient plane
      for (int i = tid; i < XY; i += PB_THREADS) dst[i] = 0.f;
      continue;
    }
    for (int i = tid; i < XY; i += PB_THREADS) plane[i] = 0.f;
    if (tid < 4) corner[tid] = 0.f;
    __syncthreads();
    const float qx = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
    const float qy = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
    float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;   // corner cells (0,0) (0,Y-1) (X-1,0) (X-1,Y-1)
#pragma unroll
    for (int k = 0; k < PB_PPT; ++k) {
      const float gk = g[k];
      const float ru = fmaf(pc[k], qx, fmaf(-ps[k], qy, ptx[k]));
      const float rv = fmaf(ps[k], qx, fmaf(pc[k], qy, pty[k]));
      const bool ulo = ru <= 0.f, uhi = ru >= Xf - 1.f;
      const bool vlo = rv <= 0.f, vhi = rv >= Yf - 1.f;
      if ((ulo || uhi) && (vlo || vhi)) {       // clamped in both directions: one corner cell
        if (ulo && vlo) c00 += gk;
        else if (ulo) c01 += gk;
        else if (vlo) c10 += gk;
        else c11 += gk;
        continue;
      }
      if (gk == 0.f) continue;
      const float cu = fminf(fmaxf(ru, 0.f), Xf - 1.f), cv = fminf(fmaxf(rv, 0.f), Yf - 1.f);
      const float fu = fminf(floorf(cu), Xf - 2.f), fv = fminf(floorf(cv), Yf - 2.f);
      const float wu = cu - fu, wv = cv - fv;
      float* q = plane + (int)fmaf(fu, Yf, fv);
      const float g0 = (1.f - wu) * gk, g1 = wu * gk;
      const float w00 = g0 * (1.f - wv), w01 = g0 * wv, w10 = g1 * (1.f - wv), w11 = g1 * wv;
      if (w00 != 0.f) atomicAdd(q, w00);
      if (w01 != 0.f) atomicAdd(q + 1, w01);
      if (w10 != 0.f) atomicAdd(q + Y, w10);
      if (w11 != 0.f) atomicAdd(q + Y + 1, w11);
    }
    c00 = wave_sum(c00); c01 = wave_sum(c01); c10 = wave_sum(c10); c11 = wave_sum(c11);
    if (lane == 0) {
      atomicAdd(&corner[0], c00); atomicAdd(&corner[1], c01);
      atomicAdd(&corner[2], c10); atomicAdd(&corner[3], c11);
    }
    __syncthreads();
    if (tid == 0) {
      plane[0] += corner[0];
      plane[Y - 1] += corner[1];
      plane[(a.X - 1) * Y] += corner[2];
      plane[(a.X - 1) * Y + Y - 1] += corner[3];
    }
    __syncthreads();
    for (int i = tid; i < XY; i += PB_THREADS) dst[i] = plane[i];
    __syncthreads();
  }
}

// Deterministic form of the fast path: the plane is accumulated in 64-bit FIXED POINT
// (ds_add_u64: integer addition is associative, so the order in which the 16 waves' atomics reach
// the LDS does not matter -- the float version's sums depend on it).  Scale: 2^(38 - e) with
// 2^e >= the largest |cotangent| of the scene, found by an order-independent max over the
// register-resident cotangents; every term |w g| <= 2^e then has 38 fractional bits (the f32
// sums it replaces carry 24), ten thousand of them stay below 2^52, and a term is converted with
// the 2^52 + 2^51 magic-number addition in f64 (exact below 2^51).  Planes up to 160 x 128 cells
// (8 bytes per cell in the 160 KB LDS); the corner cells are summed per thread, per wave (DPP)
// and then over the waves in a fixed order.
__global__ __launch_bounds__(PB_THREADS) void pose_score_bwd_det_kernel(const ScoreBwdArgs a) {
  extern __shared__ long long iplane[];
  __shared__ float corner[PB_THREADS / 64][4];
  __shared__ float wmax[PB_THREADS / 64];
  __shared__ float wbad[PB_THREADS / 64];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int Y = a.Y;
  const int XY = a.X * Y;
  const float Xf = (float)a.X, Yf = (float)Y;
  const int n_begin = blockIdx.x * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  float pc[PB_PPT], ps[PB_PPT], ptx[PB_PPT], pty[PB_PPT], g[PB_PPT];
  float gm = 0.f, bad = 0.f;
#pragma unroll
  for (int k = 0; k < PB_PPT; ++k) {
    const int p = k * PB_THREADS + tid;
    const bool live = p < a.P;
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + (live ? p : 0)];
    pc[k] = t[0]; ps[k] = t[1]; ptx[k] = t[2]; pty[k] = t[3];
    g[k] = live ? a.dscores[(int64_t)b * a.P + p] : 0.f;
    const float ag = fabsf(g[k]);
    gm = (ag <= 3.0e38f) ? fmaxf(gm, ag) : gm;              // (non-finite cotangents do not set the scale)
    bad = (ag <= 3.0e38f) ? bad : 1.f;                      // NaN / Inf cotangent (NaN fails the compare)
  }
  gm = wave_max(gm);
  bad = wave_max(bad);
  if (lane == 0) { wmax[wave] = gm; wbad[wave] = bad; }
  __syncthreads();
  gm = 0.f; bad = 0.f;
#pragma unroll
  for (int w = 0; w < PB_THREADS / 64; ++w) { gm = fmaxf(gm, wmax[w]); bad = fmaxf(bad, wbad[w]); }
  if (bad != 0.f) {
    // A non-finite cotangent of this scene cannot go through the fixed-point sums (the conversion
    // would turn it into a large FINITE number and the trainer's non-finite step skip /
    // DynamicScale back-off, trainer.py:260-277, would never see the overflow).  The float path
    // spreads NaN / Inf over the taps of that pose and from there into every parameter gradient;
    // here the scene's gradient planes are NaN outright: the step is skipped either way.
    const float nan = __int_as_float(0x7fc00000);
    for (int n = n_begin; n < n_end; ++n) {
      float* dst = a.dsim + ((int64_t)b * a.Nq + n) * XY;
      const float v = a.valid_q[(int64_t)b * a.Nq + n] ? nan : 0.f;
      for (int i = tid; i < XY; i += PB_THREADS) dst[i] = v;
    }
    return;
  }
  int e = 0;
  frexpf(fmaxf(gm, 1e-37f), &e);                             // gm < 2^e
  const double scale = ldexp(1.0, 38 - e), unscale = ldexp(1.0, e - 38);
  const double magic = 6755399441055744.0;                   // 2^52 + 2^51
  auto fixed = [&](float v) -> long long {
    const double t = fma((double)v, scale, magic);
    return __double_as_longlong(t) - __double_as_longlong(magic);
  };
  for (int n = n_begin; n < n_end; ++n) {
    float* dst = a.dsim + ((int64_t)b * a.Nq + n) * XY;
    if (!a.valid_q[(int64_t)b * a.Nq + n]) {   // block-uniform: zero gradient plane
      for (int i = tid; i < XY; i += PB_THREADS) dst[i] = 0.f;
      continue;
    }
    for (int i = tid; i < XY; i += PB_THREADS) iplane[i] = 0;
    __syncthreads();
    const float qx = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
    const float qy = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
    float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;   // corner cells (0,0) (0,Y-1) (X-1,0) (X-1,Y-1)
#pragma unroll
    for (int k = 0; k < PB_PPT; ++k) {
      const float gk = g[k];
      const float ru = fmaf(pc[k], qx, fmaf(-ps[k], qy, ptx[k]));
      const float rv = fmaf(ps[k], qx, fmaf(pc[k], qy, pty[k]));
      const bool ulo = ru <= 0.f, uhi = ru >= Xf - 1.f;
      const bool vlo = rv <= 0.f, vhi = rv >= Yf - 1.f;
      if ((ulo || uhi) && (vlo || vhi)) {       // clamped in both directions: one corner cell
        if (ulo && vlo) c00 += gk;
        else if (ulo) c01 += gk;
        else if (vlo) c10 += gk;
        else c11 += gk;
        continue;
      }
      if (gk == 0.f) continue;
      const float cu = fminf(fmaxf(ru, 0.f), Xf - 1.f), cv = fminf(fmaxf(rv, 0.f), Yf - 1.f);
      const float fu = fminf(floorf(cu), Xf - 2.f), fv = fminf(floorf(cv), Yf - 2.f);
      const float wu = cu - fu, wv = cv - fv;
      unsigned long long* q = reinterpret_cast<unsigned long long*>(iplane) + (int)fmaf(fu, Yf, fv);
      const float g0 = (1.f - wu) * gk, g1 = wu * gk;
      const float w00 = g0 * (1.f - wv), w01 = g0 * wv, w10 = g1 * (1.f - wv), w11 = g1 * wv;
      if (w00 != 0.f) atomicAdd(q, (unsigned long long)fixed(w00));
      if (w01 != 0.f) atomicAdd(q + 1, (unsigned long long)fixed(w01));
      if (w10 != 0.f) atomicAdd(q + Y, (unsigned long long)fixed(w10));
      if (w11 != 0.f) atomicAdd(q + Y + 1, (unsigned long long)fixed(w11));
    }
    c00 = wave_sum(c00); c01 = wave_sum(c01); c10 = wave_sum(c10); c11 = wave_sum(c11);
    if (lane == 0) { corner[wave][0] = c00; corner[wave][1] = c01; corner[wave][2] = c10; corner[wave][3] = c11; }
    __syncthreads();
    if (tid < 4) {                               // the waves' corner sums in a FIXED order
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < PB_THREADS / 64; ++w) t += corner[w][tid];
      const int cell = tid == 0 ? 0 : tid == 1 ? Y - 1 : tid == 2 ? (a.X - 1) * Y : (a.X - 1) * Y + Y - 1;
      corner[0][tid] = (float)((double)iplane[cell] * unscale) + t;
    }
    __syncthreads();
    for (int i = tid; i < XY; i += PB_THREADS) {
      const bool is_corner = i == 0 || i == Y - 1 || i == (a.X - 1) * Y || i == (a.X - 1) * Y + Y - 1;
      const int ci = i == 0 ? 0 : i == Y - 1 ? 1 : i == (a.X - 1) * Y ? 2 : 3;
      dst[i] = is_corner ? corner[0][ci] : (float)((double)iplane[i] * unscale);
    }
    __syncthreads();
  }
}

// G = dsim * [sim > 0] * coef[b] in place; per-block partial of sum(dsim * sim).
__global__ __launch_bounds__(256) void sim_bwd_prepare_kernel(float* __restrict__ dsim,
                                                              const float* __restrict__ sim,
                                                              int64_t per_scene, int clip,
                                                              const float* __restrict__ coef,
                                                              float* __restrict__ partial) {
  __shared__ float red[256];
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * per_scene;
  const float cf = coef[b];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_scene;
       i += (int64_t)gridDim.x * 256) {
    const float g = dsim[base + i];
    const float s = sim[base + i];
    acc += g * s;
    dsim[base + i] = (!clip || s > 0.f) ? g * cf : 0.f;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = red[0];
}

// add_confidence_query (bev_localizer.py:165-168): sim = relu(fq . fm) * scale * w[b, n].  One wave per
// (scene, point): rowdot[b, n] = sum_cells dsim * sim (fixed order: a lane's cells ascending, then
// the DPP wave sum) -- it is both the row's share of the temperature gradient and, divided by
// w[b, n], the gradient of the weight --, then G = dsim * [sim > 0] * coef[b, n] in place.
__global__ __launch_bounds__(256) void sim_bwd_prepare_rows_kernel(float* __restrict__ dsim,
                                                                   const float* __restrict__ sim,
                                                                   int64_t rows, int XY, int clip,
                                                                   const float* __restrict__ row_coef,
                                                                   float* __restrict__ rowdot) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* g = dsim + r * XY;
  const float* s = sim + r * XY;
  const float cf = row_coef[r];
  float acc = 0.f;
  for (int i = lane; i < XY; i += 64) {
    const float gv = g[i], sv = s[i];
    acc += gv * sv;
    g[i] = (!clip || sv > 0.f) ? gv * cf : 0.f;
  }
  acc = wave_sum(acc);
  if (lane == 0) rowdot[r] = acc;
}

// VJP of layers.masked_softmax over the last axis (layers.py:38-43; an all-false mask acts as
// all-true): dx = w * (dw - sum_n w dw) on the softmax's support, 0 elsewhere.  One workgroup
// per row, fixed-order sums.
__global__ __launch_bounds__(256) void masked_softmax_rows_bwd_kernel(const float* __restrict__ w,
                                                                      const float* __restrict__ dw,
                                                                      int N, float* __restrict__ dx) {
  __shared__ float red[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* wr = w + (int64_t)b * N;
  const float* dr = dw + (int64_t)b * N;
  const int seg = (N + 255) / 256;
  const int i0 = min(t * seg, N), i1 = min(i0 + seg, N);
  float loc = 0.f;
  for (int i = i0; i < i1; ++i) loc += wr[i] * dr[i];          // (w = 0 outside the support)
  red[t] = loc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float dot = red[0];
  for (int i = i0; i < i1; ++i) dx[(int64_t)b * N + i] = wr[i] * (dr[i] - dot);
}

__global__ void pose_table_cells_bwd_kernel(const float* __restrict__ poses, int64_t total,
                                            float cell, float* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float th = poses[i * 3 + 0];
  f32x4 t;
  t[0] = cosf(th) / cell; t[1] = sinf(th) / cell;
  t[2] = poses[i * 3 + 1] / cell - 0.5f; t[3] = poses[i * 3 + 2] / cell - 0.5f;
  reinterpret_cast<f32x4*>(table)[i] = t;
}

}  // namespace

extern "C" size_t snap_pose_score_bwd_workspace_bytes(int32_t B, int32_t P) {
  return (size_t)B * P * 4 * sizeof(float);
}

extern "C" int snap_pose_score_bwd_f32(const float* dscores, const float* poses, const float* q_xy,
                                       const uint8_t* valid_q, const uint8_t* map_valid, int32_t B,
                                       int32_t Nq, int32_t X, int32_t Y, int32_t P,
                                       float cell_size, int32_t mask_oob, float* dsim,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  return snap_pose_score_bwd_ex_f32(dscores, poses, q_xy, valid_q, map_valid, B, Nq, X, Y, P, cell_size,
                                    mask_oob, 0, dsim, workspace, workspace_bytes, stream);
}

extern "C" int snap_pose_score_bwd_ex_f32(const float* dscores, const float* poses, const float* q_xy,
                                          const uint8_t* valid_q, const uint8_t* map_valid, int32_t B,
                                          int32_t Nq, int32_t X, int32_t Y, int32_t P,
                                          float cell_size, int32_t mask_oob, int32_t flags, float* dsim,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!dscores || !poses || !q_xy || !valid_q || !dsim || !workspace) return SNAP_ERR_NULL;
  if (mask_oob && !map_valid) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || X <= 0 || Y <= 0 || P <= 0) return SNAP_ERR_BAD_SHAPE;
  if ((int64_t)X * Y > PB_LDS_FLOATS) return SNAP_ERR_UNSUPPORTED;   // band tiling: forward only
  if (workspace_bytes < snap_pose_score_bwd_workspace_bytes(B, P)) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* table = static_cast<float*>(workspace);
  hipLaunchKernelGGL(pose_table_cells_bwd_kernel, dim3((unsigned)snap_cdiv((int64_t)B * P, 256)),
                     dim3(256), 0, s, poses, (int64_t)B * P, cell_size, table);
  SNAP_CHECK_LAUNCH();
  ScoreBwdArgs a;
  a.dscores = dscores; a.table = table; a.q_xy = q_xy; a.valid_q = valid_q; a.map_valid = map_valid;
  a.B = B; a.Nq = Nq; a.X = X; a.Y = Y; a.P = P; a.mask_oob = mask_oob; a.dsim = dsim;
  int nch = (512 + B - 1) / B;
  if (nch > Nq) nch = Nq;
  a.points_per_chunk = (Nq + nch - 1) / nch;
  nch = (Nq + a.points_per_chunk - 1) / a.points_per_chunk;
  size_t lds = (size_t)X * Y * sizeof(float);
  const bool fast = !mask_oob && P <= PB_THREADS * PB_PPT && X >= 2 && Y >= 2;
  // deterministic (fixed-point) form where the 8-byte plane fits: bit 1 of mask_oob's word is
  // NOT used -- the choice is the shape's; SNAP_POSE_BWD_FLOAT_ATOMICS in `flags` asks for the
  // float form (A/B timing)
  const bool det = fast && (size_t)X * Y * 8 <= 152 * 1024 && !(flags & 1);
  const void* fn = det ? (const void*)&pose_score_bwd_det_kernel
                       : fast ? (const void*)&pose_score_bwd_fast_kernel
                              : (mask_oob ? (const void*)&pose_score_bwd_kernel<true>
                                          : (const void*)&pose_score_bwd_kernel<false>);
  if (det) lds = (size_t)X * Y * 8;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                            det ? (int)lds : (int)(PB_LDS_FLOATS * sizeof(float))) != hipSuccess)
      return SNAP_ERR_LAUNCH;
  }
  void* kargs[] = {(void*)&a};
  if (hipLaunchKernel(fn, dim3(nch, B), dim3(PB_THREADS), kargs, lds, s) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_sim_bwd_prepare_f32(float* dsim, const float* sim, int32_t B, int64_t per_scene,
                                        int32_t clip_negative, const float* coef, float* partial,
                                        int32_t num_partial, void* stream) {
  if (!dsim || !sim || !coef || !partial) return SNAP_ERR_NULL;
  if (B <= 0 || per_scene <= 0 || num_partial <= 0 || num_partial > 65535) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sim_bwd_prepare_kernel, dim3(num_partial, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dsim, sim, per_scene, clip_negative, coef,
                     partial);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_sim_bwd_prepare_rows_f32(float* dsim, const float* sim, int32_t B, int32_t Nq,
                                             int32_t XY, int32_t clip_negative, const float* row_coef,
                                             float* rowdot, void* stream) {
  if (!dsim || !sim || !row_coef || !rowdot) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || XY <= 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t rows = (int64_t)B * Nq;
  hipLaunchKernelGGL(sim_bwd_prepare_rows_kernel, dim3((unsigned)snap_cdiv(rows, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dsim, sim, rows, XY, clip_negative, row_coef, rowdot);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_masked_softmax_rows_bwd_f32(const float* weights, const float* dweights, int32_t B,
                                                int32_t N, float* dx, void* stream) {
  if (!weights || !dweights || !dx) return SNAP_ERR_NULL;
  if (B <= 0 || N <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(masked_softmax_rows_bwd_kernel, dim3((unsigned)B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), weights, dweights, N, dx);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
