// Bandwidth-bound helpers of the BiT ResNet-v2 / FPN encoder:
//   * StdConv weight standardisation   (snap/models/resnet.py:34-41,73-79)
//   * GroupNorm statistics (two-pass)  (snap/models/resnet.py:46-60)
//   * stand-alone GroupNorm(+ReLU) apply (tests / returned tensors)
//   * 3x3/2 max-pool with -inf padding (snap/models/resnet.py:99)
// The normalisation itself is fused into the conv engine's A-operand staging
// (conv_igemm.hip), so activations are read once for statistics and once by the
// consumer conv -- never written back normalised.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// Weight standardisation: w[K, Cout], statistics over K per column.
// block = 32 columns x 32 k-slices (1024 threads): a row segment of 32 columns is one full
// 128-byte line (8-column groups fetched 32-byte pieces of every line: 2.2 GB of fabric reads for
// the 0.2 GB of weights of the two encoders, 0.5 ms per step), and the statistics take ONE pass:
// sum(v - p) and sum((v - p)^2) around the pivot p = w[0, col] (a sample of the column: no
// cancellation), combined in double into the two-pass mean and mean((v - mean)^2) of
// resnet.py:73-79.  Fixed-order reductions: deterministic.
// ---------------------------------------------------------------------------
constexpr int WS_COLS = 32;
constexpr int WS_SLICES = 32;

__device__ __forceinline__ void weight_std_body(const float* __restrict__ w,
                                                float* __restrict__ out, int K, int Cout,
                                                float eps, int blk) {
  __shared__ float red[2][WS_SLICES][WS_COLS + 1];
  __shared__ float stat[2][WS_COLS];
  const int tc = threadIdx.x % WS_COLS, tk = threadIdx.x / WS_COLS;
  const int col = blk * WS_COLS + tc;
  const bool ok = col < Cout;
  const float piv = ok ? w[col] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int k = tk; k < K; k += WS_SLICES) {
      const float t = w[(int64_t)k * Cout + col] - piv;
      s1 += t;
      s2 += t * t;
    }
  red[0][tk][tc] = s1;
  red[1][tk][tc] = s2;
  __syncthreads();
  if (tk == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < WS_SLICES; ++i) {
      t1 += (double)red[0][i][tc];
      t2 += (double)red[1][i][tc];
    }
    const double m = t1 / (double)K;                 // mean - pivot
    const double var = t2 / (double)K - m * m;       // mean((v - mean)^2)
    stat[0][tc] = (float)((double)piv + m);
    stat[1][tc] = sqrtf((float)fmax(var, 0.0) + eps);
  }
  __syncthreads();
  const float mean = stat[0][tc];
  const float denom = stat[1][tc];
  if (ok)
    for (int k = tk; k < K; k += WS_SLICES) {
      const int64_t o = (int64_t)k * Cout + col;
      out[o] = (w[o] - mean) / denom;
    }
}

// The same for Cout % 4 == 0 (every StdConv of the encoders): a thread owns a QUAD of columns and 16-byte
// loads, four of them in flight, 128 k-slices per workgroup -- the scalar body above keeps one 4-byte load in
// flight per thread and ran at 1.4 TB/s (131 us per encoder and step; this one: see DESIGN.md 5).  Same
// per-element arithmetic around the same pivots; the k-slices partition the sums differently (fixed order:
// deterministic; the statistics agree to f64 rounding of other partial sums).
constexpr int WS4_SLICES = 128;

__device__ __forceinline__ void weight_std_body_v4(const float* __restrict__ w, float* __restrict__ out,
                                                   int K, int Cout, float eps, int blk) {
  __shared__ float red4[2][WS4_SLICES][WS_COLS + 1];
  __shared__ float stat4[2][WS_COLS];
  const int tq = threadIdx.x & 7, tk = threadIdx.x >> 3;       // 8 column quads x 128 k-slices
  const int col0 = blk * WS_COLS + 4 * tq;
  const bool ok = col0 < Cout;                                  // (Cout % 4 == 0: whole quads)
  const f32x4 piv = ok ? *reinterpret_cast<const f32x4*>(w + col0) : f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    int k = tk;
    for (; k + 3 * WS4_SLICES < K; k += 4 * WS4_SLICES) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const f32x4*>(w + (int64_t)(k + u * WS4_SLICES) * Cout + col0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = v[u][e] - piv[e];
          s1[e] += t;
          s2[e] += t * t;
        }
    }
    for (; k < K; k += WS4_SLICES) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + (int64_t)k * Cout + col0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = v[e] - piv[e];
        s1[e] += t;
        s2[e] += t * t;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red4[0][tk][4 * tq + e] = s1[e];
    red4[1][tk][4 * tq + e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < WS_COLS) {
    const int tc = threadIdx.x;
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < WS4_SLICES; ++i) {
      t1 += (double)red4[0][i][tc];
      t2 += (double)red4[1][i][tc];
    }
    const float pv = (blk * WS_COLS + tc) < Cout ? w[blk * WS_COLS + tc] : 0.f;
    const double m = t1 / (double)K;                 // mean - pivot
    const double var = t2 / (double)K - m * m;       // mean((v - mean)^2)
    stat4[0][tc] = (float)((double)pv + m);
    stat4[1][tc] = sqrtf((float)fmax(var, 0.0) + eps);
  }
  __syncthreads();
  if (ok) {
    f32x4 mean, denom;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mean[e] = stat4[0][4 * tq + e]; denom[e] = stat4[1][4 * tq + e]; }
    for (int k = tk; k < K; k += WS4_SLICES) {
      const int64_t o = (int64_t)k * Cout + col0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + o);
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = (v[e] - mean[e]) / denom[e];
      *reinterpret_cast<f32x4*>(out + o) = r;
    }
  }
}

__device__ __forceinline__ bool weight_std_quads_ok(const float* w, const float* out, int Cout) {
  return (Cout & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

__global__ __launch_bounds__(1024) void weight_std_kernel(const float* __restrict__ w,
                                                          float* __restrict__ out, int K,
                                                          int Cout, float eps) {
  if (weight_std_quads_ok(w, out, Cout)) weight_std_body_v4(w, out, K, Cout, eps, blockIdx.x);
  else weight_std_body(w, out, K, Cout, eps, blockIdx.x);
}

// every StdConv kernel of an encoder in ONE launch: workgroup -> (item, column group of 32) by
// binary search over the items' first-workgroup table.
__global__ __launch_bounds__(1024) void weight_std_multi_kernel(const SnapWstdItem* __restrict__ items,
                                                                int n_items, float eps) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SnapWstdItem it = items[lo];
  if (weight_std_quads_ok(it.w, it.out, it.Cout)) weight_std_body_v4(it.w, it.out, it.K, it.Cout, eps, blockIdx.x - it.block_begin);
  else weight_std_body(it.w, it.out, it.K, it.Cout, eps, blockIdx.x - it.block_begin);
}

// ---------------------------------------------------------------------------
// GroupNorm statistics -- ONE pass over the activation.
// grid = (slabs, N, channel chunks of <=1024); block = 256 threads laid out as
// QW float4-quads x PW pixels.  Every thread accumulates sum(v - p) and
// sum((v - p)^2) around a per-channel pivot p = v[pixel 0] (a sample of the data, so
// no catastrophic cancellation); the finalize kernel combines slabs and channels
// in double precision into the exact two-pass quantities mean and
// mean((v - mean)^2) of resnet.py:38-40.  Deterministic: fixed-order reductions.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(
    const float* __restrict__ x, int HW, int C, int Cs, int relu_first, int ppb,
    float* __restrict__ partial /*[N,S,C,2]*/) {
  __shared__ float part[256 * 8];
  const int S = gridDim.x;
  const int n = blockIdx.y;
  const int cbase = blockIdx.z * 1024;
  const int cchunk = min(C - cbase, 1024);
  const int QW = cchunk >> 2;
  const int PW = 256 / QW;
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int c0 = cbase + 4 * tq;
  const int p_begin = blockIdx.x * ppb;
  const int p_end = min(p_begin + ppb, HW);
  const float* xb = x + ((int64_t)n * HW) * Cs + c0;
  f32x4 piv = *reinterpret_cast<const f32x4*>(xb);
  if (relu_first) {
#pragma unroll
    for (int e = 0; e < 4; ++e) piv[e] = snap_relu(piv[e]);
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = p_begin + tp; p < p_end; p += PW) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (int64_t)p * Cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = (relu_first ? snap_relu(v[e]) : v[e]) - piv[e];
      s1[e] += t;
      s2[e] += t * t;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    part[threadIdx.x * 8 + e] = s1[e];
    part[threadIdx.x * 8 + 4 + e] = s2[e];
  }
  __syncthreads();
  // per-channel sums over the PW pixel lanes (fixed order)
  for (int c = threadIdx.x; c < cchunk; c += 256) {
    const int q = c >> 2, e = c & 3;
    float a1 = 0.f, a2 = 0.f;
    for (int pp = 0; pp < PW; ++pp) {
      a1 += part[(pp * QW + q) * 8 + e];
      a2 += part[(pp * QW + q) * 8 + 4 + e];
    }
    float* o = partial + (((int64_t)n * S + blockIdx.x) * C + cbase + c) * 2;
    o[0] = a1;
    o[1] = a2;
  }
}

// one wave per (image, group): a flat fp64 reduction over (slab, channel).
//   T1 = sum a1, T2 = sum a2, U = sum p_c*a1, P1 = sum_c p_c, P2 = sum_c p_c^2
//   mean = (T1 + HW*P1) / (HW*cpg)
//   M2   = T2 - 2*mean*T1 + 2*U + HW*(cpg*mean^2 - 2*mean*P1 + P2)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// tile_rows > 0: `partial` comes from the conv engine's epilogue (conv_igemm.hip): plain
// sums (pivot 0) per row tile of `tile_rows` output pixels, S = HW / tile_rows + 2 slots
// per image of which only the tiles overlapping the image are live; x is not read.
__global__ __launch_bounds__(256) void gn_finalize_kernel(
    const float* __restrict__ x, const float* __restrict__ partial, int S, int HW, int C, int Cs,
    int groups, int relu_first, float eps, const float* __restrict__ gamma,
    float* __restrict__ mu, float* __restrict__ sc, float* __restrict__ rstd_out, int total,
    int tile_rows) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // over N*groups
  if (i >= total) return;
  const int n = i / groups, g = i - n * groups;
  const int cpg = C / groups;
  const int c_lo = g * cpg;
  const bool tiled = tile_rows > 0;
  const float* xp = tiled ? nullptr : x + ((int64_t)n * HW) * Cs + c_lo;
  double t1 = 0.0, t2 = 0.0, u = 0.0, p1 = 0.0, p2 = 0.0;
  int live = S;
  if (tiled)
    live = (int)((((int64_t)(n + 1) * HW - 1) / tile_rows) - (((int64_t)n * HW) / tile_rows)) + 1;
  const int count = live * cpg;
  for (int e = lane; e < count; e += 64) {
    const int s = e / cpg, cc = e - s * cpg;
    float pv = tiled ? 0.f : xp[cc];
    if (relu_first && !tiled) pv = snap_relu(pv);
    const float* pp = partial + (((int64_t)n * S + s) * C + c_lo + cc) * 2;
    const double a1 = (double)pp[0];
    t1 += a1;
    t2 += (double)pp[1];
    u += (double)pv * a1;
  }
  for (int cc = lane; cc < cpg && !tiled; cc += 64) {
    float pv = xp[cc];
    if (relu_first) pv = snap_relu(pv);
    p1 += (double)pv;
    p2 += (double)pv * (double)pv;
  }
  t1 = wave_sum_f64(t1); t2 = wave_sum_f64(t2); u = wave_sum_f64(u);
  p1 = wave_sum_f64(p1); p2 = wave_sum_f64(p2);
  const double cnt = (double)HW;
  const double mean = (t1 + cnt * p1) / (cnt * cpg);
  const double m2 = t2 - 2.0 * mean * t1 + 2.0 * u + cnt * (cpg * mean * mean - 2.0 * mean * p1 + p2);
  const float meanf = (float)mean;
  const float var = (float)(m2 / (cnt * cpg));
  // x / sqrt(mean(x^2) + eps): division by the sqrt, as in resnet.py:40.
  const float rstd = 1.0f / sqrtf(snap_relu(var) + eps);
  for (int cc = lane; cc < cpg; cc += 64) {
    mu[(int64_t)n * C + c_lo + cc] = meanf;
    sc[(int64_t)n * C + c_lo + cc] = rstd * gamma[c_lo + cc];
    if (rstd_out) rstd_out[(int64_t)n * C + c_lo + cc] = rstd;
  }
}

// The tiled case (statistics emitted by a conv epilogue) with a WORKGROUP per (image, group):
// the one-wave-per-group form above walks up to ~1000 partial sums in 16 dependent rounds of
// loads on a grid of N * groups / 4 workgroups (a 256-CU part mostly idle, ~7-12 us per launch,
// ~100 launches per forward pass); here every thread takes <= 4 of them and the waves' totals
// are combined in fixed order (deterministic).
__global__ __launch_bounds__(256) void gn_finalize_tiled_kernel(
    const float* __restrict__ partial, int S, int HW, int C, int groups, float eps,
    const float* __restrict__ gamma, float* __restrict__ mu, float* __restrict__ sc,
    float* __restrict__ rstd_out, int tile_rows) {
  __shared__ double wsum[4][2];
  const int i = blockIdx.x;                 // over N * groups
  const int n = i / groups, g = i - n * groups;
  const int cpg = C / groups;
  const int c_lo = g * cpg;
  // (tile_rows < 0: S = -tile_rows slabs per image, all of them live)
  const int live = tile_rows < 0 ? S
                   : (int)((((int64_t)(n + 1) * HW - 1) / tile_rows) - (((int64_t)n * HW) / tile_rows)) + 1;
  const int count = live * cpg;
  // (gamma travels with the partial sums instead of after the barrier: one round trip less per launch)
  const float gam = (int)threadIdx.x < cpg ? gamma[c_lo + threadIdx.x] : 0.f;
  double t1 = 0.0, t2 = 0.0;
  // four entries in flight per thread (the weights-stationary conv kernel emits 32-row slabs: up to
  // ~4600 entries per (image, group)); the order of the sum per thread is unchanged
  int e = threadIdx.x;
  for (; e + 768 < count; e += 1024) {
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ee = e + 256 * u;
      const int s = ee / cpg, cc = ee - s * cpg;
      v[u] = *reinterpret_cast<const float2*>(partial + (((int64_t)n * S + s) * C + c_lo + cc) * 2);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      t1 += (double)v[u].x;
      t2 += (double)v[u].y;
    }
  }
  for (; e < count; e += 256) {
    const int s = e / cpg, cc = e - s * cpg;
    const float* pp = partial + (((int64_t)n * S + s) * C + c_lo + cc) * 2;
    t1 += (double)pp[0];
    t2 += (double)pp[1];
  }
  t1 = wave_sum_f64(t1);
  t2 = wave_sum_f64(t2);
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = t1; wsum[threadIdx.x >> 6][1] = t2; }
  __syncthreads();
  t1 = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
  t2 = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
  const double cnt = (double)HW;
  const double mean = t1 / (cnt * cpg);
  const double m2 = t2 - 2.0 * mean * t1 + cnt * (cpg * mean * mean);
  const float meanf = (float)mean;
  const float var = (float)(m2 / (cnt * cpg));
  const float rstd = 1.0f / sqrtf(snap_relu(var) + eps);
  for (int cc = threadIdx.x; cc < cpg; cc += 256) {
    mu[(int64_t)n * C + c_lo + cc] = meanf;
    sc[(int64_t)n * C + c_lo + cc] = rstd * (cc < 256 ? gam : gamma[c_lo + cc]);
    if (rstd_out) rstd_out[(int64_t)n * C + c_lo + cc] = rstd;
  }
}

__global__ void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total4,
                                int HW, int C, const float* __restrict__ mu,
                                const float* __restrict__ sc, const float* __restrict__ beta,
                                int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2;
  const int q = (int)(i % C4);
  const int64_t pix = i / C4;
  const int n = (int)(pix / HW);
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 m = *reinterpret_cast<const f32x4*>(mu + (int64_t)n * C + 4 * q);
  const f32x4 s = *reinterpret_cast<const f32x4*>(sc + (int64_t)n * C + 4 * q);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * q);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (mode == SNAP_PRO_GN_RELU)
      o[e] = snap_relu((v[e] - m[e]) * s[e] + b[e]);
    else
      o[e] = (snap_relu(v[e]) - m[e]) * s[e] + b[e];
  }
  reinterpret_cast<f32x4*>(y)[i] = o;
}

__global__ void max_pool_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                int W, int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % C4);
  int64_t r = i / C4;
  const int wo = (int)(r % Wo); r /= Wo;
  const int ho = (int)(r % Ho);
  const int n = (int)(r / Ho);
  f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dh = 0; dh < 3; ++dh) {
    const int hi = ho * 2 - 1 + dh;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int wi = wo * 2 - 1 + dw;
      if (wi < 0 || wi >= W) continue;
      const f32x4 v =
          *reinterpret_cast<const f32x4*>(x + (((int64_t)n * H + hi) * W + wi) * C + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) best[e] = snap_max_nan(best[e], v[e]);
    }
  }
  reinterpret_cast<f32x4*>(y)[i] = best;
}

// slabs per image: enough workgroups (>= ~1024) to fill the chip, >= 8 pixels each.
struct GnPlan { int S, ppb; };
inline GnPlan gn_plan(int N, int HW, int C) {
  const int chunks = (C + 1023) / 1024;
  int S = (1024 + N * chunks - 1) / (N * chunks);
  const int smax = (HW + 7) / 8;
  if (S > smax) S = smax;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  GnPlan p;
  p.ppb = (HW + S - 1) / S;
  p.S = (HW + p.ppb - 1) / p.ppb;
  return p;
}

}  // namespace

extern "C" int snap_weight_standardize_f32(const float* w, float* out, int32_t K, int32_t Cout,
                                           float eps, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  if (K <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weight_std_kernel, dim3((unsigned)snap_cdiv(Cout, WS_COLS)), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), w, out, K, Cout, eps);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_weight_standardize_multi_f32(const SnapWstdItem* items, int32_t n_items,
                                                 int32_t total_blocks, float eps,
                                                 void* stream) {
  if (!items) return SNAP_ERR_NULL;
  if (n_items <= 0 || total_blocks <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weight_std_multi_kernel, dim3((unsigned)total_blocks), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), items, n_items, eps);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_group_norm_stats_workspace_bytes(int32_t N, int32_t HW, int32_t C,
                                                        int32_t groups) {
  (void)groups;
  const GnPlan pl = gn_plan(N, HW, C);
  return (size_t)N * pl.S * C * 2 * sizeof(float);
}

extern "C" int snap_group_norm_stats_f32(const float* x, int32_t N, int32_t HW, int32_t C,
                                         int32_t C_stride, int32_t groups, float eps,
                                         int32_t relu_first, const float* gamma, float* mu,
                                         float* sc, float* rstd, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!x || !gamma || !mu || !sc || !workspace) return SNAP_ERR_NULL;
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0) return SNAP_ERR_BAD_SHAPE;
  if (C % 4 != 0 || C_stride % 4 != 0 || C_stride < C) return SNAP_ERR_BAD_SHAPE;
  const int nchunks = (C + 1023) / 1024;
  if (nchunks > 1 && C % 1024 != 0) return SNAP_ERR_BAD_SHAPE;
  const int cchunk = nchunks > 1 ? 1024 : C;
  const int QW = cchunk / 4;
  if (256 % QW != 0) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_group_norm_stats_workspace_bytes(N, HW, C, groups))
    return SNAP_ERR_WORKSPACE;
  const GnPlan pl = gn_plan(N, HW, C);
  float* partial = static_cast<float*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(pl.S, N, nchunks), dim3(256), 0, s, x, HW, C,
                     C_stride, relu_first, pl.ppb, partial);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)snap_cdiv(N * groups, 4)), dim3(256), 0, s,
                     x, (const float*)partial, pl.S, HW, C, C_stride, groups, relu_first, eps,
                     gamma, mu, sc, rstd, N * groups, 0);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_group_norm_stats_from_partial_f32(const float* partial, int32_t N, int32_t HW,
                                                      int32_t C, int32_t groups, float eps,
                                                      int32_t tile_rows, const float* gamma,
                                                      float* mu, float* sc, float* rstd,
                                                      void* stream) {
  if (!partial || !gamma || !mu || !sc) return SNAP_ERR_NULL;
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0) return SNAP_ERR_BAD_SHAPE;
  if (tile_rows == 0 || HW < tile_rows) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(gn_finalize_tiled_kernel, dim3((unsigned)(N * groups)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), partial, tile_rows < 0 ? -tile_rows : HW / tile_rows + 2,
                     HW, C, groups, eps, gamma, mu, sc, rstd, tile_rows);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_group_norm_apply_f32(const float* x, float* y, int32_t N, int32_t HW,
                                         int32_t C, const float* mu, const float* sc,
                                         const float* beta, int32_t mode, void* stream) {
  if (!x || !y || !mu || !sc || !beta) return SNAP_ERR_NULL;
  if (C % 4 != 0 || N <= 0 || HW <= 0) return SNAP_ERR_BAD_SHAPE;
  if (mode != SNAP_PRO_GN_RELU && mode != SNAP_PRO_RELU_GN) return SNAP_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)N * HW * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, total4, HW, C, mu, sc, beta, mode);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_max_pool_3x3s2_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W,
                                       int32_t C, void* stream) {
  if (!x || !y) return SNAP_ERR_NULL;
  if (C % 4 != 0 || N <= 0 || H <= 0 || W <= 0) return SNAP_ERR_BAD_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(max_pool_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, N, H, W, C, Ho, Wo);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// pad_to_multiple (image_encoder.py:32-39) + optional zero channels in ONE pass: x [N, H, W, C] ->
// y [N, H + ph, W + pw, C + cp], zeros at the bottom / right / in the extra channels (what
// jnp.pad does; torch's F.pad takes a fill launch and a strided copy for it).
namespace {
template <int CO>
__global__ __launch_bounds__(256) void pad_image_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int64_t total, int H, int W, int C, int Ho,
                                                        int Wo, int Co) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // output pixel
  if (i >= total) return;
  const int wo = (int)(i % Wo);
  const int64_t t = i / Wo;
  const int ho = (int)(t % Ho);
  const int64_t n = t / Ho;
  const bool in = ho < H && wo < W;
  const float* src = x + ((n * H + ho) * W + wo) * C;
  if constexpr (CO == 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) v[c] = src[c];
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  } else {
    float* dst = y + i * Co;
    for (int c = 0; c < Co; ++c) dst[c] = (in && c < C) ? src[c] : 0.f;
  }
}
}  // namespace

extern "C" int snap_pad_image_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                  int32_t pad_h, int32_t pad_w, int32_t pad_c, float* y, void* stream) {
  if (!x || !y) return SNAP_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad_h < 0 || pad_w < 0 || pad_c < 0) return SNAP_ERR_BAD_SHAPE;
  const int Ho = H + pad_h, Wo = W + pad_w, Co = C + pad_c;
  const int64_t total = (int64_t)N * Ho * Wo;
  if (total * Co > 0x7fffffff0LL) return SNAP_ERR_BAD_SHAPE;
  const dim3 grid((unsigned)snap_cdiv(total, 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (Co == 4 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
    hipLaunchKernelGGL(pad_image_kernel<4>, grid, dim3(256), 0, s, x, y, total, H, W, C, Ho, Wo, Co);
  else
    hipLaunchKernelGGL(pad_image_kernel<0>, grid, dim3(256), 0, s, x, y, total, H, W, C, Ho, Wo, Co);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

