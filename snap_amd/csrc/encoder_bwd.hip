// Backward kernels of the encoder building blocks (training path).
//   * GroupNorm(+ReLU) backward, both orders (VJP of resnet.py:46-70 fused with the
//     ReLU of resnet.py:118,125,130 / image_encoder.py:80-83)
//   * StdConv weight-standardisation backward (VJP of resnet.py:34-41,73-79)
//   * 3x3/2 max-pool backward (VJP of resnet.py:99)
//   * bilinear x2 up-sample backward (VJP of image_encoder.py:90)
//   * epilogue backward (ReLU / row-mask gating of dY) and column sums (bias grads)
// All reductions are fixed-order (deterministic).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// GroupNorm backward.  z = f(x):  GN_RELU: z = relu(xh*g + b);  RELU_GN: z = xh'*g + b
// with xh = (x - mu)*rstd (xh' uses relu(x)).  Pass 1: per (n, slab, c) sums of
//   A = sum dyp, Bs = sum dyp*xh   (dyp = dz gated by the ReLU for GN_RELU).
// Finalize: per (n, c) totals -> per (n, g) S1 = sum_c g_c A, S2 = sum_c g_c Bs and the
// parameter gradients d gamma_c = sum_n Bs, d beta_c = sum_n A.
// Pass 2: dx = rstd * (dyp*g - S1/cnt - xh*S2/cnt)  [* (x>0) for RELU_GN]  (+ add).
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, int HW, int C,
    const float* __restrict__ mu, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int ppb, float* __restrict__ partial /*[N,S,C,2]*/) {
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
  const f32x4 m4 = *reinterpret_cast<const f32x4*>(mu + (int64_t)n * C + c0);
  const f32x4 r4 = *reinterpret_cast<const f32x4*>(rstd + (int64_t)n * C + c0);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c0);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + c0);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t base = ((int64_t)n * HW) * C + c0;
  for (int p = p_begin + tp; p < p_end; p += PW) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (int64_t)p * C);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(dz + base + (int64_t)p * C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh, dyp;
      if (MODE == SNAP_PRO_GN_RELU) {
        xh = (xv[e] - m4[e]) * r4[e];
        dyp = (xh * g4[e] + b4[e] > 0.f) ? gv[e] : 0.f;
      } else {
        xh = (fmaxf(xv[e], 0.f) - m4[e]) * r4[e];
        dyp = gv[e];
      }
      s1[e] += dyp;
      s2[e] += dyp * xh;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    part[threadIdx.x * 8 + e] = s1[e];
    part[threadIdx.x * 8 + 4 + e] = s2[e];
  }
  __syncthreads();
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

// tile_rows > 0: `partial` comes from a convolution's epilogue ([N][HW / tile_rows + 2][C][2], slot = row tile
// within the image): only the slots of the row tiles that touch image n are written (as gn_finalize_tiled_kernel).
__device__ __forceinline__ int gn_live_slots(int n, int S, int HW, int tile_rows) {
  return tile_rows <= 0 ? S
                        : (int)((((int64_t)(n + 1) * HW - 1) / tile_rows) - (((int64_t)n * HW) / tile_rows)) + 1;
}

// thread per (n, c): slab totals -> AB[n,c,2]
__global__ void gn_bwd_reduce_kernel(const float* __restrict__ partial, int S, int C, int total,
                                     float* __restrict__ ab, int HW = 0, int tile_rows = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over N*C
  if (i >= total) return;
  const int n = i / C, c = i - n * C;
  float a1 = 0.f, a2 = 0.f;
  const int live = gn_live_slots(n, S, HW, tile_rows);
#pragma unroll 8
  for (int s = 0; s < live; ++s) {   // unrolled: 8 independent loads in flight (pure latency)
    const float2 p = *reinterpret_cast<const float2*>(partial + (((int64_t)n * S + s) * C + c) * 2);
    a1 += p.x;
    a2 += p.y;
  }
  ab[(int64_t)i * 2 + 0] = a1;
  ab[(int64_t)i * 2 + 1] = a2;
}

// thread per (n, g): group sums;  thread per c (second launch dimension): param grads
__global__ void gn_bwd_group_kernel(const float* __restrict__ ab, const float* __restrict__ gamma,
                                    int N, int C, int groups, float* __restrict__ s12 /*[N,G,2]*/,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cpg = C / groups;
  if (i < N * groups) {
    const int n = i / groups, g = i - n * groups;
    float t1 = 0.f, t2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      t1 += gamma[c] * ab[((int64_t)n * C + c) * 2 + 0];
      t2 += gamma[c] * ab[((int64_t)n * C + c) * 2 + 1];
    }
    s12[(int64_t)i * 2 + 0] = t1;
    s12[(int64_t)i * 2 + 1] = t2;
  }
  if (i < C) {
    float da = 0.f, db = 0.f;
    for (int n = 0; n < N; ++n) {
      da += ab[((int64_t)n * C + i) * 2 + 1];
      db += ab[((int64_t)n * C + i) * 2 + 0];
    }
    dgamma[i] = accumulate ? dgamma[i] + da : da;
    dbeta[i] = accumulate ? dbeta[i] + db : db;
  }
}

// HALF: 0 = f32 output only; 1 / 2 = also the same values rounded (RNE) to bf16 / IEEE half into
// dx_half -- the operand image of the producing layer's backward GEMMs
// gn_bwd_reduce_kernel + gn_bwd_group_kernel in ONE launch (two dependent ~8 us launches per
// GroupNorm layer, ~100 layers per training step): a workgroup owns CB channels (whole groups) of
// every image -- slab totals of (n, c) into LDS in the reduce kernel's order (slabs ascending), then
// the group sums (channels ascending) and the parameter gradients (images ascending) from LDS.
// Same sums, same orders, same bits as the two kernels.
template <int CB>
__global__ __launch_bounds__(256) void gn_bwd_reduce_group_kernel(
    const float* __restrict__ partial, int S, int N, int C, int groups, const float* __restrict__ gamma,
    float* __restrict__ s12, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    int HW = 0, int tile_rows = 0) {
  extern __shared__ float ab[];                       // [N][CB][2]
  const int c0 = blockIdx.x * CB;
  const int cpg = C / groups;
  for (int i = threadIdx.x; i < N * CB; i += 256) {
    const int n = i / CB, cl = i - n * CB;
    const int c = c0 + cl;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
      const int live = gn_live_slots(n, S, HW, tile_rows);
#pragma unroll 8
      for (int sl = 0; sl < live; ++sl) {
        const float2 p = *reinterpret_cast<const float2*>(partial + (((int64_t)n * S + sl) * C + c) * 2);
        a1 += p.x;
        a2 += p.y;
      }
    }
    ab[i * 2 + 0] = a1;
    ab[i * 2 + 1] = a2;
  }
  __syncthreads();
  const int gpb = CB / cpg;                           // groups per block (CB % cpg == 0)
  for (int i = threadIdx.x; i < N * gpb; i += 256) {
    const int n = i / gpb, gl = i - n * gpb;
    const int g = c0 / cpg + gl;
    if (g >= groups) continue;
    float t1 = 0.f, t2 = 0.f;
    for (int cl = gl * cpg; cl < (gl + 1) * cpg; ++cl) {
      const float gm = gamma[c0 + cl];
      t1 += gm * ab[(n * CB + cl) * 2 + 0];
      t2 += gm * ab[(n * CB + cl) * 2 + 1];
    }
    s12[((int64_t)n * groups + g) * 2 + 0] = t1;
    s12[((int64_t)n * groups + g) * 2 + 1] = t2;
  }
  for (int cl = threadIdx.x; cl < CB; cl += 256) {
    const int c = c0 + cl;
    if (c >= C) continue;
    float da = 0.f, db = 0.f;
    for (int n = 0; n < N; ++n) {
      da += ab[(n * CB + cl) * 2 + 1];
      db += ab[(n * CB + cl) * 2 + 0];
    }
    dgamma[c] = accumulate ? dgamma[c] + da : da;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
  }
}

template <int MODE, int HALF = 0>
__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                    const float* __restrict__ add, float* __restrict__ dx,
                                    int64_t total4, int HW, int C, int groups,
                                    const float* __restrict__ mu, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ s12, void* __restrict__ dx_half = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2;
  const int q = (int)(i % C4);
  const int n = (int)((i / C4) / HW);
  const int cpg = C / groups;
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 gv = reinterpret_cast<const f32x4*>(dz)[i];
  const f32x4 m4 = *reinterpret_cast<const f32x4*>(mu + (int64_t)n * C + 4 * q);
  const f32x4 r4 = *reinterpret_cast<const f32x4*>(rstd + (int64_t)n * C + 4 * q);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * q);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + 4 * q);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int g = (4 * q + e) / cpg;
    const float S1 = s12[((int64_t)n * groups + g) * 2 + 0];
    const float S2 = s12[((int64_t)n * groups + g) * 2 + 1];
    float xh, dyp;
    if (MODE == SNAP_PRO_GN_RELU) {
      xh = (xv[e] - m4[e]) * r4[e];
      dyp = (xh * g4[e] + b4[e] > 0.f) ? gv[e] : 0.f;
    } else {
      xh = (fmaxf(xv[e], 0.f) - m4[e]) * r4[e];
      dyp = gv[e];
    }
    float d = r4[e] * (dyp * g4[e] - S1 * inv_cnt - xh * (S2 * inv_cnt));
    if (MODE == SNAP_PRO_RELU_GN) d = xv[e] > 0.f ? d : 0.f;
    o[e] = d;
  }
  if (add) {
    const f32x4 av = reinterpret_cast<const f32x4*>(add)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += av[e];
  }
  reinterpret_cast<f32x4*>(dx)[i] = o;
  if constexpr (HALF == 1) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<bf16x4*>(dx_half)[i] = __builtin_convertvector(o, bf16x4);
  } else if constexpr (HALF == 2) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<f16x4*>(dx_half)[i] = __builtin_convertvector(o, f16x4);
  }
}

// ---------------------------------------------------------------------------
// StdConv backward: ws = (w - mean)/sigma per column;  dw = (dws - mean(dws) - ws*mean(dws*ws))/sigma
// ---------------------------------------------------------------------------
// Geometry of the forward kernel (encoder_ops.hip): 32 columns x 32 k-slices on 1024 threads -- a wave reads two
// 128-byte row runs per load -- and the forward's statistics (pivoted sums, combined in double: the same mean and
// sigma bits).  Three passes over the column block: statistics, the two gradient means, the output.
constexpr int WSB_COLS = 32, WSB_SLICES = 32;
__device__ __forceinline__ void weight_std_bwd_body(const float* __restrict__ w,
                                                    const float* __restrict__ dws,
                                                    float* __restrict__ dw, int K, int Cout,
                                                    float eps, int blk) {
  __shared__ float red[2][WSB_SLICES][WSB_COLS + 1];
  __shared__ float stat[4][WSB_COLS];
  const int tc = threadIdx.x % WSB_COLS, tk = threadIdx.x / WSB_COLS;
  const int col = blk * WSB_COLS + tc;
  const bool ok = col < Cout;
  const float piv = ok ? w[col] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int k = tk; k < K; k += WSB_SLICES) {
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
    for (int i = 0; i < WSB_SLICES; ++i) {
      t1 += (double)red[0][i][tc];
      t2 += (double)red[1][i][tc];
    }
    const double m = t1 / (double)K;
    const double var = t2 / (double)K - m * m;
    stat[0][tc] = (float)((double)piv + m);
    stat[1][tc] = sqrtf((float)fmax(var, 0.0) + eps);
  }
  __syncthreads();
  const float mean = stat[0][tc];
  const float sigma = stat[1][tc];
  float g1 = 0.f, g2 = 0.f;
  if (ok)
    for (int k = tk; k < K; k += WSB_SLICES) {
      const int64_t o = (int64_t)k * Cout + col;
      const float ws = (w[o] - mean) / sigma;
      const float g = dws[o];
      g1 += g;
      g2 += g * ws;
    }
  red[0][tk][tc] = g1;
  red[1][tk][tc] = g2;
  __syncthreads();
  if (tk == 0) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < WSB_SLICES; ++i) {
      t1 += red[0][i][tc];
      t2 += red[1][i][tc];
    }
    stat[2][tc] = t1 * (1.0f / (float)K);
    stat[3][tc] = t2 * (1.0f / (float)K);
  }
  __syncthreads();
  const float mg = stat[2][tc], mgw = stat[3][tc];
  // ws = (w - mean) / sigma;  dw = (dws - mean(dws) - ws * mean(dws * ws)) / sigma   (sigma^2 = var + eps)
  if (ok)
    for (int k = tk; k < K; k += WSB_SLICES) {
      const int64_t o = (int64_t)k * Cout + col;
      const float ws = (w[o] - mean) / sigma;
      dw[o] = (dws[o] - mg - ws * mgw) / sigma;
    }
}

__global__ __launch_bounds__(1024) void weight_std_bwd_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ dws,
                                                             float* __restrict__ dw, int K,
                                                             int Cout, float eps) {
  weight_std_bwd_body(w, dws, dw, K, Cout, eps, blockIdx.x);
}

__global__ __launch_bounds__(1024) void weight_std_bwd_multi_kernel(
    const SnapWstdItem* __restrict__ items, int n_items, float eps) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SnapWstdItem it = items[lo];
  weight_std_bwd_body(it.w, it.dws, it.out, it.K, it.Cout, eps, blockIdx.x - it.block_begin);
}

// ---------------------------------------------------------------------------
// max-pool 3x3/2 pad 1 backward (gather form; first maximum in window scan order).
// ---------------------------------------------------------------------------
// One thread = one input pixel x four channels (float4 loads: the 36 window reads per pixel come
// from L1 / L2 as 16-byte requests instead of 4-byte ones; 1.9 -> ~0.6 ms on the C3 root output).
template <int VEC>
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int H, int W,
                                                           int C, int Ho, int Wo) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int CV = C / VEC;
  const int64_t total = (int64_t)N * H * W * CV;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % CV);
  int64_t r = i / CV;
  const int wi = (int)(r % W); r /= W;
  const int hi = (int)(r % H);
  const int n = (int)(r / H);
  const float* const xn = x + (int64_t)n * H * W * C + VEC * cq;
  const float* const dyn = dy + (int64_t)n * Ho * Wo * C + VEC * cq;
  const vec_t xv = *reinterpret_cast<const vec_t*>(xn + ((int64_t)hi * W + wi) * C);
  vec_t g;
#pragma unroll
  for (int e = 0; e < VEC; ++e) g[e] = 0.f;
  // windows (ho, wo) with ho*2-1 <= hi <= ho*2+1
  for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
    if (ho >= Ho) continue;
    for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
      if (wo >= Wo) continue;
      // is (hi, wi) the first maximum of window (ho, wo)?  (per channel)
      bool first[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) first[e] = true;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int h2 = ho * 2 - 1 + dh;
        if (h2 < 0 || h2 >= H) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int w2 = wo * 2 - 1 + dw;
          if (w2 < 0 || w2 >= W) continue;
          const vec_t v = *reinterpret_cast<const vec_t*>(xn + ((int64_t)h2 * W + w2) * C);
          const bool before = (h2 < hi) || (h2 == hi && w2 < wi);
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (v[e] > xv[e] || (v[e] == xv[e] && before)) first[e] = false;
        }
      }
      const vec_t d = *reinterpret_cast<const vec_t*>(dyn + ((int64_t)ho * Wo + wo) * C);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (first[e]) g[e] += d[e];
    }
  }
  *reinterpret_cast<vec_t*>(dx + ((int64_t)n * H * W + (int64_t)hi * W + wi) * C + VEC * cq) = g;
}

// The same VJP, tiled (C % 4 == 0): a workgroup owns 16 x 16 input pixels x 16 channel quads.  Phase 1 reads every
// window that touches them ONCE (9 x 9 windows: nine float4 per window and quad instead of 36 per input pixel) and
// keeps, per (window, channel), the 9-bit set of positions that receive the window's gradient -- the first maximum
// in scan order, plus every NaN position (exactly the set the per-pixel test above selects: a NaN compares neither
// greater nor equal).  Phase 2: an input pixel adds dy of its (at most four) windows, in the same order as above:
// bit-identical.  0.89 -> ~0.3 ms on the C3 root output (20 x 272 x 272 x 64).
__global__ __launch_bounds__(256) void max_pool_bwd_tiled_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ dx, int N, int H, int W,
                                                                 int C, int Ho, int Wo, int tiles_x) {
  __shared__ uint16_t sel[81 * 16 * 4];          // [window 9 x 9][quad][channel]: position set
  const int n = blockIdx.z;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int q0 = blockIdx.y * 16;                // first channel quad of this workgroup
  const int CQ = C >> 2;
  const int ho0 = ty * 8, wo0 = tx * 8;          // first window; owned input pixels: rows 2 ho0 .., cols 2 wo0 ..
  const float* const xn = x + (int64_t)n * H * W * C;
  const float* const dyn = dy + (int64_t)n * Ho * Wo * C;
  for (int it = threadIdx.x; it < 81 * 16; it += 256) {
    const int q = it & 15, w = it >> 4;
    const int wy = w / 9, wx = w - wy * 9;
    const int ho = ho0 + wy, wo = wo0 + wx;
    uint16_t m[4] = {0, 0, 0, 0};
    if (ho < Ho && wo < Wo && q0 + q < CQ) {
      float best[4] = {0.f, 0.f, 0.f, 0.f};
      int bi[4] = {-1, -1, -1, -1};
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int h2 = ho * 2 - 1 + dh;
        if (h2 < 0 || h2 >= H) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int w2 = wo * 2 - 1 + dw;
          if (w2 < 0 || w2 >= W) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xn + ((int64_t)h2 * W + w2) * C + 4 * (q0 + q));
          const int p = dh * 3 + dw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (v[e] != v[e]) m[e] |= (uint16_t)(1u << p);
            else if (bi[e] < 0 || v[e] > best[e]) { best[e] = v[e]; bi[e] = p; }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (bi[e] >= 0) m[e] |= (uint16_t)(1u << bi[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sel[it * 4 + e] = m[e];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 256 * 16; it += 256) {
    const int q = it & 15, px = it >> 4;
    const int py = px >> 4, pxx = px & 15;
    const int hi = 2 * ho0 + py, wi = 2 * wo0 + pxx;
    if (hi >= H || wi >= W || q0 + q >= CQ) continue;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const int p = (hi - (ho * 2 - 1)) * 3 + (wi - (wo * 2 - 1));
        const uint16_t* const mm = sel + (((ho - ho0) * 9 + (wo - wo0)) * 16 + q) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dyn + ((int64_t)ho * Wo + wo) * C + 4 * (q0 + q));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((mm[e] >> p) & 1) g[e] += d[e];
      }
    }
    *reinterpret_cast<f32x4*>(dx + ((int64_t)n * H * W + (int64_t)hi * W + wi) * C + 4 * (q0 + q)) = g;
  }
}

// ---------------------------------------------------------------------------
// bilinear x2 up-sample backward: dprev[n,hp,wp,:] = sum over fine pixels whose taps hit it.
// ---------------------------------------------------------------------------
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dprev,
                                      int N, int Hp, int Wp, int C) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * Hp * Wp * C4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % C4);
  int64_t r = i / C4;
  const int wp = (int)(r % Wp); r /= Wp;
  const int hp = (int)(r % Hp);
  const int n = (int)(r / Hp);
  const int Ho = 2 * Hp, Wo = 2 * Wp;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ho = 2 * hp - 2; ho <= 2 * hp + 2; ++ho) {
    if (ho < 0 || ho >= Ho) continue;
    const float sh = (ho + 0.5f) * 0.5f - 0.5f;
    const float fh = floorf(sh);
    const float wh1 = sh - fh, wh0 = 1.f - wh1;
    const int h0 = min(max((int)fh, 0), Hp - 1), h1 = min(max((int)fh + 1, 0), Hp - 1);
    const float wh = (h0 == hp ? wh0 : 0.f) + (h1 == hp ? wh1 : 0.f);
    if (wh == 0.f) continue;
    for (int wo = 2 * wp - 2; wo <= 2 * wp + 2; ++wo) {
      if (wo < 0 || wo >= Wo) continue;
      const float sw = (wo + 0.5f) * 0.5f - 0.5f;
      const float fw = floorf(sw);
      const float ww1 = sw - fw, ww0 = 1.f - ww1;
      const int w0 = min(max((int)fw, 0), Wp - 1), w1 = min(max((int)fw + 1, 0), Wp - 1);
      const float ww = (w0 == wp ? ww0 : 0.f) + (w1 == wp ? ww1 : 0.f);
      if (ww == 0.f) continue;
      const f32x4 g =
          *reinterpret_cast<const f32x4*>(dy + (((int64_t)n * Ho + ho) * Wo + wo) * C + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (wh * ww) * g[e];
    }
  }
  reinterpret_cast<f32x4*>(dprev)[i] = acc;
}

// dy_lin = dy * [y > 0 if relu] * rowmask  (float4)
__global__ void epilogue_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                    const uint8_t* __restrict__ row_mask, float* __restrict__ out,
                                    int64_t total4, int C4, int relu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
  if (row_mask && !row_mask[i / C4]) g = f32x4{0.f, 0.f, 0.f, 0.f};
  if (relu) {
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
  }
  reinterpret_cast<f32x4*>(out)[i] = g;
}

// The gate and the column sums of its OUTPUT in one pass (a Dense / conv layer with a bias behind a
// ReLU: the bias gradient is the column sum of the gated gradient -- a second full read of it
// otherwise).  One workgroup = a slab of rows x a chunk of <= 1024 columns; a thread = a column quad
// and every PW-th row of the slab; partial[s][c] as colsum_partial_kernel writes it (rows beyond
// *row_count are neither read nor written).
__global__ __launch_bounds__(256) void epilogue_bwd_colsum_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const uint8_t* __restrict__ row_mask,
    float* __restrict__ out, int64_t M, int C, int relu, int64_t rows_per_block,
    const int32_t* __restrict__ row_count, float* __restrict__ partial) {
  __shared__ float red[256 * 4];
  const int cbase = blockIdx.y * 1024;
  const int cchunk = min(C - cbase, 1024);
  const int QW = cchunk >> 2;                         // (C % 4 == 0; QW divides 256 or equals it)
  const int PW = 256 / QW;
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int c0 = cbase + 4 * tq;
  const int64_t Msum = row_count ? min((int64_t)*row_count, M) : M;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  // (with a row count only the listed rows exist: the compact buffers of the masked MLP hold
  //  nothing beyond them that anybody reads -- 40 % of the rows at C3)
  const int64_t r1 = min(Msum, r0 + rows_per_block);
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (tp < PW) {
    for (int64_t r = r0 + tp; r < r1; r += PW) {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * C + c0);
      if (row_mask && !row_mask[r]) g = f32x4{0.f, 0.f, 0.f, 0.f};
      if (relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * C + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(out + r * C + c0) = g;
      if (r < Msum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += g[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = t[e];
  __syncthreads();
  if (tp == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
      for (int pp = 0; pp < PW; ++pp) a += red[(pp * QW + tq) * 4 + e];
      partial[(int64_t)blockIdx.x * C + c0 + e] = a;
    }
  }
}

// The same gate + column sums on 2-byte tensors of the training engine's element type (ET = __bf16 /
// _Float16): dy, y and out are [rows, C] of ET -- the inter-layer gradient and hidden activation of
// the masked MLP.  The gate is exact on the rounded values (an element is kept or zeroed); the bias
// gradient sums the kept elements in f32, in the f32 kernel's order.
// WSUM: also partial2[s][c] = sum_r w(r) * out[r][c] with w(r) = wsrc[wrows[r] * wstride] (ReLU'd if
// wrelu): the kernel-gradient row of ONE extra input channel of the layer in front (the 257th channel
// of the fusion MLP: dW0[256, :] = x[:, 256]^T g) taken in this pass instead of a third 128-channel
// tile in the GEMM.
// TAIL (with WSUM, C == 256: a row is one wave): also the DATA gradient of that extra channel,
// dtail[wrows[r] * dstride .. +3] = (sum_c out[r, c] * round(wtail[c]), 0, 0, 0) -- the four columns of the
// layer-0 input gradient (row stride 260 = 257 padded to quads) that would otherwise cost the GEMM in front a
// third, almost empty, 128-column tile.
template <typename ET, bool WSUM = false, bool TAIL = false>
__global__ __launch_bounds__(256) void epilogue_bwd_colsum_half_kernel(
    const ET* __restrict__ dy, const ET* __restrict__ y, ET* __restrict__ out, int64_t M, int C, int relu,
    int64_t rows_per_block, const int32_t* __restrict__ row_count, float* __restrict__ partial,
    const float* __restrict__ wsrc = nullptr, const int32_t* __restrict__ wrows = nullptr, int64_t wstride = 0,
    int wrelu = 0, float* __restrict__ partial2 = nullptr, const float* __restrict__ wtail = nullptr,
    float* __restrict__ dtail = nullptr, int64_t dstride = 0) {
  typedef ET etx4 __attribute__((ext_vector_type(4)));
  __shared__ float red[256 * 4];
  const int cbase = blockIdx.y * 1024;
  const int cchunk = min(C - cbase, 1024);
  const int QW = cchunk >> 2;
  const int PW = 256 / QW;
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int c0 = cbase + 4 * tq;
  const int64_t Msum = row_count ? min((int64_t)*row_count, M) : M;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(Msum, r0 + rows_per_block);
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  float u[4] = {0.f, 0.f, 0.f, 0.f};
  float wt[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (TAIL) {
#pragma unroll
    for (int e = 0; e < 4; ++e) wt[e] = (float)(ET)wtail[c0 + e];      // (rounded like the GEMM's weight image)
  }
  // four rows per iteration, every load of the four issued before the first use (one row per iteration ran at the
  // latency of its two dependent loads: 1.65 TB/s); the sums are taken in row order as before: same bits
  auto rows4 = [&](int64_t r, int nr) {
    etx4 g[4], yv[4];
    int64_t wi[4] = {0, 0, 0, 0};
    float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t rr = r + (int64_t)k * PW;
      if (k < nr) {
        g[k] = *reinterpret_cast<const etx4*>(dy + rr * C + c0);
        if (relu) yv[k] = *reinterpret_cast<const etx4*>(y + rr * C + c0);
        if constexpr (WSUM) wi[k] = wrows ? (int64_t)wrows[rr] : rr;
      }
    }
    if constexpr (WSUM) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nr) w[k] = wsrc[wi[k] * wstride];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < nr) {
        const int64_t rr = r + (int64_t)k * PW;
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[k][e] = (float)yv[k][e] > 0.f ? g[k][e] : (ET)0.f;
        }
        *reinterpret_cast<etx4*>(out + rr * C + c0) = g[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += (float)g[k][e];
        if constexpr (WSUM) {
          float wv = w[k];
          if (wrelu) wv = snap_relu(wv);
          // (the engine would multiply the ROUNDED operands: round w to ET like the GEMM's loader does)
          wv = (float)(ET)wv;
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] += wv * (float)g[k][e];
        }
        if constexpr (TAIL) {
          float dp = ((float)g[k][0] * wt[0] + (float)g[k][1] * wt[1]) + ((float)g[k][2] * wt[2] + (float)g[k][3] * wt[3]);
          dp = wave_sum(dp);                                              // QW == 64: the row is this wave
          if (tq == 0) *reinterpret_cast<f32x4*>(dtail + wi[k] * dstride) = f32x4{dp, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  if (tp < PW) {
    int64_t r = r0 + tp;
    for (; r + 3 * PW < r1; r += 4 * PW) rows4(r, 4);
    if (r < r1) rows4(r, (int)((r1 - r + PW - 1) / PW));
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = t[e];
  __syncthreads();
  if (tp == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
      for (int pp = 0; pp < PW; ++pp) a += red[(pp * QW + tq) * 4 + e];
      partial[(int64_t)blockIdx.x * C + c0 + e] = a;
    }
  }
  if constexpr (WSUM) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = u[e];
    __syncthreads();
    if (tp == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f;
        for (int pp = 0; pp < PW; ++pp) a += red[(pp * QW + tq) * 4 + e];
        partial2[(int64_t)blockIdx.x * C + c0 + e] = a;
      }
    }
  }
}

// column sums: partial[s][c] over row slabs, then fixed-order reduce.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int64_t M,
                                                             int C, int64_t rows_per_block,
                                                             float* __restrict__ partial,
                                                             const int32_t* __restrict__ rows,
                                                             const int32_t* __restrict__ row_count) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t Meff = row_count ? min((int64_t)*row_count, M) : M;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(Meff, r0 + rows_per_block);
  float t = 0.f;
  if (rows) {
    for (int64_t r = r0; r < r1; ++r) t += a[(int64_t)rows[r] * C + c];
  } else {
    for (int64_t r = r0; r < r1; ++r) t += a[r * C + c];
  }
  partial[(int64_t)blockIdx.x * C + c] = t;
}
// The same partial sums with (nearly) every lane busy and several rows in flight (C % 4 == 0, C <= 1024): a thread
// owns one column quad and every PW-th row of the slab, four rows per iteration (independent float4 loads; the
// per-column loop above has ONE 4-byte load in flight per thread and half the workgroup idle at C = 128: it ran at
// the latency of 1920 dependent iterations, 294 us per GB).  The PW row phases are combined in a fixed order.
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(const float* __restrict__ a, int64_t M, int C,
                                                                int64_t rows_per_block, float* __restrict__ partial,
                                                                const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ row_count) {
  __shared__ float red[256 * 4];
  const int QW = C >> 2, PW = 256 / QW;            // (threads beyond QW * PW idle: C / 4 need not divide 256)
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int64_t Meff = row_count ? min((int64_t)*row_count, M) : M;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = tp < PW ? min(Meff, r0 + rows_per_block) : r0;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  int64_t r = r0 + tp;
  for (; r + 3 * PW < r1; r += 4 * PW) {
    int64_t i0 = r, i1 = r + PW, i2 = r + 2 * PW, i3 = r + 3 * PW;
    if (rows) { i0 = rows[i0]; i1 = rows[i1]; i2 = rows[i2]; i3 = rows[i3]; }
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(a + i0 * C + 4 * tq);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(a + i1 * C + 4 * tq);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(a + i2 * C + 4 * tq);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(a + i3 * C + 4 * tq);
    t += (v0 + v1) + (v2 + v3);
  }
  for (; r < r1; r += PW) {
    const int64_t i = rows ? (int64_t)rows[r] : r;
    t += *reinterpret_cast<const f32x4*>(a + i * C + 4 * tq);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = t[e];
  __syncthreads();
  if (tp == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
      for (int pp = 0; pp < PW; ++pp) s += red[(pp * QW + tq) * 4 + e];
      partial[(int64_t)blockIdx.x * C + 4 * tq + e] = s;
    }
  }
}
// 32 columns x 8 slab groups per workgroup; fixed summation order (deterministic).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial,
                                                            int S, int C, float* __restrict__ out,
                                                            int accumulate) {
  __shared__ float red[8][33];
  const int tc = threadIdx.x & 31, tg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tc;
  float t = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int s = tg; s < S; s += 8) t += partial[(int64_t)s * C + c];
  }
  red[tg][tc] = t;
  __syncthreads();
  if (tg == 0 && c < C) {
    float r = accumulate ? out[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += red[i][tc];
    out[c] = r;
  }
}

struct GnPlanB { int S, ppb; };
inline GnPlanB gn_plan_b(int N, int HW, int C) {
  const int chunks = (C + 1023) / 1024;
  int S = (1024 + N * chunks - 1) / (N * chunks);
  const int smax = (HW + 7) / 8;
  if (S > smax) S = smax;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  GnPlanB p;
  p.ppb = (HW + S - 1) / S;
  p.S = (HW + p.ppb - 1) / p.ppb;
  return p;
}
inline int colsum_slabs(int64_t M) {
  int64_t s = (M + 511) / 512;
  if (s > 2048) s = 2048;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

extern "C" size_t snap_group_norm_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C,
                                                      int32_t groups) {
  const GnPlanB pl = gn_plan_b(N, HW, C);
  return ((size_t)N * pl.S * C * 2 + (size_t)N * C * 2 + (size_t)N * groups * 2) * sizeof(float);
}

extern "C" int snap_group_norm_bwd_f32(const float* x, const float* dz, const float* add,
                                       float* dx, int32_t N, int32_t HW, int32_t C, int32_t groups,
                                       const float* mu, const float* rstd, const float* gamma,
                                       const float* beta, int32_t mode, float* dgamma,
                                       float* dbeta, int32_t accumulate, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return snap_group_norm_bwd_ex_f32(x, dz, add, dx, N, HW, C, groups, mu, rstd, gamma, beta, mode, dgamma,
                                    dbeta, accumulate, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int snap_group_norm_bwd_ex_f32(const float* x, const float* dz, const float* add,
                                          float* dx, int32_t N, int32_t HW, int32_t C, int32_t groups,
                                          const float* mu, const float* rstd, const float* gamma,
                                          const float* beta, int32_t mode, float* dgamma,
                                          float* dbeta, int32_t accumulate, void* workspace,
                                          size_t workspace_bytes, void* dx_half, int32_t half_kind,
                                          void* stream) {
  return snap_group_norm_bwd_stats_f32(x, dz, add, dx, N, HW, C, groups, mu, rstd, gamma, beta, mode, dgamma, dbeta,
                                       accumulate, workspace, workspace_bytes, dx_half, half_kind, nullptr, 0, stream);
}

extern "C" int snap_group_norm_bwd_stats_f32(const float* x, const float* dz, const float* add,
                                             float* dx, int32_t N, int32_t HW, int32_t C, int32_t groups,
                                             const float* mu, const float* rstd, const float* gamma,
                                             const float* beta, int32_t mode, float* dgamma,
                                             float* dbeta, int32_t accumulate, void* workspace,
                                             size_t workspace_bytes, void* dx_half, int32_t half_kind,
                                             const float* stats, int32_t tile_rows, void* stream) {
  if (stats && (tile_rows <= 0 || HW < tile_rows || (reinterpret_cast<uintptr_t>(stats) & 7))) return SNAP_ERR_BAD_SHAPE;
  if (half_kind < 0 || half_kind > 2 || (half_kind != 0) != (dx_half != nullptr)) return SNAP_ERR_BAD_SHAPE;
  if (dx_half && (reinterpret_cast<uintptr_t>(dx_half) & 7)) return SNAP_ERR_BAD_SHAPE;
  if (!x || !dz || !dx || !mu || !rstd || !gamma || !beta || !dgamma || !dbeta || !workspace)
    return SNAP_ERR_NULL;
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || C % 4 != 0)
    return SNAP_ERR_BAD_SHAPE;
  if (mode != SNAP_PRO_GN_RELU && mode != SNAP_PRO_RELU_GN) return SNAP_ERR_UNSUPPORTED;
  const int nchunks = (C + 1023) / 1024;
  if (nchunks > 1 && C % 1024 != 0) return SNAP_ERR_BAD_SHAPE;
  const int cchunk = nchunks > 1 ? 1024 : C;
  if (256 % (cchunk / 4) != 0) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_group_norm_bwd_workspace_bytes(N, HW, C, groups))
    return SNAP_ERR_WORKSPACE;
  const GnPlanB pl = gn_plan_b(N, HW, C);
  float* partial = static_cast<float*>(workspace);
  float* ab = partial + (size_t)N * pl.S * C * 2;
  float* s12 = ab + (size_t)N * C * 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(pl.S, N, nchunks);
  // the sums the reduce pass reads: this launch's own first pass, or the epilogue statistics of the convolution
  // that wrote dz (one slot per row tile of tile_rows pixels)
  const float* sums = partial;
  int S = pl.S, tr = 0;
  if (stats) {
    sums = stats;
    S = HW / tile_rows + 2;
    tr = tile_rows;
  } else {
    if (mode == SNAP_PRO_GN_RELU)
      hipLaunchKernelGGL(gn_bwd_partial_kernel<SNAP_PRO_GN_RELU>, grid, dim3(256), 0, s, x, dz, HW, C,
                         mu, rstd, gamma, beta, pl.ppb, partial);
    else
      hipLaunchKernelGGL(gn_bwd_partial_kernel<SNAP_PRO_RELU_GN>, grid, dim3(256), 0, s, x, dz, HW, C,
                         mu, rstd, gamma, beta, pl.ppb, partial);
    SNAP_CHECK_LAUNCH();
  }
  const int cpg = C / groups;
  const size_t lds16 = (size_t)N * 16 * 2 * sizeof(float), lds64 = (size_t)N * 64 * 2 * sizeof(float);
  if (cpg <= 16 && 16 % cpg == 0 && C % 16 == 0 && lds16 <= 64 * 1024) {
    // slab totals, group sums and parameter gradients in one launch (16 channels per workgroup)
    hipLaunchKernelGGL(gn_bwd_reduce_group_kernel<16>, dim3((unsigned)(C / 16)), dim3(256), lds16, s,
                       sums, S, N, C, groups, gamma, s12, dgamma, dbeta, accumulate, HW, tr);
  } else if (cpg <= 64 && 64 % cpg == 0 && C % 64 == 0 && lds64 <= 64 * 1024) {
    hipLaunchKernelGGL(gn_bwd_reduce_group_kernel<64>, dim3((unsigned)(C / 64)), dim3(256), lds64, s,
                       sums, S, N, C, groups, gamma, s12, dgamma, dbeta, accumulate, HW, tr);
  } else {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((unsigned)snap_cdiv((int64_t)N * C, 256)),
                       dim3(256), 0, s, sums, S, C, N * C, ab, HW, tr);
    SNAP_CHECK_LAUNCH();
    const int work = N * groups > C ? N * groups : C;
    hipLaunchKernelGGL(gn_bwd_group_kernel, dim3((unsigned)snap_cdiv(work, 256)), dim3(256), 0, s,
                       (const float*)ab, gamma, N, C, groups, s12, dgamma, dbeta, accumulate);
  }
  SNAP_CHECK_LAUNCH();
  const int64_t total4 = (int64_t)N * HW * (C / 4);
#define SNAP_GN_APPLY(MODE_, HALF_)                                                                   \
  hipLaunchKernelGGL((gn_bwd_apply_kernel<MODE_, HALF_>), dim3((unsigned)snap_cdiv(total4, 256)),       \
                     dim3(256), 0, s, x, dz, add, dx, total4, HW, C, groups, mu, rstd, gamma, beta,     \
                     (const float*)s12, dx_half)
  if (mode == SNAP_PRO_GN_RELU) {
    if (half_kind == 1) SNAP_GN_APPLY(SNAP_PRO_GN_RELU, 1);
    else if (half_kind == 2) SNAP_GN_APPLY(SNAP_PRO_GN_RELU, 2);
    else SNAP_GN_APPLY(SNAP_PRO_GN_RELU, 0);
  } else {
    if (half_kind == 1) SNAP_GN_APPLY(SNAP_PRO_RELU_GN, 1);
    else if (half_kind == 2) SNAP_GN_APPLY(SNAP_PRO_RELU_GN, 2);
    else SNAP_GN_APPLY(SNAP_PRO_RELU_GN, 0);
  }
#undef SNAP_GN_APPLY
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_weight_standardize_bwd_f32(const float* w, const float* dws, float* dw,
                                               int32_t K, int32_t Cout, float eps, void* stream) {
  if (!w || !dws || !dw) return SNAP_ERR_NULL;
  if (K <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weight_std_bwd_kernel, dim3((unsigned)snap_cdiv(Cout, WSB_COLS)), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), w, dws, dw, K, Cout, eps);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_weight_standardize_bwd_multi_f32(const SnapWstdItem* items, int32_t n_items,
                                                     int32_t total_blocks, float eps,
                                                     void* stream) {
  if (!items) return SNAP_ERR_NULL;
  if (n_items <= 0 || total_blocks <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weight_std_bwd_multi_kernel, dim3((unsigned)total_blocks), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), items, n_items, eps);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_max_pool_3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int32_t N,
                                           int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !dy || !dx) return SNAP_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return SNAP_ERR_BAD_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const bool v4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                                   reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  const int64_t total = (int64_t)N * H * W * (v4 ? C / 4 : C);
  if (v4 && N <= 65535 && snap_cdiv(C, 64) <= 65535) {
    const int tiles_y = (int)snap_cdiv(H, 16), tiles_x = (int)snap_cdiv(W, 16);
    hipLaunchKernelGGL(max_pool_bwd_tiled_kernel, dim3((unsigned)(tiles_y * tiles_x), (unsigned)snap_cdiv(C, 64), (unsigned)N),
                       dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, dx, N, H, W, C, Ho, Wo, tiles_x);
  } else if (v4)
    hipLaunchKernelGGL(max_pool_bwd_kernel<4>, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dy, dx, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(max_pool_bwd_kernel<1>, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dy, dx, N, H, W, C, Ho, Wo);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_upsample2x_bwd_f32(const float* dy, float* dprev, int32_t N, int32_t Hp,
                                       int32_t Wp, int32_t C, void* stream) {
  if (!dy || !dprev) return SNAP_ERR_NULL;
  if (N <= 0 || Hp <= 0 || Wp <= 0 || C <= 0 || C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)N * Hp * Wp * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dy, dprev, N, Hp, Wp, C);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_epilogue_bwd_f32(const float* dy, const float* y, const uint8_t* row_mask,
                                     float* out, int64_t M, int32_t C, int32_t relu, void* stream) {
  if (!dy || !out) return SNAP_ERR_NULL;
  if (relu && !y) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total4 = M * (C / 4);
  hipLaunchKernelGGL(epilogue_bwd_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dy, y, row_mask, out, total4, C / 4, relu);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_epilogue_bwd_colsum_f32(const float* dy, const float* y, const uint8_t* row_mask,
                                           float* out, int64_t M, int32_t C, int32_t relu,
                                           const int32_t* row_count, float* colsum, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!dy || !out || !colsum || !workspace) return SNAP_ERR_NULL;
  if (relu && !y) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int q = C < 1024 ? C / 4 : 256;
  if (256 % q != 0 || (C > 1024 && C % 1024 != 0)) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < snap_colsum_workspace_bytes(M, C)) return SNAP_ERR_WORKSPACE;
  const int S = colsum_slabs(M);
  const int64_t rpb = (M + S - 1) / S;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(epilogue_bwd_colsum_kernel, dim3(S, (unsigned)snap_cdiv(C, 1024)), dim3(256), 0, s, dy, y,
                     row_mask, out, M, C, relu, rpb, row_count, static_cast<float*>(workspace));
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)snap_cdiv(C, 32)), dim3(256), 0, s,
                     (const float*)workspace, S, C, colsum, 0);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_epilogue_bwd_colsum_half(const void* dy, const void* y, void* out, int64_t M,
                                             int32_t C, int32_t relu, const int32_t* row_count,
                                             float* colsum, void* workspace, size_t workspace_bytes,
                                             int32_t half_kind, void* stream) {
  return snap_epilogue_bwd_colsum_wsum_half(dy, y, out, M, C, relu, row_count, colsum, workspace,
                                            workspace_bytes, half_kind, nullptr, nullptr, 0, 0, nullptr, stream);
}

extern "C" int snap_epilogue_bwd_colsum_wsum_half(const void* dy, const void* y, void* out, int64_t M,
                                                  int32_t C, int32_t relu, const int32_t* row_count,
                                                  float* colsum, void* workspace, size_t workspace_bytes,
                                                  int32_t half_kind, const float* wsrc,
                                                  const int32_t* wrows, int64_t wstride, int32_t wrelu,
                                                  float* wsum, void* stream) {
  return snap_epilogue_bwd_colsum_wsum_tail_half(dy, y, out, M, C, relu, row_count, colsum, workspace, workspace_bytes,
                                                 half_kind, wsrc, wrows, wstride, wrelu, wsum, nullptr, nullptr, 0,
                                                 stream);
}

extern "C" int snap_epilogue_bwd_colsum_wsum_tail_half(const void* dy, const void* y, void* out, int64_t M,
                                                       int32_t C, int32_t relu, const int32_t* row_count,
                                                       float* colsum, void* workspace, size_t workspace_bytes,
                                                       int32_t half_kind, const float* wsrc,
                                                       const int32_t* wrows, int64_t wstride, int32_t wrelu,
                                                       float* wsum, const float* wtail, float* dtail,
                                                       int64_t dstride, void* stream) {
  if ((wsrc != nullptr) != (wsum != nullptr)) return SNAP_ERR_NULL;
  if ((wtail != nullptr) != (dtail != nullptr)) return SNAP_ERR_NULL;
  if (wtail && (!wsrc || C != 256 || dstride < 4 || dstride % 4 != 0 || (reinterpret_cast<uintptr_t>(dtail) & 15)))
    return SNAP_ERR_UNSUPPORTED;
  if (!dy || !out || !colsum || !workspace) return SNAP_ERR_NULL;
  if (relu && !y) return SNAP_ERR_NULL;
  if (half_kind != 1 && half_kind != 2) return SNAP_ERR_UNSUPPORTED;
  if (M <= 0 || C <= 0 || C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int q = C < 1024 ? C / 4 : 256;
  if (256 % q != 0 || (C > 1024 && C % 1024 != 0)) return SNAP_ERR_UNSUPPORTED;
  // (with the weighted sums the workspace holds two partial arrays)
  if (workspace_bytes < snap_colsum_workspace_bytes(M, C) * (wsrc ? 2 : 1)) return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(y)) & 7)
    return SNAP_ERR_BAD_SHAPE;
  const int S = colsum_slabs(M);
  const int64_t rpb = (M + S - 1) / S;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(S, (unsigned)snap_cdiv(C, 1024));
  float* p1 = static_cast<float*>(workspace);
  float* p2 = p1 + (size_t)S * C;
#define SNAP_GATE_LAUNCH(ET_, W_, T_)                                                                   \
  hipLaunchKernelGGL((epilogue_bwd_colsum_half_kernel<ET_, W_, T_>), grid, dim3(256), 0, s,              \
                     static_cast<const ET_*>(dy), static_cast<const ET_*>(y), static_cast<ET_*>(out), M, C, \
                     relu, rpb, row_count, p1, wsrc, wrows, wstride, wrelu, p2, wtail, dtail, dstride)
  if (half_kind == 2) {
    if (wtail) SNAP_GATE_LAUNCH(_Float16, true, true);
    else if (wsrc) SNAP_GATE_LAUNCH(_Float16, true, false);
    else SNAP_GATE_LAUNCH(_Float16, false, false);
  } else {
    if (wtail) SNAP_GATE_LAUNCH(__bf16, true, true);
    else if (wsrc) SNAP_GATE_LAUNCH(__bf16, true, false);
    else SNAP_GATE_LAUNCH(__bf16, false, false);
  }
#undef SNAP_GATE_LAUNCH
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)snap_cdiv(C, 32)), dim3(256), 0, s,
                     (const float*)p1, S, C, colsum, 0);
  SNAP_CHECK_LAUNCH();
  if (wsrc) {
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)snap_cdiv(C, 32)), dim3(256), 0, s,
                       (const float*)p2, S, C, wsum, 0);
    SNAP_CHECK_LAUNCH();
  }
  return SNAP_OK;
}

extern "C" size_t snap_colsum_workspace_bytes(int64_t M, int32_t C) {
  return (size_t)colsum_slabs(M) * C * sizeof(float);
}

extern "C" int snap_colsum_f32(const float* a, int64_t M, int32_t C, float* out, int32_t accumulate,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return snap_colsum_rows_f32(a, M, C, nullptr, nullptr, out, accumulate, workspace,
                              workspace_bytes, stream);
}

extern "C" int snap_colsum_rows_f32(const float* a, int64_t M, int32_t C, const int32_t* rows,
                                    const int32_t* row_count, float* out, int32_t accumulate,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !out || !workspace) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_colsum_workspace_bytes(M, C)) return SNAP_ERR_WORKSPACE;
  const int S = colsum_slabs(M);
  const int64_t rpb = (M + S - 1) / S;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (C % 4 == 0 && C <= 1024 && (reinterpret_cast<uintptr_t>(a) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(S), dim3(256), 0, s, a, M, C, rpb,
                       static_cast<float*>(workspace), rows, row_count);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(S, (unsigned)snap_cdiv(C, 256)), dim3(256), 0, s, a,
                       M, C, rpb, static_cast<float*>(workspace), rows, row_count);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)snap_cdiv(C, 32)), dim3(256), 0, s,
                     (const float*)workspace, S, C, out, accumulate);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
