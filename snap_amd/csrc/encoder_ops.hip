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
// block = 32 columns x 8 k-slices.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_std_kernel(const float* __restrict__ w,
                                                         float* __restrict__ out, int K,
                                                         int Cout, float eps) {
  __shared__ float red[8][33];
  __shared__ float stat[2][32];
  const int tc = threadIdx.x & 31, tk = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tc;
  const bool ok = col < Cout;
  float s = 0.f;
  if (ok)
    for (int k = tk; k < K; k += 8) s += w[(int64_t)k * Cout + col];
  red[tk][tc] = s;
  __syncthreads();
  if (tk == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][tc];
    stat[0][tc] = t / (float)K;
  }
  __syncthreads();
  const float mean = stat[0][tc];
  float q = 0.f;
  if (ok)
    for (int k = tk; k < K; k += 8) {
      const float dlt = w[(int64_t)k * Cout + col] - mean;
      q += dlt * dlt;
    }
  red[tk][tc] = q;
  __syncthreads();
  if (tk == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][tc];
    stat[1][tc] = sqrtf(t / (float)K + eps);
  }
  __syncthreads();
  const float denom = stat[1][tc];
  if (ok)
    for (int k = tk; k < K; k += 8) {
      const int64_t o = (int64_t)k * Cout + col;
      out[o] = (w[o] - mean) / denom;
    }
}

// ---------------------------------------------------------------------------
// GroupNorm statistics.
// grid = (slabs, N, channel chunks of <=1024); block = 256 threads laid out as
// QW float4-quads x PW pixels.  PASS 0 accumulates sum(v); PASS 1 accumulates
// sum((v-mean)^2) with mean from pass 0.  Deterministic: fixed-order LDS trees
// and a fixed-order slab reduction in the finalize kernels.
// ---------------------------------------------------------------------------
constexpr int GN_PIX_PER_BLOCK = 512;

template <int PASS>
__global__ __launch_bounds__(256) void gn_partial_kernel(
    const float* __restrict__ x, int HW, int C, int Cs, int groups, int relu_first,
    const float* __restrict__ mean /*[N,groups]*/, float* __restrict__ partial /*[N,S,groups]*/) {
  __shared__ float part[256 * 4];
  __shared__ float csum[1024];
  const int S = gridDim.x;
  const int n = blockIdx.y;
  const int cbase = blockIdx.z * 1024;
  const int cchunk = min(C - cbase, 1024);
  const int QW = cchunk >> 2;        // quads in this chunk (<=256), power of two or 16..256
  const int PW = 256 / QW;
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int cpg = C / groups;
  const int c0 = cbase + 4 * tq;
  const int p_begin = blockIdx.x * GN_PIX_PER_BLOCK;
  const int p_end = min(p_begin + GN_PIX_PER_BLOCK, HW);
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  if (PASS == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = mean[n * groups + (c0 + e) / cpg];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (tp < PW) {
    const float* xb = x + ((int64_t)n * HW) * Cs + c0;
    for (int p = p_begin + tp; p < p_end; p += PW) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xb + (int64_t)p * Cs);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = relu_first ? fmaxf(v[e], 0.f) : v[e];
        if (PASS == 1) { t -= m[e]; t *= t; }
        acc[e] += t;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  // per-channel sums over the PW pixel lanes (fixed order)
  for (int c = threadIdx.x; c < cchunk; c += 256) {
    const int q = c >> 2, e = c & 3;
    float t = 0.f;
    for (int pp = 0; pp < PW; ++pp) t += part[(pp * QW + q) * 4 + e];
    csum[c] = t;
  }
  __syncthreads();
  const int g0 = cbase / cpg;
  const int ng = cchunk / cpg;
  for (int g = threadIdx.x; g < ng; g += 256) {
    float t = 0.f;
    for (int c = 0; c < cpg; ++c) t += csum[g * cpg + c];
    partial[((int64_t)n * S + blockIdx.x) * groups + g0 + g] = t;
  }
}

__global__ void gn_finalize_mean_kernel(const float* __restrict__ partial, int S, int groups,
                                        float count, float* __restrict__ mean, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over N*groups
  if (i >= total) return;
  const int n = i / groups, g = i - n * groups;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += partial[((int64_t)n * S + s) * groups + g];
  mean[i] = t / count;
}

__global__ void gn_finalize_kernel(const float* __restrict__ partial, int S, int groups, int C,
                                   float count, float eps, const float* __restrict__ mean,
                                   const float* __restrict__ gamma, float* __restrict__ mu,
                                   float* __restrict__ sc, int total /*N*C*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over N*C
  if (i >= total) return;
  const int n = i / C, c = i - n * C;
  const int g = c / (C / groups);
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += partial[((int64_t)n * S + s) * groups + g];
  const float var = t / count;
  // x / sqrt(mean(x^2) + eps): division by the sqrt, as in resnet.py:40.
  const float rstd = 1.0f / sqrtf(var + eps);
  mu[i] = mean[n * groups + g];
  sc[i] = rstd * gamma[c];
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
      o[e] = fmaxf((v[e] - m[e]) * s[e] + b[e], 0.f);
    else
      o[e] = (fmaxf(v[e], 0.f) - m[e]) * s[e] + b[e];
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
      for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v[e]);
    }
  }
  reinterpret_cast<f32x4*>(y)[i] = best;
}

inline int gn_slabs(int HW) { return (HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK; }

}  // namespace

extern "C" int snap_weight_standardize_f32(const float* w, float* out, int32_t K, int32_t Cout,
                                           float eps, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  if (K <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weight_std_kernel, dim3((unsigned)snap_cdiv(Cout, 32)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, out, K, Cout, eps);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_group_norm_stats_workspace_bytes(int32_t N, int32_t HW, int32_t C,
                                                        int32_t groups) {
  (void)C;
  const size_t S = (size_t)gn_slabs(HW);
  return ((size_t)N * S * groups + (size_t)N * groups) * sizeof(float);
}

extern "C" int snap_group_norm_stats_f32(const float* x, int32_t N, int32_t HW, int32_t C,
                                         int32_t C_stride, int32_t groups, float eps,
                                         int32_t relu_first, const float* gamma, float* mu,
                                         float* sc, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (!x || !gamma || !mu || !sc || !workspace) return SNAP_ERR_NULL;
  if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0) return SNAP_ERR_BAD_SHAPE;
  if (C % 4 != 0 || C_stride % 4 != 0 || C_stride < C) return SNAP_ERR_BAD_SHAPE;
  // channel chunking: chunks of 1024 channels must hold whole groups and a
  // power-of-two number of quads dividing 256.
  const int cpg = C / groups;
  const int nchunks = (C + 1023) / 1024;
  if (nchunks > 1 && (C % 1024 != 0 || 1024 % cpg != 0)) return SNAP_ERR_BAD_SHAPE;
  const int cchunk = nchunks > 1 ? 1024 : C;
  const int QW = cchunk / 4;
  if (256 % QW != 0) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_group_norm_stats_workspace_bytes(N, HW, C, groups))
    return SNAP_ERR_WORKSPACE;
  const int S = gn_slabs(HW);
  float* partial = static_cast<float*>(workspace);
  float* mean = partial + (size_t)N * S * groups;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(S, N, nchunks);
  const float count = (float)HW * (float)cpg;
  hipLaunchKernelGGL(gn_partial_kernel<0>, grid, dim3(256), 0, s, x, HW, C, C_stride, groups,
                     relu_first, (const float*)nullptr, partial);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize_mean_kernel, dim3((unsigned)snap_cdiv(N * groups, 256)),
                     dim3(256), 0, s, partial, S, groups, count, mean, N * groups);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_partial_kernel<1>, grid, dim3(256), 0, s, x, HW, C, C_stride, groups,
                     relu_first, (const float*)mean, partial);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)snap_cdiv((int64_t)N * C, 256)),
                     dim3(256), 0, s, partial, S, groups, C, count, eps, (const float*)mean,
                     gamma, mu, sc, N * C);
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
