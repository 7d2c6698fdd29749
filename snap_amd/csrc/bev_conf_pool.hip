// Confidence-weighted vertical pooling ('weighted' / 'softmax' modes of VerticalPooling,
// snap/models/bev_mapper.py:63-78; non-default):
//   score_z  = f_z . w + b                 ('weighted': log_sigmoid of it)
//   weight_z = softmax over the VALID levels (all levels when none is valid; shift
//              max(0, max score) as jax.nn.softmax(where=, initial=0)), zero on invalid levels
//   plane    = sum_z weight_z f_z          (zero for columns without a valid level)
// One half-wave (32 lanes x float4 = D <= 128 channels) owns one BEV column and reads its
// Z x D voxel features ONCE (online softmax); the per-level scores live one per lane.
// The reference materialises scores, weights and the weighted volume as separate XLA ops.
#include "common.h"

namespace {

__device__ __forceinline__ float hsum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__device__ __forceinline__ float hmax(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 32));
  return v;
}
__device__ __forceinline__ float log_sigmoid(float x) {
  // jax.nn.log_sigmoid = -softplus(-x) = min(x, 0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

constexpr int CP_MAXZ = 64;   // levels per column (one or two per lane)

__global__ __launch_bounds__(256) void conf_pool_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, const float* __restrict__ w,
    const float* __restrict__ bias, int64_t M, int Z, int D, int log_sig,
    float* __restrict__ plane, uint8_t* __restrict__ pvalid, float* __restrict__ scores,
    float* __restrict__ weights) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  const int nq = D >> 2;
  const bool lane_on = hl < nq;
  const uint8_t* vv = vvalid + m * Z;
  const float* base = vol + m * Z * D;
  const f32x4 wv = lane_on ? *reinterpret_cast<const f32x4*>(w + 4 * hl) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float b0 = bias[0];
  int a = 0;
  for (int z = hl; z < Z; z += 32) a |= vv[z] != 0 ? 1 : 0;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a |= __shfl_xor(a, o, 32);
  const bool any = a != 0;   // some level of this column is valid (half-wave uniform)
  // per-lane copies of the scores of levels hl and hl + 32
  float sc0 = 0.f, sc1 = 0.f;
  float run_max = 0.f, run_sum = 0.f;   // shift starts at 0 (initial=0)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < Z; ++z) {
    const f32x4 f = lane_on ? *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * hl)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    float s = hsum(((f[0] * wv[0] + f[1] * wv[1]) + f[2] * wv[2]) + f[3] * wv[3]) + b0;
    if (log_sig) s = log_sigmoid(s);
    if ((z & 31) == hl) { if (z < 32) sc0 = s; else sc1 = s; }
    const bool use = any ? vv[z] != 0 : true;   // where = valid, or everything if none valid
    if (use) {
      const float nm = fmaxf(run_max, s);
      const float scale = expf(run_max - nm);
      const float e = expf(s - nm);
      run_sum = run_sum * scale + e;
      const float ev = (any && vv[z]) ? e : 0.f;    // invalid levels carry no feature mass
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = acc[k] * scale + ev * f[k];
      run_max = nm;
    }
  }
  // outputs
  if (lane_on) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = any ? acc[k] / run_sum : 0.f;
    *reinterpret_cast<f32x4*>(plane + m * D + 4 * hl) = o;
  }
  if (hl == 0) pvalid[m] = any ? 1 : 0;
  for (int z = hl; z < Z; z += 32) {
    const float s = z < 32 ? sc0 : sc1;
    scores[m * Z + z] = s;
    weights[m * Z + z] = (any && vv[z]) ? expf(s - run_max) / run_sum : 0.f;
  }
}

// backward: d plane -> d vol, per-workgroup partial (d w, d b)
__global__ __launch_bounds__(256) void conf_pool_bwd_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, const float* __restrict__ w,
    const float* __restrict__ weights,
    const float* __restrict__ dplane, int64_t M, int Z, int D, int log_sig,
    const float* __restrict__ bias, float* __restrict__ dvol, float* __restrict__ dw_partial) {
  __shared__ float red[8][132];
  const int hl = threadIdx.x & 31;
  const int hw = threadIdx.x >> 5;
  const int64_t m = (int64_t)blockIdx.x * 8 + hw;
  const int nq = D >> 2;
  const bool lane_on = hl < nq;
  f32x4 dwacc = {0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  if (m < M) {
    const float* base = vol + m * Z * D;
    float* dbase = dvol + m * Z * D;
    const f32x4 wv = lane_on ? *reinterpret_cast<const f32x4*>(w + 4 * hl) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 g = lane_on ? *reinterpret_cast<const f32x4*>(dplane + m * D + 4 * hl)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    const float b0 = bias[0];
    // pass 1: dp_z = g . f_z for the weighted levels, and sum_k p_k dp_k
    float pdp = 0.f;
    for (int z = 0; z < Z; ++z) {
      const float p = weights[m * Z + z];
      if (p == 0.f) continue;   // half-wave uniform (weights are per column)
      const f32x4 f = lane_on ? *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * hl)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
      const float dp = hsum(((g[0] * f[0] + g[1] * f[1]) + g[2] * f[2]) + g[3] * f[3]);
      pdp += p * dp;
    }
    // pass 2: d f_z = p_z g + d s_z w ;  d s_z = p_z (dp_z - pdp) [* sigmoid(-s_z)]
    for (int z = 0; z < Z; ++z) {
      const float p = weights[m * Z + z];
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (p != 0.f) {
        const f32x4 f = lane_on ? *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * hl)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
        const float dp = hsum(((g[0] * f[0] + g[1] * f[1]) + g[2] * f[2]) + g[3] * f[3]);
        float ds = p * (dp - pdp);
        if (log_sig) {
          const float s = hsum(((f[0] * wv[0] + f[1] * wv[1]) + f[2] * wv[2]) + f[3] * wv[3]) + b0;
          ds *= 1.f / (1.f + expf(s));   // d log_sigmoid(s) / ds = sigmoid(-s)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = p * g[k] + ds * wv[k];
          dwacc[k] += ds * f[k];
        }
        dbacc += ds;
      }
      if (lane_on) *reinterpret_cast<f32x4*>(dbase + (int64_t)z * D + 4 * hl) = o;
    }
  }
  // fixed-order reduction of the 8 columns of this workgroup -> one partial row [D + 1]
#pragma unroll
  for (int k = 0; k < 4; ++k) red[hw][4 * hl + k] = lane_on ? dwacc[k] : 0.f;
  if (hl == 0) red[hw][128] = dbacc;
  __syncthreads();
  for (int c = threadIdx.x; c < D + 4; c += 256) {
    float t = 0.f;
    if (c <= D) {
      const int src = c < D ? c : 128;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += red[i][src];
    }
    dw_partial[(int64_t)blockIdx.x * (D + 4) + c] = t;   // columns D+1.. are padding (zero)
  }
}

}  // namespace

extern "C" int snap_vertical_pool_conf_f32(const float* vol, const uint8_t* vvalid, const float* w,
                                           const float* bias, int64_t M, int32_t Z, int32_t D,
                                           int32_t log_sigmoid_scores, float* plane, uint8_t* pvalid,
                                           float* scores, float* weights, void* stream) {
  if (!vol || !vvalid || !w || !bias || !plane || !pvalid || !scores || !weights) return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || Z > CP_MAXZ || D <= 0 || D % 4 != 0 || D > 128) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(conf_pool_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), vol, vvalid, w, bias, M, Z, D,
                     log_sigmoid_scores, plane, pvalid, scores, weights);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_vertical_pool_conf_bwd_partial_rows(int64_t M) { return (size_t)snap_cdiv(M, 8); }

extern "C" int snap_vertical_pool_conf_bwd_f32(const float* vol, const uint8_t* vvalid, const float* w,
                                               const float* bias, const float* weights,
                                               const float* dplane, int64_t M, int32_t Z, int32_t D,
                                               int32_t log_sigmoid_scores, float* dvol,
                                               float* dw_partial, void* stream) {
  if (!vol || !vvalid || !w || !bias || !weights || !dplane || !dvol || !dw_partial)
    return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || Z > CP_MAXZ || D <= 0 || D % 4 != 0 || D > 128) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(conf_pool_bwd_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), vol, vvalid, w, weights,
                     dplane, M, Z, D, log_sigmoid_scores, bias, dvol, dw_partial);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
