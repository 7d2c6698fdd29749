// Semantic-raster embedding (snap/models/semantic_raster_encoder.py:63-79): the multi-channel
// boolean raster of a map tile becomes the input feature image of the semantic encoder.
//   surfel-road classes are mutually exclusive: label = argmax over their channels (first set
//   channel, 0 if none) -> one row of `table_road`;
//   every other class j is binary: row (j + bit_j) of `table_other` (the reference adds the bit
//   to arange(n), so classes j and j+1 share a row -- kept as is), features concatenated.
// out[m, :] = [ table_road[label] | table_other[0 + b_0] | table_other[1 + b_1] | ... ].
// The VJP w.r.t. the two tables goes through the deterministic wgrad engine on the one-hot
// matrix written by semantic_onehot_kernel (column c < nr: label == c; column nr + 2 j + b:
// class j has bit b).
#include "common.h"

namespace {

constexpr int SEM_MAX = 32;   // classes per group

struct SemArgs {
  const uint8_t* rasters;   // [M, N]
  int64_t M;
  int N, nr, no, E;
  int idx_road[SEM_MAX], idx_other[SEM_MAX];
};

__device__ __forceinline__ int road_label(const SemArgs& a, const uint8_t* r) {
  int label = 0;
  bool found = false;
  for (int c = 0; c < a.nr; ++c) {
    const bool on = r[a.idx_road[c]] != 0;
    if (on && !found) { label = c; found = true; }
  }
  return label;
}

__global__ __launch_bounds__(256) void semantic_embed_kernel(
    const SemArgs a, const float* __restrict__ table_road, const float* __restrict__ table_other,
    float* __restrict__ out) {
  const int EQ = a.E >> 2;                 // float4 per embedding
  const int groups = 1 + a.no;             // embeddings per pixel
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = a.M * groups * EQ;
  if (i >= total) return;
  const int q = (int)(i % EQ);
  const int g = (int)((i / EQ) % groups);
  const int64_t m = i / ((int64_t)EQ * groups);
  const uint8_t* r = a.rasters + m * a.N;
  const float* src;
  if (g == 0) {
    src = table_road + (int64_t)road_label(a, r) * a.E;
  } else {
    const int j = g - 1;
    src = table_other + (int64_t)(j + (r[a.idx_other[j]] != 0 ? 1 : 0)) * a.E;
  }
  *reinterpret_cast<f32x4*>(out + (m * groups + g) * a.E + 4 * q) =
      *reinterpret_cast<const f32x4*>(src + 4 * q);
}

__global__ __launch_bounds__(256) void semantic_onehot_kernel(const SemArgs a, float* __restrict__ onehot,
                                                              int KP) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.M) return;
  const uint8_t* r = a.rasters + m * a.N;
  float* o = onehot + m * KP;
  for (int c = 0; c < KP; ++c) o[c] = 0.f;
  if (a.nr > 0) o[road_label(a, r)] = 1.f;
  for (int j = 0; j < a.no; ++j) o[a.nr + 2 * j + (r[a.idx_other[j]] != 0 ? 1 : 0)] = 1.f;
}

int fill_args(SemArgs& a, const uint8_t* rasters, int64_t M, int32_t N, const int32_t* idx_road,
              int32_t nr, const int32_t* idx_other, int32_t no, int32_t E) {
  if (!rasters || (nr > 0 && !idx_road) || (no > 0 && !idx_other)) return SNAP_ERR_NULL;
  if (M <= 0 || N <= 0 || nr < 0 || no < 0 || nr + no == 0 || nr > SEM_MAX || no > SEM_MAX ||
      E <= 0 || E % 4 != 0)
    return SNAP_ERR_BAD_SHAPE;
  a.rasters = rasters; a.M = M; a.N = N; a.nr = nr; a.no = no; a.E = E;
  for (int i = 0; i < nr; ++i) {
    if (idx_road[i] < 0 || idx_road[i] >= N) return SNAP_ERR_BAD_SHAPE;
    a.idx_road[i] = idx_road[i];
  }
  for (int i = 0; i < no; ++i) {
    if (idx_other[i] < 0 || idx_other[i] >= N) return SNAP_ERR_BAD_SHAPE;
    a.idx_other[i] = idx_other[i];
  }
  return SNAP_OK;
}

}  // namespace

extern "C" int snap_semantic_embed_f32(const uint8_t* rasters, int64_t M, int32_t N,
                                       const int32_t* idx_road, int32_t nr,
                                       const int32_t* idx_other, int32_t no,
                                       const float* table_road, const float* table_other,
                                       int32_t E, float* out, void* stream) {
  SemArgs a;
  const int rc = fill_args(a, rasters, M, N, idx_road, nr, idx_other, no, E);
  if (rc != SNAP_OK) return rc;
  if (!table_road || !table_other || !out) return SNAP_ERR_NULL;
  if (nr == 0) return SNAP_ERR_UNSUPPORTED;   // (the reference always embeds the road label)
  const int64_t total = M * (1 + no) * (E / 4);
  hipLaunchKernelGGL(semantic_embed_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, table_road, table_other, out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_semantic_onehot_f32(const uint8_t* rasters, int64_t M, int32_t N,
                                        const int32_t* idx_road, int32_t nr,
                                        const int32_t* idx_other, int32_t no, float* onehot,
                                        int32_t KP, void* stream) {
  SemArgs a;
  const int rc = fill_args(a, rasters, M, N, idx_road, nr, idx_other, no, 4);
  if (rc != SNAP_OK) return rc;
  if (!onehot) return SNAP_ERR_NULL;
  if (KP < nr + 2 * no || KP % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(semantic_onehot_kernel, dim3((unsigned)snap_cdiv(M, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, onehot, KP);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
