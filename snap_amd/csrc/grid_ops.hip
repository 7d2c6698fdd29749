// Generic N-D grid operators of snap/utils/grids.py:116-153 as stand-alone entry points (inside
// the hot path the same interpolation is fused into lift.hip / pose.hip / voting.hip):
//   interpolate_nd  -- N-linear interpolation at corner-origin coordinates (half-pixel centres),
//                      jax.scipy.ndimage.map_coordinates(order=1, mode='nearest') semantics: the
//                      weights come from the UNCLIPPED coordinate, every tap index is clipped, the
//                      2^n products are added in itertools.product order; validity = in bounds
//                      AND no tap (even a zero-weight one) is invalid (the 0 * NaN trick of :131).
//   expectation_nd  -- sum_cells index(cell) * pdf(cell) per leading row (:148-153).
// argmax_nd (:140-145) is snap_argmax_rows_f32 + index arithmetic on the host side.
#include "common.h"

namespace {

struct InterpArgs {
  const float* array;
  const uint8_t* valid_array;
  const float* points;
  float* values;
  uint8_t* valid;
  int64_t K;
  int n, D;
  int size[3];
};

template <int N>
__global__ __launch_bounds__(256) void interpolate_nd_kernel(const InterpArgs a) {
  // one thread per (point, channel); the 2^N taps are recomputed per channel (not a hot path)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = a.K * a.D;
  if (i >= total) return;
  const int64_t k = i / a.D;
  const int d = (int)(i - k * a.D);
  int idx[N][2];
  float w[N][2];
  bool inb = true;
#pragma unroll
  for (int t = 0; t < N; ++t) {
    const float p = a.points[k * N + t];
    inb = inb && (p >= 0.f) && (p < (float)a.size[t]);
    const float c = p - 0.5f;
    const float lo = floorf(c);
    const float whi = c - lo;
    w[t][0] = 1.f - whi;
    w[t][1] = whi;
    const int il = (int)lo;
    idx[t][0] = min(max(il, 0), a.size[t] - 1);
    idx[t][1] = min(max(il + 1, 0), a.size[t] - 1);
  }
  float acc = 0.f;
  bool taps_ok = true;
#pragma unroll
  for (int c = 0; c < (1 << N); ++c) {
    int64_t off = 0;
    float ww = 1.f;
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const int bit = (c >> (N - 1 - t)) & 1;      // product order: the first axis varies slowest
      off = off * a.size[t] + idx[t][bit];
      ww = t == 0 ? w[t][bit] : ww * w[t][bit];
    }
    const float contrib = ww * a.array[off * a.D + d];
    acc = c == 0 ? contrib : acc + contrib;
    if (a.valid_array && a.valid_array[off] == 0) taps_ok = false;
  }
  a.values[i] = acc;
  if (d == 0) a.valid[k] = (inb && taps_ok) ? 1 : 0;
}

// one workgroup per row; fixed-order tree reduction (deterministic)
template <int N>
__global__ __launch_bounds__(256) void expectation_nd_kernel(const float* __restrict__ pdf,
                                                             int64_t cells, int s0, int s1, int s2,
                                                             float* __restrict__ out) {
  __shared__ float red[N][256];
  const int64_t row = blockIdx.x;
  const float* p = pdf + row * cells;
  float acc[N];
#pragma unroll
  for (int t = 0; t < N; ++t) acc[t] = 0.f;
  for (int64_t c = threadIdx.x; c < cells; c += 256) {
    const float v = p[c];
    int64_t r = c;
    int id[3] = {0, 0, 0};
    if (N == 3) { id[2] = (int)(r % s2); r /= s2; }
    if (N >= 2) { id[1] = (int)(r % s1); r /= s1; }
    id[0] = (int)r;
#pragma unroll
    for (int t = 0; t < N; ++t) acc[t] += (float)id[t] * v;
  }
#pragma unroll
  for (int t = 0; t < N; ++t) red[t][threadIdx.x] = acc[t];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int t = 0; t < N; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < N) out[row * N + threadIdx.x] = red[threadIdx.x][0];
}

}  // namespace

extern "C" int snap_interpolate_nd_f32(const float* array, const int32_t* size, int32_t n, int32_t D,
                                       const uint8_t* valid_array, const float* points, int64_t K,
                                       float* values, uint8_t* valid, void* stream) {
  if (!array || !size || !points || !values || !valid) return SNAP_ERR_NULL;
  if (n < 1 || n > 3 || D <= 0 || K <= 0) return SNAP_ERR_BAD_SHAPE;
  InterpArgs a{array, valid_array, points, values, valid, K, n, D, {1, 1, 1}};
  for (int t = 0; t < n; ++t) {
    if (size[t] <= 0) return SNAP_ERR_BAD_SHAPE;
    a.size[t] = size[t];
  }
  const dim3 grid((unsigned)snap_cdiv(K * D, 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 1) hipLaunchKernelGGL(interpolate_nd_kernel<1>, grid, dim3(256), 0, s, a);
  if (n == 2) hipLaunchKernelGGL(interpolate_nd_kernel<2>, grid, dim3(256), 0, s, a);
  if (n == 3) hipLaunchKernelGGL(interpolate_nd_kernel<3>, grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_expectation_nd_f32(const float* pdf, int64_t rows, const int32_t* size, int32_t n,
                                       float* out, void* stream) {
  if (!pdf || !size || !out) return SNAP_ERR_NULL;
  if (n < 1 || n > 3 || rows <= 0) return SNAP_ERR_BAD_SHAPE;
  int s[3] = {1, 1, 1};
  int64_t cells = 1;
  for (int t = 0; t < n; ++t) {
    if (size[t] <= 0) return SNAP_ERR_BAD_SHAPE;
    s[t] = size[t];
    cells *= size[t];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)rows);
  if (n == 1) hipLaunchKernelGGL(expectation_nd_kernel<1>, grid, dim3(256), 0, st, pdf, cells, s[0], s[1], s[2], out);
  if (n == 2) hipLaunchKernelGGL(expectation_nd_kernel<2>, grid, dim3(256), 0, st, pdf, cells, s[0], s[1], s[2], out);
  if (n == 3) hipLaunchKernelGGL(expectation_nd_kernel<3>, grid, dim3(256), 0, st, pdf, cells, s[0], s[1], s[2], out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
