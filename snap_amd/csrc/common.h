// Shared device/host helpers for libsnap_hip (gfx950 only).
#ifndef SNAP_CSRC_COMMON_H_
#define SNAP_CSRC_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snap_hip.h"

#define SNAP_WAVE 64

#define SNAP_CHECK_LAUNCH()                                \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return SNAP_ERR_LAUNCH;          \
  } while (0)

static inline int64_t snap_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave-level reductions over all 64 lanes (xor butterflies; result in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#endif  // SNAP_CSRC_COMMON_H_
