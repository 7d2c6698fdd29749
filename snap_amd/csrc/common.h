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
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave-level reductions over all 64 lanes; the (bitwise identical) result lands in every
// lane.  Within each 16-lane row: four DPP rotate steps (row_ror 8/4/2/1 -- no LDS traffic,
// unlike __shfl_xor, which lowers to ds_bpermute_b32 and a full LDS round trip per step);
// across the four rows: v_readlane of one lane per row.  Rotations keep every step symmetric
// (a + b == b + a), so all lanes hold the same bits.
template <int CTRL>
__device__ __forceinline__ float snap_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float snap_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += snap_dpp<0x128>(v);   // row_ror:8
  v += snap_dpp<0x124>(v);   // row_ror:4
  v += snap_dpp<0x122>(v);   // row_ror:2
  v += snap_dpp<0x121>(v);   // row_ror:1
  return (snap_lane(v, 0) + snap_lane(v, 16)) + (snap_lane(v, 32) + snap_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, snap_dpp<0x128>(v));
  v = fmaxf(v, snap_dpp<0x124>(v));
  v = fmaxf(v, snap_dpp<0x122>(v));
  v = fmaxf(v, snap_dpp<0x121>(v));
  return fmaxf(fmaxf(snap_lane(v, 0), snap_lane(v, 16)), fmaxf(snap_lane(v, 32), snap_lane(v, 48)));
}

// jnp.max / jnp.maximum semantics: a NaN operand makes the result NaN (fmaxf alone is IEEE maxNum
// and DROPS it).  The pooling kernels (bev_mapper.py:63-78: jnp.max(where=...)) and the ReLUs of
// the MLP / epilogue paths use these, so that a non-finite forward pass reaches the plane -- and
// from there the loss and the trainer's non-finite step skip (trainer.py:260-277) -- as it does
// in the reference.  snap_max_nan returns the canonical POSITIVE quiet NaN: through the
// sign-split integer atomic max of mlp_pool.hip its bit pattern beats every finite value.
__device__ __forceinline__ float snap_max_nan(float a, float b) {
  const float m = fmaxf(a, b);
  return (a != a || b != b) ? __int_as_float(0x7fc00000) : m;
}
__device__ __forceinline__ float snap_relu(float v) { return v < 0.f ? 0.f : v; }   // relu(NaN) = NaN

#endif  // SNAP_CSRC_COMMON_H_
