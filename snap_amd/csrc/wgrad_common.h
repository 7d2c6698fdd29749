// Shared pieces of the weight-gradient engines (wgrad.hip: exact f32; wgrad_bf16.hip: bf16
// operands / f32 accumulate): launch arguments, the fused prologues and the launch plan.
#ifndef SNAP_CSRC_WGRAD_COMMON_H_
#define SNAP_CSRC_WGRAD_COMMON_H_

#include <stdlib.h>

#include "common.h"

namespace snapwg {

struct WgradArgs {
  SnapConvDesc d;
  const float* x;
  const float* dy;
  float* partial;  // [S, K, Cout]
  const float* gn_mu;
  const float* gn_sc;
  const float* gn_beta;
  int M, K;
  int ctiles;      // channel tiles per (kh,kw) tap
  int ncol;        // Cout tiles
  int slabs_per_chunk;
  const int32_t* rows_z;     // optional: reduction row m reads x row rows_z[m] (flat 1x1 only)
  const int32_t* rows_dy;    // optional: ... and dy row rows_dy[m]
  const int32_t* row_count;  // optional device scalar: only the first *row_count rows exist
  int x_is_half;             // wgrad_bf16.hip: `x` holds 2-byte elements of the engine's type (prologue NONE,
                             //   Cin_stride % 4 == 0): the hidden activations of the masked MLP
  int dy_is_half;            // ... and / or `dy` does (the inter-layer gradients of the masked MLP)
};

struct WgPlan { int bkt, bn, ctiles, ncol, ktiles, S, slabs_per_chunk; };

inline WgPlan wg_plan(const SnapConvDesc& d, bool vec) {
  WgPlan p;
  p.bkt = (vec && d.Cin > 64) ? 128 : 64;
  p.bn = d.Cout > 64 ? 128 : 64;
  p.ctiles = (d.Cin + p.bkt - 1) / p.bkt;
  p.ncol = (d.Cout + p.bn - 1) / p.bn;
  const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
  // row tiles: per-tap channel tiles (VEC) or flat-k tiles (scalar path)
  p.ktiles = vec ? d.KH * d.KW * p.ctiles : (d.KH * d.KW * d.Cin + p.bkt - 1) / p.bkt;
  // the M split of the VEC plan is sized on 64-channel tiles so that it (and the
  // workspace) does not depend on BKT.
  const int64_t tiles = vec ? (int64_t)d.KH * d.KW * ((d.Cin + 63) / 64) * p.ncol
                            : (int64_t)p.ktiles * p.ncol;
  int64_t S = (1024 + tiles - 1) / tiles;
  const int64_t smax = (M + 255) / 256;   // >= 16 slabs per chunk
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  const int64_t slabs = (M + 15) / 16;
  p.slabs_per_chunk = (int)((slabs + S - 1) / S);
  p.S = (int)((slabs + p.slabs_per_chunk - 1) / p.slabs_per_chunk);
  return p;
}

// WIDE plan of the half-precision engine (wgrad_bf16.hip, 512 threads): flat / 1 x 1 kernel gradients with
// many rows and >= 192 input channels -- one workgroup owns a 256 x 256 (256 x 128) tile of dW, i.e. for the
// fusion / projection MLPs ALL of it, and reads its chunk of rows once.
inline bool wg_wide_ok(const SnapConvDesc& d, bool vec, int math) {
  const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
  return vec && math != SNAP_MATH_F32 && d.KH == 1 && d.KW == 1 && d.Cin >= 192 && d.Cout > 64 && M >= 65536 &&
         (d.prologue == SNAP_PRO_NONE || d.prologue == SNAP_PRO_RELU || d.prologue == SNAP_PRO_AFFINE);
}

inline WgPlan wg_plan_wide(const SnapConvDesc& d) {
  WgPlan p;
  p.bkt = 256;
  p.bn = d.Cout > 128 ? 256 : 128;
  p.ctiles = (d.Cin + 255) / 256;
  p.ncol = (d.Cout + p.bn - 1) / p.bn;
  p.ktiles = p.ctiles;
  const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
  const int tiles = p.ktiles * p.ncol;
  int64_t S = ((p.bn == 256 ? 256 : 512) + tiles - 1) / tiles;     // one (two) workgroup(s) per CU
  const int64_t slabs = (M + 15) / 16;
  int64_t spc = (slabs + S - 1) / S;
  spc += spc & 1;                                                  // whole 32-row slabs per chunk
  p.slabs_per_chunk = (int)spc;
  p.S = (int)((slabs + spc - 1) / spc);
  return p;
}

// Fused-tap plan of the 3 x 3 / stride 1 / pad 1 kernel gradients (wgrad3x3.hip): 64 input channels x 128 (64)
// output channels x 9 taps per workgroup, the reduction in patches of 4 x 8 output pixels; slabs_per_chunk
// counts PATCHES, S counts partial slots (two per chunk for the 64-column tile).
inline bool wg_3x3_ok(const SnapConvDesc& d, bool vec, int math, bool x_half, bool dy_half, bool lists) {
  return vec && math != SNAP_MATH_F32 && dy_half && !x_half && !lists && d.KH == 3 && d.KW == 3 && d.stride == 1 &&
         d.pad_t == 1 && d.pad_l == 1 && d.Ho == d.H && d.Wo == d.W && d.Cin % 4 == 0 && d.Cin >= 32 && d.Cout >= 32 &&
         (d.prologue == SNAP_PRO_NONE || d.prologue == SNAP_PRO_GN_RELU);
}

inline WgPlan wg_plan_3x3(const SnapConvDesc& d) {
  WgPlan p;
  p.bkt = 64;
  p.bn = d.Cout > 64 ? 128 : 64;
  p.ctiles = (d.Cin + 63) / 64;
  p.ncol = (d.Cout + p.bn - 1) / p.bn;
  p.ktiles = p.ctiles;
  const int64_t NP = (int64_t)d.N * ((d.Ho + 3) / 4) * ((d.Wo + 7) / 8);
  const int tiles = p.ktiles * p.ncol;
  int64_t S = (256 + tiles - 1) / tiles;          // eight-wave workgroups: one per CU
  const int64_t smax = (NP + 7) / 8;              // >= 8 patches per chunk
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  const int64_t ppc = (NP + S - 1) / S;
  p.slabs_per_chunk = (int)ppc;
  p.S = (int)((NP + ppc - 1) / ppc) * (p.bn == 64 ? 2 : 1);
  return p;
}

// bf16-operand engine (wgrad_bf16.hip); `a` / `p` prepared by snap_conv2d_wgrad_ex_f32
int launch_bf16(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s);   // half: IEEE f16 operands
int launch_3x3(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s);    // wgrad3x3.hip (wg_plan_3x3)

}  // namespace snapwg

namespace {
using snapwg::WgradArgs;
using snapwg::WgPlan;
using snapwg::wg_plan;

template <int PRO>
__device__ __forceinline__ float wg_pro(float v, float mu, float sc, float beta, float s, float t) {
  if constexpr (PRO == SNAP_PRO_AFFINE) return v * s + t;
  if constexpr (PRO == SNAP_PRO_GN_RELU) return snap_relu((v - mu) * sc + beta);
  if constexpr (PRO == SNAP_PRO_RELU_GN) return (snap_relu(v) - mu) * sc + beta;
  if constexpr (PRO == SNAP_PRO_RELU) return snap_relu(v);
  return v;
}

}  // namespace

#endif  // SNAP_CSRC_WGRAD_COMMON_H_
