// Pose-likelihood kernels: point-vs-map similarity + softmax statistics (k10),
// correspondence sampling + 2-point Kabsch (k11), pose scoring with LDS-staged
// score planes (k12) and the refinement lattice (k13).
//
// Replaces snap/models/bev_localizer.py:157-173 and
// snap/models/pose_estimation.py:49-82,100-165,168-205.
//
// Data layout in HBM (per scene b):
//   fq   [Nq, Dm]      query matching features (L2-normalised, masked)
//   fm   [X*Y, Dm]     map matching features
//   sim  [Nq, X*Y]     sim_points, one contiguous X*Y plane per query point
//   chunk_stats [Nq, ceil(XY/64), 2]  per-64-cell (max, sum exp) of the softmax
// The prob_points tensor of the reference is never materialised on the product
// path: its row-softmax is represented by chunk_stats (1/32 of the bytes) and
// re-evaluated on the fly by the sampler.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SIM_CH = 64;  // cells per softmax chunk == one wave
constexpr int SIM_TQ = 64;  // query rows per workgroup tile

// ---------------------------------------------------------------------------
// k10: similarity.  MODE 0: write sim + chunk stats.  MODE 1: write prob.
// grid = (ceil(XY/256), ceil(Nq/TQ), B), block = 256 (thread <-> map cell).
// Each thread keeps its map cell's Dm-vector in registers and streams the TQ
// query rows (LDS broadcast reads); sim rows are written 1 KiB-coalesced.
// ---------------------------------------------------------------------------
template <int DM, int MODE>
__global__ __launch_bounds__(256) void sim_kernel(
    const float* __restrict__ fq, const float* __restrict__ fm, int Nq, int XY, float scale,
    int clip, const float* __restrict__ num_valid, float* __restrict__ sim,
    float* __restrict__ stats, const float* __restrict__ rowstats, float* __restrict__ prob,
    const float* __restrict__ row_weight) {
  __shared__ float q_s[SIM_TQ * DM];
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SIM_TQ;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int NC = (XY + SIM_CH - 1) / SIM_CH;
  const bool cvalid = cell < XY;
  // query rows staged K-MAJOR ([k][row]) so that two consecutive rows of one k are an
  // aligned 8-byte pair: the dot products of TWO rows advance with one v_pk_fma_f32.
  for (int i = threadIdx.x; i < SIM_TQ * DM; i += 256) {
    const int r = i / DM, k = i - r * DM;
    q_s[k * SIM_TQ + r] = (n0 + r < Nq) ? fq[((int64_t)b * Nq + n0) * DM + i] : 0.f;
  }
  f32x2 mv[DM];
  if (cvalid) {
    const f32x4* src = reinterpret_cast<const f32x4*>(fm + ((int64_t)b * XY + cell) * DM);
#pragma unroll
    for (int k = 0; k < DM / 4; ++k) {
      const f32x4 t = src[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) mv[4 * k + e] = f32x2{t[e], t[e]};
    }
  } else {
#pragma unroll
    for (int k = 0; k < DM; ++k) mv[k] = f32x2{0.f, 0.f};
  }
  __syncthreads();
  const float nv = num_valid[b];
  const float rnv = 1.0f / nv;      // sim = x * (1 / num_valid): one rounding away from x / num_valid
  const int rows = min(SIM_TQ, Nq - n0);
  for (int r0 = 0; r0 < rows; r0 += 2) {
    f32x2 dot2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DM; ++k)   // per component the same k-ordered fmaf chain as before
      dot2 = __builtin_elementwise_fma(*reinterpret_cast<const f32x2*>(q_s + k * SIM_TQ + r0), mv[k], dot2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = r0 + h;
      if (r >= rows) break;
      const int64_t row = (int64_t)b * Nq + n0 + r;
      const float dot = h ? dot2.y : dot2.x;
      float x = clip ? fmaxf(dot, 0.f) : dot;
      x *= scale;
      // bev_localizer.py:165-172: confidence weights (masked softmax over the query points)
      // replace the 1 / num_valid normalisation when add_confidence_query is set
      const float wrow = row_weight ? row_weight[row] : rnv;
      if (MODE == 0) {
        if (cvalid) sim[row * XY + cell] = x * wrow;
        const float m = wave_max(cvalid ? x : -INFINITY);
        const float s = wave_sum(cvalid ? expf(x - m) : 0.f);
        if (lane == 0 && chunk < NC) {
          stats[(row * NC + chunk) * 2 + 0] = m;
          stats[(row * NC + chunk) * 2 + 1] = s;
        }
      } else {
        const float M = rowstats[row * 2 + 0], T = rowstats[row * 2 + 1];
        if (cvalid) prob[row * XY + cell] = row_weight ? (expf(x - M) / T) * wrow : (expf(x - M) / T) / nv;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k10 on the f32 matrix cores (MODE 0 of the kernel above: sim + chunk statistics).
// D[n, cell] = sum_k fq[n, k] fm[cell, k] as 32x32x2 f32 MFMAs (each the exact k-ordered fmaf
// chain the VALU kernel runs, so sim and the sampler's re-evaluation keep their bits).
// grid as above; a wave owns the 64 cells of one softmax chunk (B operand, held in registers for
// the whole tile) and the tile's 64 query rows (A operand): 2 x 2 tiles of 32 x 32, DM / 2 MFMAs
// each.  Neither operand touches LDS: lane (l31, lhi) of an MFMA holds element k = 2 j + lhi of
// row / column l31, picked out of the lane's own contiguous DM-vector.  Epilogue straight from the
// accumulators: a register holds, per half-wave, 32 consecutive cells of one query row -> one
// 128-byte line per half-wave and store; the chunk's (max, sum exp) pair is a half-wave
// reduction (4 DPP row rotates + one cross-row exchange).
// The VALU kernel was bound by its packed-FMA issue (97 % VALU busy, 1.65 ms at C2 for a
// 2.44 GB write that the HBM can take in 0.4 ms).
// ---------------------------------------------------------------------------
template <int DM>
__device__ __forceinline__ void sim_load_operand(const float* __restrict__ row, bool valid, int lhi,
                                                 float (&out)[DM / 2]) {
  if (!valid) {
#pragma unroll
    for (int j = 0; j < DM / 2; ++j) out[j] = 0.f;
    return;
  }
  const f32x4* src = reinterpret_cast<const f32x4*>(row);
#pragma unroll
  for (int q = 0; q < DM / 4; ++q) {
    const f32x4 t = src[q];
    out[2 * q] = lhi ? t[1] : t[0];
    out[2 * q + 1] = lhi ? t[3] : t[2];
  }
}

__device__ __forceinline__ float half_wave_max(float v) {
  v = fmaxf(v, snap_dpp<0x128>(v));
  v = fmaxf(v, snap_dpp<0x124>(v));
  v = fmaxf(v, snap_dpp<0x122>(v));
  v = fmaxf(v, snap_dpp<0x121>(v));
  return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v += snap_dpp<0x128>(v);
  v += snap_dpp<0x124>(v);
  v += snap_dpp<0x122>(v);
  v += snap_dpp<0x121>(v);
  return v + __shfl_xor(v, 16, 64);
}

template <int DM>
__global__ __launch_bounds__(256, DM <= 32 ? 4 : 2) void sim_mfma_kernel(
    const float* __restrict__ fq, const float* __restrict__ fm, int Nq, int XY, float scale,
    int clip, const float* __restrict__ num_valid, float* __restrict__ sim,
    float* __restrict__ stats, const float* __restrict__ row_weight) {
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SIM_TQ;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int chunk = blockIdx.x * 4 + wave;
  const int cell0 = chunk * SIM_CH;
  const int NC = (XY + SIM_CH - 1) / SIM_CH;
  if (cell0 >= XY) return;                      // (whole wave; no barrier in this kernel)
  float bq[2][DM / 2], aq[2][DM / 2];
  bool cv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int cell = cell0 + 32 * t + l31;
    cv[t] = cell < XY;
    sim_load_operand<DM>(fm + ((int64_t)b * XY + (cv[t] ? cell : 0)) * DM, cv[t], lhi, bq[t]);
    const int row = n0 + 32 * t + l31;
    sim_load_operand<DM>(fq + ((int64_t)b * Nq + (row < Nq ? row : 0)) * DM, row < Nq, lhi, aq[t]);
  }
#ifndef SNAP_SIM_ABLATE
#define SNAP_SIM_ABLATE 0          // timing experiments only: 1 no sim stores, 2 no MFMAs, 4 no exp
#endif
  // Epilogue through a per-wave LDS tile (32 rows x 64 cells, one ti at a time): the MFMA C
  // layout gives a lane ONE cell of 16 rows -- 64 dword stores per tile, and narrow stores are
  // issue-bound.  Staged, 16 lanes own one query row's 64 cells as float4: a wave store covers
  // four 256-byte row segments with dwordx4 (8 stores per ti instead of 32), and the chunk's
  // (max, sum exp) is a reduction inside one 16-lane DPP row.
  __shared__ __attribute__((aligned(16))) float stage[4][32][64 + 4];   // +4: rows 4 apart hit distinct banks
  float (*st)[64 + 4] = stage[wave];
  const float rnv = 1.0f / num_valid[b];   // as the VALU kernel: the two stay bit-identical
  const int sub = lane >> 4;            // row within a 4-row pass
  const int c4 = (lane & 15) * 4;       // first of the lane's 4 cells
  const int ncell = XY - cell0;         // live cells of this chunk (>= 1)
  // One 32-row half at a time: the stores of half 0 drain while the matrix pipe works on half 1
  // (measured: the contraction costs 0.46 ms and the stores 0.38 ms of the kernel's 1.17 ms when
  // all MFMAs run first -- they did not overlap), and the kernel holds 32 accumulator registers
  // instead of 64 (one more wave per SIMD).
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < ((SNAP_SIM_ABLATE & 2) ? 1 : DM / 2); ++j)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
        acc[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[ti][j], bq[tj][j], acc[tj], 0, 0, 0);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st[(r & 3) + 8 * (r >> 2) + 4 * lhi][32 * tj + l31] = acc[tj][r];   // MFMA C layout
    // (the tile is private to the wave: LDS ops of one wave complete in order, no barrier)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int rr = 4 * p + sub;
      const int n = n0 + 32 * ti + rr;
      const bool live = n < Nq;
      const int64_t row = (int64_t)b * Nq + (live ? n : 0);
      const float wrow = row_weight ? row_weight[row] : rnv;
      f32x4 x = *reinterpret_cast<const f32x4*>(&st[rr][c4]);
      float m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (clip) x[e] = fmaxf(x[e], 0.f);
        x[e] *= scale;
        if (c4 + e < ncell) m = fmaxf(m, x[e]);
      }
      m = fmaxf(m, snap_dpp<0x128>(m));
      m = fmaxf(m, snap_dpp<0x124>(m));
      m = fmaxf(m, snap_dpp<0x122>(m));
      m = fmaxf(m, snap_dpp<0x121>(m));
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c4 + e < ncell) sum += (SNAP_SIM_ABLATE & 4) ? (x[e] - m) : __expf(x[e] - m);   // v_exp_f32: ~1e-6 relative on the chunk mass
      sum += snap_dpp<0x128>(sum);
      sum += snap_dpp<0x124>(sum);
      sum += snap_dpp<0x122>(sum);
      sum += snap_dpp<0x121>(sum);
      if (live) {
        float* o = sim + row * XY + cell0 + c4;
        if (SNAP_SIM_ABLATE & 1) {
        } else if (c4 + 3 < ncell && ((XY & 3) == 0)) {
          *reinterpret_cast<f32x4*>(o) = f32x4{x[0] * wrow, x[1] * wrow, x[2] * wrow, x[3] * wrow};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c4 + e < ncell) o[e] = x[e] * wrow;
        }
        if ((lane & 15) == 0) {
          stats[(row * NC + chunk) * 2 + 0] = m;
          stats[(row * NC + chunk) * 2 + 1] = sum;
        }
      }
    }
  }
}

// ---- the same similarity with the contraction on the BF16 matrix cores, f32-grade -------------
// f32 MFMAs cost 0.46 ms of the kernel above at C2 (39 GFLOP at half the f32 matrix peak).  Here
// fq / fm are split once into NS bf16 parts per element (hi = bf16(v), mid = bf16(v - hi), lo =
// bf16(v - hi - mid): conv_split.hip's arithmetic; NS = 3 keeps 24 significand bits per operand,
// six part products per MAC, error ~2^-24 per product -- inside the f32 chain's own rounding) by
// sim_presplit_kernel, and the tile is NS (NS + 1) / 2 x DM / 16 v_mfma_f32_32x32x16_bf16 per
// 32 x 32 block instead of DM / 2 f32 MFMAs: 2.7 x fewer matrix cycles at NS = 3.  Same epilogue.
typedef __bf16 sim_bf16x8 __attribute__((ext_vector_type(8)));

// x [R, DM] f32 -> [R][NS][DM] bf16
template <int NS>
__global__ __launch_bounds__(256) void sim_presplit_kernel(const float* __restrict__ x, int64_t n,
                                                           int DM, __bf16* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t r = i / DM;
  const int k = (int)(i - r * DM);
  float v = x[i];
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    const __bf16 b = (__bf16)v;
    out[(r * NS + p) * DM + k] = b;
    v -= (float)b;
  }
}

template <int DM, int NS>
__global__ __launch_bounds__(256, 3) void sim_split_kernel(
    const __bf16* __restrict__ fqs, const __bf16* __restrict__ fms, int Nq, int XY, float scale,
    int clip, const float* __restrict__ num_valid, float* __restrict__ sim,
    float* __restrict__ stats, const float* __restrict__ row_weight) {
  constexpr int KS = DM / 16;
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SIM_TQ;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int chunk = blockIdx.x * 4 + wave;
  const int cell0 = chunk * SIM_CH;
  const int NC = (XY + SIM_CH - 1) / SIM_CH;
  if (cell0 >= XY) return;
  // B operand (map cells), both 32-cell tiles, kept for both row halves
  sim_bf16x8 bq[2][KS][NS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int cell = cell0 + 32 * t + l31;
    const bool cv = cell < XY;
    const __bf16* src = fms + ((int64_t)b * XY + (cv ? cell : 0)) * (NS * DM) + 8 * lhi;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        bq[t][s][p] = *reinterpret_cast<const sim_bf16x8*>(src + p * DM + 16 * s);
        if (!cv) bq[t][s][p] = sim_bf16x8{};
      }
  }
  __shared__ __attribute__((aligned(16))) float stage[4][32][64 + 4];
  float (*st)[64 + 4] = stage[wave];
  const float rnv = 1.0f / num_valid[b];
  const int sub = lane >> 4;
  const int c4 = (lane & 15) * 4;
  const int ncell = XY - cell0;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    sim_bf16x8 aq[KS][NS];
    {
      const int row = n0 + 32 * ti + l31;
      const bool rv = row < Nq;
      const __bf16* src = fqs + ((int64_t)b * Nq + (rv ? row : 0)) * (NS * DM) + 8 * lhi;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          aq[s][p] = *reinterpret_cast<const sim_bf16x8*>(src + p * DM + 16 * s);
          if (!rv) aq[s][p] = sim_bf16x8{};
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#define SNAP_SIM_PRODUCT(PA, PB)                                                               \
  _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)                                               \
      acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[s][PA], bq[tj][s][PB], acc[tj], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (NS == 3) {               // smallest terms first, as conv_split.hip
        SNAP_SIM_PRODUCT(2, 0)
        SNAP_SIM_PRODUCT(0, 2)
        SNAP_SIM_PRODUCT(1, 1)
        SNAP_SIM_PRODUCT(1, 0)
        SNAP_SIM_PRODUCT(0, 1)
        SNAP_SIM_PRODUCT(0, 0)
      } else {
        SNAP_SIM_PRODUCT(1, 0)
        SNAP_SIM_PRODUCT(0, 1)
        SNAP_SIM_PRODUCT(0, 0)
      }
    }
#undef SNAP_SIM_PRODUCT
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st[(r & 3) + 8 * (r >> 2) + 4 * lhi][32 * tj + l31] = acc[tj][r];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int rr = 4 * p + sub;
      const int n = n0 + 32 * ti + rr;
      const bool live = n < Nq;
      const int64_t row = (int64_t)b * Nq + (live ? n : 0);
      const float wrow = row_weight ? row_weight[row] : rnv;
      f32x4 x = *reinterpret_cast<const f32x4*>(&st[rr][c4]);
      float m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (clip) x[e] = fmaxf(x[e], 0.f);
        x[e] *= scale;
        if (c4 + e < ncell) m = fmaxf(m, x[e]);
      }
      m = fmaxf(m, snap_dpp<0x128>(m));
      m = fmaxf(m, snap_dpp<0x124>(m));
      m = fmaxf(m, snap_dpp<0x122>(m));
      m = fmaxf(m, snap_dpp<0x121>(m));
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c4 + e < ncell) sum += __expf(x[e] - m);
      sum += snap_dpp<0x128>(sum);
      sum += snap_dpp<0x124>(sum);
      sum += snap_dpp<0x122>(sum);
      sum += snap_dpp<0x121>(sum);
      if (live) {
        float* o = sim + row * XY + cell0 + c4;
        if (c4 + 3 < ncell && ((XY & 3) == 0)) {
          *reinterpret_cast<f32x4*>(o) = f32x4{x[0] * wrow, x[1] * wrow, x[2] * wrow, x[3] * wrow};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c4 + e < ncell) o[e] = x[e] * wrow;
        }
        if ((lane & 15) == 0) {
          stats[(row * NC + chunk) * 2 + 0] = m;
          stats[(row * NC + chunk) * 2 + 1] = sum;
        }
      }
    }
  }
}

// The same kernel for XY % 256 == 0 (every chunk of every workgroup full: C2 128 x 128, C4 256 x 256,
// the reference's 120 x 160 map), leaner around the same arithmetic -- identical bits in sim and in
// the chunk statistics (tests/test_gpu_kernels.py compares the two kernels bit for bit):
//   * operand roles swapped in the MFMA (A = map cells, B = query rows: the same products in the same
//     k positions, the transposed accumulator tile): a lane then holds FOUR CONSECUTIVE cells of one
//     row per accumulator quad, so the tile goes to the staging buffer in 8 ds_write_b128 instead of
//     32 ds_write_b32 (stride 68 floats: the 16 rows of a lane group land on distinct banks; the old
//     b32 pattern had a 25 % conflict rate);
//   * no tail masks (4 compare / select per element) and the clip flag at compile time;
//   * the (max, sum) pairs of the workgroup's four chunks leave as ONE 32-byte store per row (they
//     are adjacent in [row][chunk][2]) instead of four 8-byte partial-line writes: the statistics
//     were a third of the kernel's write requests;
//   * sim is written with non-temporal stores (2.5 GB streamed once, read back later by the scoring
//     kernel: nothing of it is worth an L2 line now).
template <int DM, int NS, bool CLIP>
__global__ __launch_bounds__(256, 3) void sim_split_fast_kernel(
    const __bf16* __restrict__ fqs, const __bf16* __restrict__ fms, int Nq, int XY, float scale,
    const float* __restrict__ num_valid, float* __restrict__ sim, float* __restrict__ stats,
    const float* __restrict__ row_weight) {
  constexpr int KS = DM / 16;
  constexpr int LD = 64 + 4;
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SIM_TQ;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int chunk = blockIdx.x * 4 + wave;
  const int cell0 = chunk * SIM_CH;
  const int NC = XY / SIM_CH;
  sim_bf16x8 bq[2][KS][NS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const __bf16* src = fms + ((int64_t)b * XY + cell0 + 32 * t + l31) * (NS * DM) + 8 * lhi;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int p = 0; p < NS; ++p) bq[t][s][p] = *reinterpret_cast<const sim_bf16x8*>(src + p * DM + 16 * s);
  }
  __shared__ __attribute__((aligned(16))) float stage[4][32][LD];
  __shared__ __attribute__((aligned(16))) float sstat[SIM_TQ][8];
  float (*st)[LD] = stage[wave];
  const float rnv = 1.0f / num_valid[b];
  const int sub = lane >> 4;
  const int c4 = (lane & 15) * 4;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    sim_bf16x8 aq[KS][NS];
    {
      const int row = n0 + 32 * ti + l31;
      const bool rv = row < Nq;
      const __bf16* src = fqs + ((int64_t)b * Nq + (rv ? row : 0)) * (NS * DM) + 8 * lhi;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          aq[s][p] = *reinterpret_cast<const sim_bf16x8*>(src + p * DM + 16 * s);
          if (!rv) aq[s][p] = sim_bf16x8{};
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // (cells as the A operand, rows as B: acc[tj][r] = sim[row l31][cell 32 tj + (r & 3) + 8 (r >> 2) + 4 lhi])
#define SNAP_SIM_PRODUCT(PA, PB)                                                               \
  _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)                                               \
      acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[tj][s][PB], aq[s][PA], acc[tj], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (NS == 3) {
        SNAP_SIM_PRODUCT(2, 0)
        SNAP_SIM_PRODUCT(0, 2)
        SNAP_SIM_PRODUCT(1, 1)
        SNAP_SIM_PRODUCT(1, 0)
        SNAP_SIM_PRODUCT(0, 1)
        SNAP_SIM_PRODUCT(0, 0)
      } else {
        SNAP_SIM_PRODUCT(1, 0)
        SNAP_SIM_PRODUCT(0, 1)
        SNAP_SIM_PRODUCT(0, 0)
      }
    }
#undef SNAP_SIM_PRODUCT
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(&st[l31][32 * tj + 8 * g + 4 * lhi]) =
            f32x4{acc[tj][4 * g], acc[tj][4 * g + 1], acc[tj][4 * g + 2], acc[tj][4 * g + 3]};
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int rr = 4 * p + sub;
      const int n = n0 + 32 * ti + rr;
      const bool live = n < Nq;
      const int64_t row = (int64_t)b * Nq + (live ? n : 0);
      const float wrow = row_weight ? row_weight[row] : rnv;
      f32x4 x = *reinterpret_cast<const f32x4*>(&st[rr][c4]);
      float m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (CLIP) x[e] = fmaxf(x[e], 0.f);
        x[e] *= scale;
        m = fmaxf(m, x[e]);
      }
      m = fmaxf(m, snap_dpp<0x128>(m));
      m = fmaxf(m, snap_dpp<0x124>(m));
      m = fmaxf(m, snap_dpp<0x122>(m));
      m = fmaxf(m, snap_dpp<0x121>(m));
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += __expf(x[e] - m);
      sum += snap_dpp<0x128>(sum);
      sum += snap_dpp<0x124>(sum);
      sum += snap_dpp<0x122>(sum);
      sum += snap_dpp<0x121>(sum);
      if (live) {
        const f32x4 o = f32x4{x[0] * wrow, x[1] * wrow, x[2] * wrow, x[3] * wrow};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(sim + row * XY + cell0 + c4));
      }
      if ((lane & 15) == 0) {
        sstat[32 * ti + rr][2 * wave] = m;
        sstat[32 * ti + rr][2 * wave + 1] = sum;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * SIM_TQ) {
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    const int n = n0 + r;
    if (n < Nq)
      *reinterpret_cast<f32x4*>(stats + (((int64_t)b * Nq + n) * NC + blockIdx.x * 4) * 2 + 4 * h) =
          *reinterpret_cast<const f32x4*>(&sstat[r][4 * h]);
  }
}

// layers.masked_softmax over the query points (snap/models/layers.py:38-43) + its inclusive CDF
// (the sampler's row distribution).  One workgroup per scene; fixed-order sums.
__global__ __launch_bounds__(256) void masked_softmax_rows_kernel(
    const float* __restrict__ x, const uint8_t* __restrict__ mask, int N, float* __restrict__ w,
    float* __restrict__ cdf) {
  __shared__ float red[256];
  __shared__ int any_s;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* xr = x + (int64_t)b * N;
  const uint8_t* mr = mask + (int64_t)b * N;
  const int seg = (N + 255) / 256;
  const int i0 = min(t * seg, N), i1 = min(i0 + seg, N);
  int any = 0;
  for (int i = i0; i < i1; ++i) any |= mr[i];
  if (t == 0) any_s = 0;
  __syncthreads();
  if (any) atomicOr(&any_s, 1);
  __syncthreads();
  const bool all = any_s == 0;                 // nothing valid: the mask becomes all-true
  float m = -INFINITY;
  for (int i = i0; i < i1; ++i)
    if (all || mr[i]) m = fmaxf(m, xr[i]);
  red[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmaxf(red[t], red[t + o]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float loc = 0.f;
  for (int i = i0; i < i1; ++i) loc += (all || mr[i]) ? expf(xr[i] - m) : 0.f;
  red[t] = loc;
  __syncthreads();
  // inclusive scan of the 256 segment sums (Hillis-Steele, fixed order)
  for (int o = 1; o < 256; o <<= 1) {
    const float add = t >= o ? red[t - o] : 0.f;
    __syncthreads();
    red[t] += add;
    __syncthreads();
  }
  const float total = red[255];
  float run = red[t] - loc;
  for (int i = i0; i < i1; ++i) {
    const float e = (all || mr[i]) ? expf(xr[i] - m) : 0.f;
    run += e;
    w[(int64_t)b * N + i] = e / total;
    cdf[(int64_t)b * N + i] = run / total;
  }
}

// one wave per (b, n): row max M and total T = sum_c s_c * exp(m_c - M).
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ stats,
                                                        int64_t rows, int NC,
                                                        float* __restrict__ rowstats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* st = stats + row * NC * 2;
  float m = -INFINITY;
  for (int c = lane; c < NC; c += 64) m = fmaxf(m, st[2 * c]);
  m = wave_max(m);
  float t = 0.f;
  for (int c = lane; c < NC; c += 64) t += st[2 * c + 1] * expf(st[2 * c] - m);
  t = wave_sum(t);
  if (lane == 0) {
    rowstats[row * 2 + 0] = m;
    rowstats[row * 2 + 1] = t;
  }
}

// ---------------------------------------------------------------------------
// k11a: iid categorical sampling of correspondences ~ prob_points.
// prob mass is uniform over query points (every row softmax sums to 1), so:
// n = floor(u1 * Nq); cell ~ softmax row n by two-level inverse CDF:
// chunk from chunk_stats, then the 64 cells of that chunk re-evaluated.
// One wave per sample.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t* out) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// one wave per (b, n): the row maximum M and, per lane l, the inclusive prefix over lanes of
// the mass of lane l's chunk range (w_c = s_c * exp(m_c - M)) -- exactly the quantities the
// sampler used to rebuild per SAMPLE (2 passes over the row's statistics, 2 x cpl expf, a
// wave max and a wave scan).  ~34 samples share a row at C2; with this table a sample reads 64
// floats and only the selected lane re-evaluates its own chunks.  Bit-identical samples.
__global__ __launch_bounds__(256) void chunk_prefix_kernel(const float* __restrict__ stats,
                                                           int64_t rows, int NC,
                                                           float* __restrict__ lane_incl,
                                                           float* __restrict__ rowmax) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* st = stats + row * NC * 2;
  const int cpl = (NC + 63) / 64;
  const int cb = lane * cpl, ce = min(cb + cpl, NC);
  float M = -INFINITY;
  for (int c = cb; c < ce; ++c) M = fmaxf(M, st[2 * c]);
  M = wave_max(M);
  float local = 0.f;
  for (int c = cb; c < ce; ++c) local += st[2 * c + 1] * __expf(st[2 * c] - M);
  lane_incl[row * 64 + lane] = wave_scan_incl(local, lane);
  if (lane == 0) rowmax[row] = M;
}

template <int DM>
__global__ __launch_bounds__(256) void ransac_sample_kernel(
    const float* __restrict__ fq, const float* __restrict__ fm, const float* __restrict__ stats,
    int Nq, int X, int Y, float scale, int clip, int S, uint64_t seed,
    const float* __restrict__ uniforms, int32_t* __restrict__ corr,
    const float* __restrict__ lane_incl, const float* __restrict__ rowmax,
    const float* __restrict__ row_cdf, const float* __restrict__ sim,
    const float* __restrict__ row_unscale) {
  const int lane = threadIdx.x & 63;
  // one wave = one correspondence: everything that depends only on (scene, sample) is wave-uniform.
  // readfirstlane tells the compiler so: the Philox rounds, the row / chunk bookkeeping and the
  // address arithmetic then run on the scalar unit instead of 64-wide on the (saturated) VALU
  const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int b = blockIdx.y;
  if (s >= S) return;
  const int XY = X * Y;
  const int NC = (XY + SIM_CH - 1) / SIM_CH;
  float u1, u2;
  if (uniforms) {
    u1 = uniforms[((int64_t)b * S + s) * 2 + 0];
    u2 = uniforms[((int64_t)b * S + s) * 2 + 1];
  } else {
    uint32_t rnd[4];
    philox4x32_10((uint32_t)s, (uint32_t)b, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
    u1 = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
    u2 = (float)(rnd[1] >> 8) * (1.0f / 16777216.0f);
  }
  int n;
  if (row_cdf) {
    // rows carry the confidence weights as mass (bev_localizer.py:165-168): first n whose
    // inclusive CDF exceeds u1 * total (wave-uniform binary search)
    const float* cdf = row_cdf + (int64_t)b * Nq;
    const float tgt = u1 * cdf[Nq - 1];
    int lo = 0, hi = Nq - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > tgt) hi = mid; else lo = mid + 1;
    }
    n = lo;
  } else {
    n = min((int)(u1 * (float)Nq), Nq - 1);
  }
  n = __builtin_amdgcn_readfirstlane(n);
  const int64_t row = (int64_t)b * Nq + n;
  const float* st = stats + row * NC * 2;

  // level 1: pick the chunk.  lane l owns the contiguous chunk range [l*cpl, (l+1)*cpl).
  const int cpl = (NC + 63) / 64;
  const int cb = lane * cpl, ce = min(cb + cpl, NC);
  float M, local = 0.f, incl;
  const bool table = lane_incl != nullptr;
  if (table) {                   // per-row prefix table (chunk_prefix_kernel)
    M = rowmax[row];
    incl = lane_incl[row * 64 + lane];
  } else {
    M = -INFINITY;
    for (int c = cb; c < ce; ++c) M = fmaxf(M, st[2 * c]);
    M = wave_max(M);
    for (int c = cb; c < ce; ++c) local += st[2 * c + 1] * __expf(st[2 * c] - M);
    incl = wave_scan_incl(local, lane);
  }
  const float total = __shfl(incl, 63, 64);
  const float target = u2 * total;
  unsigned long long bal = __ballot(incl > target && ce > cb);
  int L;
  if (bal) {
    L = __ffsll((long long)bal) - 1;
  } else {  // rounding: fall back to the last non-empty lane
    const unsigned long long ne = __ballot(ce > cb);
    L = 63 - __clzll((long long)ne);
  }
  // the chunks of lane L: lane i evaluates the mass of chunk cbL + i (one exp per lane, side by
  // side), then the walk runs over those values in the same order and with the same additions as
  // a serial walk by lane L would
  L = __builtin_amdgcn_readfirstlane(L);
  const int cbL = L * cpl, ceL = min(cbL + cpl, NC);
  const int nL = ceL - cbL;                                 // 1 .. cpl chunks (uniform)
  const float inclL = __shfl(incl, L, 64);
  float localL = __shfl(local, L, 64);
  float run = 0.f;
  int cstar = ceL - 1;
  float resid = INFINITY;
  bool found = false;
  // (pass 0 re-sums lane L's chunk masses when only the table's prefix is known; pass 1 walks)
  for (int pass = table ? 0 : 1; pass < 2; ++pass) {
    if (pass == 1) run = inclL - localL;
    float acc = 0.f;
    for (int base = 0; base < nL; base += 64) {             // (nL <= 64 up to 512 x 512 maps)
      float ex = 0.f, wgt = 0.f;
      if (base + lane < nL) {
        ex = __expf(st[2 * (cbL + base + lane)] - M);
        wgt = st[2 * (cbL + base + lane) + 1] * ex;
      }
      const int m = min(64, nL - base);
      for (int i = 0; i < m; ++i) {
        const float wi = __shfl(wgt, i, 64);
        if (pass == 0) {
          acc += wi;
        } else {
          if (!found && run + wi > target) {
            cstar = cbL + base + i;
            resid = (target - run) / __shfl(ex, i, 64);     // relative to the chunk's own max
            found = true;
          }
          run += wi;
        }
      }
    }
    if (pass == 0) localL = acc;
  }
  const float mc = st[2 * cstar];

  // level 2: the 64 cells of the chunk.
  const int cell = cstar * SIM_CH + lane;
  const bool cvalid = cell < XY;
  float e = 0.f;
  if (cvalid && sim) {
    // the chunk's scores were written by the similarity kernel: 256 contiguous bytes of sim
    // (x = sim * num_valid, or sim / weight, to within one rounding) instead of re-evaluating 64
    // dot products from 8 KB of map features -- the sampler was bound by that L2 traffic
    // (10.8 GB per C2 step, 1.75 ms)
    e = __expf(sim[row * (int64_t)XY + cell] * row_unscale[row] - mc);
  } else if (cvalid) {
    const f32x4* mp = reinterpret_cast<const f32x4*>(fm + ((int64_t)b * XY + cell) * DM);
    const f32x4* qp = reinterpret_cast<const f32x4*>(fq + row * DM);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < DM / 4; ++k) {
      const f32x4 a = qp[k], c = mp[k];
      dot = fmaf(a[0], c[0], dot); dot = fmaf(a[1], c[1], dot); dot = fmaf(a[2], c[2], dot); dot = fmaf(a[3], c[3], dot);
    }
    float x = clip ? fmaxf(dot, 0.f) : dot;
    x *= scale;
    e = __expf(x - mc);
  }
  const float ci = wave_scan_incl(e, lane);
  unsigned long long b2 = __ballot(cvalid && ci > resid);
  int pick;
  if (b2) {
    pick = __ffsll((long long)b2) - 1;
  } else {
    const unsigned long long nv = __ballot(cvalid);
    pick = 63 - __clzll((long long)nv);
  }
  if (lane == 0) {
    const int c = cstar * SIM_CH + pick;
    int32_t* o = corr + ((int64_t)b * S + s) * 3;
    o[0] = n;
    o[1] = c / Y;
    o[2] = c - (c / Y) * Y;
  }
}

// The same draw for the common configuration (per-row prefix table + chunk scores read from sim),
// NQ correspondences per wave in lock step.  One correspondence is three DEPENDENT memory round
// trips (row table -> chunk statistics -> the chunk's 64 scores) and ~450 instructions: with one
// per wave the kernel ran at the pace of those round trips (32 waves per CU / ~5 us each = the
// measured 0.8 ms for 1.28 M correspondences at C2).  Here the NQ chains of a wave are issued
// phase by phase, so each round trip serves NQ correspondences.  Same arithmetic per
// correspondence, in the same order: identical samples.
template <int NQ>
__global__ __launch_bounds__(256) void ransac_sample_fast_kernel(
    const float* __restrict__ stats, int Nq, int X, int Y, int S, uint64_t seed,
    const float* __restrict__ uniforms, int32_t* __restrict__ corr,
    const float* __restrict__ lane_incl, const float* __restrict__ rowmax,
    const float* __restrict__ row_cdf, const float* __restrict__ sim,
    const float* __restrict__ row_unscale) {
  const int lane = threadIdx.x & 63;
  const int s0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * NQ);
  const int b = blockIdx.y;
  if (s0 >= S) return;
  const int XY = X * Y;
  const int NC = (XY + SIM_CH - 1) / SIM_CH;
  const int cpl = (NC + 63) / 64;                           // <= 64 (checked by the launcher)
  const int cb = lane * cpl, ce = min(cb + cpl, NC);
  bool live[NQ];
  int n[NQ];
  float u2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int s = s0 + q;
    live[q] = s < S;
    float u1;
    if (uniforms) {
      u1 = live[q] ? uniforms[((int64_t)b * S + s) * 2 + 0] : 0.f;
      u2[q] = live[q] ? uniforms[((int64_t)b * S + s) * 2 + 1] : 0.f;
    } else {
      uint32_t rnd[4];
      philox4x32_10((uint32_t)s, (uint32_t)b, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
      u1 = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
      u2[q] = (float)(rnd[1] >> 8) * (1.0f / 16777216.0f);
    }
    if (row_cdf) {
      const float* cdf = row_cdf + (int64_t)b * Nq;
      const float tgt = u1 * cdf[Nq - 1];
      int lo = 0, hi = Nq - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > tgt) hi = mid; else lo = mid + 1;
      }
      n[q] = lo;
    } else {
      n[q] = min((int)(u1 * (float)Nq), Nq - 1);
    }
    n[q] = __builtin_amdgcn_readfirstlane(n[q]);
  }
  // round trip 1: the row's maximum, its per-lane inclusive prefix, its un-scale factor
  float M[NQ], incl[NQ], ru[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int64_t row = (int64_t)b * Nq + n[q];
    M[q] = rowmax[row];
    incl[q] = lane_incl[row * 64 + lane];
    ru[q] = row_unscale[row];
  }
  // level 1: the lane whose chunk range holds the target; round trip 2: that range's statistics
  int cbL[NQ], nL[NQ];
  float target[NQ], inclL[NQ], sm[NQ], ss[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float total = __shfl(incl[q], 63, 64);
    target[q] = u2[q] * total;
    const unsigned long long bal = __ballot(incl[q] > target[q] && ce > cb);
    int L;
    if (bal) {
      L = __ffsll((long long)bal) - 1;
    } else {
      const unsigned long long ne = __ballot(ce > cb);
      L = 63 - __clzll((long long)ne);
    }
    L = __builtin_amdgcn_readfirstlane(L);
    cbL[q] = L * cpl;
    nL[q] = min(cbL[q] + cpl, NC) - cbL[q];
    inclL[q] = __shfl(incl[q], L, 64);
    const float* st = stats + ((int64_t)b * Nq + n[q]) * NC * 2;
    sm[q] = ss[q] = 0.f;
    if (lane < nL[q]) {
      sm[q] = st[2 * (cbL[q] + lane)];
      ss[q] = st[2 * (cbL[q] + lane) + 1];
    }
  }
  // the walk over lane L's chunks (masses side by side, additions in the serial order);
  // round trip 3: the selected chunk's 64 scores
  int cstar[NQ];
  float resid[NQ], mc[NQ], x[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float ex = lane < nL[q] ? __expf(sm[q] - M[q]) : 0.f;
    const float wgt = ss[q] * ex;
    float localL = 0.f;
    for (int i = 0; i < nL[q]; ++i) localL += __shfl(wgt, i, 64);
    float run = inclL[q] - localL;
    int cs = cbL[q] + nL[q] - 1;
    float rs = INFINITY;
    bool found = false;
    for (int i = 0; i < nL[q]; ++i) {
      const float wi = __shfl(wgt, i, 64);
      if (!found && run + wi > target[q]) {
        cs = cbL[q] + i;
        rs = (target[q] - run) / __shfl(ex, i, 64);
        found = true;
      }
      run += wi;
    }
    cstar[q] = cs;
    resid[q] = rs;
    mc[q] = __shfl(sm[q], cs - cbL[q], 64);
    const int cell = cs * SIM_CH + lane;
    x[q] = (cell < XY && live[q]) ? sim[((int64_t)b * Nq + n[q]) * XY + cell] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int cell = cstar[q] * SIM_CH + lane;
    const bool cvalid = cell < XY;
    const float e = cvalid ? __expf(x[q] * ru[q] - mc[q]) : 0.f;
    const float ci = wave_scan_incl(e, lane);
    const unsigned long long b2 = __ballot(cvalid && ci > resid[q]);
    int pick;
    if (b2) {
      pick = __ffsll((long long)b2) - 1;
    } else {
      const unsigned long long nv = __ballot(cvalid);
      pick = 63 - __clzll((long long)nv);
    }
    if (lane == 0 && live[q]) {
      const int c = cstar[q] * SIM_CH + pick;
      int32_t* o = corr + ((int64_t)b * S + s0 + q) * 3;
      o[0] = n[q];
      o[1] = c / Y;
      o[2] = c - (c / Y) * Y;
    }
  }
}

// ---------------------------------------------------------------------------
// k11b: best-of-retries + 2-point Kabsch (closed form of the 2x2 SVD).
// ---------------------------------------------------------------------------
__global__ void poses_from_corr_kernel(const int32_t* __restrict__ corr,
                                       const float* __restrict__ q_xy, int B, int Nq, int P,
                                       int retries, float cell, float* __restrict__ poses) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * P) return;
  const int b = (int)(i / P), p = (int)(i - (int64_t)b * P);
  const int32_t* cb = corr + ((int64_t)b * P * retries * 2) * 3;
  float best = INFINITY;
  float bi0x = 0, bi0y = 0, bi1x = 0, bi1y = 0, bj0x = 0, bj0y = 0, bj1x = 0, bj1y = 0;
  for (int r = 0; r < retries; ++r) {
    const int32_t* c0 = cb + ((int64_t)(p * retries + r) * 2 + 0) * 3;
    const int32_t* c1 = c0 + 3;
    const float i0x = q_xy[((int64_t)b * Nq + c0[0]) * 2 + 0], i0y = q_xy[((int64_t)b * Nq + c0[0]) * 2 + 1];
    const float i1x = q_xy[((int64_t)b * Nq + c1[0]) * 2 + 0], i1y = q_xy[((int64_t)b * Nq + c1[0]) * 2 + 1];
    const float j0x = ((float)c0[1] + 0.5f) * cell, j0y = ((float)c0[2] + 0.5f) * cell;
    const float j1x = ((float)c1[1] + 0.5f) * cell, j1y = ((float)c1[2] + 0.5f) * cell;
    float ratio = 0.f;
    if (retries > 1) {
      const float dix = i1x - i0x, diy = i1y - i0y, djx = j1x - j0x, djy = j1y - j0y;
      const float d_i = sqrtf(dix * dix + diy * diy), d_j = sqrtf(djx * djx + djy * djy);
      ratio = fmaxf(d_i / fmaxf(d_j, 1e-5f), d_j / fmaxf(d_i, 1e-5f));
    }
    if (r == 0 || ratio < best) {
      best = ratio;
      bi0x = i0x; bi0y = i0y; bi1x = i1x; bi1y = i1y;
      bj0x = j0x; bj0y = j0y; bj1x = j1x; bj1y = j1y;
    }
  }
  // kabsch_algorithm_2d(j_xy (map), i_xy (query)) -> map_t_query.
  const float mux = (bj0x + bj1x) * 0.5f, muy = (bj0y + bj1y) * 0.5f;   // map mean
  const float nux = (bi0x + bi1x) * 0.5f, nuy = (bi0y + bi1y) * 0.5f;   // query mean
  const float ax = bj1x - bj0x, ay = bj1y - bj0y;                       // map diff
  const float qx = bi1x - bi0x, qy = bi1y - bi0y;                       // query diff
  const float dot = ax * qx + ay * qy;
  const float crs = qx * ay - qy * ax;
  const float nrm = sqrtf(dot * dot + crs * crs);
  float c = 1.f, s = 0.f;
  if (nrm > 0.f) { c = dot / nrm; s = crs / nrm; }
  const float tx = mux - (c * nux - s * nuy);
  const float ty = muy - (s * nux + c * nuy);
  float* o = poses + i * 3;
  o[0] = atan2f(s, c);
  o[1] = tx;
  o[2] = ty;
}

// ---------------------------------------------------------------------------
// k12: pose scoring.  grid = (point chunks, pose chunks, B), block = 1024.
// The score plane sim[b, n] (X*Y fp32, or a row band of it) is staged in LDS once
// and gathered from there by every pose; each thread keeps PPT poses (cos, sin,
// t) and their accumulators in registers.  Compulsory HBM traffic: sim read once
// per pose chunk.  Partial sums per point chunk are reduced in fixed order
// (deterministic).
// ---------------------------------------------------------------------------
constexpr int PS_THREADS = 1024;
constexpr int PS_LDS_FLOATS = 24 * 1024;  // 96 KiB plane / band buffer

struct ScoreArgs {
  const float* sim;
  const float* poses;
  const float* q_xy;
  const uint8_t* valid_q;
  const uint8_t* map_valid;
  int B, Nq, X, Y, P;
  float cell;
  int mask_oob;
  int points_per_chunk;
  int RB, NB;      // rows per band, number of bands
  float* partial;  // [B, NCH, P]
  const float* table;  // [B, P, 4] = (cos, sin, tx, ty)
};

__global__ void pose_table_kernel(const float* __restrict__ poses, int64_t total,
                                  float* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float th = poses[i * 3 + 0];
  f32x4 t;
  t[0] = cosf(th); t[1] = sinf(th); t[2] = poses[i * 3 + 1]; t[3] = poses[i * 3 + 2];
  reinterpret_cast<f32x4*>(table)[i] = t;
}

// Same poses as affine maps in CELL units, half-cell shift folded in:
//   cu = A*qx - B*qy + Cx,  cv = B*qx + A*qy + Cy   with A = cos/cell, B = sin/cell,
//   Cx = tx/cell - 0.5, Cy = ty/cell - 0.5.
// (Transform2D.transform followed by "/ cell_size" and the "- 0.5" of interpolate_nd,
// pose_estimation.py:74 + grids.py:129, re-associated: <= 1 ulp on the coordinate.)
__global__ void pose_table_cells_kernel(const float* __restrict__ poses, int64_t total,
                                        float cell, float* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float th = poses[i * 3 + 0];
  f32x4 t;
  t[0] = cosf(th) / cell; t[1] = sinf(th) / cell;
  t[2] = poses[i * 3 + 1] / cell - 0.5f; t[3] = poses[i * 3 + 2] / cell - 0.5f;
  reinterpret_cast<f32x4*>(table)[i] = t;
}

template <int PPT, bool MASK, bool BANDS>
__global__ __launch_bounds__(PS_THREADS) void pose_score_kernel(const ScoreArgs a) {
  extern __shared__ float plane[];
  const int b = blockIdx.z;
  // pose chunk fastest: workgroups sharing the same score planes are dispatched
  // back to back and hit L2 / Infinity Cache on the re-read.
  const int chunk = blockIdx.y;
  const int NCH = gridDim.y;
  const int p_base = blockIdx.x * (PS_THREADS * PPT);
  const int tid = threadIdx.x;
  float pc[PPT], ps[PPT], ptx[PPT], pty[PPT], acc[PPT];
  bool pok[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    pok[k] = p < a.P;
    acc[k] = 0.f;
    // out-of-range slots score pose P-1 again; only the final store is guarded.
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + min(p, a.P - 1)];
    pc[k] = t[0]; ps[k] = t[1]; ptx[k] = t[2]; pty[k] = t[3];
  }
  const int n_begin = chunk * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  const float Xf = (float)a.X, Yf = (float)a.Y;
  const uint8_t* mvalid = a.map_valid ? a.map_valid + (int64_t)b * a.X * a.Y : nullptr;
  for (int n = n_begin; n < n_end; ++n) {
    if (!a.valid_q[(int64_t)b * a.Nq + n]) continue;  // block-uniform
    const float qx = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
    const float qy = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
    const float* src = a.sim + ((int64_t)b * a.Nq + n) * a.X * a.Y;
    const int nbands = BANDS ? a.NB : 1;
    for (int band = 0; band < nbands; ++band) {
      const int r0 = band * (a.RB - 1);
      const int nrows = min(a.RB, a.X - r0);
      const int nfl = nrows * a.Y;
      __syncthreads();  // previous consumers done
      {
        const float* s0 = src + (int64_t)r0 * a.Y;
        if (((a.Y & 3) == 0)) {
          const f32x4* s4 = reinterpret_cast<const f32x4*>(s0);
          f32x4* d4 = reinterpret_cast<f32x4*>(plane);
          for (int i = tid; i < (nfl >> 2); i += PS_THREADS) d4[i] = s4[i];
        } else {
          for (int i = tid; i < nfl; i += PS_THREADS) plane[i] = s0[i];
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        // (j_t_i @ xy) / cell_size  -- Transform2D.transform, geometry.py:137-139
        const float xm = (pc[k] * qx - ps[k] * qy) + ptx[k];
        const float ym = (ps[k] * qx + pc[k] * qy) + pty[k];
        const float u = xm / a.cell, v = ym / a.cell;
        const float cu = u - 0.5f, cv = v - 0.5f;
        const float fu = floorf(cu), fv = floorf(cv);
        const int i0 = (int)fminf(fmaxf(fu, 0.f), Xf - 1.f);
        bool mine = true;
        if (BANDS) mine = min(i0 / (a.RB - 1), a.NB - 1) == band;
        const int i1 = (int)fminf(fmaxf(fu + 1.f, 0.f), Xf - 1.f);
        const int j0 = (int)fminf(fmaxf(fv, 0.f), Yf - 1.f);
        const int j1 = (int)fminf(fmaxf(fv + 1.f, 0.f), Yf - 1.f);
        const float wu1 = cu - fu, wu0 = 1.f - wu1;
        const float wv1 = cv - fv, wv0 = 1.f - wv1;
        const int l0 = mine ? (i0 - r0) * a.Y : 0, l1 = mine ? (i1 - r0) * a.Y : 0;
        const float s00 = plane[l0 + j0], s01 = plane[l0 + j1];
        const float s10 = plane[l1 + j0], s11 = plane[l1 + j1];
        const float val =
            (((wu0 * wv0) * s00 + (wu0 * wv1) * s01) + (wu1 * wv0) * s10) + (wu1 * wv1) * s11;
        bool ok = mine;
        if (MASK) {
          ok = ok && (u >= 0.f) && (u < Xf) && (v >= 0.f) && (v < Yf);
          ok = ok && mvalid[i0 * a.Y + j0] && mvalid[i0 * a.Y + j1] && mvalid[i1 * a.Y + j0] &&
               mvalid[i1 * a.Y + j1];
        }
        acc[k] += ok ? val : 0.f;
        // keep the PPT bodies sequential: interleaving them spills (16 live
        // bilinear contexts); 4 waves/SIMD already hide the LDS latency.
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    if (pok[k]) a.partial[((int64_t)b * NCH + chunk) * a.P + p] = acc[k];
  }
}

// Double-buffered variant for planes that fit twice in LDS (X*Y*4 <= 64 KiB, the
// 128x128 map): the plane of the NEXT valid query point is streamed HBM -> LDS by
// the LDS-DMA path (global_load_lds_dwordx4: no VGPR round trip, wave-uniform LDS
// base + lane*16, i.e. a linear copy) while every thread gathers from the current
// one; one barrier per point.  This keeps the HBM stream busy during the gather
// phase -- the kernel's roofline is the single read of sim.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

__device__ __forceinline__ float uniform_f(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
constexpr int PS_DB_MAX_POINTS = 1024;  // valid points of one chunk staged in LDS
constexpr int PS_DB_ROUNDS = 5;           // 16-byte DMA chunks per thread per plane (max)
constexpr int PS_DB_PLANE_BYTES = PS_DB_ROUNDS * PS_THREADS * 16 - 12 * 1024;  // 68 KiB padded plane

// YC: compile-time row length (0 = run time).  The 2x2 footprint is two ds_read2_b32 off
// ONE address (offsets {0,1} and {YC,YC+1}); each lands as a (col j, col j+1) register
// pair, so the coordinate and row-lerp arithmetic is packed f32 (v_pk_fma/v_pk_add).
template <int PPT, bool MASK, int YC>
__global__ __launch_bounds__(PS_THREADS) void pose_score_db_kernel(const ScoreArgs a) {
  extern __shared__ float plane[];
  const int b = blockIdx.z;
  const int chunk = blockIdx.y;
  const int NCH = gridDim.y;
  const int p_base = blockIdx.x * (PS_THREADS * PPT);
  const int tid = threadIdx.x;
  const int Y = YC ? YC : a.Y;
  const int XY = a.X * Y;
  // LDS rows are padded by one 16-byte chunk (stride S = Y + 4 floats): bank(i, j) =
  // (4 i + j) mod 32, so samples clamped to the first / last COLUMN (out-of-map poses:
  // a large share of RANSAC hypotheses) spread over 8 banks instead of all hitting one.
  // The DMA writes LDS contiguously but each lane may fetch any global chunk, so the pad
  // costs one duplicate chunk per row and no extra instructions.
  const int S = Y + 4;
  const int CRP = (Y >> 2) + 1;            // chunks per padded row
  const int nchunks = a.X * CRP;
  const int PLANE = a.X * S;               // floats per LDS plane
  f32x2 pcs[PPT], pt[PPT];
  float acc[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    acc[k] = 0.f;
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + min(p, a.P - 1)];
    pcs[k] = f32x2{t[0], t[1]};
    pt[k] = f32x2{t[2], t[3]};
  }
  const int n_begin = chunk * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  const float Xf = (float)a.X, Yf = (float)Y;
  const float Sf = (float)S;
  const f32x2 lim1 = {Xf - 1.f, Yf - 1.f}, lim2 = {Xf - 2.f, Yf - 2.f}, zero2 = {0.f, 0.f};
  const uint8_t* vq = a.valid_q + (int64_t)b * a.Nq;
  const uint8_t* mvalid = a.map_valid ? a.map_valid + (int64_t)b * XY : nullptr;

  // global float offset of the chunk this thread fetches in DMA round k (plane-invariant)
  int goff[PS_DB_ROUNDS];
#pragma unroll
  for (int k = 0; k < PS_DB_ROUNDS; ++k) {
    const int c = min(k * PS_THREADS + tid, nchunks - 1);
    const int row = c / CRP;
    goff[k] = row * Y + 4 * min(c - row * CRP, CRP - 2);
  }
  auto issue = [&](int n, int buf) {
    const float* src = a.sim + ((int64_t)b * a.Nq + n) * XY;
    float* dst = plane + buf * PLANE;
#pragma unroll
    for (int k = 0; k < PS_DB_ROUNDS; ++k) {
      const int c = k * PS_THREADS + tid;
      if (c < nchunks)
        __builtin_amdgcn_global_load_lds((global_void_t*)(src + goff[k]),
                                         (lds_void_t*)(dst + 4 * c), 16, 0, 0);
    }
  };
  // The chunk's valid points, compacted (ascending) into LDS up front: the plane loop
  // then touches global memory ONLY through the LDS-DMA stream, so the vmcnt(0) at the
  // top of an iteration waits for exactly the plane it is about to read and the next
  // plane's DMA flies under the arithmetic (any other global load inside the loop would
  // sit behind the in-order vmcnt and serialise load and compute).
  __shared__ int pt_n[PS_DB_MAX_POINTS];
  __shared__ float pt_x[PS_DB_MAX_POINTS], pt_y[PS_DB_MAX_POINTS];
  __shared__ int pt_count;
  if (tid < 64) {
    int count = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
      const int n = n0 + tid;
      const bool v = n < n_end && vq[n] != 0;
      const unsigned long long m = __ballot(v);
      if (v) {
        const int slot = count + __popcll(m & ((1ull << tid) - 1ull));
        pt_n[slot] = n;
        pt_x[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
        pt_y[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
      }
      count += __popcll(m);
    }
    if (tid == 0) pt_count = count;
  }
  __syncthreads();
  const int count = pt_count;
  int buf = 0;
  if (count > 0) issue(pt_n[0], 0);

  // One sample = coords() (pure VALU, needs only the pose and the point) + gather()
  // (two ds_read2_b32 + lerp).  The loop is software-pipelined across the barrier: the
  // coordinates of the first G poses for plane i+1 are computed at the end of iteration
  // i, so every wave fires LDS reads right after the barrier instead of all 16 waves
  // doing address arithmetic while the LDS sits idle.
  struct Coord { f32x2 w; int off; f32x2 r; };
  auto coords = [&](int k, float qx, float qy) {
    // 'nearest' extension == sampling at the clamped coordinate; the cell is capped at
    // X-2 (weight 1 there) so the 2x2 footprint always lies inside the plane.
    const f32x2 qx2 = {qx, qx}, qyn = {-qy, qy};
    Coord o;
    o.r = __builtin_elementwise_fma(pcs[k], qx2, __builtin_elementwise_fma(pcs[k].yx, qyn, pt[k]));
    const f32x2 c = __builtin_elementwise_min(__builtin_elementwise_max(o.r, zero2), lim1);
    const f32x2 f = __builtin_elementwise_min(f32x2{floorf(c.x), floorf(c.y)}, lim2);
    o.w = c - f;
    o.off = (int)fmaf(f.x, Sf, f.y);
    return o;
  };
  struct Taps { f32x2 s0, s1; };
  auto fetch = [&](const Coord& o, const float* pl) {
    const float* q = pl + o.off;
    Taps t;
    t.s0 = f32x2{q[0], q[1]};  // one ds_read2_b32 each: (row i, cols j,j+1)
    t.s1 = f32x2{q[S], q[S + 1]};
    return t;
  };
  auto blend = [&](int k, const Coord& o, const Taps& tp) {
    const f32x2 t = __builtin_elementwise_fma(f32x2{o.w.x, o.w.x}, tp.s1 - tp.s0, tp.s0);
    const float val = fmaf(o.w.y, t.y - t.x, t.x);
    bool ok = true;
    if (MASK) {
      const float u = o.r.x + 0.5f, v = o.r.y + 0.5f;
      ok = (u >= 0.f) && (u < Xf) && (v >= 0.f) && (v < Yf);
      const float gu = floorf(o.r.x), gv = floorf(o.r.y);
      const int i0 = (int)fminf(fmaxf(gu, 0.f), Xf - 1.f);
      const int i1 = (int)fminf(fmaxf(gu + 1.f, 0.f), Xf - 1.f);
      const int j0 = (int)fminf(fmaxf(gv, 0.f), Yf - 1.f);
      const int j1 = (int)fminf(fmaxf(gv + 1.f, 0.f), Yf - 1.f);
      ok = ok && mvalid[i0 * Y + j0] && mvalid[i0 * Y + j1] && mvalid[i1 * Y + j0] &&
           mvalid[i1 * Y + j1];
    }
    acc[k] += ok ? val : 0.f;
  };
  constexpr int G = PPT / 2;
  Coord pre[G];
  {
    const float qx = count > 0 ? pt_x[0] : 0.f, qy = count > 0 ? pt_y[0] : 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) pre[k] = coords(k, qx, qy);
  }
  for (int i = 0; i < count; ++i) {
    const int inext = min(i + 1, count - 1);
    const float qx = uniform_f(pt_x[i]), qy = uniform_f(pt_y[i]);  // wave-uniform: keep in SGPRs
    const float nqx = uniform_f(pt_x[inext]), nqy = uniform_f(pt_y[inext]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // plane i landed for every wave; the other buffer is free
    if (i + 1 < count) issue(pt_n[i + 1], buf ^ 1);
    const float* pl = plane + buf * PLANE;
    // group 0 reads go out back to back (addresses were ready before the barrier); group
    // 1's coordinates are computed under their latency.  Same again for group 1 with the
    // next plane's group-0 coordinates as the cover.
    Taps tp[G];
    Coord c1[PPT - G];
#pragma unroll
    for (int k = 0; k < G; ++k) tp[k] = fetch(pre[k], pl);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = G; k < PPT; ++k) c1[k - G] = coords(k, qx, qy);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < G; ++k) blend(k, pre[k], tp[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = G; k < PPT; ++k) tp[k - G] = fetch(c1[k - G], pl);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < G; ++k) pre[k] = coords(k, nqx, nqy);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = G; k < PPT; ++k) blend(k, c1[k - G], tp[k - G]);
    buf ^= 1;
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    if (p < a.P) a.partial[((int64_t)b * NCH + chunk) * a.P + p] = acc[k];
  }
}

// Banded variant for planes that do not fit LDS (256x256 maps of the eval path, BASELINE
// configs[3]): the plane of a point streams through LDS in NB bands of RB rows (+1 halo row)
// with the same LDS-DMA double buffering, padded rows and LDS point list as the whole-plane
// kernel; the clamped coordinates / weights of the thread's 10 poses are computed once per
// point and kept in registers across its bands, every band then costs a membership test, two
// ds_read2_b32 and the lerp.  No validity mask (MASK launches take the older band kernel).
template <int PPT>
__global__ __launch_bounds__(PS_THREADS) void pose_score_band_db_kernel(const ScoreArgs a) {
  extern __shared__ float plane[];
  const int b = blockIdx.z;
  const int chunk = blockIdx.y;
  const int NCH = gridDim.y;
  const int p_base = blockIdx.x * (PS_THREADS * PPT);
  const int tid = threadIdx.x;
  const int Y = a.Y;
  const int XY = a.X * Y;
  const int S = Y + 4;
  const int CRP = (Y >> 2) + 1;
  const int RB = a.RB, NB = a.NB;          // band rows (without the halo row), band count
  const int PLANE = (RB + 1) * S;          // floats per LDS band buffer
  f32x2 pcs[PPT], pt[PPT];
  float acc[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    acc[k] = 0.f;
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + min(p, a.P - 1)];
    pcs[k] = f32x2{t[0], t[1]};
    pt[k] = f32x2{t[2], t[3]};
  }
  const int n_begin = chunk * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  const float Xf = (float)a.X, Yf = (float)Y, Sf = (float)S;
  const f32x2 lim1 = {Xf - 1.f, Yf - 1.f}, lim2 = {Xf - 2.f, Yf - 2.f}, zero2 = {0.f, 0.f};
  const uint8_t* vq = a.valid_q + (int64_t)b * a.Nq;

  // chunk -> global float offset inside a band (band-invariant; rows past the band end are
  // masked by the chunk count of the band)
  int goff[PS_DB_ROUNDS];
#pragma unroll
  for (int k = 0; k < PS_DB_ROUNDS; ++k) {
    const int c = k * PS_THREADS + tid;
    const int row = c / CRP;
    goff[k] = row * Y + 4 * min(c - row * CRP, CRP - 2);
  }
  auto issue = [&](int n, int band, int buf) {
    const int r0 = band * RB;
    const int rows = min(RB + 1, a.X - r0);
    const int nchunks = rows * CRP;
    const float* src = a.sim + ((int64_t)b * a.Nq + n) * XY + (int64_t)r0 * Y;
    float* dst = plane + buf * PLANE;
#pragma unroll
    for (int k = 0; k < PS_DB_ROUNDS; ++k) {
      const int c = k * PS_THREADS + tid;
      if (c < nchunks)
        __builtin_amdgcn_global_load_lds((global_void_t*)(src + goff[k]),
                                         (lds_void_t*)(dst + 4 * c), 16, 0, 0);
    }
  };
  __shared__ int pt_n[PS_DB_MAX_POINTS];
  __shared__ float pt_x[PS_DB_MAX_POINTS], pt_y[PS_DB_MAX_POINTS];
  __shared__ int pt_count;
  if (tid < 64) {
    int count = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
      const int n = n0 + tid;
      const bool v = n < n_end && vq[n] != 0;
      const unsigned long long m = __ballot(v);
      if (v) {
        const int slot = count + __popcll(m & ((1ull << tid) - 1ull));
        pt_n[slot] = n;
        pt_x[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
        pt_y[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
      }
      count += __popcll(m);
    }
    if (tid == 0) pt_count = count;
  }
  __syncthreads();
  const int count = pt_count;
  const int steps = count * NB;            // (point, band) pairs, band fastest
  int buf = 0;
  if (steps > 0) issue(pt_n[0], 0, 0);
  float wu[PPT], wv[PPT];
  int off[PPT];   // cell offset in full-plane padded rows; its row = off / S (fv < Y < S)
  int i = 0, band = 0;
  for (int t = 0; t < steps; ++t) {
    if (band == 0) {
      // per-point state of the 10 poses: cell row, lerp weights, offset in full-plane rows
      const float qx = uniform_f(pt_x[i]), qy = uniform_f(pt_y[i]);
      const f32x2 qx2 = {qx, qx}, qyn = {-qy, qy};
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const f32x2 r = __builtin_elementwise_fma(pcs[k], qx2, __builtin_elementwise_fma(pcs[k].yx, qyn, pt[k]));
        const f32x2 c = __builtin_elementwise_min(__builtin_elementwise_max(r, zero2), lim1);
        const f32x2 f = __builtin_elementwise_min(f32x2{floorf(c.x), floorf(c.y)}, lim2);
        wu[k] = c.x - f.x;
        wv[k] = c.y - f.y;
        off[k] = (int)fmaf(f.x, Sf, f.y);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // band t landed for every wave; the other buffer is free
    {
      int ni = i, nb = band + 1;
      if (nb == NB) { nb = 0; ++ni; }
      if (t + 1 < steps) issue(pt_n[ni], nb, buf ^ 1);
    }
    const float* pl = plane + buf * PLANE;
    const int shift = band * RB * S;         // band rows [band*RB, band*RB + RB)
    const int span = RB * S;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const bool in = (unsigned)(off[k] - shift) < (unsigned)span;
      const float* q = pl + (in ? off[k] - shift : 0);
      const f32x2 s0 = {q[0], q[1]};
      const f32x2 s1 = {q[S], q[S + 1]};
      const f32x2 tt = __builtin_elementwise_fma(f32x2{wu[k], wu[k]}, s1 - s0, s0);
      const float val = fmaf(wv[k], tt.y - tt.x, tt.x);
      acc[k] += in ? val : 0.f;
    }
    buf ^= 1;
    if (++band == NB) { band = 0; ++i; }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    if (p < a.P) a.partial[((int64_t)b * NCH + chunk) * a.P + p] = acc[k];
  }
}

// Windowed variant for pose sets whose samples of a point all fall within `rad` cells of ONE centre
// pose per scene -- the 41 x 41 x 41 refinement lattice of grid_refinement (pose_estimation.py:168-205:
// +-4 m, +-5 degrees around the RANSAC pose).  The general kernels above stream the WHOLE score plane of
// every point through LDS once per pose chunk (256 x 256 maps: 262 KB per point and chunk, nine chunks
// for 68 921 poses = 11 GB per scene); here a point's plane is read as ONE window of <= (2 rad + 3) rows x
// (2 rad + 6) columns around the centre pose's image of the point (~80 x 84 cells = 27 KB): 0.14 GB per
// pose chunk.  Arithmetic per sample (coordinates from the same cell-unit pose table, clamp, floor, the
// two lerps) and the order of the sums (points ascending inside the same point chunks, chunks ascending in
// the reduce pass) are those of pose_score_band_db_kernel: the scores are the same bits.  No validity
// mask (MASK launches take the general kernels).
struct ScoreWinArgs {
  ScoreArgs s;
  const float* ctable;   // [B, 4] the centre poses as cell-unit affine maps (pose_table_cells_kernel)
  int rad;               // every sample of a point lies within `rad` cells of the centre's
  int WR, WC;            // window rows / columns (WC % 4 == 0)
};
constexpr int PS_WIN_ROUNDS = 3;          // 16-byte DMA chunks per thread and window (max): 48 KB per buffer
constexpr int PS_WIN_PPT = 12;

template <int PPT>
__global__ __launch_bounds__(PS_THREADS) void pose_score_window_kernel(const ScoreWinArgs w) {
  extern __shared__ float plane[];
  const ScoreArgs& a = w.s;
  const int b = blockIdx.z;
  const int chunk = blockIdx.y;
  const int NCH = gridDim.y;
  const int p_base = blockIdx.x * (PS_THREADS * PPT);
  const int tid = threadIdx.x;
  const int X = a.X, Y = a.Y;
  const int XY = X * Y;
  const int WR = w.WR, WC = w.WC, rad = w.rad;
  const int S = WC + 4;                    // LDS row pitch (floats): the window row + one pad chunk
  const int CRP = (WC >> 2) + 1;           // 16-byte chunks per LDS row
  const int PLANE = WR * S;
  f32x2 pcs[PPT], pt[PPT];
  float acc[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    acc[k] = 0.f;
    const f32x4 t = reinterpret_cast<const f32x4*>(a.table)[(int64_t)b * a.P + min(p, a.P - 1)];
    pcs[k] = f32x2{t[0], t[1]};
    pt[k] = f32x2{t[2], t[3]};
  }
  const f32x4 ct = reinterpret_cast<const f32x4*>(w.ctable)[b];
  const int n_begin = chunk * a.points_per_chunk;
  const int n_end = min(n_begin + a.points_per_chunk, a.Nq);
  const float Xf = (float)X, Yf = (float)Y, Sf = (float)S;
  const f32x2 lim1 = {Xf - 1.f, Yf - 1.f}, lim2 = {Xf - 2.f, Yf - 2.f}, zero2 = {0.f, 0.f};
  const uint8_t* vq = a.valid_q + (int64_t)b * a.Nq;

  // chunk -> float offset relative to the window's first cell (window-invariant)
  int goff[PS_WIN_ROUNDS];
#pragma unroll
  for (int k = 0; k < PS_WIN_ROUNDS; ++k) {
    const int c = k * PS_THREADS + tid;
    const int row = c / CRP;
    goff[k] = row * Y + 4 * min(c - row * CRP, CRP - 2);
  }
  const int nchunks = WR * CRP;
  // first cell (row0, col0) of the window of point (qx, qy): the centre pose's cell minus the radius,
  // one more for the floor, kept inside the plane; col0 % 4 == 0 (16-byte DMA chunks)
  auto origin = [&](float qx, float qy) {
    const float cu = fmaf(ct[0], qx, fmaf(ct[1], -qy, ct[2]));
    const float cv = fmaf(ct[1], qx, fmaf(ct[0], qy, ct[3]));
    const float ru = fminf(fmaxf(floorf(cu) - (float)(rad + 1), 0.f), (float)(X - WR));
    const float rv = fminf(fmaxf(floorf(cv) - (float)(rad + 1), 0.f), (float)(Y - WC));
    return make_int2((int)ru, ((int)rv) & ~3);
  };
  auto issue = [&](int n, int2 o, int buf) {
    const float* src = a.sim + ((int64_t)b * a.Nq + n) * XY + (int64_t)o.x * Y + o.y;
    float* dst = plane + buf * PLANE;
#pragma unroll
    for (int k = 0; k < PS_WIN_ROUNDS; ++k) {
      const int c = k * PS_THREADS + tid;
      if (c < nchunks)
        __builtin_amdgcn_global_load_lds((global_void_t*)(src + goff[k]),
                                         (lds_void_t*)(dst + 4 * c), 16, 0, 0);
    }
  };
  __shared__ int pt_n[PS_DB_MAX_POINTS];
  __shared__ float pt_x[PS_DB_MAX_POINTS], pt_y[PS_DB_MAX_POINTS];
  __shared__ int pt_count;
  if (tid < 64) {
    int count = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
      const int n = n0 + tid;
      const bool v = n < n_end && vq[n] != 0;
      const unsigned long long m = __ballot(v);
      if (v) {
        const int slot = count + __popcll(m & ((1ull << tid) - 1ull));
        pt_n[slot] = n;
        pt_x[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 0];
        pt_y[slot] = a.q_xy[((int64_t)b * a.Nq + n) * 2 + 1];
      }
      count += __popcll(m);
    }
    if (tid == 0) pt_count = count;
  }
  __syncthreads();
  const int count = pt_count;
  int buf = 0;
  int2 o_cur = make_int2(0, 0);
  if (count > 0) {
    o_cur = origin(uniform_f(pt_x[0]), uniform_f(pt_y[0]));
    issue(pt_n[0], o_cur, 0);
  }
  const int off_max = PLANE - S - 2;
  for (int i = 0; i < count; ++i) {
    const float qx = uniform_f(pt_x[i]), qy = uniform_f(pt_y[i]);
    const f32x2 qx2 = {qx, qx}, qyn = {-qy, qy};
    const int obase = o_cur.x * S + o_cur.y;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // window i landed for every wave; the other buffer is free
    if (i + 1 < count) {
      o_cur = origin(uniform_f(pt_x[i + 1]), uniform_f(pt_y[i + 1]));
      issue(pt_n[i + 1], o_cur, buf ^ 1);
    }
    const float* pl = plane + buf * PLANE;
    // four poses at a time: coordinates, the eight LDS reads, the lerps (per-pose state stays in 5 registers)
#pragma unroll
    for (int g = 0; g < PPT; g += 4) {
      float wu[4], wv[4];
      const float* q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = g + j;
        const f32x2 r = __builtin_elementwise_fma(pcs[k], qx2, __builtin_elementwise_fma(pcs[k].yx, qyn, pt[k]));
        const f32x2 c = __builtin_elementwise_min(__builtin_elementwise_max(r, zero2), lim1);
        const f32x2 f = __builtin_elementwise_min(f32x2{floorf(c.x), floorf(c.y)}, lim2);
        wu[j] = c.x - f.x;
        wv[j] = c.y - f.y;
        // (a pose outside the promised radius reads a clamped cell of the window: wrong, never out of bounds)
        q[j] = pl + min(max((int)fmaf(f.x, Sf, f.y) - obase, 0), off_max);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 s0 = {q[j][0], q[j][1]};
        const f32x2 s1 = {q[j][S], q[j][S + 1]};
        const f32x2 tt = __builtin_elementwise_fma(f32x2{wu[j], wu[j]}, s1 - s0, s0);
        acc[g + j] += fmaf(wv[j], tt.y - tt.x, tt.x);
      }
    }
    buf ^= 1;
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = p_base + k * PS_THREADS + tid;
    if (p < a.P) a.partial[((int64_t)b * NCH + chunk) * a.P + p] = acc[k];
  }
}

__global__ void pose_score_reduce_kernel(const float* __restrict__ partial, int NCH, int P,
                                         int64_t total, float* __restrict__ scores) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*P
  if (i >= total) return;
  const int64_t b = i / P, p = i - b * P;
  float t = 0.f;
#pragma unroll 8
  for (int c = 0; c < NCH; ++c) t += partial[(b * NCH + c) * P + p];
  scores[i] = t;
}

inline int score_chunks(int B, int Nq, int pose_chunks) {
  // aim for >= 512 workgroups overall, at least 8 points per chunk.
  int nch = (512 + B * pose_chunks - 1) / (B * pose_chunks);
  nch = nch < 1 ? 1 : nch;
  const int maxch = (Nq + 7) / 8;
  if (nch > maxch) nch = maxch;
  const int minch = (Nq + PS_DB_MAX_POINTS - 1) / PS_DB_MAX_POINTS;  // LDS point list of the db kernel
  if (nch < minch) nch = minch;
  if (nch < 1) nch = 1;
  return nch;
}
constexpr int PS_PPT = 10;
constexpr int PS_BAND_PPT = 10;  // the banded kernel keeps per-pose sampling state in registers (128 VGPRs at 10: 20 001 hypotheses = two pose chunks, not three)
inline int score_pose_chunks(int P) { return (P + PS_THREADS * PS_PPT - 1) / (PS_THREADS * PS_PPT); }

__global__ void refine_lattice_kernel(const float* __restrict__ init,
                                      const float* __restrict__ offs_r,
                                      const float* __restrict__ offs_p, int B, int nr, int np_,
                                      float* __restrict__ out) {
  const int64_t per = (int64_t)nr * np_ * np_;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * B) return;
  const int b = (int)(i / per);
  int64_t r = i - (int64_t)b * per;
  const int iy = (int)(r % np_); r /= np_;
  const int ix = (int)(r % np_);
  const int ir = (int)(r / np_);
  const float th0 = init[b * 3 + 0], tx0 = init[b * 3 + 1], ty0 = init[b * 3 + 2];
  const float c = cosf(th0), s = sinf(th0);
  const float ox = offs_p[ix], oy = offs_p[iy];
  // compose: angle = a0 + a1; t = t0 + R(a0) t1   (geometry.py:141-144)
  out[i * 3 + 0] = th0 + offs_r[ir];
  out[i * 3 + 1] = tx0 + (c * ox - s * oy);
  out[i * 3 + 2] = ty0 + (s * ox + c * oy);
}

// NT threads per row: long rows (the 68 921-pose refinement lattice with one scene per GPU) take 1024 -- one
// workgroup is all the parallelism a row has here, and the scan is bound by its loads in flight
template <int NT>
__global__ __launch_bounds__(NT) void argmax_rows_kernel(const float* __restrict__ scores, int P,
                                                         int start, int32_t* __restrict__ idx) {
  __shared__ float bv[NT];
  __shared__ int bi[NT];
  const int b = blockIdx.x;
  const float* row = scores + (int64_t)b * P;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int p = start + threadIdx.x; p < P; p += NT) {
    const float v = row[p];
    if (v > best || besti == 0x7fffffff) { best = v; besti = p; }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = besti;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float ov = bv[threadIdx.x + o];
      const int oi = bi[threadIdx.x + o];
      const float mv = bv[threadIdx.x];
      const int mi = bi[threadIdx.x];
      if (oi != 0x7fffffff && (mi == 0x7fffffff || ov > mv || (ov == mv && oi < mi))) {
        bv[threadIdx.x] = ov;
        bi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) idx[b] = bi[0] - start;
}

template <int MODE>
int launch_sim(int Dm, dim3 grid, hipStream_t s, const float* fq, const float* fm, int Nq, int XY,
               float scale, int clip, const float* nv, float* sim, float* stats,
               const float* rowstats, float* prob, const float* row_weight) {
  switch (Dm) {
    case 8: hipLaunchKernelGGL((sim_kernel<8, MODE>), grid, dim3(256), 0, s, fq, fm, Nq, XY, scale, clip, nv, sim, stats, rowstats, prob, row_weight); break;
    case 16: hipLaunchKernelGGL((sim_kernel<16, MODE>), grid, dim3(256), 0, s, fq, fm, Nq, XY, scale, clip, nv, sim, stats, rowstats, prob, row_weight); break;
    case 32: hipLaunchKernelGGL((sim_kernel<32, MODE>), grid, dim3(256), 0, s, fq, fm, Nq, XY, scale, clip, nv, sim, stats, rowstats, prob, row_weight); break;
    case 64: hipLaunchKernelGGL((sim_kernel<64, MODE>), grid, dim3(256), 0, s, fq, fm, Nq, XY, scale, clip, nv, sim, stats, rowstats, prob, row_weight); break;
    default: return SNAP_ERR_UNSUPPORTED;
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

}  // namespace

extern "C" size_t snap_sim_rowstats_bytes(int32_t B, int32_t Nq) {
  return (size_t)B * Nq * 2 * sizeof(float);
}

extern "C" int snap_sim_softmax_f32(const float* fq, const float* fm, int32_t B, int32_t Nq,
                                    int32_t XY, int32_t Dm, float scale, int32_t clip_negative,
                                    const float* num_valid, float* sim, float* chunk_stats,
                                    float* prob, float* rowstats, void* stream) {
  return snap_sim_softmax_weighted_f32(fq, fm, B, Nq, XY, Dm, scale, clip_negative, num_valid,
                                       nullptr, sim, chunk_stats, prob, rowstats, stream);
}

extern "C" int snap_sim_softmax_weighted_f32(const float* fq, const float* fm, int32_t B,
                                             int32_t Nq, int32_t XY, int32_t Dm, float scale,
                                             int32_t clip_negative, const float* num_valid,
                                             const float* row_weight, float* sim,
                                             float* chunk_stats, float* prob, float* rowstats,
                                             void* stream) {
  if (!fq || !fm || !num_valid || !sim || !chunk_stats) return SNAP_ERR_NULL;
  if (prob && !rowstats) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || XY <= 0) return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)snap_cdiv(XY, 256), (unsigned)snap_cdiv(Nq, SIM_TQ), (unsigned)B);
  // (the VALU kernel takes unaligned inputs and other Dm; tests reach it through a 4-byte
  //  offset view and compare the two bit for bit)
  const bool use_mfma = true;
  const bool aligned = ((reinterpret_cast<uintptr_t>(fq) | reinterpret_cast<uintptr_t>(fm)) & 15) == 0;
  int rc = SNAP_OK;
  if (use_mfma && aligned && (Dm == 8 || Dm == 16 || Dm == 32 || Dm == 64)) {
#define SNAP_SIM_MFMA_CASE(D)                                                                     \
  case D:                                                                                         \
    hipLaunchKernelGGL(sim_mfma_kernel<D>, grid, dim3(256), 0, s, fq, fm, Nq, XY, scale,          \
                       clip_negative, num_valid, sim, chunk_stats, row_weight);                   \
    break;
    switch (Dm) {
      SNAP_SIM_MFMA_CASE(8)
      SNAP_SIM_MFMA_CASE(16)
      SNAP_SIM_MFMA_CASE(32)
      SNAP_SIM_MFMA_CASE(64)
    }
#undef SNAP_SIM_MFMA_CASE
    SNAP_CHECK_LAUNCH();
  } else {
    rc = launch_sim<0>(Dm, grid, s, fq, fm, Nq, XY, scale, clip_negative, num_valid, sim,
                       chunk_stats, nullptr, nullptr, row_weight);
  }
  if (rc != SNAP_OK) return rc;
  if (rowstats) {
    const int64_t rows = (int64_t)B * Nq;
    const int NC = (XY + SIM_CH - 1) / SIM_CH;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)snap_cdiv(rows, 4)), dim3(256), 0, s,
                       (const float*)chunk_stats, rows, NC, rowstats);
    SNAP_CHECK_LAUNCH();
  }
  if (prob) {
    rc = launch_sim<1>(Dm, grid, s, fq, fm, Nq, XY, scale, clip_negative, num_valid, nullptr,
                       nullptr, rowstats, prob, row_weight);
  }
  return rc;
}

extern "C" size_t snap_sim_split_workspace_bytes(int32_t B, int32_t Nq, int32_t XY, int32_t Dm,
                                                 int32_t parts) {
  if (B <= 0 || Nq <= 0 || XY <= 0 || Dm <= 0 || parts < 2 || parts > 3) return 0;
  return ((size_t)B * Nq + (size_t)B * XY) * parts * Dm * sizeof(__bf16);
}

extern "C" int snap_sim_softmax_split_f32(const float* fq, const float* fm, int32_t B, int32_t Nq,
                                          int32_t XY, int32_t Dm, float scale, int32_t clip_negative,
                                          const float* num_valid, const float* row_weight,
                                          int32_t parts, float* sim, float* chunk_stats,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!fq || !fm || !num_valid || !sim || !chunk_stats || !workspace) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || XY <= 0) return SNAP_ERR_BAD_SHAPE;
  if ((Dm != 16 && Dm != 32 && Dm != 64) || (parts != 2 && parts != 3)) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < snap_sim_split_workspace_bytes(B, Nq, XY, Dm, parts)) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return SNAP_ERR_BAD_SHAPE;
  const bool force_general = (clip_negative & 2) != 0;   // bit 1: tests / tools pin the general kernel
  clip_negative &= 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  __bf16* fqs = static_cast<__bf16*>(workspace);
  __bf16* fms = fqs + (size_t)B * Nq * parts * Dm;
  const int64_t nq = (int64_t)B * Nq * Dm, nm = (int64_t)B * XY * Dm;
  if (parts == 3) {
    hipLaunchKernelGGL(sim_presplit_kernel<3>, dim3((unsigned)snap_cdiv(nq, 256)), dim3(256), 0, s, fq, nq, Dm, fqs);
    hipLaunchKernelGGL(sim_presplit_kernel<3>, dim3((unsigned)snap_cdiv(nm, 256)), dim3(256), 0, s, fm, nm, Dm, fms);
  } else {
    hipLaunchKernelGGL(sim_presplit_kernel<2>, dim3((unsigned)snap_cdiv(nq, 256)), dim3(256), 0, s, fq, nq, Dm, fqs);
    hipLaunchKernelGGL(sim_presplit_kernel<2>, dim3((unsigned)snap_cdiv(nm, 256)), dim3(256), 0, s, fm, nm, Dm, fms);
  }
  SNAP_CHECK_LAUNCH();
  const dim3 grid((unsigned)snap_cdiv(XY, 256), (unsigned)snap_cdiv(Nq, SIM_TQ), (unsigned)B);
  // full-chunk shapes (XY % 256 == 0, 16-byte aligned rows) take the lean kernel: same bits
  const bool fast = (XY % 256 == 0) && !(reinterpret_cast<uintptr_t>(sim) & 15) &&
                    !(reinterpret_cast<uintptr_t>(chunk_stats) & 15) && !force_general;
#define SNAP_SIM_SPLIT_CASE(D, P)                                                                  \
  do {                                                                                             \
    if (fast && clip_negative)                                                                     \
      hipLaunchKernelGGL((sim_split_fast_kernel<D, P, true>), grid, dim3(256), 0, s, (const __bf16*)fqs, \
                         (const __bf16*)fms, Nq, XY, scale, num_valid, sim, chunk_stats, row_weight); \
    else if (fast)                                                                                 \
      hipLaunchKernelGGL((sim_split_fast_kernel<D, P, false>), grid, dim3(256), 0, s, (const __bf16*)fqs, \
                         (const __bf16*)fms, Nq, XY, scale, num_valid, sim, chunk_stats, row_weight); \
    else                                                                                           \
      hipLaunchKernelGGL((sim_split_kernel<D, P>), grid, dim3(256), 0, s, (const __bf16*)fqs,      \
                         (const __bf16*)fms, Nq, XY, scale, clip_negative, num_valid, sim, chunk_stats, \
                         row_weight);                                                              \
  } while (0)
  if (parts == 3) {
    if (Dm == 16) SNAP_SIM_SPLIT_CASE(16, 3); else if (Dm == 32) SNAP_SIM_SPLIT_CASE(32, 3); else SNAP_SIM_SPLIT_CASE(64, 3);
  } else {
    if (Dm == 16) SNAP_SIM_SPLIT_CASE(16, 2); else if (Dm == 32) SNAP_SIM_SPLIT_CASE(32, 2); else SNAP_SIM_SPLIT_CASE(64, 2);
  }
#undef SNAP_SIM_SPLIT_CASE
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_ransac_sample_f32(const float* fq, const float* fm, const float* chunk_stats,
                                      int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t Dm,
                                      float scale, int32_t clip_negative, int32_t S, uint64_t seed,
                                      const float* uniforms, int32_t* corr, void* stream) {
  return snap_ransac_sample_ws_f32(fq, fm, chunk_stats, B, Nq, X, Y, Dm, scale, clip_negative, S,
                                   seed, uniforms, corr, nullptr, 0, stream);
}

extern "C" size_t snap_ransac_sample_workspace_bytes(int32_t B, int32_t Nq) {
  return (size_t)B * Nq * 65 * sizeof(float);   // 64 lane prefixes + the row maximum
}

extern "C" int snap_ransac_sample_ws_f32(const float* fq, const float* fm, const float* chunk_stats,
                                         int32_t B, int32_t Nq, int32_t X, int32_t Y, int32_t Dm,
                                         float scale, int32_t clip_negative, int32_t S,
                                         uint64_t seed, const float* uniforms, int32_t* corr,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return snap_ransac_sample_rows_f32(fq, fm, chunk_stats, nullptr, B, Nq, X, Y, Dm, scale,
                                     clip_negative, S, seed, uniforms, corr, workspace,
                                     workspace_bytes, stream);
}

extern "C" int snap_ransac_sample_rows_f32(const float* fq, const float* fm,
                                           const float* chunk_stats, const float* row_cdf,
                                           int32_t B, int32_t Nq, int32_t X, int32_t Y,
                                           int32_t Dm, float scale, int32_t clip_negative,
                                           int32_t S, uint64_t seed, const float* uniforms,
                                           int32_t* corr, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  return snap_ransac_sample_sim_f32(fq, fm, chunk_stats, row_cdf, nullptr, nullptr, B, Nq, X, Y, Dm,
                                    scale, clip_negative, S, seed, uniforms, corr, workspace,
                                    workspace_bytes, stream);
}

extern "C" int snap_ransac_sample_sim_f32(const float* fq, const float* fm,
                                          const float* chunk_stats, const float* row_cdf,
                                          const float* sim, const float* row_unscale, int32_t B,
                                          int32_t Nq, int32_t X, int32_t Y, int32_t Dm, float scale,
                                          int32_t clip_negative, int32_t S, uint64_t seed,
                                          const float* uniforms, int32_t* corr, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (!fq || !fm || !chunk_stats || !corr) return SNAP_ERR_NULL;
  if ((sim == nullptr) != (row_unscale == nullptr)) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || X <= 0 || Y <= 0 || S <= 0) return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* lane_incl = nullptr;
  float* rowmax = nullptr;
  if (workspace) {
    if (workspace_bytes < snap_ransac_sample_workspace_bytes(B, Nq)) return SNAP_ERR_WORKSPACE;
    const int64_t rows = (int64_t)B * Nq;
    lane_incl = static_cast<float*>(workspace);
    rowmax = lane_incl + rows * 64;
    const int NC = (X * Y + SIM_CH - 1) / SIM_CH;
    hipLaunchKernelGGL(chunk_prefix_kernel, dim3((unsigned)snap_cdiv(rows, 4)), dim3(256), 0, s,
                       chunk_stats, rows, NC, lane_incl, rowmax);
    SNAP_CHECK_LAUNCH();
  }
  {
    const int NC = (X * Y + SIM_CH - 1) / SIM_CH;
    // (without a workspace: the table-free kernel, one correspondence per wave, same samples)
    if (lane_incl && sim && row_unscale && (NC + 63) / 64 <= 64) {
#ifndef SNAP_RANSAC_NQ
#define SNAP_RANSAC_NQ 4
#endif
      constexpr int NQ = SNAP_RANSAC_NQ;
      const dim3 fgrid((unsigned)snap_cdiv(S, 4 * NQ), (unsigned)B);
      hipLaunchKernelGGL(ransac_sample_fast_kernel<NQ>, fgrid, dim3(256), 0, s, chunk_stats, Nq, X, Y, S,
                         seed, uniforms, corr, (const float*)lane_incl, (const float*)rowmax, row_cdf,
                         sim, row_unscale);
      SNAP_CHECK_LAUNCH();
      return SNAP_OK;
    }
  }
  const dim3 grid((unsigned)snap_cdiv(S, 4), (unsigned)B);
  switch (Dm) {
    case 8: hipLaunchKernelGGL(ransac_sample_kernel<8>, grid, dim3(256), 0, s, fq, fm, chunk_stats, Nq, X, Y, scale, clip_negative, S, seed, uniforms, corr, (const float*)lane_incl, (const float*)rowmax, row_cdf, sim, row_unscale); break;
    case 16: hipLaunchKernelGGL(ransac_sample_kernel<16>, grid, dim3(256), 0, s, fq, fm, chunk_stats, Nq, X, Y, scale, clip_negative, S, seed, uniforms, corr, (const float*)lane_incl, (const float*)rowmax, row_cdf, sim, row_unscale); break;
    case 32: hipLaunchKernelGGL(ransac_sample_kernel<32>, grid, dim3(256), 0, s, fq, fm, chunk_stats, Nq, X, Y, scale, clip_negative, S, seed, uniforms, corr, (const float*)lane_incl, (const float*)rowmax, row_cdf, sim, row_unscale); break;
    case 64: hipLaunchKernelGGL(ransac_sample_kernel<64>, grid, dim3(256), 0, s, fq, fm, chunk_stats, Nq, X, Y, scale, clip_negative, S, seed, uniforms, corr, (const float*)lane_incl, (const float*)rowmax, row_cdf, sim, row_unscale); break;
    default: return SNAP_ERR_UNSUPPORTED;
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_masked_softmax_rows_f32(const float* x, const uint8_t* mask, int32_t B,
                                            int32_t N, float* weights, float* cdf, void* stream) {
  if (!x || !mask || !weights || !cdf) return SNAP_ERR_NULL;
  if (B <= 0 || N <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(masked_softmax_rows_kernel, dim3((unsigned)B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, mask, N, weights, cdf);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_poses_from_corr_f32(const int32_t* corr, const float* q_xy, int32_t B,
                                        int32_t Nq, int32_t P, int32_t retries, float cell_size,
                                        float* poses, void* stream) {
  if (!corr || !q_xy || !poses) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || P <= 0 || retries <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(poses_from_corr_kernel, dim3((unsigned)snap_cdiv((int64_t)B * P, 256)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), corr, q_xy, B, Nq, P, retries,
                     cell_size, poses);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_pose_score_workspace_bytes(int32_t B, int32_t Nq, int32_t P, int32_t X,
                                                  int32_t Y) {
  (void)X; (void)Y;
  const int pch = score_pose_chunks(P);
  const int nch = score_chunks(B, Nq, pch);
  // pose table (16-byte aligned, first) + per-chunk partial sums.
  return (size_t)B * P * 4 * sizeof(float) + (size_t)B * nch * P * sizeof(float);
}

extern "C" int snap_pose_score_f32(const float* sim, const float* poses, const float* q_xy,
                                   const uint8_t* valid_q, const uint8_t* map_valid, int32_t B,
                                   int32_t Nq, int32_t X, int32_t Y, int32_t P, float cell_size,
                                   int32_t mask_oob, float* scores, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (!sim || !poses || !q_xy || !valid_q || !scores || !workspace) return SNAP_ERR_NULL;
  if (mask_oob && !map_valid) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || X <= 0 || Y <= 0 || P <= 0) return SNAP_ERR_BAD_SHAPE;
  if (Y > PS_LDS_FLOATS / 2) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < snap_pose_score_workspace_bytes(B, Nq, P, X, Y)) return SNAP_ERR_WORKSPACE;
  ScoreArgs a;
  a.sim = sim; a.poses = poses; a.q_xy = q_xy; a.valid_q = valid_q; a.map_valid = map_valid;
  a.B = B; a.Nq = Nq; a.X = X; a.Y = Y; a.P = P;
  a.cell = cell_size; a.mask_oob = mask_oob;
  const int pch = score_pose_chunks(P);
  const int nch = score_chunks(B, Nq, pch);
  a.points_per_chunk = (Nq + nch - 1) / nch;
  if ((int64_t)X * Y <= PS_LDS_FLOATS) {
    a.RB = X; a.NB = 1;
  } else {
    a.RB = PS_LDS_FLOATS / Y;
    a.NB = (X - 1 + (a.RB - 1) - 1) / (a.RB - 1);
  }
  float* table = static_cast<float*>(workspace);
  a.table = table;
  a.partial = table + (size_t)B * P * 4;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return SNAP_ERR_BAD_SHAPE;
  const size_t lds = (size_t)min((int64_t)a.RB * Y, (int64_t)PS_LDS_FLOATS) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool bands = a.NB > 1;
  // double-buffered LDS-DMA variant: two whole planes must fit (2 * XY * 4 <= 128 KiB).
  constexpr bool db_enabled = true;
  const bool use_db = db_enabled && !bands && (Y % 4 == 0) && X >= 2 && Y >= 2 &&
                      ((int64_t)X * (Y + 4) * 4 <= PS_DB_PLANE_BYTES);
  // banded double-buffered variant: a plane that does not fit streams through in row bands
  const int band_rows = (int)(PS_DB_PLANE_BYTES / ((size_t)(Y + 4) * sizeof(float))) - 1;
  const bool use_band_db = db_enabled && !use_db && !mask_oob && (Y % 4 == 0) && X >= 2 &&
                           Y >= 2 && band_rows >= 1 && X > band_rows;
  const void* fn = nullptr;
  size_t lds_bytes = lds;
  if (use_band_db) {
    a.RB = band_rows;
    a.NB = (X - 1 + band_rows - 1) / band_rows;     // cell rows 0 .. X-2
    fn = (const void*)&pose_score_band_db_kernel<PS_BAND_PPT>;
    lds_bytes = (size_t)2 * (band_rows + 1) * (Y + 4) * sizeof(float);
  } else if (use_db) {
    if (Y == 128)
      fn = mask_oob ? (const void*)&pose_score_db_kernel<PS_PPT, true, 128>
                    : (const void*)&pose_score_db_kernel<PS_PPT, false, 128>;
    else
      fn = mask_oob ? (const void*)&pose_score_db_kernel<PS_PPT, true, 0>
                    : (const void*)&pose_score_db_kernel<PS_PPT, false, 0>;
    lds_bytes = (size_t)2 * X * (Y + 4) * sizeof(float);
  } else if (mask_oob) fn = bands ? (const void*)&pose_score_kernel<PS_PPT, true, true>
                           : (const void*)&pose_score_kernel<PS_PPT, true, false>;
  else fn = bands ? (const void*)&pose_score_kernel<PS_PPT, false, true>
                  : (const void*)&pose_score_kernel<PS_PPT, false, false>;
  if (lds_bytes > 64 * 1024) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((use_db || use_band_db) ? 2 * PS_DB_PLANE_BYTES
                                                          : PS_LDS_FLOATS * sizeof(float))) !=
        hipSuccess)
      return SNAP_ERR_LAUNCH;
  }
  if (use_db || use_band_db) {
    hipLaunchKernelGGL(pose_table_cells_kernel, dim3((unsigned)snap_cdiv((int64_t)B * P, 256)),
                       dim3(256), 0, s, poses, (int64_t)B * P, cell_size, table);
  } else {
    hipLaunchKernelGGL(pose_table_kernel, dim3((unsigned)snap_cdiv((int64_t)B * P, 256)),
                       dim3(256), 0, s, poses, (int64_t)B * P, table);
  }
  SNAP_CHECK_LAUNCH();
  {
    void* kargs[] = {(void*)&a};
    // (the banded kernel carries PS_BAND_PPT poses per thread; the point chunking and the partial buffer
    // layout [B, nch, P] are the same)
    const int gx = use_band_db ? (P + PS_THREADS * PS_BAND_PPT - 1) / (PS_THREADS * PS_BAND_PPT) : pch;
    if (hipLaunchKernel(fn, dim3(gx, nch, B), dim3(PS_THREADS), kargs, lds_bytes, s) != hipSuccess)
      return SNAP_ERR_LAUNCH;
  }
  SNAP_CHECK_LAUNCH();
  const int64_t total = (int64_t)B * P;
  hipLaunchKernelGGL(pose_score_reduce_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     s, (const float*)a.partial, nch, P, total, scores);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_pose_score_window_workspace_bytes(int32_t B, int32_t Nq, int32_t P, int32_t X,
                                                         int32_t Y) {
  return snap_pose_score_workspace_bytes(B, Nq, P, X, Y) + (size_t)B * 4 * sizeof(float) + 16;
}

extern "C" int32_t snap_pose_score_window_supported(int32_t X, int32_t Y, int32_t radius_cells) {
  if (X < 2 || Y < 4 || Y % 4 != 0 || radius_cells < 0) return 0;
  const int WR = min(2 * radius_cells + 3, X);
  const int WC = min((2 * radius_cells + 6 + 3) & ~3, Y);
  return (int64_t)WR * ((WC >> 2) + 1) <= (int64_t)PS_WIN_ROUNDS * PS_THREADS ? 1 : 0;
}

extern "C" int snap_pose_score_window_f32(const float* sim, const float* poses, const float* centers,
                                          int32_t radius_cells, const float* q_xy,
                                          const uint8_t* valid_q, int32_t B, int32_t Nq, int32_t X,
                                          int32_t Y, int32_t P, float cell_size, float* scores,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!sim || !poses || !centers || !q_xy || !valid_q || !scores || !workspace) return SNAP_ERR_NULL;
  if (B <= 0 || Nq <= 0 || X <= 0 || Y <= 0 || P <= 0) return SNAP_ERR_BAD_SHAPE;
  if (!snap_pose_score_window_supported(X, Y, radius_cells)) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < snap_pose_score_window_workspace_bytes(B, Nq, P, X, Y)) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return SNAP_ERR_BAD_SHAPE;
  ScoreWinArgs w;
  ScoreArgs& a = w.s;
  a.sim = sim; a.poses = poses; a.q_xy = q_xy; a.valid_q = valid_q; a.map_valid = nullptr;
  a.B = B; a.Nq = Nq; a.X = X; a.Y = Y; a.P = P;
  a.cell = cell_size; a.mask_oob = 0;
  // the point chunking (and so the order of every sum) of snap_pose_score_f32
  const int pch = score_pose_chunks(P);
  const int nch = score_chunks(B, Nq, pch);
  a.points_per_chunk = (Nq + nch - 1) / nch;
  a.RB = X; a.NB = 1;
  float* table = static_cast<float*>(workspace);
  a.table = table;
  a.partial = table + (size_t)B * P * 4;
  float* ctable = a.partial + (size_t)B * nch * P;
  ctable = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ctable) + 15) & ~(uintptr_t)15);
  w.ctable = ctable;
  w.rad = radius_cells;
  w.WR = min(2 * radius_cells + 3, X);
  w.WC = min((2 * radius_cells + 6 + 3) & ~3, Y);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pose_table_cells_kernel, dim3((unsigned)snap_cdiv((int64_t)B * P, 256)), dim3(256), 0,
                     s, poses, (int64_t)B * P, cell_size, table);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(pose_table_cells_kernel, dim3((unsigned)snap_cdiv((int64_t)B, 256)), dim3(256), 0, s,
                     centers, (int64_t)B, cell_size, ctable);
  SNAP_CHECK_LAUNCH();
  const size_t lds_bytes = (size_t)2 * w.WR * (w.WC + 4) * sizeof(float);
  const void* fn = (const void*)&pose_score_window_kernel<PS_WIN_PPT>;
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
    return SNAP_ERR_LAUNCH;
  {
    void* kargs[] = {(void*)&w};
    const int gx = (P + PS_THREADS * PS_WIN_PPT - 1) / (PS_THREADS * PS_WIN_PPT);
    if (hipLaunchKernel(fn, dim3(gx, nch, B), dim3(PS_THREADS), kargs, lds_bytes, s) != hipSuccess)
      return SNAP_ERR_LAUNCH;
  }
  SNAP_CHECK_LAUNCH();
  const int64_t total = (int64_t)B * P;
  hipLaunchKernelGGL(pose_score_reduce_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     s, (const float*)a.partial, nch, P, total, scores);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_refine_lattice_f32(const float* init, const float* offs_r, const float* offs_p,
                                       int32_t B, int32_t nr, int32_t np_, float* out,
                                       void* stream) {
  if (!init || !offs_r || !offs_p || !out) return SNAP_ERR_NULL;
  if (B <= 0 || nr <= 0 || np_ <= 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)B * nr * np_ * np_;
  hipLaunchKernelGGL(refine_lattice_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), init, offs_r, offs_p, B, nr, np_, out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_argmax_rows_f32(const float* scores, int32_t B, int32_t P, int32_t start,
                                    int32_t* idx, void* stream) {
  if (!scores || !idx) return SNAP_ERR_NULL;
  if (B <= 0 || P <= 0 || start < 0 || start >= P) return SNAP_ERR_BAD_SHAPE;
  if (P - start > 4096)
    hipLaunchKernelGGL(argmax_rows_kernel<1024>, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       scores, P, start, idx);
  else
    hipLaunchKernelGGL(argmax_rows_kernel<256>, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       scores, P, start, idx);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
