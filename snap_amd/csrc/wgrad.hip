// Weight-gradient engine (training path): dW[k, co] = sum_m Z[m, k] * dY[m, co]
// with Z the implicit im2col of prologue(x) -- the transpose-A companion of
// conv_igemm.hip on the same f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32).
//
// GEMM view: rows = K = KH*KW*Cin (one workgroup owns BKT consecutive input channels
// of ONE (kh, kw) tap), cols = Cout (BN per workgroup), reduction over the M = N*Ho*Wo
// output pixels in slabs of 16, split over gridDim.y chunks of M whose partial tiles
// are summed by a second, fixed-order kernel (deterministic; no float atomics).
// Both operands are staged row-major ([16 m][BKT] and [16 m][BN]): the loader writes
// float4 rows straight to LDS and both MFMA operand fetches are conflict-free
// ds_read_b32 over consecutive lanes -- no transpose anywhere.
//
// Used for: every conv / Dense kernel gradient (VJP of snap/models/resnet.py StdConv,
// image_encoder.py skip convs, layers.py Dense), d fm = G^T fq of the similarity VJP
// (bev_localizer.py:157) and the matching-head kernel gradient (bev_mapper.py:285).
#include "wgrad_common.h"

namespace {

template <int BKT, int BN, bool VEC, int PRO>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  constexpr int RS = 16;                   // reduction slab (output pixels)
  constexpr int TM = BKT / 64, TN = BN / 64;
  constexpr int ZQ = BKT / 4;              // float4 per Z row
  constexpr int ZRPP = 256 / ZQ;           // Z rows per pass
  constexpr int ZPASS = RS / ZRPP;
  constexpr int DQ = BN / 4;
  constexpr int DRPP = 256 / DQ;
  constexpr int DPASS = RS / DRPP;
  constexpr int ZELEMS = (RS * BKT) / 256;  // scalar path: elements per thread
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  __shared__ __attribute__((aligned(16))) float smem[2 * RS * BKT + 2 * RS * BN];
  float* const Zs0 = smem;
  float* const Ds0 = smem + 2 * RS * BKT;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int kt = blockIdx.x / a.ncol, col_t = blockIdx.x - kt * a.ncol;
  // VEC: a tile is BKT channels of ONE (kh, kw) tap.  Scalar path (Cin < 4 or unaligned,
  // i.e. the 7x7x3 root conv): a tile is BKT consecutive rows of the FLAT k = (kh*KW+kw)*Cin+c
  // axis, so its 147 rows fill 3 tiles instead of 49 tiles that are 95 % padding.
  const int kpos = VEC ? kt / a.ctiles : 0, ct = VEC ? kt - kpos * a.ctiles : 0;
  const int kh = kpos / d.KW, kw = kpos - kh * d.KW;
  const int c0 = ct * BKT;
  const int k0 = kt * BKT;  // scalar path: first flat k of the tile
  const int n0 = col_t * BN;
  const int HoWo = d.Ho * d.Wo;
  const int64_t Meff = a.row_count ? min((int64_t)*a.row_count, (int64_t)a.M) : (int64_t)a.M;
  // with a device-side row count the chunks are re-cut over the rows that exist, so every
  // workgroup keeps an equal share (the launch was sized for the upper bound M)
  const int64_t spc = a.row_count ? (((Meff + RS - 1) / RS) + gridDim.y - 1) / gridDim.y
                                  : (int64_t)a.slabs_per_chunk;
  const int64_t m_begin = (int64_t)blockIdx.y * spc * RS;
  const int64_t m_end = min(Meff, m_begin + spc * RS);
  const int nslab = m_end > m_begin ? (int)((m_end - m_begin + RS - 1) / RS) : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 zr[VEC ? ZPASS : 1], zmu[VEC ? ZPASS : 1], zsc[VEC ? ZPASS : 1], zbeta;
  bool zin[VEC ? ZPASS : 1];
  float se[VEC ? 1 : ZELEMS], smu[VEC ? 1 : ZELEMS], ssc[VEC ? 1 : ZELEMS], sbeta[VEC ? 1 : ZELEMS];
  bool sin_[VEC ? 1 : ZELEMS];
  f32x4 dr[DPASS];
  bool din[DPASS];
  const int zq = tid % ZQ, zrow0 = tid / ZQ;
  const int dq = tid % DQ, drow0 = tid / DQ;
  const int zc = c0 + 4 * zq;  // VEC: first channel of this thread's quad
  // scalar path: (kh, kw, c) of each of this thread's slab elements (slab-invariant)
  int ekh[VEC ? 1 : ZELEMS], ekw[VEC ? 1 : ZELEMS], ec[VEC ? 1 : ZELEMS];
  bool ekok[VEC ? 1 : ZELEMS];
  if constexpr (!VEC) {
#pragma unroll
    for (int e = 0; e < ZELEMS; ++e) {
      const int idx = tid + 256 * e;
      const int k = k0 + (idx % BKT);
      ekok[e] = k < a.K;
      const int kk = ekok[e] ? k : 0;
      const int kp = kk / d.Cin;
      ec[e] = kk - kp * d.Cin;
      ekh[e] = kp / d.KW;
      ekw[e] = kp - ekh[e] * d.KW;
    }
  }
  if constexpr (VEC && need_gn) {
    zbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + (zc < d.Cin ? zc : 0));
  }

  // (image, ho, wo) of each Z row this thread stages, advanced by RS rows per slab: the
  // reduction walks M, so a per-slab m -> (n, ho, wo) decode would cost two integer
  // divisions per row per slab (more VALU time than the loads themselves).
  constexpr int NZ = VEC ? ZPASS : ZELEMS;
  int rn[NZ], rho[NZ], rwo[NZ];
#pragma unroll
  for (int p = 0; p < NZ; ++p) {
    const int64_t m = m_begin + (VEC ? zrow0 + p * ZRPP : (tid + 256 * p) / BKT);
    const int mm = (int)min(m, (int64_t)a.M - 1);
    rn[p] = mm / HoWo;
    const int r = mm - rn[p] * HoWo;
    rho[p] = r / d.Wo;
    rwo[p] = r - rho[p] * d.Wo;
  }
  auto advance_rows = [&]() {
#pragma unroll
    for (int p = 0; p < NZ; ++p) {
      rwo[p] += RS;
      while (rwo[p] >= d.Wo) { rwo[p] -= d.Wo; ++rho[p]; }
      while (rho[p] >= d.Ho) { rho[p] -= d.Ho; ++rn[p]; }
    }
  };

  auto load_slab = [&](int sl) {
    const int64_t ms = m_begin + (int64_t)sl * RS;
    if constexpr (VEC) {
#pragma unroll
      for (int p = 0; p < ZPASS; ++p) {
        const int64_t m = ms + zrow0 + p * ZRPP;
        const bool mok = m < m_end;
        const int n = rn[p], ho = rho[p], wo = rwo[p];   // walked incrementally (no divisions)
        const int hi = ho * d.stride - d.pad_t + kh, wi = wo * d.stride - d.pad_l + kw;
        const bool inb = mok && zc < d.Cin && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
        zin[p] = inb;
        int64_t off = inb ? (((int64_t)n * d.H + hi) * d.W + wi) * d.Cin_stride + zc : (int64_t)0;
        if (a.rows_z)  // row list over a flat [1,1,M,C] tensor
          off = inb ? (int64_t)a.rows_z[m] * d.Cin_stride + zc : (int64_t)0;
        zr[p] = *reinterpret_cast<const f32x4*>(a.x + off);
        if constexpr (need_gn) {
          const int64_t so = inb ? (int64_t)n * d.Cin + zc : (int64_t)0;
          zmu[p] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
          zsc[p] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < ZELEMS; ++e) {
        const int idx = tid + 256 * e;           // over RS x BKT
        const int rr = idx / BKT;
        const int64_t m = ms + rr;
        const bool mok = m < m_end;
        const int n = rn[e], ho = rho[e], wo = rwo[e];
        const int hi = ho * d.stride - d.pad_t + ekh[e], wi = wo * d.stride - d.pad_l + ekw[e];
        const int c = ec[e];
        const bool inb = mok && ekok[e] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
        sin_[e] = inb;
        int64_t off = inb ? (((int64_t)n * d.H + hi) * d.W + wi) * d.Cin_stride + c : (int64_t)0;
        if (a.rows_z) off = inb ? (int64_t)a.rows_z[m] * d.Cin_stride + c : (int64_t)0;
        se[e] = a.x[off];
        if constexpr (need_gn) {
          const int64_t so = inb ? (int64_t)n * d.Cin + c : (int64_t)0;
          smu[e] = a.gn_mu[so];
          ssc[e] = a.gn_sc[so];
          sbeta[e] = a.gn_beta[inb ? c : 0];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p) {
      const int64_t m = ms + drow0 + p * DRPP;
      const int col = n0 + 4 * dq;
      const bool ok = m < m_end && col < d.Cout;
      din[p] = ok;
      const int64_t drow = (ok && a.rows_dy) ? (int64_t)a.rows_dy[m] : m;
      dr[p] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? drow * d.Cout_stride + col : (int64_t)0));
    }
  };

  auto store_slab = [&](int buf) {
    float* zs = Zs0 + buf * (RS * BKT);
    float* ds = Ds0 + buf * (RS * BN);
    if constexpr (VEC) {
#pragma unroll
      for (int p = 0; p < ZPASS; ++p) {
        f32x4 v = zr[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pv;
          if constexpr (need_gn)
            pv = wg_pro<PRO>(v[e], zmu[p][e], zsc[p][e], zbeta[e], d.in_scale, d.in_shift);
          else
            pv = wg_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
          v[e] = (zin[p] && (zc + e < d.Cin)) ? pv : 0.f;
        }
        *reinterpret_cast<f32x4*>(zs + (zrow0 + p * ZRPP) * BKT + 4 * zq) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < ZELEMS; ++e) {
        const int idx = tid + 256 * e;
        float pv;
        if constexpr (need_gn)
          pv = wg_pro<PRO>(se[e], smu[e], ssc[e], sbeta[e], d.in_scale, d.in_shift);
        else
          pv = wg_pro<PRO>(se[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        zs[idx] = sin_[e] ? pv : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p)
      *reinterpret_cast<f32x4*>(ds + (drow0 + p * DRPP) * BN + 4 * dq) =
          din[p] ? dr[p] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  if (nslab > 0) {
    load_slab(0);
    advance_rows();
    store_slab(0);
  }
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    const int cur = sl & 1;
    const bool more = sl + 1 < nslab;
    if (more) {
      load_slab(sl + 1);
      advance_rows();
    }
    const float* zs = Zs0 + cur * (RS * BKT);
    const float* ds = Ds0 + cur * (RS * BN);
    // LDS -> register operand fetch runs one k-pair ahead of the MFMAs (as in conv_igemm.hip).
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = zs[lhi * BKT + wr * (BKT / 2) + i * 32 + l31];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = ds[lhi * BN + wc * (BN / 2) + j * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < RS / 2; ++kk) {
      const int cu = kk & 1, nx = cu ^ 1;
      if (kk + 1 < RS / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[nx][i] = zs[(2 * (kk + 1) + lhi) * BKT + wr * (BKT / 2) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[nx][j] = ds[(2 * (kk + 1) + lhi) * BN + wc * (BN / 2) + j * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cu][i], bv[cu][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
    for (int kk = 0; kk < RS / 2; ++kk) {
      if (kk + 1 < RS / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
    if (more) store_slab(cur ^ 1);
    __syncthreads();
  }

  // partial tile -> workspace [chunk][K][Cout]
  float* out = a.partial + (int64_t)blockIdx.y * a.K * d.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int c = c0 + wr * (BKT / 2) + i * 32 + ri;
      int64_t krow;
      if constexpr (VEC) {
        if (c >= d.Cin) continue;
        krow = (int64_t)kpos * d.Cin + c;
      } else {
        krow = k0 + wr * (BKT / 2) + i * 32 + ri;
        if (krow >= a.K) continue;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 32 + l31;
        if (col < d.Cout) out[krow * d.Cout + col] = acc[i][j][r];
      }
    }
}

// Fixed-order reduction of the partial tiles over the chunks.  A workgroup owns 64 consecutive elements; its four
// waves take the slots s = p, p + 4, ... (eight loads in flight each) and are combined in the order p = 0..3: four
// times the loads in flight of the one-thread-per-element loop, which ran at the latency of S / 8 dependent rounds
// on a handful of workgroups (126 launches, 1.3 ms per C3 step).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int S, int64_t total,
                                                           float* __restrict__ dw, int accumulate) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + l;
  float t = 0.f;
  if (i < total) {
#pragma unroll 8
    for (int s = p; s < S; s += 4) t += partial[(int64_t)s * total + i];
  }
  red[p][l] = t;
  __syncthreads();
  if (p == 0 && i < total) {
    float r = accumulate ? dw[i] : 0.f;
    r += ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
    dw[i] = r;
  }
}

template <int BKT, int BN, bool VEC, int PRO>
int wg_launch(const WgradArgs& a, const WgPlan& p, hipStream_t s) {
  const dim3 grid((unsigned)(p.ktiles * p.ncol), (unsigned)p.S);
  hipLaunchKernelGGL((wgrad_kernel<BKT, BN, VEC, PRO>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

template <int BKT, int BN, bool VEC>
int wg_launch_pro(const WgradArgs& a, const WgPlan& p, hipStream_t s) {
  switch (a.d.prologue) {
    case SNAP_PRO_NONE: return wg_launch<BKT, BN, VEC, SNAP_PRO_NONE>(a, p, s);
    case SNAP_PRO_AFFINE: return wg_launch<BKT, BN, VEC, SNAP_PRO_AFFINE>(a, p, s);
    case SNAP_PRO_GN_RELU:
      if constexpr (VEC) return wg_launch<BKT, BN, VEC, SNAP_PRO_GN_RELU>(a, p, s);
      return SNAP_ERR_UNSUPPORTED;
    case SNAP_PRO_RELU_GN:
      if constexpr (VEC) return wg_launch<BKT, BN, VEC, SNAP_PRO_RELU_GN>(a, p, s);
      return SNAP_ERR_UNSUPPORTED;
    case SNAP_PRO_RELU: return wg_launch<BKT, BN, VEC, SNAP_PRO_RELU>(a, p, s);
    default: return SNAP_ERR_UNSUPPORTED;
  }
}

}  // namespace

// (the A/B switches of the wide / fused-tap plans travel PER CALL in desc->tile_hint:
//  SNAP_WGRAD_NO_WIDE / SNAP_WGRAD_NO_FUSED3 -- no process-wide state; the workspace is always sized
//  for every plan)

extern "C" size_t snap_conv2d_wgrad_workspace_bytes(const SnapConvDesc* desc) {
  if (!desc) return 0;
  // the loader variant depends on the pointer alignment seen at launch: size for both.
  int S = max(wg_plan(*desc, true).S, wg_plan(*desc, false).S);
  if (snapwg::wg_wide_ok(*desc, true, SNAP_MATH_BF16)) S = max(S, snapwg::wg_plan_wide(*desc).S);
  if (snapwg::wg_3x3_ok(*desc, true, SNAP_MATH_BF16, false, true, false)) S = max(S, snapwg::wg_plan_3x3(*desc).S);
  return (size_t)S * desc->KH * desc->KW * desc->Cin * desc->Cout * sizeof(float);
}

extern "C" int snap_conv2d_wgrad_f32(const SnapConvDesc* desc, const float* x, const float* dy,
                                     float* dw, const float* gn_mu, const float* gn_sc,
                                     const float* gn_beta, int32_t accumulate, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return snap_conv2d_wgrad_rows_f32(desc, x, dy, dw, gn_mu, gn_sc, gn_beta, accumulate, workspace,
                                    workspace_bytes, nullptr, nullptr, nullptr, stream);
}

extern "C" int snap_conv2d_wgrad_rows_f32(const SnapConvDesc* desc, const float* x,
                                          const float* dy, float* dw, const float* gn_mu,
                                          const float* gn_sc, const float* gn_beta,
                                          int32_t accumulate, void* workspace,
                                          size_t workspace_bytes, const int32_t* rows_z,
                                          const int32_t* rows_dy, const int32_t* row_count,
                                          void* stream) {
  return snap_conv2d_wgrad_ex_f32(desc, x, dy, dw, gn_mu, gn_sc, gn_beta, accumulate, workspace,
                                  workspace_bytes, rows_z, rows_dy, row_count, SNAP_MATH_F32, stream);
}

extern "C" int snap_conv2d_wgrad_ex_f32(const SnapConvDesc* desc, const float* x,
                                        const float* dy, float* dw, const float* gn_mu,
                                        const float* gn_sc, const float* gn_beta,
                                        int32_t accumulate, void* workspace,
                                        size_t workspace_bytes, const int32_t* rows_z,
                                        const int32_t* rows_dy, const int32_t* row_count,
                                        int32_t math, void* stream) {
  return snap_conv2d_wgrad_half_f32(desc, x, dy, dw, gn_mu, gn_sc, gn_beta, accumulate, workspace,
                                    workspace_bytes, rows_z, rows_dy, row_count, math, 0, 0, stream);
}

extern "C" int snap_conv2d_wgrad_half_f32(const SnapConvDesc* desc, const void* x_,
                                          const void* dy_, float* dw, const float* gn_mu,
                                          const float* gn_sc, const float* gn_beta,
                                          int32_t accumulate, void* workspace,
                                          size_t workspace_bytes, const int32_t* rows_z,
                                          const int32_t* rows_dy, const int32_t* row_count,
                                          int32_t math, int32_t x_is_half, int32_t dy_is_half,
                                          void* stream) {
  const float* x = static_cast<const float*>(x_);
  const float* dy = static_cast<const float*>(dy_);
  if (math != SNAP_MATH_F32 && math != SNAP_MATH_BF16 && math != SNAP_MATH_F16) return SNAP_ERR_UNSUPPORTED;
  if ((x_is_half || dy_is_half) &&
      (math == SNAP_MATH_F32 || (x_is_half && dy_is_half) || (x_is_half && desc && desc->prologue != SNAP_PRO_NONE) ||
       (x_is_half && desc && (desc->Cin % 4 != 0 || desc->Cin_stride % 4 != 0))))
    return SNAP_ERR_UNSUPPORTED;
  if (!desc || !x || !dy || !dw || !workspace) return SNAP_ERR_NULL;
  if ((rows_z || rows_dy) && !(desc->KH == 1 && desc->KW == 1 && desc->stride == 1 &&
                               desc->N == 1 && desc->H == 1 && desc->pad_t == 0 &&
                               desc->pad_l == 0))
    return SNAP_ERR_UNSUPPORTED;  // row lists address a flat [1,1,M,C] (Dense) layout
  const SnapConvDesc& d = *desc;
  if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.KH <= 0 || d.KW <= 0 ||
      d.stride <= 0 || d.Ho <= 0 || d.Wo <= 0)
    return SNAP_ERR_BAD_SHAPE;
  if (d.Cin_stride < d.Cin || d.Cout_stride < d.Cout || d.Cout % 4 != 0 || d.Cout_stride % 4 != 0)
    return SNAP_ERR_BAD_SHAPE;
  if ((int64_t)d.N * d.Ho * d.Wo > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(dy) & (dy_is_half ? 7 : 15)) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return SNAP_ERR_BAD_SHAPE;
  const bool gn = d.prologue == SNAP_PRO_GN_RELU || d.prologue == SNAP_PRO_RELU_GN;
  if (gn && (!gn_mu || !gn_sc || !gn_beta)) return SNAP_ERR_NULL;
  if (gn && math != SNAP_MATH_F32 && d.Ho * d.Wo < 4) math = SNAP_MATH_F32;   // (the half engines' loaders assume four
                                                                              //  consecutive rows span <= two images)
  if (workspace_bytes < snap_conv2d_wgrad_workspace_bytes(desc)) return SNAP_ERR_WORKSPACE;
  const bool vec = (d.Cin_stride % 4 == 0) && (d.Cin >= 4) &&
                   ((reinterpret_cast<uintptr_t>(x) & (x_is_half ? 7 : 15)) == 0) && (!gn || (d.Cin % 4 == 0));
  if ((x_is_half || dy_is_half) && !vec) return SNAP_ERR_UNSUPPORTED;
  const bool allow_wide = !(d.tile_hint & SNAP_WGRAD_NO_WIDE), allow_fused3 = !(d.tile_hint & SNAP_WGRAD_NO_FUSED3);
  const bool fused3 = allow_fused3 &&
                      snapwg::wg_3x3_ok(d, vec, math, x_is_half != 0, dy_is_half != 0, rows_z || rows_dy || row_count);
  const WgPlan p = fused3 ? snapwg::wg_plan_3x3(d)
                   : (allow_wide && snapwg::wg_wide_ok(d, vec, math)) ? snapwg::wg_plan_wide(d) : wg_plan(d, vec);
  WgradArgs a;
  a.d = d;
  a.x = x; a.dy = dy; a.partial = static_cast<float*>(workspace);
  a.gn_mu = gn_mu; a.gn_sc = gn_sc; a.gn_beta = gn_beta;
  a.M = d.N * d.Ho * d.Wo;
  a.K = d.KH * d.KW * d.Cin;
  a.ctiles = p.ctiles; a.ncol = p.ncol; a.slabs_per_chunk = p.slabs_per_chunk;
  a.rows_z = rows_z; a.rows_dy = rows_dy; a.row_count = row_count;
  a.x_is_half = x_is_half ? 1 : 0;
  a.dy_is_half = dy_is_half ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (fused3) {
    rc = snapwg::launch_3x3(a, p, math == SNAP_MATH_F16, s);
  } else if (vec && math != SNAP_MATH_F32) {
    rc = snapwg::launch_bf16(a, p, math == SNAP_MATH_F16, s);
  } else if (vec) {
    if (p.bkt == 128) rc = p.bn == 128 ? wg_launch_pro<128, 128, true>(a, p, s) : wg_launch_pro<128, 64, true>(a, p, s);
    else rc = p.bn == 128 ? wg_launch_pro<64, 128, true>(a, p, s) : wg_launch_pro<64, 64, true>(a, p, s);
  } else {
    rc = p.bn == 128 ? wg_launch_pro<64, 128, false>(a, p, s) : wg_launch_pro<64, 64, false>(a, p, s);
  }
  if (rc != SNAP_OK) return rc;
  const int64_t total = (int64_t)a.K * d.Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)snap_cdiv(total, 64)), dim3(256), 0, s,
                     (const float*)a.partial, p.S, total, dw, accumulate);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
