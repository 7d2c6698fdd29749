// Weight-gradient engine with bf16 operands and f32 accumulation
// (v_mfma_f32_32x32x16_bf16): dW[k, co] = sum_m bf16(Z[m, k]) * bf16(dY[m, co]) -- the
// training-precision companion of conv_bf16.hip (reference analogue: the float16 train
// config, snap/configs/train_localization.py:25); wgrad.hip stays the exact path.
//
// Same tiling, M split and fixed-order reduction as wgrad.hip.  The reduction runs over the
// output pixels m, so both MFMA fragments need 8 CONSECUTIVE m of one channel / one output
// column: the loader transposes in registers.  A thread owns one channel quad and 4
// consecutive rows m (lane % 8 picks the row group, lane / 8 the quad: every row is read in
// 128-byte runs), applies the prologue in f32, rounds, and writes each channel's 4 values as
// one ds_write_b64 into a [channel][32 m] image with an 80-byte row stride -- 5 x 16 bytes,
// so both the 8-byte transposing stores and the ds_read_b128 fragment fetches are
// conflict-free without a swizzle.
#include "wgrad_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// element type of the rounded operands (as conv_bf16.hip): bf16, or IEEE half for SNAP_MATH_F16
template <bool F16> struct Elem;
template <> struct Elem<false> {
  typedef bf16x8 x8; typedef bf16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<true> {
  typedef f16x8 x8; typedef f16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// ZH / DH: the Z / dY operand is read in the engine's element type (2 bytes) instead of f32: 8-byte
// row quads, transposed with byte permutes (two v_perm_b32 per output dword) instead of converted --
// half the bytes of that operand, no rounding work (the values ARE the rounded ones)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

//
// NT = 512 (eight waves, 4 x 2): the WIDE tile of the flat (Dense) kernel gradients -- a whole 256 x 256 (or
// 256 x 128) dW per workgroup, so a chunk of rows is read ONCE (the 128 x 128 tiling reads Z per column tile
// and dY per channel tile: 2.3 x the bytes at 256 x 256), and the row-list entries travel one slab ahead of
// the rows they address (one memory round trip per slab instead of two).
template <int BKT, int BN, int PRO, bool F16 = false, bool ZH = false, bool DH = false, int NT = 256>
__global__ __launch_bounds__(NT) void wgrad_bf16_kernel(const WgradArgs a) {
  static_assert(!ZH || PRO == SNAP_PRO_NONE, "a half Z operand has no prologue");
  static_assert(BKT / 4 <= NT / 8 && BN / 4 <= NT / 8, "one channel quad per loader thread");
  typedef Elem<F16> E;
  typedef typename E::x8 etx8;
  typedef typename E::x4 etx4;
  constexpr int RS = 32;                   // reduction slab (output pixels) = two MFMA k-steps
  constexpr int RSB = 80;                  // LDS row stride in bytes (32 bf16 + 16 B pad)
  constexpr int WR = NT / 128;             // wave rows (x 2 wave columns)
  constexpr int ZROWS = BKT / WR;          // channels per wave row
  constexpr int TM = ZROWS / 32, TN = BN / 64;
  constexpr int ZQ = BKT / 4, DQ = BN / 4; // float4 quads per row
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  constexpr int Z_ST = BKT * RSB, D_ST = BN * RSB;   // bytes per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * Z_ST + 2 * D_ST];
  char* const Zs0 = smem;
  char* const Ds0 = smem + 2 * Z_ST;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int kt = blockIdx.x / a.ncol, col_t = blockIdx.x - kt * a.ncol;
  const int kpos = kt / a.ctiles, ct = kt - kpos * a.ctiles;
  const int kh = kpos / d.KW, kw = kpos - kh * d.KW;
  const int c0 = ct * BKT;
  const int n0 = col_t * BN;
  const int HoWo = d.Ho * d.Wo;
  const int64_t Meff = a.row_count ? min((int64_t)*a.row_count, (int64_t)a.M) : (int64_t)a.M;
  // chunk of the M axis owned by this workgroup (the plan counts 16-row slabs)
  const int64_t rpc = a.row_count ? ((((Meff + 15) / 16) + gridDim.y - 1) / gridDim.y) * 16
                                  : (int64_t)a.slabs_per_chunk * 16;
  const int64_t m_begin = (int64_t)blockIdx.y * rpc;
  const int64_t m_end = min(Meff, m_begin + rpc);
  const int nslab = m_end > m_begin ? (int)((m_end - m_begin + RS - 1) / RS) : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loader coordinates: row group (4 consecutive m) and quad
  const int mg = tid & 7;
  const int quad = tid >> 3;                  // 0..NT/8-1
  const bool z_on = quad < ZQ, d_on = quad < DQ;
  const int zc = c0 + 4 * quad;
  const int dcol = n0 + 4 * quad;
  f32x4 zr[4], zmu[2], zsc[2], zbeta = {0.f, 0.f, 0.f, 0.f};
  bool zfirst[4] = {true, true, true, true};
  bool zin[4];
  f32x4 dr[4];
  bool din[4];
  u32x2 zrh[4], drh[4];          // ZH / DH: the row quads as two dwords of packed 2-byte elements
  if constexpr (need_gn) {
    zbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + ((z_on && zc < d.Cin) ? zc : 0));
  }

  int rn[4], rho[4], rwo[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int64_t m = m_begin + 4 * mg + p;
    const int mm = (int)min(m, (int64_t)a.M - 1);
    rn[p] = mm / HoWo;
    const int r = mm - rn[p] * HoWo;
    rho[p] = r / d.Wo;
    rwo[p] = r - rho[p] * d.Wo;
  }
  auto advance_rows = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rwo[p] += RS;
      while (rwo[p] >= d.Wo) { rwo[p] -= d.Wo; ++rho[p]; }
      while (rho[p] >= d.Ho) { rho[p] -= d.Ho; ++rn[p]; }
    }
  };

  // wide plan: row-list entries of a slab, fetched one slab ahead of the rows they address (the 128 x 128
  // tiles sit right at 128 registers = four workgroups per CU and read the entries in place)
  constexpr bool IDX = NT == 512;
  int zidx[IDX ? 4 : 1] = {0}, didx[IDX ? 4 : 1] = {0};
  auto load_idx = [&](int sl) {
    if constexpr (IDX) {
      const int64_t ms = m_begin + (int64_t)sl * RS + 4 * mg;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int64_t m = ms + p;
        if (a.rows_z) zidx[p] = (m < m_end && z_on) ? a.rows_z[m] : 0;
        if (a.rows_dy) didx[p] = (m < m_end && d_on) ? a.rows_dy[m] : 0;
      }
    }
  };

  auto load_slab = [&](int sl) {
    const int64_t ms = m_begin + (int64_t)sl * RS + 4 * mg;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t m = ms + p;
      const bool mok = m < m_end;
      const int n = rn[p], ho = rho[p], wo = rwo[p];
      const int hi = ho * d.stride - d.pad_t + kh, wi = wo * d.stride - d.pad_l + kw;
      const bool inb = mok && z_on && zc < d.Cin && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
      zin[p] = inb;
      int64_t off = inb ? (((int64_t)n * d.H + hi) * d.W + wi) * d.Cin_stride + zc : (int64_t)0;
      if (a.rows_z) off = inb ? (int64_t)(IDX ? zidx[IDX ? p : 0] : a.rows_z[m]) * d.Cin_stride + zc : (int64_t)0;
      if constexpr (ZH)
        zrh[p] = inb ? *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.x) + off) : u32x2{0u, 0u};
      else
        zr[p] = *reinterpret_cast<const f32x4*>(a.x + off);
      if constexpr (need_gn) {
        // the thread's four rows are consecutive output pixels: they lie in at most two images (Ho Wo >= 4,
        // checked by the entry point) -- the table rows of the first and the last row's image, a select per row
        // (half the table loads; they were half of the loop's vector-memory instructions)
        if (p == 0 || p == 3) {
          const bool zok = z_on && zc < d.Cin;
          const int64_t so = zok ? (int64_t)min(n, d.N - 1) * d.Cin + zc : (int64_t)0;   // (rows past the end: masked)
          zmu[p ? 1 : 0] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
          zsc[p ? 1 : 0] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
        }
        zfirst[p] = n == rn[0];
      }
      const bool ok = mok && d_on && dcol < d.Cout;
      din[p] = ok;
      const int64_t drow = (ok && a.rows_dy) ? (int64_t)(IDX ? didx[IDX ? p : 0] : a.rows_dy[m]) : m;
      if constexpr (DH)
        drh[p] = ok ? *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.dy) + drow * d.Cout_stride + dcol)
                    : u32x2{0u, 0u};
      else
        dr[p] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? drow * d.Cout_stride + dcol : (int64_t)0));
    }
  };

  auto store_slab = [&](int buf) {
    char* zs = Zs0 + buf * Z_ST;
    char* ds = Ds0 + buf * D_ST;
    // element e of the four rows' packed quads -> one 8-byte (4 consecutive m) store per channel
    auto transpose_store = [&](const u32x2 (&r)[4], char* base) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;   // high / low halves of (src0, src1)
        const int w = e >> 1;
        u32x2 o;
        o[0] = __builtin_amdgcn_perm(r[1][w], r[0][w], sel);        // rows 0, 1
        o[1] = __builtin_amdgcn_perm(r[3][w], r[2][w], sel);        // rows 2, 3
        *reinterpret_cast<u32x2*>(base + (4 * quad + e) * RSB + mg * 8) = o;
      }
    };
    f32x4 zv[4], dv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (!ZH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pv;
          if constexpr (need_gn)
            pv = wg_pro<PRO>(zr[p][e], zfirst[p] ? zmu[0][e] : zmu[1][e], zfirst[p] ? zsc[0][e] : zsc[1][e], zbeta[e],
                             d.in_scale, d.in_shift);
          else
            pv = wg_pro<PRO>(zr[p][e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
          zv[p][e] = (zin[p] && (zc + e < d.Cin)) ? pv : 0.f;
        }
      }
      if constexpr (!DH) dv[p] = din[p] ? dr[p] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (z_on) {
      if constexpr (ZH) {
        transpose_store(zrh, zs);        // (Cin % 4 == 0 for a half operand: whole quads or nothing)
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 t = {zv[0][e], zv[1][e], zv[2][e], zv[3][e]};   // 4 consecutive m of channel e
          *reinterpret_cast<etx4*>(zs + (4 * quad + e) * RSB + mg * 8) = __builtin_convertvector(t, etx4);
        }
      }
    }
    if (d_on) {
      if constexpr (DH) {
        transpose_store(drh, ds);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 t = {dv[0][e], dv[1][e], dv[2][e], dv[3][e]};
          *reinterpret_cast<etx4*>(ds + (4 * quad + e) * RSB + mg * 8) = __builtin_convertvector(t, etx4);
        }
      }
    }
  };

  if (nslab > 0) {
    load_idx(0);
    load_slab(0);
    advance_rows();
    if (nslab > 1) load_idx(1);
    store_slab(0);
  }
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    const int cur = sl & 1;
    const bool more = sl + 1 < nslab;
    if (more) {
      load_slab(sl + 1);
      advance_rows();
      if (sl + 2 < nslab) load_idx(sl + 2);
    }
    const char* zs = Zs0 + cur * Z_ST;
    const char* ds = Ds0 + cur * D_ST;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      etx8 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[i] = *reinterpret_cast<const etx8*>(zs + (wr * ZROWS + i * 32 + l31) * RSB + (2 * s + lhi) * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[j] = *reinterpret_cast<const etx8*>(ds + (wc * (BN / 2) + j * 32 + l31) * RSB + (2 * s + lhi) * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = E::mfma(av[i], bv[j], acc[i][j]);
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
      const int c = c0 + wr * ZROWS + i * 32 + ri;
      if (c >= d.Cin) continue;
      const int64_t krow = (int64_t)kpos * d.Cin + c;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 32 + l31;
        if (col < d.Cout) out[krow * d.Cout + col] = acc[i][j][r];
      }
    }
}

template <int BKT, int BN, int PRO>
int wg_launch(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  const dim3 grid((unsigned)(p.ktiles * p.ncol), (unsigned)p.S);
  if (a.x_is_half || a.dy_is_half) {
    // half operands (the masked MLP's hidden activations / inter-layer gradients): one of the two
    if (a.x_is_half && a.dy_is_half) return SNAP_ERR_UNSUPPORTED;
    if (a.x_is_half) {
      if constexpr (PRO == SNAP_PRO_NONE) {
        if (half) hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, true, true, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, false, true, false>), grid, dim3(256), 0, s, a);
      } else {
        return SNAP_ERR_UNSUPPORTED;
      }
    } else {
      // (every prologue: the ResNet's kernel gradients read the half twin of the GroupNorm VJP's output)
      if (half) hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, true, false, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, false, false, true>), grid, dim3(256), 0, s, a);
    }
  } else if (half)
    hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((wgrad_bf16_kernel<BKT, BN, PRO, false>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

template <int BKT, int BN>
int wg_launch_pro(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  switch (a.d.prologue) {
    case SNAP_PRO_NONE: return wg_launch<BKT, BN, SNAP_PRO_NONE>(a, p, half, s);
    case SNAP_PRO_AFFINE: return wg_launch<BKT, BN, SNAP_PRO_AFFINE>(a, p, half, s);
    case SNAP_PRO_GN_RELU: return wg_launch<BKT, BN, SNAP_PRO_GN_RELU>(a, p, half, s);
    case SNAP_PRO_RELU_GN: return wg_launch<BKT, BN, SNAP_PRO_RELU_GN>(a, p, half, s);
    case SNAP_PRO_RELU: return wg_launch<BKT, BN, SNAP_PRO_RELU>(a, p, half, s);
    default: return SNAP_ERR_UNSUPPORTED;
  }
}



// the wide tiles (wg_plan_wide): prologues NONE / RELU / AFFINE
template <int BN, int PRO, bool ZH, bool DH>
int wg_launch_wide(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  const dim3 grid((unsigned)(p.ktiles * p.ncol), (unsigned)p.S);
  if (half) hipLaunchKernelGGL((wgrad_bf16_kernel<256, BN, PRO, true, ZH, DH, 512>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((wgrad_bf16_kernel<256, BN, PRO, false, ZH, DH, 512>), grid, dim3(512), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

template <int BN>
int wg_launch_wide_pro(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  if (a.x_is_half && a.dy_is_half) return SNAP_ERR_UNSUPPORTED;
  const int pro = a.d.prologue;
  if (a.x_is_half) return pro == SNAP_PRO_NONE ? wg_launch_wide<BN, SNAP_PRO_NONE, true, false>(a, p, half, s)
                                                : SNAP_ERR_UNSUPPORTED;
  if (a.dy_is_half) {
    if (pro == SNAP_PRO_NONE) return wg_launch_wide<BN, SNAP_PRO_NONE, false, true>(a, p, half, s);
    if (pro == SNAP_PRO_RELU) return wg_launch_wide<BN, SNAP_PRO_RELU, false, true>(a, p, half, s);
    return SNAP_ERR_UNSUPPORTED;
  }
  if (pro == SNAP_PRO_NONE) return wg_launch_wide<BN, SNAP_PRO_NONE, false, false>(a, p, half, s);
  if (pro == SNAP_PRO_RELU) return wg_launch_wide<BN, SNAP_PRO_RELU, false, false>(a, p, half, s);
  if (pro == SNAP_PRO_AFFINE) return wg_launch_wide<BN, SNAP_PRO_AFFINE, false, false>(a, p, half, s);
  return SNAP_ERR_UNSUPPORTED;
}

}  // namespace

int snapwg::launch_bf16(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  if (p.bkt == 256) return p.bn == 256 ? wg_launch_wide_pro<256>(a, p, half, s) : wg_launch_wide_pro<128>(a, p, half, s);
  if (p.bkt == 128)
    return p.bn == 128 ? wg_launch_pro<128, 128>(a, p, half, s) : wg_launch_pro<128, 64>(a, p, half, s);
  return p.bn == 128 ? wg_launch_pro<64, 128>(a, p, half, s) : wg_launch_pro<64, 64>(a, p, half, s);
}
