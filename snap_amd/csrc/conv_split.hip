// Implicit-GEMM convolution / dense engine with f32-GRADE accuracy on the bf16 matrix cores:
// every f32 operand is split into NS bf16 parts (hi = bf16(v), mid = bf16(v - hi),
// lo = bf16(v - hi - mid); each subtraction is exact in f32) and the product a*b is taken as the
// sum of the part products whose weight is above the f32 rounding level, accumulated in f32 by
// v_mfma_f32_32x32x16_bf16:
//   NS = 2 ("bf16x3"): a_lo b_hi + a_hi b_lo + a_hi b_hi        per-product error ~ 2^-17
//   NS = 3 ("bf16x6"): a_lo b_hi + a_hi b_lo + a_mid b_mid
//                      + a_mid b_hi + a_hi b_mid + a_hi b_hi     per-product error ~ 2^-24
// (NS = 3 keeps 24 significand bits per operand, i.e. the operands themselves are exact; what is
// dropped -- a_mid b_lo, a_lo b_mid, a_lo b_lo -- is below 2^-24 of the product.)  The bf16 matrix
// cores run at 16x the rate of v_mfma_f32_32x32x2_f32, so six products still leave 2.7x the f32
// matrix peak (417 TFLOP/s f32-equivalent), three leave 5.3x.  The exact-f32 engine of
// conv_igemm.hip stays selectable; this one replaces flax.linen.Conv / Dense
// (snap/models/resnet.py:73-132, image_encoder.py:67-94, layers.py:55-78) on the same footing.
//
// Same GEMM view, fusions, tile order, split-K and epilogue as the other two engines
// (conv_common.h).  Operand pipeline:
//   * A (im2col rows): global -> registers (float4), fused prologue in f32, split into NS bf16
//     images, each stored as [row][16 k] (32 B per row; the two 16-byte k-octets XOR-swizzled by
//     (row >> 3) & 1 so the MFMA fragment fetch -- one ds_read_b128 per lane -- is conflict-free).
//   * B: the weights are split once by snap_conv2d_pack_weights_split_bf16 into
//     [part][Cout][tap][cin8] bf16 and go global -> LDS by LDS-DMA (no registers, no VALU), with
//     the swizzle applied on the source side.
//   * one K slab = 16 k = one MFMA k-step of NS(NS+1)/2 products per 32x32 tile; double-buffered
//     LDS, one barrier per slab (96 MFMA-cycles x TM x TN at NS = 3 between barriers).
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NS>
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4 (&out)[NS]) {
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    out[p] = __builtin_convertvector(r, bf16x4);            // v_cvt_pk_bf16_f32, RNE
    if (p + 1 < NS) {
      const f32x4 back = __builtin_convertvector(out[p], f32x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = r[e] - back[e];    // exact
    }
  }
}

template <int BM, int BN, int PRO, int NS>
__device__ __forceinline__ void conv_split_body(const ConvArgs& a) {
  constexpr int BK = 16;
  constexpr int TM = BM / 64;
  constexpr int TN = BN / 64;
  constexpr int QPR = BK / 4;             // float4 quads per A row of the slab
  constexpr int RPP = 256 / QPR;          // rows staged per pass (64)
  constexpr int AROWS = BM / RPP;         // float4 per thread for A
  constexpr int A_PART = BM * 32;         // bytes per part image per stage
  constexpr int B_PART = BN * 32;
  constexpr int A_ST = NS * A_PART;       // bytes per stage
  constexpr int B_ST = NS * B_PART;
  constexpr int BSLOTS = NS * BN * 2;     // 16-byte DMA pieces per stage
  constexpr int BPIECES = (BSLOTS + 255) / 256;
  constexpr int kSlabBytes = 2 * (A_ST + B_ST);
  constexpr int kStageBytes = 64 * BN * 4;
  constexpr int kSmemBytes = kSlabBytes > kStageBytes ? kSlabBytes : kStageBytes;
  __shared__ __attribute__((aligned(16))) float smem[kSmemBytes / 4];
  char* const Ab = reinterpret_cast<char*>(smem);
  char* const Bb = Ab + 2 * A_ST;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  // tile order / split-K / row-indexed mode: as conv_igemm.hip
  const int ncol = a.ncol;
  const int split = a.ksplit > 1 ? blockIdx.x / a.tiles_per_split : 0;
  const int bid = a.ksplit > 1 ? blockIdx.x - split * a.tiles_per_split : blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int col_t = seq % ncol;
  const int row_t = (seq / ncol) * 8 + xcd;
  const int Meff = a.row_count ? min(*a.row_count, a.M) : a.M;
  if (row_t * BM >= Meff) return;
  const int m0 = row_t * BM;
  const int n0 = col_t * BN;
  const int HoWo = d.Ho * d.Wo;
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);

  int r_hb[AROWS], r_wb[AROWS];
  bool r_ok[AROWS];
  const float* r_px[AROWS];
  int64_t r_gn[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int row = (tid / QPR) + RPP * i;
    const int m = m0 + row;
    r_ok[i] = m < Meff;
    int mm = r_ok[i] ? m : 0;
    if (a.rows_in) mm = a.rows_in[mm];
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_hb[i] = ho * d.stride - d.pad_t;
    r_wb[i] = wo * d.stride - d.pad_l;
    r_px[i] = a.x + (((int64_t)n * d.H + r_hb[i]) * d.W + r_wb[i]) * d.Cin_stride;
    r_gn[i] = (int64_t)n * d.Cin;
  }
  const int akq = tid % QPR;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 xa[AROWS], xmu[AROWS], xsc[AROWS], xbeta;
  bool xin[AROWS];
  int cur_c = 0;

  const int taps = d.KH * d.KW;
  const int kt_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;
  const int kt_end = a.ksplit > 1 ? min(a.nk, kt_begin + a.slabs_per_split) : a.nk;
  int kpos = 0, ct = 0, kh = 0, kw = 0;
  if (kt_begin > 0) {
    kpos = kt_begin / a.ctiles;
    ct = kt_begin - kpos * a.ctiles;
    kh = kpos / d.KW;
    kw = kpos - kh * d.KW;
  }
  const float* tap_px[AROWS];
  bool tap_in[AROWS];
  auto set_tap = [&]() {
    const int64_t delta = ((int64_t)kh * d.W + kw) * d.Cin_stride;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int hi = r_hb[i] + kh, wi = r_wb[i] + kw;
      tap_in[i] = r_ok[i] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
      tap_px[i] = r_px[i] + delta;
    }
  };
  set_tap();

  auto load_a = [&]() {
    const int c = ct * BK + 4 * akq;
    cur_c = c;
    const bool cvalid = c < d.Cin;
    if constexpr (need_gn) xbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + (cvalid ? c : 0));
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const bool inb = tap_in[i] && cvalid;
      xin[i] = inb;
      const float* px = inb ? tap_px[i] + c : a.x;
      xa[i] = *reinterpret_cast<const f32x4*>(px);
      if constexpr (need_gn) {
        const int64_t so = inb ? r_gn[i] + c : (int64_t)0;
        xmu[i] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
        xsc[i] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
      }
    }
  };
  const __bf16* const wt = static_cast<const __bf16*>(a.w_bf16);
  const int64_t part_stride = (int64_t)d.Cout * taps * a.cin8;
  auto issue_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      const int slot = tid + 256 * p;
      if (BSLOTS % 256 != 0 && slot >= BSLOTS) break;       // wave-uniform (BSLOTS % 64 == 0)
      const int part = slot / (2 * BN);
      const int rem = slot - part * (2 * BN);
      const int col = rem >> 1;
      const int oct = (rem & 1) ^ ((col >> 3) & 1);         // logical k-octet held by this slot
      const int kc = ct * BK + 8 * oct;
      const bool ok = kc < a.cin8 && (n0 + col) < d.Cout;
      const void* src =
          ok ? static_cast<const void*>(wt + part * part_stride +
                                        ((int64_t)(n0 + col) * taps + kpos) * a.cin8 + kc)
             : static_cast<const void*>(kZeroChunk);
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                       (lds_void_t*)(Bb + buf * B_ST + 16 * slot), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++ct == a.ctiles) {
      ct = 0;
      ++kpos;
      if (++kw == d.KW) { kw = 0; ++kh; }
      set_tap();
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = (tid / QPR) + RPP * i;
      f32x4 v = xa[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv;
        if constexpr (need_gn)
          pv = apply_pro<PRO>(v[e], xmu[i][e], xsc[i][e], xbeta[e], d.in_scale, d.in_shift);
        else
          pv = apply_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        v[e] = (xin[i] && (cur_c + e < d.Cin)) ? pv : 0.f;
      }
      bf16x4 parts[NS];
      split_bf16<NS>(v, parts);
      const int oct = (akq >> 1) ^ ((row >> 3) & 1);
      char* dst = Ab + buf * A_ST + row * 32 + oct * 16 + (akq & 1) * 8;
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<bf16x4*>(dst + p * A_PART) = parts[p];
    }
  };

  const int l31 = lane & 31, lhi = lane >> 5;
  if (kt_begin < kt_end) {
    load_a();
    issue_b(0);
    advance();
    store_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) {
      load_a();
      issue_b(cur ^ 1);
      advance();
    }
    const char* as = Ab + cur * A_ST;
    const char* bs = Bb + cur * B_ST;
    bf16x8 av[TM][NS], bv[TN][NS];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int R = wr * (BM / 2) + i * 32 + l31;
      const char* p0 = as + R * 32 + ((lhi ^ ((R >> 3) & 1)) * 16);
#pragma unroll
      for (int p = 0; p < NS; ++p) av[i][p] = *reinterpret_cast<const bf16x8*>(p0 + p * A_PART);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int C = wc * (BN / 2) + j * 32 + l31;
      const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
#pragma unroll
      for (int p = 0; p < NS; ++p) bv[j][p] = *reinterpret_cast<const bf16x8*>(p0 + p * B_PART);
    }
    // smallest terms first; the four (i, j) accumulators interleave so that two MFMAs on the
    // same accumulator are TM*TN issues apart
#define SNAP_SPLIT_PRODUCT(PA, PB)                                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA], bv[j][PB], acc[i][j], 0, 0, 0);
    if constexpr (NS == 3) {
      SNAP_SPLIT_PRODUCT(2, 0)
      SNAP_SPLIT_PRODUCT(0, 2)
      SNAP_SPLIT_PRODUCT(1, 1)
      SNAP_SPLIT_PRODUCT(1, 0)
      SNAP_SPLIT_PRODUCT(0, 1)
      SNAP_SPLIT_PRODUCT(0, 0)
    } else {
      SNAP_SPLIT_PRODUCT(1, 0)
      SNAP_SPLIT_PRODUCT(0, 1)
      SNAP_SPLIT_PRODUCT(0, 0)
    }
#undef SNAP_SPLIT_PRODUCT
    if (more) store_a(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the B octets of slab kt+1 landed
    __syncthreads();
  }

  conv_epilogue<BM, BN>(a, acc, smem, m0, n0, Meff, row_t, split);
}

template <int BM, int BN, int PRO, int NS>
__global__ __launch_bounds__(256) void conv_split_kernel(const ConvArgs a) {
  conv_split_body<BM, BN, PRO, NS>(a);
}

template <int BM, int BN, int PRO, int NS>
int launch(ConvArgs a, hipStream_t s) {
  constexpr int BK = 16;
  a.ctiles = (a.d.Cin + BK - 1) / BK;
  a.nk = a.d.KH * a.d.KW * a.ctiles;
  const int64_t nrow = snap_cdiv(a.M, BM);
  a.ncol = (int)snap_cdiv(a.d.Cout, BN);
  a.gn_slabs = (a.d.Ho * a.d.Wo) / BM + 2;
  int64_t nblocks = snap_cdiv(nrow, 8) * 8 * a.ncol;
  if (nblocks > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  a.ksplit = 1;
  a.tiles_per_split = (int)nblocks;
  a.slabs_per_split = a.nk;
  const int64_t tiles = nrow * a.ncol;
  const int target = splitk_target();
  if (a.kpartial && target > 0 && tiles <= splitk_max_tiles() && a.nk >= 16 && !a.rows_in &&
      !a.rows_out && !a.row_count && !a.gn_partial &&
      !(a.d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) {
    int64_t S = (target + tiles - 1) / tiles;
    S = S < a.nk / 8 ? S : a.nk / 8;                        // >= 8 slabs (128 k) per split
    const int64_t fit = (int64_t)(a.kpartial_bytes / ((size_t)a.M * a.d.Cout * sizeof(float)));
    S = S < fit ? S : fit;
    if (S >= 2) {
      a.slabs_per_split = (int)((a.nk + S - 1) / S);
      a.ksplit = (a.nk + a.slabs_per_split - 1) / a.slabs_per_split;
      nblocks *= a.ksplit;
    }
  }
  hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  if (a.ksplit > 1) {
    const int64_t total4 = (int64_t)a.M * (a.d.Cout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0, s,
                       (const float*)a.kpartial, a.ksplit, (int64_t)a.M, a.d.Cout, a.d.Cout_stride,
                       a.d.epilogue, a.bias, a.residual, a.row_mask, a.y);
    SNAP_CHECK_LAUNCH();
  }
  return SNAP_OK;
}

template <int BM, int BN, int NS>
int launch_pro(const ConvArgs& a, hipStream_t s) {
  switch (a.d.prologue) {
    case SNAP_PRO_NONE: return launch<BM, BN, SNAP_PRO_NONE, NS>(a, s);
    case SNAP_PRO_AFFINE: return launch<BM, BN, SNAP_PRO_AFFINE, NS>(a, s);
    case SNAP_PRO_GN_RELU: return launch<BM, BN, SNAP_PRO_GN_RELU, NS>(a, s);
    case SNAP_PRO_RELU_GN: return launch<BM, BN, SNAP_PRO_RELU_GN, NS>(a, s);
    case SNAP_PRO_RELU: return launch<BM, BN, SNAP_PRO_RELU, NS>(a, s);
    default: return SNAP_ERR_UNSUPPORTED;
  }
}

template <int NS>
int launch_tile(const ConvArgs& a, hipStream_t s) {
  const TileChoice t = choose_tile(a.M, a.d.Cout);
  if (t.bm == 128 && t.bn == 128) return launch_pro<128, 128, NS>(a, s);
  if (t.bm == 128) return launch_pro<128, 64, NS>(a, s);
  if (t.bn == 128) return launch_pro<64, 128, NS>(a, s);
  return launch_pro<64, 64, NS>(a, s);
}

// w [taps*Cin, Cout] f32 -> out [parts][Cout][taps][cin8] bf16: part 0 = bf16(w) (RNE), part p =
// bf16 of the exact f32 residual left by parts 0..p-1; channels Cin..cin8 zero.  One 32 x 32
// (k x n) tile per workgroup through LDS: coalesced along n on the way in, along k on the way out.
__global__ __launch_bounds__(256) void pack_weights_split_kernel(
    const float* __restrict__ w, __bf16* __restrict__ out, int taps, int Cin, int cin8, int Cout,
    int parts) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32, t = blockIdx.z;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    tile[j][tx] = (c < Cin && n < Cout) ? w[((int64_t)t * Cin + c) * Cout + n] : 0.f;
  }
  __syncthreads();
  const int64_t part_stride = (int64_t)Cout * taps * cin8;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, c = c0 + tx;
    if (n < Cout && c < cin8) {
      float r = tile[tx][j];
      __bf16* o = out + ((int64_t)n * taps + t) * cin8 + c;
      for (int p = 0; p < parts; ++p) {
        const __bf16 b = (__bf16)r;
        o[p * part_stride] = b;
        r -= (float)b;
      }
    }
  }
}

}  // namespace

int snapconv::launch_split(ConvArgs a, int parts, hipStream_t s) {
  if (parts == 2) return launch_tile<2>(a, s);
  if (parts == 3) return launch_tile<3>(a, s);
  return SNAP_ERR_UNSUPPORTED;
}

extern "C" size_t snap_conv2d_packed_weights_split_bytes(int32_t taps, int32_t Cin, int32_t Cout,
                                                         int32_t parts) {
  if (parts < 1 || parts > 3) return 0;
  return (size_t)parts * snap_conv2d_packed_weights_bytes(taps, Cin, Cout);
}

extern "C" int snap_conv2d_pack_weights_split_bf16(const float* w, int32_t taps, int32_t Cin,
                                                   int32_t Cout, int32_t parts, void* out,
                                                   size_t out_bytes, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  if (taps <= 0 || Cin <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  if (parts < 1 || parts > 3) return SNAP_ERR_UNSUPPORTED;
  if (out_bytes < snap_conv2d_packed_weights_split_bytes(taps, Cin, Cout, parts)) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(out) & 15) return SNAP_ERR_BAD_SHAPE;
  const int cin8 = (Cin + 7) / 8 * 8;
  const dim3 grid((unsigned)snap_cdiv(cin8, 32), (unsigned)snap_cdiv(Cout, 32), (unsigned)taps);
  hipLaunchKernelGGL(pack_weights_split_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, static_cast<__bf16*>(out), taps, Cin, cin8, Cout, parts);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
