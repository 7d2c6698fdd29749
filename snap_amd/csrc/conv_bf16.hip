// Implicit-GEMM convolution / dense engine with bf16 operands and f32 accumulation
// (v_mfma_f32_32x32x16_bf16) -- the training-precision analogue of the reference's float16
// train config (snap/configs/train_localization.py:25, trainer.py:391 DynamicScale); the
// exact-f32 engine of conv_igemm.hip stays the inference / parity path.
//
// Same GEMM view, fusions, tile order, split-K and epilogue as conv_igemm.hip (shared through
// conv_common.h).  What differs is the operand pipeline:
//   * tensors stay f32 in HBM.  A (im2col rows) is gathered global -> registers, the fused
//     prologue runs in f32, the result is rounded to bf16 (v_cvt_pk_bf16_f32, RNE) and stored
//     as [row][32 k] (64 B per row, 16-byte k-octets XOR-swizzled by (row >> 2) & 3): one
//     conflict-free ds_read_b128 fetches a lane's 8 consecutive k of the MFMA A fragment.
//   * B comes from a bf16 copy of the weights packed [Cout][tap][cin8] (k contiguous per
//     output channel, snap_conv2d_pack_weights_bf16) and goes global -> LDS by LDS-DMA with the
//     same swizzle applied on the source side: no registers, no conversion, no store phase.
//   * one K slab = 32 k = two MFMA k-steps; double-buffered LDS, one barrier per slab.
// Products of bf16 values are exact in f32, so the result equals an f32-accumulated dot product
// of the rounded operands (up to summation order): the parity tests compare against exactly
// that restatement.
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Element type of the rounded operands: bf16 (default) or IEEE half (F16: the reference's
// dtype=float16 train config, train_localization.py:93 -- 11 significand bits instead of 8, the
// exponent range of half: values beyond 65504 round to inf and reach the trainer's non-finite
// check, gradients below 6e-8 flush -- which is what DynamicScale is for, trainer.py:391-392).
template <bool F16> struct Elem;
template <> struct Elem<false> {
  typedef __bf16 T; typedef bf16x8 x8; typedef bf16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<true> {
  typedef _Float16 T; typedef f16x8 x8; typedef f16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// GNT (as conv_split.hip): the GroupNorm operands of the slab's 32 channels -- mean and rstd * gamma of the (at
// most two) images a row tile of BM <= Ho Wo pixels touches, and beta -- come from a 640-byte LDS table that 40
// lanes fetch by LDS-DMA two slabs ahead (three-slot ring; the slab's closing barrier publishes it), instead of
// two float4 global loads per staged ROW: those were two thirds of the loop's vector-memory instructions.
template <int BM, int BN, int PRO, bool F16, bool YH = false, bool GNT = false>
__device__ __forceinline__ void conv_bf16_body(const ConvArgs& a) {
  typedef Elem<F16> E;
  typedef typename E::T ET;
  typedef typename E::x8 etx8;
  typedef typename E::x4 etx4;
  constexpr int BK = 32;
  constexpr int TM = BM / 64;
  constexpr int TN = BN / 64;
  constexpr int QPR = BK / 4;             // float4 quads per A row of the slab
  constexpr int RPP = 256 / QPR;          // rows staged per pass
  constexpr int AROWS = BM / RPP;         // float4 per thread for A
  constexpr int BPIECES = (BN * 4) / 256; // 16-byte DMA pieces per thread for B (BN cols x 4 octets)
  constexpr int A_ST = BM * 16;           // floats per A stage (64 B per row)
  constexpr int B_ST = BN * 16;
  constexpr int kSlabFloats = 2 * (A_ST + B_ST);
  constexpr int kStageFloats = 64 * BN;
  constexpr int kSmemFloats = kSlabFloats > kStageFloats ? kSlabFloats : kStageFloats;
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  constexpr bool gn_tab = need_gn && GNT;
  constexpr int kGnRing = 640;            // bytes per table: [mu n0 | sc n0 | mu n1 | sc n1 | beta] x 32 channels
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats + (gn_tab ? 3 * kGnRing / 4 : 0)];
  char* const Ab = reinterpret_cast<char*>(smem);
  char* const Bb = reinterpret_cast<char*>(smem + 2 * A_ST);
  char* const Gt = reinterpret_cast<char*>(smem + kSmemFloats);

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
  const int n_first = m0 / HoWo;
  const int m_split = (n_first + 1) * HoWo;   // first row of the tile's second image
  const int n_second = min(n_first + 1, d.N - 1);

  int r_hb[AROWS], r_wb[AROWS];
  bool r_ok[AROWS];
  const float* r_px[AROWS];
  int64_t r_gn[AROWS];
  int r_slot[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int row = (tid / QPR) + RPP * i;
    const int m = m0 + row;
    r_slot[i] = m >= m_split ? 1 : 0;
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
    if constexpr (need_gn && !gn_tab) xbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + (cvalid ? c : 0));
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const bool inb = tap_in[i] && cvalid;
      xin[i] = inb;
      const float* px = inb ? tap_px[i] + c : a.x;
      xa[i] = *reinterpret_cast<const f32x4*>(px);
      if constexpr (need_gn && !gn_tab) {
        const int64_t so = inb ? r_gn[i] + c : (int64_t)0;
        xmu[i] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
        xsc[i] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
      }
    }
  };
  const ET* const wt = static_cast<const ET*>(a.w_bf16);
  auto issue_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      const int slot = tid + 256 * p;
      const int col = slot >> 2;
      const int oct = (slot & 3) ^ ((col >> 2) & 3);    // logical k-octet held by this slot
      const int kc = ct * BK + 8 * oct;
      const bool ok = kc < a.cin8 && (n0 + col) < d.Cout;
      const void* src = ok ? static_cast<const void*>(wt + ((int64_t)(n0 + col) * taps + kpos) * a.cin8 + kc)
                           : static_cast<const void*>(kZeroChunk);
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                       (lds_void_t*)(Bb + buf * (B_ST * 4) + 16 * slot), 16, 0, 0);
    }
  };
  // GroupNorm table of channel tile `ctile` -> ring slot `ring` (lanes 0..39 of wave 0)
  auto issue_gn = [&](int ring, int ctile) {
    if constexpr (gn_tab) {
      if (tid < 40) {
        const int seg = tid >> 3;
        const int c = ctile * BK + 4 * (tid & 7);
        const float* base = seg == 4 ? a.gn_beta
                                     : ((seg & 1) ? a.gn_sc : a.gn_mu) +
                                           (int64_t)(seg >= 2 ? n_second : n_first) * d.Cin;
        const void* src = c < d.Cin ? static_cast<const void*>(base + c)
                                    : static_cast<const void*>(kZeroChunk);
        __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                         (lds_void_t*)(Gt + ring * kGnRing + 16 * tid), 16, 0, 0);
      }
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
  auto store_a = [&](int buf, int ring) {
    const float* const tb = reinterpret_cast<const float*>(Gt + ring * kGnRing);
    if constexpr (gn_tab) xbeta = *reinterpret_cast<const f32x4*>(tb + 128 + 4 * akq);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = (tid / QPR) + RPP * i;
      f32x4 v = xa[i];
      if constexpr (gn_tab) {
        xmu[i] = *reinterpret_cast<const f32x4*>(tb + r_slot[i] * 64 + 4 * akq);
        xsc[i] = *reinterpret_cast<const f32x4*>(tb + r_slot[i] * 64 + 32 + 4 * akq);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv;
        if constexpr (need_gn)
          pv = apply_pro<PRO>(v[e], xmu[i][e], xsc[i][e], xbeta[e], d.in_scale, d.in_shift);
        else
          pv = apply_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        v[e] = (xin[i] && (cur_c + e < d.Cin)) ? pv : 0.f;
      }
      const etx4 b = __builtin_convertvector(v, etx4);
      const int oct = (akq >> 1) ^ ((row >> 2) & 3);
      *reinterpret_cast<etx4*>(Ab + buf * (A_ST * 4) + row * 64 + oct * 16 + (akq & 1) * 8) = b;
    }
  };

  const int l31 = lane & 31, lhi = lane >> 5;
  int g_ct = ct;                 // channel tile of the next table to fetch
  auto next_gct = [&]() { if (++g_ct == a.ctiles) g_ct = 0; };
  int ring_cur = 0;              // ring slot of the slab being multiplied
  if (kt_begin < kt_end) {
    if constexpr (gn_tab) {
      issue_gn(0, g_ct); next_gct();
      issue_gn(1, g_ct); next_gct();
    }
    load_a();
    issue_b(0);
    advance();
    if constexpr (gn_tab) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();           // tables of slabs 0 and 1 visible to every wave
    }
    store_a(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    const int ring_next = ring_cur == 2 ? 0 : ring_cur + 1;
    if (more) {
      load_a();
      issue_b(cur ^ 1);
      advance();
      if constexpr (gn_tab) {
        if (kt + 2 < kt_end) { issue_gn(ring_next == 2 ? 0 : ring_next + 1, g_ct); next_gct(); }
      }
    }
    const char* as = Ab + cur * (A_ST * 4);
    const char* bs = Bb + cur * (B_ST * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      etx8 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int R = wr * (BM / 2) + i * 32 + l31;
        av[i] = *reinterpret_cast<const etx8*>(as + R * 64 + (((2 * s + lhi) ^ ((R >> 2) & 3)) * 16));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int C = wc * (BN / 2) + j * 32 + l31;
        bv[j] = *reinterpret_cast<const etx8*>(bs + C * 64 + (((2 * s + lhi) ^ ((C >> 2) & 3)) * 16));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = E::mfma(av[i], bv[j], acc[i][j]);
    }
    if (more) store_a(cur ^ 1, ring_next);
    ring_cur = ring_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the B octets of slab kt+1 landed
    __syncthreads();
  }

  conv_epilogue<BM, BN, false, 256, false, YH ? (F16 ? 2 : 1) : 0>(a, acc, smem, m0, n0, Meff, row_t, split);
}

// The same GEMM with the A operand ALREADY in the engine's element type in HBM (ConvArgs.x_half:
// [N, H, W, Cin_stride] bf16 / f16, Cin_stride % 8 == 0, prologue NONE): the data-gradient
// convolutions of the training step, whose input -- the gradient a GroupNorm VJP just wrote -- has
// a half-precision twin (snap_group_norm_bwd_ex_f32).  Both operands then travel global -> LDS by
// LDS-DMA in 16-byte pieces (A: one piece = 8 channels of one tap of one output row, the k-octet
// swizzle applied on the source side exactly as for B; a tap outside the image or a row beyond M
// fetches the zero chunk): no staging registers, no conversion, no store phase, half the bytes.
// Same products, same k order and accumulation as conv_bf16_body on the rounded operand: the
// results are bit-identical to the f32-input launch of the same tensor (tested).
template <int BM, int BN, bool F16, bool GNB = false>
__device__ __forceinline__ void conv_bf16_xh_body(const ConvArgs& a) {
  typedef Elem<F16> E;
  typedef typename E::T ET;
  typedef typename E::x8 etx8;
  constexpr int BK = 32;
  constexpr int TM = BM / 64;
  constexpr int TN = BN / 64;
  constexpr int APIECES = (BM * 4) / 256;
  constexpr int BPIECES = (BN * 4) / 256;
  constexpr int A_ST = BM * 16;           // floats per A stage (64 B per row)
  constexpr int B_ST = BN * 16;
  constexpr int NST = 3;                  // ring depth
  constexpr int kSlabFloats = NST * (A_ST + B_ST);
  constexpr int kStageFloats = 64 * BN;
  constexpr int kSmemFloats = kSlabFloats > kStageFloats ? kSlabFloats : kStageFloats;
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];
  char* const Ab = reinterpret_cast<char*>(smem);
  char* const Bb = reinterpret_cast<char*>(smem + NST * A_ST);

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int ncol = a.ncol;
  const int split = a.ksplit > 1 ? blockIdx.x / a.tiles_per_split : 0;
  const int bid = a.ksplit > 1 ? blockIdx.x - split * a.tiles_per_split : blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int col_t = seq % ncol;
  const int row_t = (seq / ncol) * 8 + xcd;
  const int Meff = a.row_count ? min(*a.row_count, a.M) : a.M;     // (compact row buffers: the first *row_count rows)
  if (row_t * BM >= Meff) return;
  const int m0 = row_t * BM;
  const int n0 = col_t * BN;
  const int HoWo = d.Ho * d.Wo;
  const ET* const xh = static_cast<const ET*>(a.x_half);

  // A pieces of this thread: slot = tid + 256 p -> row slot >> 2, physical octet slot & 3
  int r_hb[APIECES], r_wb[APIECES], r_oct[APIECES];
  bool r_ok[APIECES];
  const ET* r_px[APIECES];
#pragma unroll
  for (int p = 0; p < APIECES; ++p) {
    const int slot = tid + 256 * p;
    const int row = slot >> 2;
    const int m = m0 + row;
    r_ok[p] = m < Meff;
    const int mm = r_ok[p] ? m : 0;
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_hb[p] = ho * d.stride - d.pad_t;
    r_wb[p] = wo * d.stride - d.pad_l;
    r_oct[p] = (slot & 3) ^ ((row >> 2) & 3);     // logical k-octet this LDS slot holds
    r_px[p] = xh + (((int64_t)n * d.H + r_hb[p]) * d.W + r_wb[p]) * d.Cin_stride + 8 * r_oct[p];
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
  const ET* const wt = static_cast<const ET*>(a.w_bf16);
  auto issue = [&](int buf) {
    const int64_t delta = ((int64_t)kh * d.W + kw) * d.Cin_stride + ct * BK;
#pragma unroll
    for (int p = 0; p < APIECES; ++p) {
      const int slot = tid + 256 * p;
      const int hi = r_hb[p] + kh, wi = r_wb[p] + kw;
      const bool ok = r_ok[p] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W &&
                      (ct * BK + 8 * r_oct[p]) < d.Cin;          // (Cin % 8 == 0: whole octets)
      const void* src = ok ? static_cast<const void*>(r_px[p] + delta) : static_cast<const void*>(kZeroChunk);
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                       (lds_void_t*)(Ab + buf * (A_ST * 4) + 16 * slot), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      const int slot = tid + 256 * p;
      const int col = slot >> 2;
      const int oct = (slot & 3) ^ ((col >> 2) & 3);
      const int kc = ct * BK + 8 * oct;
      const bool ok = kc < a.cin8 && (n0 + col) < d.Cout;
      const void* src = ok ? static_cast<const void*>(wt + ((int64_t)(n0 + col) * taps + kpos) * a.cin8 + kc)
                           : static_cast<const void*>(kZeroChunk);
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                       (lds_void_t*)(Bb + buf * (B_ST * 4) + 16 * slot), 16, 0, 0);
    }
    if (++ct == a.ctiles) {
      ct = 0;
      ++kpos;
      if (++kw == d.KW) { kw = 0; ++kh; }
    }
  };

  const int l31 = lane & 31, lhi = lane >> 5;
  const int nslab = kt_end - kt_begin;
  // three-stage ring: slabs i + 1 and i + 2 are in flight while slab i is multiplied; every thread
  // issues the same APIECES + BPIECES loads per slab, so the waits are compile-time counts
  if (nslab > 0) issue(0);
  if (nslab > 1) issue(1);
  for (int i = 0; i < nslab; ++i) {
    const int cur = i % NST;
    if (i + 1 < nslab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APIECES + BPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // slab i landed everywhere; stage (i + 2) % 3 was consumed in iteration i - 1
    if (i + 2 < nslab) issue((i + 2) % NST);
    const char* as = Ab + cur * (A_ST * 4);
    const char* bs = Bb + cur * (B_ST * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      etx8 av[TM], bv[TN];
#pragma unroll
      for (int ii = 0; ii < TM; ++ii) {
        const int R = wr * (BM / 2) + ii * 32 + l31;
        av[ii] = *reinterpret_cast<const etx8*>(as + R * 64 + (((2 * s + lhi) ^ ((R >> 2) & 3)) * 16));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int C = wc * (BN / 2) + j * 32 + l31;
        bv[j] = *reinterpret_cast<const etx8*>(bs + C * 64 + (((2 * s + lhi) ^ ((C >> 2) & 3)) * 16));
      }
#pragma unroll
      for (int ii = 0; ii < TM; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[ii][j] = E::mfma(av[ii], bv[j], acc[ii][j]);
    }
  }
  __syncthreads();                        // the ring is drained: the epilogue reuses it as its staging tile
  conv_epilogue<BM, BN, false, 256, false, 0, GNB>(a, acc, smem, m0, n0, Meff, row_t, split);
}

// GNB: the epilogue also emits the statistics of the GroupNorm VJP this gradient feeds (ConvArgs.gnb_*)
template <int BM, int BN, bool F16, bool GNB = false>
__global__ __launch_bounds__(256) void conv_bf16_xh_kernel(const ConvArgs a) {
  conv_bf16_xh_body<BM, BN, F16, GNB>(a);
}

template <int BM, int BN, int PRO, bool F16 = false, bool YH = false, bool GNT = false>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvArgs a) {
  conv_bf16_body<BM, BN, PRO, F16, YH, GNT>(a);
}

template <int BM, int BN, int PRO>
int launch(ConvArgs a, hipStream_t s) {
  constexpr int BK = 32;
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
  if (a.kpartial && target > 0 && tiles <= splitk_max_tiles() && a.nk >= 8 && !a.rows_in &&
      !a.rows_out && !a.row_count && (!a.gn_partial || (a.gn_rows32 && a.d.Cout_stride == a.d.Cout)) && !a.y_half &&
      !a.gnb_mode && !(a.d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) {
    int64_t S = (target + tiles - 1) / tiles;
    S = S < a.nk / 4 ? S : a.nk / 4;                        // >= 4 slabs (128 k) per split
    const int64_t fit = (int64_t)(a.kpartial_bytes / ((size_t)a.M * a.d.Cout * sizeof(float)));
    S = S < fit ? S : fit;
    if (S >= 2) {
      a.slabs_per_split = (int)((a.nk + S - 1) / S);
      a.ksplit = (a.nk + a.slabs_per_split - 1) / a.slabs_per_split;
      nblocks *= a.ksplit;
    }
  }
  if (a.gn_partial && a.gn_rows32 && a.ksplit == 1) return SNAP_ERR_WORKSPACE;   // (the caller sized for a split-K launch)
  if (a.x_half) {
    if constexpr (PRO == SNAP_PRO_NONE) {
      if (a.gnb_mode) {
        if (a.ksplit > 1 || !a.gn_partial) return SNAP_ERR_UNSUPPORTED;
        if (a.half)
          hipLaunchKernelGGL((conv_bf16_xh_kernel<BM, BN, true, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
        else
          hipLaunchKernelGGL((conv_bf16_xh_kernel<BM, BN, false, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
      } else if (a.half)
        hipLaunchKernelGGL((conv_bf16_xh_kernel<BM, BN, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((conv_bf16_xh_kernel<BM, BN, false>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
    } else {
      return SNAP_ERR_UNSUPPORTED;
    }
  } else if (a.y_half) {
    // half (also / only) output: the dense layers of the masked MLP (prologue NONE / RELU)
    if constexpr (PRO == SNAP_PRO_NONE || PRO == SNAP_PRO_RELU) {
      if (a.ksplit > 1) return SNAP_ERR_UNSUPPORTED;
      if (a.half)
        hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, true, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, false, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
    } else {
      return SNAP_ERR_UNSUPPORTED;
    }
  } else {
    constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
    // the LDS statistics table needs "a row tile touches at most two images"
    const bool table = need_gn && a.d.Ho * a.d.Wo >= BM && !a.rows_in && !a.no_plain;   // (SNAP_TUNE_NO_PLAIN: A/B)
    if constexpr (need_gn) {
      if (table) {
        if (a.half)
          hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, true, false, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
        else
          hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, false, false, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
      }
    }
    if (!table) {
      if (a.half)
        hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, true>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, PRO, false>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
    }
  }
  SNAP_CHECK_LAUNCH();
  // (the reduce pass also emits the GroupNorm partial sums of a launch that owes them: per 32-row slab)
  if (a.ksplit > 1) return launch_splitk_reduce(a, s);
  return SNAP_OK;
}

template <int BM, int BN>
int launch_pro(const ConvArgs& a, hipStream_t s) {
  switch (a.d.prologue) {
    case SNAP_PRO_NONE: return launch<BM, BN, SNAP_PRO_NONE>(a, s);
    case SNAP_PRO_AFFINE: return launch<BM, BN, SNAP_PRO_AFFINE>(a, s);
    case SNAP_PRO_GN_RELU: return launch<BM, BN, SNAP_PRO_GN_RELU>(a, s);
    case SNAP_PRO_RELU_GN: return launch<BM, BN, SNAP_PRO_RELU_GN>(a, s);
    case SNAP_PRO_RELU: return launch<BM, BN, SNAP_PRO_RELU>(a, s);
    default: return SNAP_ERR_UNSUPPORTED;
  }
}

// w [taps*Cin, Cout] f32 -> out [Cout][taps][cin8] bf16 (RNE), channels Cin..cin8 zero.
// One 32 x 32 (k x n) tile per workgroup through LDS: coalesced along n on the way in,
// along k on the way out.
template <typename ET>
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(
    const float* __restrict__ w, ET* __restrict__ out, int taps, int Cin, int cin8, int Cout) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32, t = blockIdx.z;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    tile[j][tx] = (c < Cin && n < Cout) ? w[((int64_t)t * Cin + c) * Cout + n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, c = c0 + tx;
    if (n < Cout && c < cin8) out[((int64_t)n * taps + t) * cin8 + c] = (ET)tile[tx][j];
  }
}

// every weight image of a training step in ONE launch (the forward images and, items with taps < 0,
// the ROTATED images the data-gradient convolutions read: out [Cin4][taps][cout8] = w[T-1-t][ci][co],
// i.e. the image of w.flip(0, 1).permute(0, 1, 3, 2) padded to four output channels -- both sides
// contiguous along co, no transposition through LDS).  Items sorted by block_begin.
template <typename ET>
__global__ __launch_bounds__(256) void pack_weights_bf16_multi_kernel(const SnapPackItem* __restrict__ items,
                                                                      int n_items) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SnapPackItem it = items[lo];
  const bool rot = it.taps < 0;
  const int taps = rot ? -it.taps : it.taps;
  const int Cin = it.Cin, Cout = it.Cout;
  const int local = blockIdx.x - it.block_begin;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  ET* const out = static_cast<ET*>(it.out);
  if (!rot) {
    const int cin8 = (Cin + 7) / 8 * 8;
    const int gx = (cin8 + 31) / 32, gy = (Cout + 31) / 32;
    const int t = local / (gx * gy), rem = local - t * (gx * gy);
    const int c0 = (rem % gx) * 32, n0 = (rem / gx) * 32;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, n = n0 + tx;
      tile[j][tx] = (c < Cin && n < Cout) ? it.w[((int64_t)t * Cin + c) * Cout + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
      const int n = n0 + j, c = c0 + tx;
      if (n < Cout && c < cin8) out[((int64_t)n * taps + t) * cin8 + c] = (ET)tile[tx][j];
    }
  } else {
    const int cin4 = (Cin + 3) / 4 * 4, cout8 = (Cout + 7) / 8 * 8;
    const int gx = (cout8 + 31) / 32, gy = (cin4 + 31) / 32;
    const int t = local / (gx * gy), rem = local - t * (gx * gy);
    const int co0 = (rem % gx) * 32, ci0 = (rem / gx) * 32;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
      const int ci = ci0 + j, co = co0 + tx;
      if (ci < cin4 && co < cout8) {
        const float v = (ci < Cin && co < Cout) ? it.w[((int64_t)(taps - 1 - t) * Cin + ci) * Cout + co] : 0.f;
        out[((int64_t)ci * taps + t) * cout8 + co] = (ET)v;
      }
    }
  }
}

}  // namespace

extern "C" int32_t snap_conv2d_pack_weights_blocks(int32_t taps, int32_t Cin, int32_t Cout) {
  if (taps == 0 || Cin <= 0 || Cout <= 0) return 0;
  const bool rot = taps < 0;
  const int T = rot ? -taps : taps;
  if (rot) return ((((Cin + 3) / 4 * 4) + 31) / 32) * ((((Cout + 7) / 8 * 8) + 31) / 32) * T;
  return ((((Cin + 7) / 8 * 8) + 31) / 32) * ((Cout + 31) / 32) * T;
}

extern "C" int snap_conv2d_pack_weights_multi_bf16(const SnapPackItem* items, int32_t n_items,
                                                   int32_t total_blocks, void* stream) {
  if (!items) return SNAP_ERR_NULL;
  if (n_items <= 0 || total_blocks <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pack_weights_bf16_multi_kernel<__bf16>, dim3((unsigned)total_blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, n_items);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_conv2d_pack_weights_multi_f16(const SnapPackItem* items, int32_t n_items,
                                                  int32_t total_blocks, void* stream) {
  if (!items) return SNAP_ERR_NULL;
  if (n_items <= 0 || total_blocks <= 0) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pack_weights_bf16_multi_kernel<_Float16>, dim3((unsigned)total_blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, n_items);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

int snapconv::launch_bf16(ConvArgs a, hipStream_t s) {
  const TileChoice t = choose_tile(a.M, a.d.Cout, a.d.tile_hint, desc_k(a.d));
  if (t.bm == 128 && t.bn == 128) return launch_pro<128, 128>(a, s);
  if (t.bm == 128) return launch_pro<128, 64>(a, s);
  if (t.bn == 128) return launch_pro<64, 128>(a, s);
  return launch_pro<64, 64>(a, s);
}

extern "C" size_t snap_conv2d_packed_weights_bytes(int32_t taps, int32_t Cin, int32_t Cout) {
  if (taps <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const size_t cin8 = ((size_t)Cin + 7) / 8 * 8;
  return (size_t)Cout * taps * cin8 * 2;
}

static int pack_weights_one(const float* w, int32_t taps, int32_t Cin, int32_t Cout, void* out,
                            size_t out_bytes, bool half, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  if (taps <= 0 || Cin <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  if (out_bytes < snap_conv2d_packed_weights_bytes(taps, Cin, Cout)) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(out) & 15) return SNAP_ERR_BAD_SHAPE;
  const int cin8 = (Cin + 7) / 8 * 8;
  const dim3 grid((unsigned)snap_cdiv(cin8, 32), (unsigned)snap_cdiv(Cout, 32), (unsigned)taps);
  if (half)
    hipLaunchKernelGGL(pack_weights_bf16_kernel<_Float16>, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       w, static_cast<_Float16*>(out), taps, Cin, cin8, Cout);
  else
    hipLaunchKernelGGL(pack_weights_bf16_kernel<__bf16>, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       w, static_cast<__bf16*>(out), taps, Cin, cin8, Cout);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_conv2d_pack_weights_bf16(const float* w, int32_t taps, int32_t Cin,
                                             int32_t Cout, void* out, size_t out_bytes,
                                             void* stream) {
  return pack_weights_one(w, taps, Cin, Cout, out, out_bytes, false, stream);
}

extern "C" int snap_conv2d_pack_weights_f16(const float* w, int32_t taps, int32_t Cin,
                                            int32_t Cout, void* out, size_t out_bytes,
                                            void* stream) {
  return pack_weights_one(w, taps, Cin, Cout, out, out_bytes, true, stream);
}
