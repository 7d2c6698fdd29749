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
//   * B: the weights are split once by snap_conv2d_pack_weights_split_bf16 into an image laid out
//     as the LDS stages ([column tile][tap][channel tile][part][column][16 k], swizzle applied at
//     rest) and go global -> LDS by LDS-DMA as contiguous 1 KB runs (no registers, no VALU).
//   * one K slab = 16 k = one MFMA k-step of NS(NS+1)/2 products per 32x32 tile; double-buffered
//     LDS, one barrier per slab (96 MFMA-cycles x TM x TN at NS = 3 between barriers).
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Four f32 -> NS x four bf16 (packed two per dword).  Each part is one v_cvt_pk_bf16_f32 (RNE)
// per element PAIR; the value it represents is recovered by a shift / mask of the packed dword
// and subtracted exactly in f32 (3 VALU per element and part instead of the 4 the generic
// vector conversion costs).
template <int NS>
__device__ __forceinline__ void split_bf16(const f32x4& v, u32x2 (&out)[NS]) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < NS; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 pr = {r[2 * h], r[2 * h + 1]};
      const bf16x2 b = __builtin_convertvector(pr, bf16x2);
      unsigned u;
      __builtin_memcpy(&u, &b, 4);
      out[p][h] = u;
      if (p + 1 < NS) {
        r[2 * h] = r[2 * h] - __uint_as_float(u << 16);               // exact
        r[2 * h + 1] = r[2 * h + 1] - __uint_as_float(u & 0xffff0000u);
      }
    }
  }
}

// GNT: the GroupNorm operands (mean, rstd*gamma per (image, channel); beta per channel) of the
// slab's 16 channels come from a small LDS table instead of two float4 global loads per staged
// ROW: a row tile of BM <= Ho*Wo pixels touches at most two images, so 20 lanes fetch
// [mu n0 | sc n0 | mu n1 | sc n1 | beta] (320 B) by LDS-DMA two slabs ahead (3-deep ring; the
// slab's closing barrier publishes it).  Per-row statistics loads were 2/3 of the bytes the
// vector-memory path moved per slab (they hit L1, but the path itself delivers 64 B/clk/CU).
// TAIL: Cin is not a multiple of 4 (the 257-channel fusion MLP): the last channel quad needs a
// per-element mask; otherwise one flag per 16-byte chunk does.
// ROOT: the 7 x 7 / stride 2 / pad 3 root convolution of an RGB image stored with FOUR floats per
// pixel (RGB + a zero): a K slab is then 4 consecutive pixels x 4 floats of one kernel row, i.e.
// 16 CONTIGUOUS floats, and the kernel row's 7 taps are two slabs (kw = 0..3 and 4..7, tap 7 and
// channel 3 carry zero weights: snap_conv2d_pack_weights_split_root_bf16).  The launch sets KW = 2
// "virtual taps" per kernel row; a thread's quad IS one pixel, so bounds are per thread.  K = 14 x
// 16 = 224 for 147 real products -- against Cin = 3 padded to a 16-k slab PER TAP (49 slabs).
// PLAIN: a 1 x 1 / stride 1 / unpadded convolution over whole 16-channel slabs and consecutive rows
// (most launches of a ResNet): no tap can fall outside the image and no channel outside Cin, so
// the per-element selects, the bounds compares and the 64-bit address arithmetic of the general
// loader drop out -- rows beyond M are CLAMPED to the last row (their accumulators are garbage the
// epilogue never stores) and the A rows are fetched with buffer loads: a constant 32-bit byte
// offset per thread plus the slab's offset in the scalar operand.  Same values, same bits.
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
__device__ unsigned long long g_conv_timeline[1024 * 4 * 8];
#endif

template <int BM, int BN, int PRO, int NS, bool GNT, bool TAIL, bool ROOT = false, bool DUAL = false, bool PLAIN = false>
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
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  constexpr bool gn_tab = need_gn && GNT;
  constexpr int kGnRing = 320;            // bytes per table
  __shared__ __attribute__((aligned(16))) float smem[(kSmemBytes + (gn_tab ? 3 * kGnRing : 0)) / 4];
  char* const Ab = reinterpret_cast<char*>(smem);
  char* const Bb = Ab + 2 * A_ST;
  char* const Gt = Ab + kSmemBytes;

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

  int r_hb[AROWS], r_wb[AROWS];
  bool r_ok[AROWS];
  const float* r_px[AROWS];
  int64_t r_gn[AROWS];
  int r_slot[AROWS];
  int r_voff[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int row = (tid / QPR) + RPP * i;
    const int m = m0 + row;
    r_ok[i] = m < Meff;
    int mm = r_ok[i] ? m : (PLAIN ? Meff - 1 : 0);
    if (a.rows_in) mm = a.rows_in[mm];
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_hb[i] = ho * d.stride - d.pad_t;
    r_wb[i] = wo * d.stride - d.pad_l;
    r_px[i] = a.x + (((int64_t)n * d.H + r_hb[i]) * d.W + r_wb[i]) * d.Cin_stride;
    r_gn[i] = (int64_t)n * d.Cin;
    r_slot[i] = (PLAIN ? mm : m) >= m_split ? 1 : 0;
    if constexpr (PLAIN) r_voff[i] = (int)(((int64_t)(mm - m0) * d.Cin_stride + 4 * (tid % QPR)) * 4);
  }
  const int akq = tid % QPR;
  // PLAIN: the tile's rows through one buffer resource (window from the tile's first row)
  const int64_t x_left = ((int64_t)(Meff - m0) * d.Cin_stride) * 4;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (PLAIN ? (int64_t)m0 * d.Cin_stride : 0), 0,
      (int)(x_left < 0x7ff00000LL ? x_left : 0x7ff00000LL), 0x00020000);

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
    const int64_t delta = ((int64_t)kh * d.W + (ROOT ? 4 * kw : kw)) * d.Cin_stride;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int hi = r_hb[i] + kh, wi = r_wb[i] + (ROOT ? 4 * kw + (tid % QPR) : kw);
      tap_in[i] = r_ok[i] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
      tap_px[i] = r_px[i] + delta;
    }
  };
  set_tap();

  // Timing ablations of this loop (WRONG results; tools/conv_ablate_split.py) exist ONLY in an alt
  // build (scripts/build_alt.sh ... -DSNAP_CONV_SPLIT_ABLATE=1 pulls conv_split_ablate.inc in): the
  // product source carries no hook.  SNAP_ABL(bit) is a compile-time false here.
#if defined(SNAP_CONV_SPLIT_ABLATE) && SNAP_CONV_SPLIT_ABLATE
  const int ablate = a.ablate;       // bit0 no A loads, bit1 no B DMA, bit2 no MFMAs, bit3 no prologue /
#define SNAP_ABL(bit) (ablate & (bit))   // split math, bit4 no A LDS stores, bit5 no fragment fetches
#else
#define SNAP_ABL(bit) false
#endif
  auto load_a = [&]() {
    if (SNAP_ABL(1)) return;
    if constexpr (PLAIN) {
      static_assert(!PLAIN || (GNT || !need_gn), "PLAIN takes the table variant of the GroupNorm prologue");
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_x, r_voff[i], ct * (BK * 4), 0);
        __builtin_memcpy(&xa[i], &r, 16);
      }
      return;
    }
    const int c = ct * BK + 4 * akq;
    cur_c = c;
    const bool cvalid = ROOT || c < d.Cin;
    if constexpr (need_gn && !gn_tab)
      xbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + (cvalid ? c : 0));
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
  // B: the packed weights are stored in exactly the order the LDS stage wants them --
  // [column tile of 128][tap][channel tile][part][column][two swizzled k-octets] -- so one slab of
  // one column tile is ONE contiguous block of NS * 4 KB and a wave's DMA instruction reads 1 KB of
  // consecutive bytes (8 L2 requests of 128 B instead of 32 of 32 B with a k-contiguous layout:
  // the L2 request rate, not its byte rate, bounded the loop).  The block of the next slab is
  // the next NS * 4 KB: the (tap, channel tile) walk of the K loop is the storage order.
  const char* const wt = static_cast<const char*>(a.w_bf16);
  const int64_t col_tile_bytes = (int64_t)taps * a.ctiles * (NS * 4096);
  const char* bsrc[BPIECES];
#pragma unroll
  for (int p = 0; p < BPIECES; ++p) {
    const int slot = tid + 256 * p;
    const int part = slot / (2 * BN);
    const int rem = slot - part * (2 * BN);
    const int gcol = n0 + (rem >> 1);                          // (padded columns hold zeros)
    bsrc[p] = wt + (gcol >> 7) * col_tile_bytes + (int64_t)kt_begin * (NS * 4096) +
              (part < NS ? part : 0) * 4096 + (gcol & 127) * 32 + (rem & 1) * 16;
  }
  auto issue_b = [&](int buf) {
    if (SNAP_ABL(2)) return;
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      const int slot = tid + 256 * p;
      if (BSLOTS % 256 != 0 && slot >= BSLOTS) break;       // wave-uniform (BSLOTS % 64 == 0)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)bsrc[p],
                                       (lds_void_t*)(Bb + buf * B_ST + 16 * slot), 16, 0, 0);
      bsrc[p] += NS * 4096;
    }
  };
  // GroupNorm table of channel tile `ctile` -> ring slot `ring` (lanes 0..19 of wave 0)
  const int n_second = min(n_first + 1, d.N - 1);
  auto issue_gn = [&](int ring, int ctile) {
    if constexpr (gn_tab) {
      if (tid < 20) {
        const int seg = tid >> 2;
        const int c = ctile * BK + 4 * (tid & 3);
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
    if (SNAP_ABL(16)) return;
#if defined(SNAP_CONV_SPLIT_ABLATE) && SNAP_CONV_SPLIT_ABLATE
#include "conv_split_ablate.inc"       // (bit 3: raw stores instead of prologue + split)
#endif
    const float* const tb = reinterpret_cast<const float*>(Gt + ring * kGnRing);
    if constexpr (gn_tab) xbeta = *reinterpret_cast<const f32x4*>(tb + 64 + 4 * akq);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = (tid / QPR) + RPP * i;
      f32x4 v = xa[i];
      if constexpr (gn_tab) {
        xmu[i] = *reinterpret_cast<const f32x4*>(tb + r_slot[i] * 32 + 4 * akq);
        xsc[i] = *reinterpret_cast<const f32x4*>(tb + r_slot[i] * 32 + 16 + 4 * akq);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv;
        if constexpr (need_gn)
          pv = apply_pro<PRO>(v[e], xmu[i][e], xsc[i][e], xbeta[e], d.in_scale, d.in_shift);
        else
          pv = apply_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        if constexpr (PLAIN)
          v[e] = pv;
        else if constexpr (TAIL)
          v[e] = (xin[i] && (cur_c + e < d.Cin)) ? pv : 0.f;
        else
          v[e] = xin[i] ? pv : 0.f;
      }
      u32x2 parts[NS];
      split_bf16<NS>(v, parts);
      const int oct = (akq >> 1) ^ ((row >> 3) & 1);
      char* dst = Ab + buf * A_ST + row * 32 + oct * 16 + (akq & 1) * 8;
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<u32x2*>(dst + p * A_PART) = parts[p];
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
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
  // ALT BUILD ONLY (tools/conv_timeline.py): where a wave's cycles go, phase by phase (s_memtime around
  // fenced phases: the fences themselves cost overlap, so read the SHARES, not the total)
  unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define TL_MARK(i) do { asm volatile("" ::: "memory"); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tl_acc[i] += tn_ - tl_t; tl_t = tn_; } while (0)
  unsigned long long tl_t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define TL_MARK(i) do {} while (0)
#endif
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    const int ring_next = ring_cur == 2 ? 0 : ring_cur + 1;
    TL_MARK(6);                       // loop control / back edge
    if (more) {
      load_a();
      issue_b(cur ^ 1);
      advance();
      if constexpr (gn_tab) {
        if (kt + 2 < kt_end) { issue_gn(ring_next == 2 ? 0 : ring_next + 1, g_ct); next_gct(); }
      }
    }
    TL_MARK(0);                       // issue of the loads / DMA of the next k-step
    const char* as = Ab + cur * A_ST;
    const char* bs = Bb + cur * B_ST;
    bf16x8 av[TM][NS], bv[TN][NS];
    if (!SNAP_ABL(32)) {
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
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(av[i][p]));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(bv[j][p]));
    }
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    TL_MARK(1);                       // fragment fetches landed
    // smallest terms first; the four (i, j) accumulators interleave so that two MFMAs on the
    // same accumulator are TM*TN issues apart
#define SNAP_SPLIT_PRODUCT(PA, PB)                                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA], bv[j][PB], acc[i][j], 0, 0, 0);
    if (!SNAP_ABL(4)) {
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
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NS; ++p) asm volatile("" ::"v"(av[i][p]));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < NS; ++p) asm volatile("" ::"v"(bv[j][p]));
    }
#undef SNAP_SPLIT_PRODUCT
    TL_MARK(2);                       // MFMA issue (12 x 32 cycles if the pipe is free)
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TL_MARK(3);                       // wait for the next k-step's A rows (+ B, table)
    if (more) store_a(cur ^ 1, ring_next);
    ring_cur = ring_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // B octets of slab kt+1 (+ table kt+2) landed
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    TL_MARK(4);                       // prologue + split + LDS stores
    __syncthreads();
    TL_MARK(5);                       // barrier
  }
#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
  if (PLAIN && lane == 0 && blockIdx.x < 1024) {
#pragma unroll
    for (int i = 0; i < 7; ++i) g_conv_timeline[(blockIdx.x * 4 + wid) * 8 + i] = tl_acc[i];
    g_conv_timeline[(blockIdx.x * 4 + wid) * 8 + 7] = (unsigned long long)(kt_end - kt_begin);
  }
#endif
#undef TL_MARK

  conv_epilogue<BM, BN, DUAL>(a, acc, smem, m0, n0, Meff, row_t, split);
}

// NS = 2 is held to 128 registers (64 accumulators + 64) so that four workgroups share a CU:
// single 3x3 layers lose ~10 %, the HBM-leaning 1x1 layers gain ~10 %, the C2 step's conv time
// as a whole 23.3 -> 22.1 ms.
// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 with the input staged ONCE per channel tile ("halo" body).
// The im2col body above fetches, normalises and splits every input pixel nine times (once per
// tap); its loop spends 41 % of its time on those loads and 16 % on the conversion
// (tools/conv_ablate_split.py).  Here the K loop runs channel tile OUTER, tap INNER: per channel
// tile the BM + 2W + 2 consecutive pixels (flattened (n, y, x) order) that the tile's nine taps
// touch are fetched, normalised (GroupNorm table of at most two images) and split ONCE into an
// LDS stage; tap (kh, kw) of output row r reads stage row r + kh W + kw.  A tap that falls
// outside the image would read the wrapped neighbour: the fragment is replaced by zeros per lane
// (the reference pads the NORMALISED tensor with zeros).  2.1x (W = 68) ... 1.3x (W = 17) one
// tap's worth of loads per channel tile instead of 9x.  Taken for W <= 79 (a stage of at most
// 288 rows, three workgroups per CU); at W = 128 / 136 the 3.1x stage and two workgroups per CU
// measured SLOWER than the im2col body (0.34 vs 0.29 ms, stage 1 of the C2 encoders).
// Measured at C2: 17 x 17 x 512 0.30 -> 0.20 ms, 34 x 34 x 256 0.225 -> 0.165, 68 x 68 x 128
// 0.23 -> 0.22, the conv family 15.4 -> 14.55 ms per step.  (Draining the stage loads at tap 0,
// two instead of three workgroups per CU, __syncthreads instead of the fence-less barrier: all
// within the run-to-run noise.)  Same tile order, weight image,
// accumulation order per channel... NOT the same k order as the im2col body (tap-major there,
// channel-tile-major here): results differ in the last bits, not in the error class.
// Epilogue, row tiling and GroupNorm statistics of the output: unchanged (conv_epilogue).
template <int BN, int PRO, int NS, int HPMAX>
__device__ __forceinline__ void conv3x3_halo_body(const ConvArgs& a) {
  constexpr int BM = 128, BK = 16;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int NPER = (HPMAX * 4 + 255) / 256;     // float4 per thread and channel tile
  constexpr int H_PART = HPMAX * 32;                // bytes per part image of a halo stage
  constexpr int H_ST = NS * H_PART;
  constexpr int B_PART = BN * 32;
  constexpr int B_ST = NS * B_PART;
  constexpr int BSLOTS = NS * BN * 2;
  constexpr int BPIECES = (BSLOTS + 255) / 256;
  constexpr int kSlabBytes = 2 * (H_ST + B_ST);
  constexpr int kStageBytes = 64 * BN * 4;
  constexpr int kSmemBytes = kSlabBytes > kStageBytes ? kSlabBytes : kStageBytes;
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  constexpr int kGnRing = 320;
  __shared__ __attribute__((aligned(16))) float smem[(kSmemBytes + (need_gn ? 2 * kGnRing : 0)) / 4];
  char* const Hb = reinterpret_cast<char*>(smem);
  char* const Bb = Hb + 2 * H_ST;
  char* const Gt = Hb + kSmemBytes;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ncol = a.ncol;
  // split-K (small-M layers): split `split` takes the channel tiles [ct_begin, ct_end), all nine taps
  const int split = a.ksplit > 1 ? blockIdx.x / a.tiles_per_split : 0;
  const int bid = a.ksplit > 1 ? blockIdx.x - split * a.tiles_per_split : blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int col_t = seq % ncol;
  const int row_t = (seq / ncol) * 8 + xcd;
  const int Meff = a.M;
  if (row_t * BM >= Meff) return;
  const int m0 = row_t * BM;
  const int n0 = col_t * BN;
  const int W = d.W, HW = d.H * d.W;
  const int n_first = m0 / HW;
  const int m_split = (n_first + 1) * HW;
  const int n_second = min(n_first + 1, d.N - 1);
  const int halo0 = m0 - W - 1;                       // flattened pixel of stage row 0
  const int HP = BM + 2 * W + 2;                      // stage rows in use (<= HPMAX)

  // per output row of this lane (two row tiles): 9-bit mask of the taps inside the image
  int tapmask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wr * 64 + i * 32 + l31;
    int mk = 0;
    if (m < Meff) {
      const int r = m % HW;
      const int y = r / W, x = r - y * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < d.H && xx >= 0 && xx < W) mk |= 1 << t;
      }
    }
    tapmask[i] = mk;
  }

  // halo staging: float4 f = tid + 256 j -> stage row f >> 2, channel quad f & 3
  // (addresses and flags are recomputed per channel tile: keeping them costs the registers that
  // decide between two and three workgroups per CU)
  const int Mtot = d.N * HW;
  auto halo_ok_row = [&](int hp) {
    const int mp = halo0 + hp;
    return hp < HP && mp >= 0 && mp < Mtot;
  };
  f32x4 xa[NPER];
  auto load_halo = [&](int ct) {
#pragma unroll
    for (int j = 0; j < NPER; ++j) {
      const int f = tid + 256 * j;
      const int hp = f >> 2;
      const bool ok = halo_ok_row(hp);
      const float* px = a.x + ((int64_t)(ok ? halo0 + hp : 0) * d.Cin_stride + 4 * (f & 3) + ct * BK);
      xa[j] = *reinterpret_cast<const f32x4*>(ok ? px : a.x);
    }
  };
  auto issue_gn = [&](int ring, int ct) {
    if constexpr (need_gn) {
      if (tid < 20) {
        const int seg = tid >> 2;
        const int c = ct * BK + 4 * (tid & 3);
        const float* base = seg == 4 ? a.gn_beta
                                     : ((seg & 1) ? a.gn_sc : a.gn_mu) +
                                           (int64_t)(seg >= 2 ? n_second : n_first) * d.Cin;
        __builtin_amdgcn_global_load_lds((cglobal_void_t*)(base + c),
                                         (lds_void_t*)(Gt + ring * kGnRing + 16 * tid), 16, 0, 0);
      }
    }
  };
  auto store_halo = [&](int buf, int ring) {
    const float* const tb = reinterpret_cast<const float*>(Gt + ring * kGnRing);
#pragma unroll
    for (int j = 0; j < NPER; ++j) {
      const int f = tid + 256 * j;
      const int hp = f >> 2, q = f & 3;
      if (NPER * 256 > HPMAX * 4 && hp >= HPMAX) continue;
      f32x4 v = xa[j];
      f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = mu, be = mu;
      const bool ok = halo_ok_row(hp);
      if constexpr (need_gn) {
        const int slot = halo0 + hp >= m_split ? 1 : 0;
        mu = *reinterpret_cast<const f32x4*>(tb + slot * 32 + 4 * q);
        sc = *reinterpret_cast<const f32x4*>(tb + slot * 32 + 16 + 4 * q);
        be = *reinterpret_cast<const f32x4*>(tb + 64 + 4 * q);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = apply_pro<PRO>(v[e], mu[e], sc[e], be[e], d.in_scale, d.in_shift);
        v[e] = ok ? pv : 0.f;
      }
      u32x2 parts[NS];
      split_bf16<NS>(v, parts);
      const int oct = (q >> 1) ^ ((hp >> 3) & 1);
      char* dst = Hb + buf * H_ST + hp * 32 + oct * 16 + (q & 1) * 8;
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<u32x2*>(dst + p * H_PART) = parts[p];
    }
  };

  // B: slab (tap, ct) of column tile n0 / 128 = block (tap * ctiles + ct) of the packed image
  const char* const wt = static_cast<const char*>(a.w_bf16);
  const int ctiles = a.ctiles;
  const int64_t col_tile_bytes = (int64_t)9 * ctiles * (NS * 4096);
  const char* bbase[BPIECES];
#pragma unroll
  for (int p = 0; p < BPIECES; ++p) {
    const int slot = tid + 256 * p;
    const int part = slot / (2 * BN);
    const int rem = slot - part * (2 * BN);
    const int gcol = n0 + (rem >> 1);
    bbase[p] = wt + (gcol >> 7) * col_tile_bytes + (part < NS ? part : 0) * 4096 + (gcol & 127) * 32 +
               (rem & 1) * 16;
  }
  auto issue_b = [&](int buf, int ct, int t) {
    const int64_t off = ((int64_t)t * ctiles + ct) * (NS * 4096);
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      const int slot = tid + 256 * p;
      if (BSLOTS % 256 != 0 && slot >= BSLOTS) break;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(bbase[p] + off),
                                       (lds_void_t*)(Bb + buf * B_ST + 16 * slot), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: table + halo of the first channel tile, weights of its slab (tap 0)
  const int ct_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;          // (here: channel tiles per split)
  const int ct_end = a.ksplit > 1 ? min(ctiles, ct_begin + a.slabs_per_split) : ctiles;
  issue_gn(0, ct_begin);
  load_halo(ct_begin);
  issue_b(0, ct_begin, 0);
  if constexpr (need_gn) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  store_halo(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int bcur = 0;
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    const int hb = (ct - ct_begin) & 1;
    const bool more_ct = ct + 1 < ct_end;
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      const bool last = !more_ct && t == 8;
      const bool fetch = t == 0 && more_ct;
      if (fetch) issue_gn(hb ^ 1, ct + 1);
      if (!last) issue_b(bcur ^ 1, t == 8 ? ct + 1 : ct, t == 8 ? 0 : t + 1);
      if (fetch) {
        // the next channel tile's pixels: issued AFTER this slab's DMAs (compiler barrier) so that
        // the slab-closing wait can leave exactly these NPER loads in flight
        asm volatile("" ::: "memory");
        load_halo(ct + 1);
        asm volatile("" ::: "memory");
      }
      const int kh = t / 3, kw = t - 3 * kh;
      const int toff = kh * W + kw;
      const char* hs = Hb + hb * H_ST;
      const char* bs = Bb + bcur * B_ST;
      bf16x8 av[TM][NS], bv[TN][NS];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int h = wr * 64 + i * 32 + l31 + toff;
        const char* p0 = hs + h * 32 + ((lhi ^ ((h >> 3) & 1)) * 16);
        const bool in = (tapmask[i] >> t) & 1;
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          u32x4 raw = *reinterpret_cast<const u32x4*>(p0 + p * H_PART);
          if (!in) raw = u32x4{0u, 0u, 0u, 0u};
          __builtin_memcpy(&av[i][p], &raw, 16);
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int C = wc * (BN / 2) + j * 32 + l31;
        const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
#pragma unroll
        for (int p = 0; p < NS; ++p) bv[j][p] = *reinterpret_cast<const bf16x8*>(p0 + p * B_PART);
      }
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
      // the next channel tile's pixels (fetched during tap 0) -> the other halo stage: nobody
      // reads it before the barrier that closes tap 8
      if (t == 2 && more_ct) store_halo(hb ^ 1, hb ^ 1);
      if (fetch)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      bcur ^= 1;
    }
  }
  conv_epilogue<BM, BN>(a, acc, smem, m0, n0, Meff, row_t, split);
}

template <int BN, int PRO, int NS, int HPMAX>
__global__ __launch_bounds__(256, NS == 2 ? 3 : 2) void conv3x3_halo_kernel(const ConvArgs a) {
  conv3x3_halo_body<BN, PRO, NS, HPMAX>(a);
}

// the halo body takes: 3x3, stride 1, pad 1, whole 16-channel tiles, plain row order (split-K by
// channel tiles),
// images at least as large as the stage (BM + 2W + 2 pixels: the stage then touches <= 2 images)
inline bool halo_ok(const ConvArgs& a) {
  const bool on = !a.no_halo;                      // (SNAP_TUNE_NO_HALO: im2col body for every 3x3)
  const SnapConvDesc& d = a.d;
  const int hp = 128 + 2 * d.W + 2;
  return on && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 &&
         d.Ho == d.H && d.Wo == d.W && d.Cin % 16 == 0 && !a.rows_in && !a.rows_out &&
         !a.row_count && hp <= 288 && d.H * d.W >= hp &&
         (int64_t)d.N * d.H * d.W * d.Cin_stride < 0x7fffffffLL &&
         (d.prologue == SNAP_PRO_GN_RELU || d.prologue == SNAP_PRO_NONE);
}

template <int BN, int PRO, int NS>
void launch_halo(const ConvArgs& a, dim3 grid, hipStream_t s) {
  const int hp = 128 + 2 * a.d.W + 2;
  if (hp <= 208)
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, PRO, NS, 208>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, PRO, NS, 288>), grid, dim3(256), 0, s, a);
}

template <int BM, int BN, int PRO, int NS, bool GNT, bool TAIL, bool DUAL = false, bool PLAIN = false>
__global__ __launch_bounds__(256, NS == 2 ? 4 : 3) void conv_split_kernel(const ConvArgs a) {
  conv_split_body<BM, BN, PRO, NS, GNT, TAIL, false, DUAL, PLAIN>(a);
}

template <int BN, int PRO, int NS>
__global__ __launch_bounds__(256, NS == 2 ? 4 : 3) void conv_split_root_kernel(const ConvArgs a) {
  conv_split_body<128, BN, PRO, NS, false, false, true>(a);
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
      !a.rows_out && !a.row_count && (!a.gn_partial || (a.gn_rows32 && a.d.Cout_stride == a.d.Cout)) &&
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
  if (a.gn_partial && a.gn_rows32 && a.ksplit == 1) return SNAP_ERR_WORKSPACE;   // (the caller sized for a split-K launch)
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  // the LDS statistics table needs "a row tile touches at most two images"
  const bool table_ok = !need_gn || (a.d.Ho * a.d.Wo >= BM && !a.rows_in);
  const bool tail = (a.d.Cin & 3) != 0;
  const dim3 grid((unsigned)nblocks);
  if constexpr (BM == 128 && (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_NONE)) {
    if (halo_ok(a)) {
      if (a.ksplit > 1) {
        // split by whole channel tiles (the halo body stages a channel tile once for its nine taps)
        const int per = (a.ctiles + a.ksplit - 1) / a.ksplit;
        ConvArgs h = a;
        h.slabs_per_split = per;
        h.ksplit = (a.ctiles + per - 1) / per;
        if (h.ksplit > 1) {
          const dim3 hgrid((unsigned)(h.tiles_per_split * h.ksplit));
          launch_halo<BN, PRO, NS>(h, hgrid, s);
          SNAP_CHECK_LAUNCH();
          return launch_splitk_reduce(h, s);
        }
        if (h.gn_partial) return SNAP_ERR_UNSUPPORTED;     // (cannot happen: nk >= 16 gives >= 2 channel tiles)
        h.ksplit = 1;
        h.slabs_per_split = h.nk;
        launch_halo<BN, PRO, NS>(h, dim3((unsigned)h.tiles_per_split), s);
        SNAP_CHECK_LAUNCH();
        return SNAP_OK;
      }
      launch_halo<BN, PRO, NS>(a, grid, s);
      SNAP_CHECK_LAUNCH();
      return SNAP_OK;
    }
  }
  // statistics of y AND of relu(y) (ConvArgs.gn_partial2): one kernel variant, the shape of the
  // closing 1 x 1 conv of a ResNet stage; anywhere else the second buffer is left alone and
  // *gn_partial2_done stays 0 (the caller then takes the stand-alone statistics pass)
  bool dual = false;
  if constexpr (BM == 128 && BN == 128 && PRO == SNAP_PRO_GN_RELU)
    dual = a.gn_partial2 != nullptr && a.gn_partial != nullptr && table_ok && a.ksplit == 1;
  if (a.gn_partial2_done) *a.gn_partial2_done = dual ? 1 : 0;
  if constexpr (BM == 128 && BN == 128 && PRO == SNAP_PRO_GN_RELU) {
    if (dual) {
      hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS, true, false, true>), grid, dim3(256), 0, s, a);
      SNAP_CHECK_LAUNCH();
      return SNAP_OK;
    }
  }
  // the lean loader for 1 x 1 / stride 1 / unpadded layers over whole slabs (conv_split_body: PLAIN)
  const bool plain = !a.no_plain && a.d.KH == 1 && a.d.KW == 1 && a.d.stride == 1 && a.d.pad_t == 0 &&
                     a.d.pad_l == 0 && a.d.H == a.d.Ho && a.d.W == a.d.Wo && a.d.Cin % 16 == 0 &&
                     (a.d.Cin_stride & 3) == 0 && !a.rows_in && !a.row_count && a.M > 0 &&
                     (int64_t)BM * a.d.Cin_stride * 4 < 0x7ff00000LL;
  // K >= 256 with a GroupNorm prologue: the raw rows through an LDS ring, converted at fragment fetch
  // (conv_raw.hip: two k-steps in flight instead of one memory round trip per k-step; same bits)
  if constexpr (BM == 128 && NS == 2 && (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN)) {
    if (plain && table_ok && snapconv::raw_ok(a, BM, BN, PRO)) {
      const int rc = snapconv::launch_raw(a, BN, PRO, grid, s);
      if (rc != SNAP_OK) return rc;
      return a.ksplit > 1 ? launch_splitk_reduce(a, s) : SNAP_OK;
    }
  }
  if constexpr (NS == 2 && (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_NONE || PRO == SNAP_PRO_RELU_GN)) {
    if (plain && (!need_gn || table_ok)) {
      hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS, need_gn, false, false, true>), grid, dim3(256), 0, s, a);
      SNAP_CHECK_LAUNCH();
      return a.ksplit > 1 ? launch_splitk_reduce(a, s) : SNAP_OK;
    }
  }
  if (need_gn && table_ok)           // (GroupNorm operands are per channel QUAD: Cin % 4 == 0)
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS, need_gn, false>), grid, dim3(256), 0, s, a);
  else if (tail)
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS, false, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, PRO, NS, false, false>), grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return a.ksplit > 1 ? launch_splitk_reduce(a, s) : SNAP_OK;
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
  const TileChoice t = choose_tile(a.M, a.d.Cout, a.d.tile_hint, desc_k(a.d));
  if (t.bm == 128 && t.bn == 128) return launch_pro<128, 128, NS>(a, s);
  if (t.bm == 128) return launch_pro<128, 64, NS>(a, s);
  if (t.bn == 128) return launch_pro<64, 128, NS>(a, s);
  return launch_pro<64, 64, NS>(a, s);
}

// w [taps*Cin, Cout] f32 -> the split engine's weight image
//   out[column tile of 128][tap][channel tile of 16][part][column 0..127][16 k] bf16,
// the two 8-k octets of a column swapped where (column >> 3) & 1 (the LDS swizzle, applied at
// rest); part 0 = bf16(w) (RNE), part p = bf16 of the exact f32 residual left by parts < p;
// channels >= Cin and columns >= Cout are zero.  One 32 x 32 (k x n) tile per workgroup through
// LDS: coalesced along n on the way in, 64 B runs on the way out.
__device__ __forceinline__ void pack_weights_split_body(const float* __restrict__ w,
                                                        __bf16* __restrict__ out, int taps, int Cin,
                                                        int ctiles, int Cout, int parts, int bx,
                                                        int by, int t) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = bx * 32, n0 = by * 32;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    tile[j][tx] = (c < Cin && n < Cout) ? w[((int64_t)t * Cin + c) * Cout + n] : 0.f;
  }
  __syncthreads();
  // one 16-byte chunk (8 consecutive k of one column and part) per thread and round: the image's own
  // granule -- 2-byte scattered stores ran this kernel at 1.8 TB/s.  Same conversions, same bits.
  for (int i = threadIdx.x; i < 128 * parts; i += 256) {
    const int p = i >> 7, q = i & 127;
    const int j = q >> 2, slab = (q >> 1) & 1, ko = q & 1;      // column of the tile, channel tile, k-octet
    const int n = n0 + j, cb = c0 + 16 * slab;
    if (cb >= 16 * ctiles) continue;
    const int col = n & 127;
    const int oct = ko ^ ((col >> 3) & 1);
    const int64_t blk = ((int64_t)(n >> 7) * taps + t) * ctiles + (cb >> 4);
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    bf16x8v v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float r = tile[16 * slab + 8 * ko + e][j];
      for (int pp = 0; pp < p; ++pp) r -= (float)(__bf16)r;
      v[e] = (__bf16)r;
    }
    *reinterpret_cast<bf16x8v*>(out + blk * ((int64_t)parts * 2048) + p * 2048 + col * 16 + oct * 8) = v;
  }
}

__global__ __launch_bounds__(256) void pack_weights_split_kernel(
    const float* __restrict__ w, __bf16* __restrict__ out, int taps, int Cin, int ctiles, int Cout,
    int parts) {
  pack_weights_split_body(w, out, taps, Cin, ctiles, Cout, parts, blockIdx.x, blockIdx.y, blockIdx.z);
}

// every weight of an encoder in ONE launch (items sorted by block_begin, as SnapWstdItem)
__global__ __launch_bounds__(256) void pack_weights_split_multi_kernel(
    const SnapPackItem* __restrict__ items, int n_items, int parts) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SnapPackItem it = items[lo];
  const int ctiles = (it.Cin + 15) / 16;
  const int gx = (16 * ctiles + 31) / 32;
  const int gy = ((it.Cout + 127) / 128 * 128) / 32;
  const int local = blockIdx.x - it.block_begin;
  const int t = local / (gx * gy);
  const int rem = local - t * (gx * gy);
  pack_weights_split_body(it.w, static_cast<__bf16*>(it.out), it.taps, it.Cin, ctiles, it.Cout, parts,
                          rem % gx, rem / gx, t);
}

// w [7, 7, 3, Cout] f32 -> the ROOT image [column tile of 128][14 slabs = (kh, kw group of 4)][part]
// [column][16 k], k = 4 (kw - 4 group) + channel; tap kw = 7 and channel 3 are zero
__global__ __launch_bounds__(256) void pack_weights_split_root_kernel(const float* __restrict__ w,
                                                                      __bf16* __restrict__ out,
                                                                      int Cout, int cout128, int parts) {
  const int i = blockIdx.x * 256 + threadIdx.x;            // over [14 slabs][cout128][16 k]
  if (i >= 14 * cout128 * 16) return;
  const int k = i & 15;
  const int n = (i >> 4) % cout128;
  const int t = i / (16 * cout128);
  const int kh = t >> 1, kw = 4 * (t & 1) + (k >> 2), c = k & 3;
  float r = (kw < 7 && c < 3 && n < Cout) ? w[((kh * 7 + kw) * 3 + c) * Cout + n] : 0.f;
  const int col = n & 127;
  const int oct = (k >> 3) ^ ((col >> 3) & 1);
  __bf16* o = out + ((int64_t)(n >> 7) * 14 + t) * ((int64_t)parts * 2048) + col * 16 + oct * 8 + (k & 7);
  for (int p = 0; p < parts; ++p) {
    const __bf16 b = (__bf16)r;
    o[p * 2048] = b;
    r -= (float)b;
  }
}

}  // namespace

namespace {
template <int BN, int NS>
int launch_root_pro(const ConvArgs& a, dim3 grid, hipStream_t s) {
  if (a.d.prologue == SNAP_PRO_AFFINE)
    hipLaunchKernelGGL((conv_split_root_kernel<BN, SNAP_PRO_AFFINE, NS>), grid, dim3(256), 0, s, a);
  else if (a.d.prologue == SNAP_PRO_NONE)
    hipLaunchKernelGGL((conv_split_root_kernel<BN, SNAP_PRO_NONE, NS>), grid, dim3(256), 0, s, a);
  else
    return SNAP_ERR_UNSUPPORTED;
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
}  // namespace

// 7 x 7 / stride 2 / pad 3 over x [N, H, W, 4] (Cin = 3), weights in the root image
int snapconv::launch_split_root(ConvArgs a, int parts, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  if (d.KH != 7 || d.KW != 7 || d.stride != 2 || d.pad_t != 3 || d.pad_l != 3 || d.Cin != 3 ||
      d.Cin_stride != 4 || a.rows_in || a.rows_out || a.row_count || a.gn_partial)
    return SNAP_ERR_UNSUPPORTED;
  // 64 output channels, two-part split: the weights-stationary kernel (conv_rs.hip), same bits
  if (parts == 2 && d.Cout == 64 && d.Cout_stride == 64 && hint_mode(d.tile_hint) != 1 &&
      (d.prologue == SNAP_PRO_AFFINE || d.prologue == SNAP_PRO_NONE) && !(d.epilogue & ~SNAP_EPI_RELU) &&
      (a.M >= 40000 || hint_mode(d.tile_hint) == 2 || hint_mode(d.tile_hint) == 4) && d.W >= 8)
    return launch_root_ws(a, s);
  a.d.KW = 2;                      // two 4-pixel slabs per kernel row
  a.ctiles = 1;
  a.nk = 14;
  a.ksplit = 1;
  a.slabs_per_split = a.nk;
  const bool wide = d.Cout > 64;
  const int bn = wide ? 128 : 64;
  a.ncol = (int)snap_cdiv(d.Cout, bn);
  a.gn_slabs = (d.Ho * d.Wo) / 128 + 2;
  const int64_t nblocks = snap_cdiv(snap_cdiv(a.M, 128), 8) * 8 * a.ncol;
  if (nblocks > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  a.tiles_per_split = (int)nblocks;
  const dim3 grid((unsigned)nblocks);
  if (parts == 2) return wide ? launch_root_pro<128, 2>(a, grid, s) : launch_root_pro<64, 2>(a, grid, s);
  if (parts == 3) return wide ? launch_root_pro<128, 3>(a, grid, s) : launch_root_pro<64, 3>(a, grid, s);
  return SNAP_ERR_UNSUPPORTED;
}

int snapconv::launch_split(ConvArgs a, int parts, hipStream_t s) {
  const int kind = stationary_kind(a.d, parts, a.rows_in || a.rows_out || a.row_count);
  // (a caller that sized its statistics buffer for a split-K launch asked the wrong query:
  //  the stationary kernels never split K and lay their partial sums out their own way)
  if (kind && a.gn_partial && a.gn_rows32) return SNAP_ERR_WORKSPACE;
  switch (kind) {
    case 1: return launch_rs(a, s);
    case 2: case 3: return launch_bs(a, s);
    default: break;
  }
  if (parts == 2) return launch_tile<2>(a, s);
  if (parts == 3) return launch_tile<3>(a, s);
  return SNAP_ERR_UNSUPPORTED;
}

extern "C" size_t snap_conv2d_packed_weights_split_bytes(int32_t taps, int32_t Cin, int32_t Cout,
                                                         int32_t parts) {
  if (parts < 1 || parts > 3 || taps <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const size_t cin16 = ((size_t)Cin + 15) / 16 * 16;
  const size_t cout128 = ((size_t)Cout + 127) / 128 * 128;
  const size_t elems = (size_t)parts * cout128 * taps * cin16;
  if (elems >= ((size_t)1 << 33)) return 0;   // (template banks of the exhaustive voting: f32 engine)
  return elems * 2;
}

extern "C" int snap_conv2d_pack_weights_split_bf16(const float* w, int32_t taps, int32_t Cin,
                                                   int32_t Cout, int32_t parts, void* out,
                                                   size_t out_bytes, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  if (taps <= 0 || Cin <= 0 || Cout <= 0) return SNAP_ERR_BAD_SHAPE;
  if (parts < 1 || parts > 3) return SNAP_ERR_UNSUPPORTED;
  const size_t need = snap_conv2d_packed_weights_split_bytes(taps, Cin, Cout, parts);
  if (need == 0) return SNAP_ERR_UNSUPPORTED;
  if (out_bytes < need) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(out) & 15) return SNAP_ERR_BAD_SHAPE;
  const int ctiles = (Cin + 15) / 16;
  const int cout128 = (Cout + 127) / 128 * 128;
  const dim3 grid((unsigned)snap_cdiv(16 * ctiles, 32), (unsigned)(cout128 / 32), (unsigned)taps);
  hipLaunchKernelGGL(pack_weights_split_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, static_cast<__bf16*>(out), taps, Cin, ctiles, Cout, parts);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int32_t snap_conv2d_pack_weights_split_blocks(int32_t taps, int32_t Cin, int32_t Cout) {
  if (taps <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int ctiles = (Cin + 15) / 16;
  return ((16 * ctiles + 31) / 32) * (((Cout + 127) / 128 * 128) / 32) * taps;
}

extern "C" int snap_conv2d_pack_weights_split_multi_bf16(const SnapPackItem* items, int32_t n_items,
                                                         int32_t total_blocks, int32_t parts,
                                                         void* stream) {
  if (!items) return SNAP_ERR_NULL;
  if (n_items <= 0 || total_blocks <= 0) return SNAP_ERR_BAD_SHAPE;
  if (parts < 1 || parts > 3) return SNAP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_weights_split_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, n_items, parts);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" size_t snap_conv2d_packed_weights_split_root_bytes(int32_t Cout, int32_t parts) {
  if (parts < 2 || parts > 3 || Cout <= 0) return 0;
  return (size_t)parts * ((Cout + 127) / 128 * 128) * 14 * 16 * 2;
}

extern "C" int snap_conv2d_pack_weights_split_root_bf16(const float* w, int32_t Cout, int32_t parts,
                                                        void* out, size_t out_bytes, void* stream) {
  if (!w || !out) return SNAP_ERR_NULL;
  const size_t need = snap_conv2d_packed_weights_split_root_bytes(Cout, parts);
  if (need == 0) return SNAP_ERR_UNSUPPORTED;
  if (out_bytes < need) return SNAP_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(out) & 15) return SNAP_ERR_BAD_SHAPE;
  const int cout128 = (Cout + 127) / 128 * 128;
  hipLaunchKernelGGL(pack_weights_split_root_kernel, dim3((unsigned)snap_cdiv(14 * cout128 * 16, 256)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(out),
                     Cout, cout128, parts);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

#if defined(SNAP_CONV_TIMELINE) && SNAP_CONV_TIMELINE
// alt build only: copy the phase counters of the last plain-body launch to the host (tools/conv_timeline.py)
extern "C" int snap_debug_conv_timeline(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_timeline), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
