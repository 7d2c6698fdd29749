// 1 x 1 / stride 1 / unpadded convolution (Dense) of the split-bf16 engine with the ACTIVATION ROWS
// STREAMED RAW INTO AN LDS RING: the f32 rows travel global -> LDS by LDS-DMA two k-steps ahead (no
// registers), and the GroupNorm + ReLU prologue and the two-part bf16 split run AT FRAGMENT FETCH, in
// the wave that multiplies the fragment.
//
// AN EXPERIMENT THAT ANSWERED A QUESTION (round 5; opt-in: SNAP_TUNE_RAW_RING / ops.CONV_RAW_RING, off by
// default).  Hypothesis: the tiled body of conv_split.hip fetches a k-step's A rows through registers
// ONE step ahead, so every 16-k step exposes a memory round trip; with the ring slots as the extra
// storage -- three 16 KB stages (A raw 8 KB + split weights 8 KB), TWO in flight while the third is
// multiplied, 48 KB per workgroup = three workgroups per CU -- the loop should run at the pace of its
// instructions.  Result (tools/conv_raw_bench.py, profiles/r05_conv_raw_bench.log): bit-identical, and
// LEVEL TO 20 % SLOWER on every K >= 256 layer of the C2 encoders (1024 -> 256 @ 34^2: 121 vs 119 us;
// 2048 -> 512 @ 17^2: 154 vs 124).  With two k-steps in flight the loop still takes ~4100 cycles per
// k-step and SIMD: three waves x (12 MFMAs = 384 cycles + ~250 VALU instructions of conversion at ~4
// cycles + 18 LDS fetches) -- the waves of a SIMD do NOT hide each other's VALU under their MFMAs
// here, the phases add up (what round 2's ablation build had already shown from the other side).  The
// conv family is bound by SIMD ISSUE (prologue / split VALU + MFMA + LDS), not by memory latency:
// what moves it is fewer VALU cycles per element, not more loads in flight.
//
// Same arithmetic as conv_split_body<..., PLAIN>: apply_pro on the same f32 value, the same RNE
// split (hi = bf16(v), lo = bf16(v - hi)), the same products in the same order per accumulator (lo x
// hi, hi x lo, hi x hi per k-step, k ascending): BIT-IDENTICAL output (tests/test_gpu_conv_rs.py
// compares).  Epilogue, tile order, split-K and statistics: conv_common.h's, unchanged.
//
// Replaces the same reference expressions as conv_split.hip (flax.linen.Conv / StdConv 1 x 1 with the
// GroupNorm -> ReLU in front: snap/models/resnet.py:103-132, image_encoder.py:67-94).
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void raw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// four f32 -> hi / lo bf16 pairs (conv_split.hip's split_bf16<2>: RNE, exact residual)
__device__ __forceinline__ void raw_split2(const f32x4& v, u32x2& hi, u32x2& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2 pr = {v[2 * h], v[2 * h + 1]};
    const bf16x2 b = __builtin_convertvector(pr, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &b, 4);
    hi[h] = u;
    const f32x2 rr = {pr[0] - __uint_as_float(u << 16), pr[1] - __uint_as_float(u & 0xffff0000u)};
    const bf16x2 c = __builtin_convertvector(rr, bf16x2);
    unsigned w;
    __builtin_memcpy(&w, &c, 4);
    lo[h] = w;
  }
}

// BM = 128 rows, 256 threads (wave = 64 rows x BN / 2 columns), one 16-k step per ring stage.
template <int BN, int PRO, int NST>
__device__ __forceinline__ void conv_raw_body(const ConvArgs& a) {
  constexpr int BM = 128, NT = 256, NS = 2;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int A_HALF = BM * 32;                 // bytes of one k-half (8 k) of the raw tile: [row][2 chunks of 16 B]
  constexpr int A_ST = 2 * A_HALF;                // 8 KB: 128 rows x 16 f32
  constexpr int B_PART = BN * 32, B_ST = NS * B_PART;
  constexpr int ST = A_ST + B_ST;                 // one stage = one k-step
  constexpr int BSLOTS = NS * BN * 2;             // 16-byte B pieces of a k-step
  static_assert(BSLOTS % NT == 0, "whole B pieces per thread");
  constexpr int BP = BSLOTS / NT;
  constexpr int PIECES = 2 + BP;                  // DMA instructions per thread and stage (A: two k-halves)
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
  constexpr int kGn = 320;                        // [mu n0 | sc n0 | mu n1 | sc n1 | beta] x 16 channels
  static_assert(NST == 3, "two stages in flight, one multiplied");
  constexpr int kRing = NST * ST;
  constexpr int kStageBytes = 64 * BN * 4;        // epilogue: staged output rows
  constexpr int kSmem = kRing > kStageBytes ? kRing : kStageBytes;
  __shared__ __attribute__((aligned(16))) float smem[(kSmem + (need_gn ? NST * kGn : 0)) / 4];
  char* const ring = reinterpret_cast<char*>(smem);
  char* const Gt = ring + kSmem;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ncol = a.ncol;
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
  const int HoWo = d.Ho * d.Wo;
  const int n_first = m0 / HoWo;
  const int m_split = (n_first + 1) * HoWo;       // first row of the tile's second image
  const int n_second = min(n_first + 1, d.N - 1);
  const int ctiles = a.ctiles;

  // ---- A: this thread's row and chunk position.  Buffer loads into LDS: a scalar resource windowed
  // at the tile's first row + one 32-bit byte offset per lane; rows beyond M pass an out-of-range
  // offset and receive zeros (their accumulators are never stored).  The 16-byte chunk position is
  // XOR-swizzled by (row >> 3) & 1 so that the fragment fetch below is conflict-free.
  const int arow = tid >> 1;
  const int kchunk = (tid & 1) ^ ((arow >> 3) & 1);        // logical chunk (4 k) stored at position tid & 1
  const bool r_ok = m0 + arow < Meff;
  const int64_t rowb = (int64_t)d.Cin_stride * 4;          // bytes per row of x
  constexpr int kOob = (int)0x80000000u;
  const int r_off = r_ok ? (int)(arow * rowb) + kchunk * 16 : kOob;
  const int64_t a_left = (int64_t)(Meff - m0) * rowb;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (int64_t)m0 * d.Cin_stride, 0,
      (int)(a_left < 0x7ff00000LL ? a_left : 0x7ff00000LL), 0x00020000);

  const int kt_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;
  const int kt_end = a.ksplit > 1 ? min(a.nk, kt_begin + a.slabs_per_split) : a.nk;
  const int nk_loc = kt_end - kt_begin;           // (1 x 1: k-step kt <-> channel tile kt)

  // ---- B: as conv_ps.hip -- piece p of this thread = slot tid + NT p of the stage's
  // [part][column][octet] order, 1 KB contiguous per wave in the weight image and in LDS
  const int64_t col_tile_bytes = (int64_t)ctiles * (NS * 4096);       // (taps = 1)
  __amdgpu_buffer_rsrc_t rs_b[BP];
  int b_voff[BP], b_lds[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int slot = tid + NT * p;
    const int part = slot / (2 * BN);
    const int rem = slot - part * (2 * BN);
    const int gcol = n0 + (rem >> 1);                          // (padded columns hold zeros)
    b_voff[p] = part * 4096 + (gcol & 127) * 32 + (rem & 1) * 16;
    const int slot0 = __builtin_amdgcn_readfirstlane(slot - lane);   // the wave's first slot
    b_lds[p] = A_ST + slot0 * 16;
    const int tile = __builtin_amdgcn_readfirstlane(gcol >> 7);
    rs_b[p] = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.w_bf16)) + tile * col_tile_bytes, 0,
        (int)col_tile_bytes, 0x00020000);
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wid) * 1024;     // LDS destination = base (M0) + 16 x lane
  int ikt = 0;                                    // k-steps issued so far
  auto issue = [&](int slot) {
    char* const base = ring + slot * ST;
    const bool live = ikt < nk_loc;               // (past the end: zeros / a repeated table into a slot nobody reads)
    const int kt = kt_begin + (live ? ikt : 0);
    const int voff = live ? r_off : kOob;
    char* const dst = base + wbase;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t*)dst, 16, voff, kt * 64, 0, 0);
    // (the second k-half through the SCALAR offset: an immediate offset would move the LDS address too)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t*)(dst + A_HALF), 16, voff, kt * 64 + 32, 0, 0);
#pragma unroll
    for (int p = 0; p < BP; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b[p], (lds_void_t*)(base + b_lds[p]), 16,
                                               live ? b_voff[p] : kOob, kt * (NS * 4096), 0, 0);
    if constexpr (need_gn) {
      if (tid < 20) {                             // wave 0: the k-step's GroupNorm operands (+1 piece)
        const int seg = tid >> 2;
        const int c = kt * 16 + 4 * (tid & 3);
        const float* gb = seg == 4 ? a.gn_beta
                                   : ((seg & 1) ? a.gn_sc : a.gn_mu) + (int64_t)(seg >= 2 ? n_second : n_first) * d.Cin;
        __builtin_amdgcn_global_load_lds((cglobal_void_t*)(gb + c), (lds_void_t*)(Gt + slot * kGn + 16 * tid), 16, 0, 0);
      }
    }
    ++ikt;
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows of this lane and their image slot (a row tile touches at most two images)
  int frow[TM], fslot[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    frow[i] = wr * 64 + i * 32 + l31;
    fslot[i] = (m0 + frow[i]) >= m_split ? 1 : 0;
  }

  int slot = 0;                 // ring slot of the stage being multiplied
  int islot = NST - 1;          // ring slot the next issue goes to
  for (int st = 0; st < nk_loc; ++st) {
    // own pieces of stage st landed (the younger stage's may still travel; every iteration issues a
    // full set of pieces, so the count is constant: wave 0 carries one more, the table) ...
    if constexpr (need_gn) {
      if (wid == 0) raw_wait_vm<(NST - 2) * (PIECES + 1)>();
      else raw_wait_vm<(NST - 2) * PIECES>();
    } else {
      raw_wait_vm<(NST - 2) * PIECES>();
    }
    // ... and everybody's; all waves are also past their reads of the slot issued next.  (Fence-less
    // barrier: __syncthreads() would drain the younger stage's DMAs as well.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(islot);
    const char* const as = ring + slot * ST;
    const char* const bs = as + A_ST;
    const float* const tb = reinterpret_cast<const float*>(Gt + slot * kGn);
    slot = slot + 1 == NST ? 0 : slot + 1;
    islot = islot + 1 == NST ? 0 : islot + 1;

    bf16x8 av[TM][NS], bv[TN][NS];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int C = wc * (BN / 2) + j * 32 + l31;
      const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
#pragma unroll
      for (int p = 0; p < NS; ++p) bv[j][p] = *reinterpret_cast<const bf16x8*>(p0 + p * B_PART);
    }
    f32x4 gbeta[2];
    if constexpr (need_gn) {
      gbeta[0] = *reinterpret_cast<const f32x4*>(tb + 64 + 8 * lhi);
      gbeta[1] = *reinterpret_cast<const f32x4*>(tb + 64 + 8 * lhi + 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int R = frow[i];
      const int sw = (R >> 3) & 1;
      const char* rp = as + lhi * A_HALF + R * 32;
      f32x4 v[2];
      v[0] = *reinterpret_cast<const f32x4*>(rp + (0 ^ sw) * 16);      // k = 8 lhi + 0..3
      v[1] = *reinterpret_cast<const f32x4*>(rp + (1 ^ sw) * 16);      // k = 8 lhi + 4..7
      u32x2 hi[2], lo[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 gmu, gsc;
        if constexpr (need_gn) {
          gmu = *reinterpret_cast<const f32x4*>(tb + fslot[i] * 32 + 8 * lhi + 4 * h);
          gsc = *reinterpret_cast<const f32x4*>(tb + fslot[i] * 32 + 16 + 8 * lhi + 4 * h);
        }
        f32x4 pv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (need_gn)
            pv[e] = apply_pro<PRO>(v[h][e], gmu[e], gsc[e], gbeta[h][e], d.in_scale, d.in_shift);
          else
            pv[e] = apply_pro<PRO>(v[h][e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        }
        raw_split2(pv, hi[h], lo[h]);
      }
      const u32x4 fh = {hi[0][0], hi[0][1], hi[1][0], hi[1][1]};
      const u32x4 fl = {lo[0][0], lo[0][1], lo[1][0], lo[1][1]};
      __builtin_memcpy(&av[i][0], &fh, 16);
      __builtin_memcpy(&av[i][1], &fl, 16);
    }
#define SNAP_RAW_PRODUCT(PA, PB)                                                             \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA], bv[j][PB], acc[i][j], 0, 0, 0);
    SNAP_RAW_PRODUCT(1, 0)
    SNAP_RAW_PRODUCT(0, 1)
    SNAP_RAW_PRODUCT(0, 0)
#undef SNAP_RAW_PRODUCT
  }
  raw_wait_vm<0>();             // (the dummy issues past the end)
  __syncthreads();              // the last stage is read: the ring becomes the epilogue's buffer
  conv_epilogue<BM, BN, false>(a, acc, smem, m0, n0, Meff, row_t, split);
}

template <int BN, int PRO>
__global__ __launch_bounds__(256, 3) void conv_raw_kernel(const ConvArgs a) {
  conv_raw_body<BN, PRO, 3>(a);
}

}  // namespace

// a: set up by conv_split.hip's launch<128, BN, PRO, 2> (ctiles, nk, ncol, gn_slabs, ksplit ...); the
// caller has checked raw_ok() and launches the split-K reduce pass itself.
bool snapconv::raw_ok(const ConvArgs& a, int bm, int bn, int pro) {
  const SnapConvDesc& d = a.d;
  const bool gn = pro == SNAP_PRO_GN_RELU || pro == SNAP_PRO_RELU_GN;
  return a.use_raw && bm == 128 && (bn == 128 || bn == 64) && gn && d.KH == 1 && d.KW == 1 && d.stride == 1 &&
         d.pad_t == 0 && d.pad_l == 0 && d.H == d.Ho && d.W == d.Wo && d.Cin % 16 == 0 && (d.Cin_stride & 3) == 0 &&
         !a.rows_in && !a.rows_out && !a.row_count && a.M > 0 && d.Ho * d.Wo >= bm &&
         (a.ksplit > 1 ? a.slabs_per_split : a.nk) >= 16 &&          // K >= 256 per workgroup: the ring pays from there
         (int64_t)bm * d.Cin_stride * 4 < 0x7ff00000LL && a.ctiles * (int64_t)(2 * 4096) < 0x7ff00000LL &&
         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}

int snapconv::launch_raw(const ConvArgs& a, int bn, int pro, dim3 grid, hipStream_t s) {
#define SNAP_RAW_LAUNCH(BN_, PRO_) \
  hipLaunchKernelGGL((conv_raw_kernel<BN_, PRO_>), grid, dim3(256), 0, s, a)
  if (bn == 128) {
    if (pro == SNAP_PRO_GN_RELU) SNAP_RAW_LAUNCH(128, SNAP_PRO_GN_RELU);
    else SNAP_RAW_LAUNCH(128, SNAP_PRO_RELU_GN);
  } else {
    if (pro == SNAP_PRO_GN_RELU) SNAP_RAW_LAUNCH(64, SNAP_PRO_GN_RELU);
    else SNAP_RAW_LAUNCH(64, SNAP_PRO_RELU_GN);
  }
#undef SNAP_RAW_LAUNCH
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
