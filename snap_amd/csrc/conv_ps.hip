// Implicit-GEMM convolution / dense engine for PRE-SPLIT activations: the f32-grade split-bf16
// arithmetic of conv_split.hip at NS = 2 ("bf16x3": a_lo b_hi + a_hi b_lo + a_hi b_hi per MAC on
// v_mfma_f32_32x32x16_bf16, f32 accumulate, the same slab order and the same product order per
// accumulator), with the A operand arriving ALREADY normalised and split:
//
//   x_ps [pixel][Cin / 16][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]   bf16, 64 B per (pixel, 16 channels)
//
// written once per activation by snap_gn_norm_split_f32 (GroupNorm + ReLU of a conv output whose
// statistics came out of the producing conv's epilogue) or snap_presplit_f32 (plain split).
// conv_split.hip fetches, normalises and splits every A element once per (tap, column tile) in
// VALU code that shares the issue port with the MFMAs (~9 VALU per MFMA: the loop is bound by
// it); here BOTH operands travel global -> LDS by LDS-DMA, the K loop holds no VALU work but
// address increments, and the stages form a ring NST deep (one barrier per 16-k step).
// Replaces flax.linen.Conv / Dense (snap/models/resnet.py:73-132: the 3x3 and the closing 1x1
// convolution of every bottleneck unit) and the correlation of pose_exhaustive_voting.py:72-104.
//
// Tiles: 128 x 128 / 128 x 64 on 256 threads (2 x 2 waves) or 256 x 128 on 512 threads (4 x 2
// waves), every wave a 64 x 64 (64 x 32) block of 32 x 32 MFMA tiles.  A thread owns ONE A row
// and one of its two k-octet positions: per k-step it issues the hi and the lo chunk (16 B each)
// of that row -- a tap outside the image or a row beyond M reads a zero chunk instead -- and its
// share of the weight image's contiguous stage block.  Stage layout = conv_split's
// ([part][row][2 x 16 B], octets swapped where (row >> 3) & 1): the fragment fetch is one
// conflict-free ds_read_b128 per lane and MFMA operand.
//
// RES_INIT: the residual (resnet.py:131 `residual + y`) is loaded straight into the accumulators
// before the K loop -- 4-byte loads in the MFMA C layout, 128 B runs per row -- so that it
// travels under the ring's first stages instead of as a dependent load phase of the epilogue;
// the sum is then r + p_1 + p_2 + ... instead of (p_1 + p_2 + ...) + r: same error class, not
// the same last bits.  Split-K, output statistics (also of relu(y): DUAL) and the epilogue are
// the other engines' (conv_common.h).
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Timing-only ablations (alt builds: scripts/build_alt.sh; WRONG results): bit0 no DMA inside the
// loop, bit1 no fragment fetches, bit3 no MFMAs
#ifndef SNAP_PS_ABLATE
#define SNAP_PS_ABLATE 0
#endif
#ifndef SNAP_PS_STAGGER
#define SNAP_PS_STAGGER 1     // 512-thread tiles: the two waves of a SIMD issue their DMA pieces at different points of the stage
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KS = 16-k steps per ring stage (one barrier per stage), NST = ring depth in stages
// NS = 2: the two-part pre-split image (f32-grade, three products per MAC).  NS = 1: ONE part -- x is a plain bf16
// matrix [pixel][Cin] (16 channels = the 32-byte run of a k-step), the weights the one-part image of the same
// packer: the training-precision arithmetic (operands rounded to bf16, f32 accumulate) with both operands by
// LDS-DMA through the same ring -- the dense layers of the ViT encoder, whose producers write bf16.
// OUTH (NS = 1): 1 = the stored values also / only go, rounded to bf16, to a.y_half (conv_epilogue).
template <int BM, int BN, int NT, int KS, int NST, bool RES_INIT, bool DUAL, int NS = 2, int OUTH = 0>
__device__ __forceinline__ void conv_ps_body(const ConvArgs& a) {
  constexpr int WR = NT / 128;
  constexpr int TM = BM / (32 * WR), TN = BN / 64;
  static_assert(BM * 2 == NT, "a thread owns one A row and one octet position");
  static_assert(TM == 2 && BN % 64 == 0, "64-row wave blocks, 32-column MFMA tiles");
  constexpr int A_PART = BM * 32, B_PART = BN * 32;
  constexpr int A_ST = NS * A_PART, B_ST = NS * B_PART;
  constexpr int ST1 = A_ST + B_ST;                // one k-step
  constexpr int ST = KS * ST1;                    // one stage
  constexpr int BSLOTS1 = NS * BN * 2;            // 16-byte B pieces of one k-step
  static_assert((KS * BSLOTS1) % NT == 0, "whole B pieces per thread and stage");
  constexpr int BP = KS * BSLOTS1 / NT;
  constexpr int PIECES = NS * KS + BP;            // DMA instructions per thread and stage
  static_assert(NST >= 2 && NST <= 4, "ring depth");
  constexpr int kRing = NST * ST;
  constexpr int kStageBytes = 32 * WR * BN * 4;   // epilogue: staged output rows
  constexpr int kStatBytes = NT * 16 * 4;         // epilogue: statistics reduce
  constexpr int kSmemA = kRing > kStageBytes ? kRing : kStageBytes;
  constexpr int kSmem = kSmemA > kStatBytes ? kSmemA : kStatBytes;
  __shared__ __attribute__((aligned(16))) float smem[kSmem / 4];
  char* const ring = reinterpret_cast<char*>(smem);

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
  const int ctiles = a.ctiles;
  const int64_t pixb = (int64_t)ctiles * (NS * 32);   // bytes per pixel of x_ps

  // ---- this thread's A row ----------------------------------------------------------------
  // Both operands are fetched with BUFFER loads (buffer_load_dwordx4 ... lds): a scalar resource +
  // one 32-bit byte offset per lane instead of a 64-bit address pair, and a lane whose tap lies
  // outside the image (or whose row lies beyond M) passes an out-of-range offset and receives
  // ZEROS from the range check -- no select against a zero chunk, no 64-bit VALU in the loop.
  const int arow = tid >> 1;
  const int koct = (tid & 1) ^ ((arow >> 3) & 1);
  const bool r_ok = m0 + arow < Meff;
  // Offsets are taken from the (tap 0, 0) pixel of the tile's first row -- the pixel index is
  // non-decreasing in the row index --, clamped at the start of the tensor: every valid (row, tap)
  // of the tile then lies at a small non-negative 32-bit offset, whatever the size of the tensor.
  int64_t tile_px;
  {
    const int n = m0 / HoWo;
    const int r = m0 - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    tile_px = ((int64_t)n * d.H + ho * d.stride - d.pad_t) * d.W + wo * d.stride - d.pad_l;
    if (tile_px < 0) tile_px = 0;
  }
  int r_hb, r_wb;
  int r_off;                                      // byte offset of the row's (tap 0, 0) pixel (may be < 0)
  {
    const int mm = r_ok ? m0 + arow : 0;
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_hb = ho * d.stride - d.pad_t;
    r_wb = wo * d.stride - d.pad_l;
    r_off = (int)(((((int64_t)n * d.H + r_hb) * d.W + r_wb) - tile_px) * pixb) + koct * 16;
  }
  const int64_t a_left = ((int64_t)d.N * d.H * d.W - tile_px) * pixb;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.x_ps)) + tile_px * pixb, 0,
      (int)(a_left < 0x7ff00000LL ? a_left : 0x7ff00000LL), 0x00020000);
  constexpr int kOob = (int)0x80000000u;          // beyond any window this path accepts (< 2 GB)
  // ---- issue cursor (runs NST - 1 stages ahead of the multiply cursor) ----------------------
  const int kt_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;
  const int kt_end = a.ksplit > 1 ? min(a.nk, kt_begin + a.slabs_per_split) : a.nk;
  const int nk_loc = kt_end - kt_begin;
  const int nst_loc = (nk_loc + KS - 1) / KS;     // stages (the last one may be partly empty)
  int ct = 0, kh = 0, kw = 0;
  if (kt_begin > 0) {
    const int kpos = kt_begin / ctiles;
    ct = kt_begin - kpos * ctiles;
    kh = kpos / d.KW;
    kw = kpos - kh * d.KW;
  }
  int tap_off;                                    // scalar: byte offset of tap (kh, kw)
  bool tap_ok;
  auto set_tap = [&]() {
    const int hi = r_hb + kh, wi = r_wb + kw;
    tap_ok = r_ok && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
    tap_off = (kh * d.W + kw) * (int)pixb;
  };
  set_tap();
  // B: piece p of this thread = slot tid + NT p of the stage's [k-step][part][column][octet]
  // order; a wave's 64 slots are 32 whole columns of one (k-step, part) -- 1 KB contiguous in the
  // weight image (column tile of 128 = (n0 + column) >> 7: wave-uniform) and in LDS.
  const int taps = d.KH * d.KW;
  const int64_t col_tile_bytes = (int64_t)taps * ctiles * (NS * 4096);
  __amdgpu_buffer_rsrc_t rs_b[BP];
  int b_voff[BP], b_lds[BP], b_ks[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int slot = tid + NT * p;
    const int ks = slot / BSLOTS1;
    const int rem1 = slot - ks * BSLOTS1;
    const int part = rem1 / (2 * BN);
    const int rem = rem1 - part * (2 * BN);
    const int gcol = n0 + (rem >> 1);                          // (padded columns hold zeros)
    b_voff[p] = ks * (NS * 4096) + part * 4096 + (gcol & 127) * 32 + (rem & 1) * 16;
    b_ks[p] = __builtin_amdgcn_readfirstlane(ks);
    const int slot0 = __builtin_amdgcn_readfirstlane(slot - lane);   // the wave's first slot
    const int ks0 = slot0 / BSLOTS1;
    b_lds[p] = ks0 * ST1 + A_ST + (slot0 - ks0 * BSLOTS1) * 16;
    const int tile = __builtin_amdgcn_readfirstlane(gcol >> 7);
    rs_b[p] = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.w_bf16)) + tile * col_tile_bytes, 0,
        (int)col_tile_bytes, 0x00020000);
  }
  int b_soff = kt_begin * (NS * 4096);            // scalar: the stage's first block of the weight image
  int ikt = 0;                                    // k-steps issued so far
  // LDS destination of a DMA piece = wave-uniform base (M0) + 16 x lane
  const int wbase = __builtin_amdgcn_readfirstlane(wid) * 1024;
  auto issue = [&](int slot) {
    char* const base = ring + slot * ST;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool live = ikt + ks < nk_loc;        // (beyond the end: zeros into a slot nobody reads)
      const int voff = (tap_ok && live) ? r_off + (tap_off + ct * (NS * 32)) : kOob;
      char* const dst = base + ks * ST1 + wbase;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t*)dst, 16, voff, 0, 0, 0);
      // (the lo half through the SCALAR offset: an immediate offset would move the LDS address too)
      if constexpr (NS == 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t*)(dst + A_PART), 16, voff, 32, 0, 0);
      // (branch-free: the loop body stays ONE basic block, so that the issues can be scheduled
      //  between the MFMAs)
      const bool cw = ct + 1 == ctiles;
      const bool kww = cw && kw + 1 == d.KW;
      ct = cw ? 0 : ct + 1;
      kw = kww ? 0 : (cw ? kw + 1 : kw);
      kh = kww ? kh + 1 : kh;
      set_tap();
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const bool live = ikt + b_ks[p] < nk_loc;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b[p], (lds_void_t*)(base + b_lds[p]), 16,
                                               live ? b_voff[p] : kOob, b_soff, 0, 0);
    }
    b_soff += KS * (NS * 4096);
    ikt += KS;
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);

  f32x16 acc[TM][TN];
  if constexpr (RES_INIT) {
    // acc[i][j][r] <-> row wr (BM / WR) + 32 i + (r & 3) + 8 (r >> 2) + 4 lhi, column wc (BN / 2) + 32 j + l31
    const int colb = n0 + wc * (BN / 2) + l31;
    const float* const rb = a.residual + (int64_t)(m0 + wr * (BM / WR) + 4 * lhi) * d.Cout_stride + colb;
    const int mrow = m0 + wr * (BM / WR) + 4 * lhi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = 32 * i + (r & 3) + 8 * (r >> 2);
          const bool ok = mrow + ro < Meff && colb + 32 * j < d.Cout;
          acc[i][j][r] = ok ? rb[(int64_t)ro * d.Cout_stride + 32 * j] : 0.f;
        }
    // Resolve the loads HERE (they were issued behind the ring's first stages and travel with
    // them): left pending into the loop, the compiler's own wait for them would sit in front of
    // the MFMAs of EVERY iteration and drain the ring down to one stage.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const int grp = (NT == 512 && SNAP_PS_STAGGER) ? __builtin_amdgcn_readfirstlane(wid >> 2) : 0;
  int slot = 0;                 // ring slot of the stage being multiplied
  int islot = NST - 1;          // ring slot the next issue goes to
  for (int st = 0; st < nst_loc; ++st) {
    // own pieces of stage st landed (the NST - 2 younger stages' may still travel; every iteration
    // issues a full set of pieces -- out-of-range ones past the end --, so the count is constant) ...
    wait_vm<(NST - 2) * PIECES>();
    // ... and everybody's; all waves are also past their reads of the slot issued next.  (The
    // fence-less barrier: __syncthreads() would drain the younger stages' DMAs as well.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // Waves w and w + 4 of a 512-thread workgroup share a SIMD.  A wave issuing its DMA pieces
    // blocks while the vector-memory queue takes them, and right after the barrier every wave of
    // the workgroup would do so at once with the matrix pipes idle: the upper four waves issue
    // theirs after the first block of MFMAs instead, under which the lower four issue.
    if (!(SNAP_PS_ABLATE & 1) && grp == 0) issue(islot);
    const char* const stage = ring + slot * ST;
    slot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks == KS / 2 && KS > 1) {
        if (!(SNAP_PS_ABLATE & 1) && grp == 1) issue(islot);
      }
      const char* as = stage + ks * ST1;
      const char* bs = as + A_ST;
      bf16x8 av[TM][NS], bv[TN][NS];
      if (SNAP_PS_ABLATE & 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(av[i][p]));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(bv[j][p]));
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int R = wr * (BM / WR) + i * 32 + l31;
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
      }
#define SNAP_PS_PRODUCT(PA, PB)                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA], bv[j][PB], acc[i][j], 0, 0, 0);
      if (!(SNAP_PS_ABLATE & 8)) {
        if constexpr (NS == 2) {
          SNAP_PS_PRODUCT(1, 0)
          if (KS == 1) {
            if (!(SNAP_PS_ABLATE & 1) && grp == 1) issue(islot);
          }
          SNAP_PS_PRODUCT(0, 1)
          SNAP_PS_PRODUCT(0, 0)
        } else {
          static_assert(NS == 2 || NT == 256 || KS > 1, "one-part 512-thread tiles: two k-steps per stage");
          SNAP_PS_PRODUCT(0, 0)
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
#undef SNAP_PS_PRODUCT
    }
    islot = islot + 1 == NST ? 0 : islot + 1;
  }
  __syncthreads();              // the last stage is read: the ring becomes the epilogue's buffer
  conv_epilogue<BM, BN, DUAL, NT, RES_INIT, OUTH>(a, acc, smem, m0, n0, Meff, row_t, split);
}

// workgroups per CU: three of 256 threads (48 KB), two of 512 threads (72 KB), or ONE 256 x 192
// tile (112 KB, 96 accumulator + 40 fragment registers per lane)
template <int BM, int BN, int NT, int KS, int NST, bool RES_INIT, bool DUAL>
__global__ __launch_bounds__(NT, NT == 512 ? (BN == 192 ? 2 : 4) : 3) void conv_ps_kernel(const ConvArgs a) {
  conv_ps_body<BM, BN, NT, KS, NST, RES_INIT, DUAL>(a);
}

// the one-part (bf16) engine: 256 x 128 tiles on 512 threads, two k-steps per stage (the same 24 KB stages and
// 72 KB ring as the two-part tile), or 128 x 128 on 256 threads
template <int BM, int BN, int NT, int KS, int NST, int OUTH>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 3) void conv_ps1_kernel(const ConvArgs a) {
  conv_ps_body<BM, BN, NT, KS, NST, false, false, 1, OUTH>(a);
}

template <int BM, int BN, int NT, int KS, int NST>
int launch_variant1(const ConvArgs& a, dim3 grid, hipStream_t s) {
  if (a.y_half)
    hipLaunchKernelGGL((conv_ps1_kernel<BM, BN, NT, KS, NST, 1>), grid, dim3(NT), 0, s, a);
  else
    hipLaunchKernelGGL((conv_ps1_kernel<BM, BN, NT, KS, NST, 0>), grid, dim3(NT), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

template <int BM, int BN, int NT, int KS, int NST>
int launch_variant(const ConvArgs& a, bool res_init, bool dual, dim3 grid, hipStream_t s) {
  if constexpr (BN == 128) {
    if (res_init && dual)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, KS, NST, true, true>), grid, dim3(NT), 0, s, a);
    else if (res_init)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, KS, NST, true, false>), grid, dim3(NT), 0, s, a);
    else if (dual)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, KS, NST, false, true>), grid, dim3(NT), 0, s, a);
    else
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, KS, NST, false, false>), grid, dim3(NT), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, KS, NST, false, false>), grid, dim3(NT), 0, s, a);
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

}  // namespace

// Tile: 64-wide column tiles only for Cout <= 64 (the 3x3 convolutions of the first ResNet
// stage); 256-row tiles (512 threads, two workgroups per CU) once they still give two full rounds
// of the 512 slots; force: 1 = 128 rows, 2 = 256 rows, 3 = 256 x 192 (one workgroup per CU, two
// k-steps per stage: the large-GEMM tile of the exhaustive voting; Cout % 192 == 0)
// (SnapConvExtras.ps_tile).
snapconv::PsTile snapconv::ps_choose_tile(int64_t M, int64_t N, int force) {
  if (force == 3 && N % 192 == 0) return {256, 192, 512};
  if (N <= 64) return {128, 64, 256};
  const int64_t t256 = snap_cdiv(M, 256) * snap_cdiv(N, 128);
  if (force == 2 || (force == 0 && t256 >= 1024)) return {256, 128, 512};
  return {128, 128, 256};
}

// split-K factor of a launch with few output tiles (as conv_common.h's rule: about 768
// workgroups, at least 8 k-steps per split, bounded by the workspace); 1 = none
int snapconv::ps_ksplit(int64_t M, int Cout, int64_t nk, int bm, int bn, size_t kpartial_bytes) {
  const int64_t tiles = snap_cdiv(M, bm) * snap_cdiv((int64_t)Cout, bn);
  const int64_t target = bn == 192 ? 768 : bm == 256 ? 1024 : 768;
  if (tiles > target / 2 || nk < 16) return 1;
  int64_t S = (target + tiles - 1) / tiles;
  S = S < nk / 8 ? S : nk / 8;
  const int64_t fit = (int64_t)(kpartial_bytes / ((size_t)M * Cout * sizeof(float)));
  S = S < fit ? S : fit;
  return S >= 2 ? (int)S : 1;
}

// Shapes the pre-split engine takes (shared by the launch and by snap_conv2d_presplit_supported).
// 32-bit buffer offsets with an out-of-range marker at 2^31: the input window of one row tile
// (256 output rows + the kernel's extent, two images) and one column tile of the weight image
// must stay below 2 GB.
static bool ps_shape_supported(const SnapConvDesc& d) {
  if (d.prologue != SNAP_PRO_NONE || d.Cin % 16 != 0 || d.Cin <= 0) return false;
  const int64_t Wo = d.Wo, Ho = d.Ho, st = d.stride;
  if (Wo <= 0 || Ho <= 0 || d.N <= 0) return false;
  int64_t dq = 255 / Wo + 1;                                   // output-row wraps inside a tile
  if (dq > (int64_t)d.N * Ho - 1) dq = (int64_t)d.N * Ho - 1;
  int64_t dn = dq / Ho + 1;                                    // image wraps
  if (dn > d.N - 1) dn = d.N - 1;
  const int64_t c = d.H - Ho * st > 0 ? d.H - Ho * st : 0;
  const int64_t dwo = dq == 0 ? (Wo < 256 ? Wo : 256) : Wo;
  const int64_t span = (dq * st + dn * c + d.KH - 1 + d.pad_t) * d.W + (dwo + 1) * st + d.KW + d.pad_l;
  if (span * (d.Cin / 16) * 64 >= 0x7ff00000LL) return false;
  if ((int64_t)d.KH * d.KW * (d.Cin / 16) * 8192 >= 0x7ff00000LL) return false;
  return true;
}

extern "C" int32_t snap_conv2d_presplit_supported(const SnapConvDesc* desc) {
  return (desc && ps_shape_supported(*desc)) ? 1 : 0;
}

int snapconv::launch_ps(ConvArgs a, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  if (!a.x_ps || !a.w_bf16) return SNAP_ERR_NULL;
  if (a.rows_in || a.rows_out || a.row_count) return SNAP_ERR_UNSUPPORTED;
  if (!ps_shape_supported(d)) return SNAP_ERR_UNSUPPORTED;
  if (a.ps_parts == 1) {
    // one part: no statistics / split-K / residual pre-load variants (the ViT's dense layers need none)
    if (a.gn_partial || (d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) return SNAP_ERR_UNSUPPORTED;
    const int64_t t256 = snap_cdiv((int64_t)a.M, 256) * snap_cdiv((int64_t)d.Cout, 128);
    const bool big = a.ps_tile == 2 || (a.ps_tile == 0 && t256 >= 384);
    const int bm = big ? 256 : 128;
    a.ctiles = d.Cin / 16;
    a.nk = d.KH * d.KW * a.ctiles;
    a.ncol = (int)snap_cdiv(d.Cout, 128);
    a.gn_slabs = 0;
    const int64_t nblocks = snap_cdiv(snap_cdiv((int64_t)a.M, bm), 8) * 8 * a.ncol;
    if (nblocks > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
    a.ksplit = 1;
    a.kpartial = nullptr;
    a.tiles_per_split = (int)nblocks;
    a.slabs_per_split = a.nk;
    const dim3 grid((unsigned)nblocks);
    return big ? launch_variant1<256, 128, 512, 2, 3>(a, grid, s) : launch_variant1<128, 128, 256, 1, 3>(a, grid, s);
  }
  if (a.y_half) return SNAP_ERR_UNSUPPORTED;
  const PsTile t = ps_choose_tile(a.M, d.Cout, a.ps_tile);
  a.ctiles = d.Cin / 16;
  a.nk = d.KH * d.KW * a.ctiles;
  const int64_t nrow = snap_cdiv(a.M, t.bm);
  a.ncol = (int)snap_cdiv(d.Cout, t.bn);
  a.gn_slabs = (d.Ho * d.Wo) / t.bm + 2;
  int64_t nblocks = snap_cdiv(nrow, 8) * 8 * a.ncol;
  a.ksplit = 1;
  a.tiles_per_split = (int)nblocks;
  a.slabs_per_split = a.nk;
  if (a.kpartial && !a.gn_partial && !(d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) {
    const int S = ps_ksplit(a.M, d.Cout, a.nk, t.bm, t.bn, a.kpartial_bytes);
    if (S >= 2) {
      a.slabs_per_split = (a.nk + S - 1) / S;
      a.ksplit = (a.nk + a.slabs_per_split - 1) / a.slabs_per_split;
      nblocks *= a.ksplit;
    }
  }
  if (nblocks > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  // statistics need "a row tile touches at most two images"
  if (a.gn_partial && d.Ho * d.Wo < t.bm) return SNAP_ERR_UNSUPPORTED;
  const bool res_init = a.ps_res_init && (d.epilogue & SNAP_EPI_RESIDUAL) && a.ksplit == 1 &&
                        t.bn == 128 && d.Cout % 128 == 0;
  const bool dual = a.gn_partial2 != nullptr && a.gn_partial != nullptr && !a.gn_relu &&
                    a.ksplit == 1 && t.bn == 128;
  if (a.gn_partial2_done) *a.gn_partial2_done = dual ? 1 : 0;
  const dim3 grid((unsigned)nblocks);
  int st;
  if (t.bn == 192)
    st = launch_variant<256, 192, 512, 2, 2>(a, false, false, grid, s);
  else if (t.bm == 256)
    st = launch_variant<256, 128, 512, 1, 3>(a, res_init, dual, grid, s);
  else if (t.bn == 128)
    st = launch_variant<128, 128, 256, 1, 3>(a, res_init, dual, grid, s);
  else
    st = launch_variant<128, 64, 256, 1, 4>(a, false, false, grid, s);
  if (st != SNAP_OK) return st;
  if (a.ksplit > 1) {
    const int64_t total4 = (int64_t)a.M * (d.Cout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0, s,
                       (const float*)a.kpartial, a.ksplit, (int64_t)a.M, d.Cout, d.Cout_stride,
                       d.epilogue, a.bias, a.residual, a.row_mask, a.y);
    SNAP_CHECK_LAUNCH();
  }
  return SNAP_OK;
}
