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

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int NT, int NST, bool RES_INIT, bool DUAL>
__device__ __forceinline__ void conv_ps_body(const ConvArgs& a) {
  constexpr int NS = 2;
  constexpr int WR = NT / 128;
  constexpr int TM = BM / (32 * WR), TN = BN / 64;
  static_assert(BM * 2 == NT, "a thread owns one A row and one octet position");
  static_assert(TM == 2, "64-row wave blocks");
  constexpr int A_PART = BM * 32, B_PART = BN * 32;
  constexpr int A_ST = NS * A_PART, B_ST = NS * B_PART;
  constexpr int ST = A_ST + B_ST;
  constexpr int BSLOTS = NS * BN * 2;
  static_assert(BSLOTS % NT == 0, "whole B pieces per thread");
  constexpr int BPIECES = BSLOTS / NT;
  constexpr int PIECES = NS + BPIECES;            // DMA instructions per thread and k-step
  static_assert(NST >= 3 && NST <= 4, "ring depth");
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
  const int64_t pixb = (int64_t)ctiles * 64;      // bytes per pixel of x_ps

  // ---- this thread's A row ----------------------------------------------------------------
  const int arow = tid >> 1;
  const int koct = (tid & 1) ^ ((arow >> 3) & 1);
  const bool r_ok = m0 + arow < Meff;
  int r_hb, r_wb;
  const char* r_px;
  {
    const int mm = r_ok ? m0 + arow : 0;
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_hb = ho * d.stride - d.pad_t;
    r_wb = wo * d.stride - d.pad_l;
    r_px = static_cast<const char*>(a.x_ps) + (((int64_t)n * d.H + r_hb) * d.W + r_wb) * pixb + koct * 16;
  }
  // ---- issue cursor (runs NST - 1 k-steps ahead of the multiply cursor) ---------------------
  const int kt_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;
  const int kt_end = a.ksplit > 1 ? min(a.nk, kt_begin + a.slabs_per_split) : a.nk;
  const int nk_loc = kt_end - kt_begin;
  int ct = 0, kh = 0, kw = 0;
  if (kt_begin > 0) {
    const int kpos = kt_begin / ctiles;
    ct = kt_begin - kpos * ctiles;
    kh = kpos / d.KW;
    kw = kpos - kh * d.KW;
  }
  const char* tap_px;
  bool tap_ok;
  auto set_tap = [&]() {
    const int hi = r_hb + kh, wi = r_wb + kw;
    tap_ok = r_ok && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
    tap_px = r_px + ((int64_t)kh * d.W + kw) * pixb;
  };
  set_tap();
  const char* const wt = static_cast<const char*>(a.w_bf16);
  const int taps = d.KH * d.KW;
  const int64_t col_tile_bytes = (int64_t)taps * ctiles * (NS * 4096);
  const char* bsrc[BPIECES];
#pragma unroll
  for (int p = 0; p < BPIECES; ++p) {
    const int slot = tid + NT * p;
    const int part = slot / (2 * BN);
    const int rem = slot - part * (2 * BN);
    const int gcol = n0 + (rem >> 1);                          // (padded columns hold zeros)
    bsrc[p] = wt + (gcol >> 7) * col_tile_bytes + (int64_t)kt_begin * (NS * 4096) + part * 4096 +
              (gcol & 127) * 32 + (rem & 1) * 16;
  }
  const char* const zero = reinterpret_cast<const char*>(kZeroChunk);
  auto issue = [&](int slot) {
    char* const base = ring + slot * ST;
    const char* const s0 = tap_ok ? tap_px + ct * 64 : zero;
    __builtin_amdgcn_global_load_lds((cglobal_void_t*)s0, (lds_void_t*)(base + tid * 16), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((cglobal_void_t*)(tap_ok ? s0 + 32 : zero),
                                     (lds_void_t*)(base + A_PART + tid * 16), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)bsrc[p],
                                       (lds_void_t*)(base + A_ST + 16 * (tid + NT * p)), 16, 0, 0);
      bsrc[p] += NS * 4096;
    }
    if (++ct == ctiles) {
      ct = 0;
      if (++kw == d.KW) { kw = 0; ++kh; }
      set_tap();
    }
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk_loc) issue(s);

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

  int slot = 0;                 // ring slot of the k-step being multiplied
  int islot = NST - 1;          // ring slot the next issue goes to
  for (int kt = 0; kt < nk_loc; ++kt) {
    // own pieces of k-step kt landed (the younger k-steps' may still travel) ...
    const int ahead = nk_loc - 1 - kt;
    if (ahead >= NST - 2) wait_vm<(NST - 2) * PIECES>();
    else if (NST == 4 && ahead == 1) wait_vm<PIECES>();
    else wait_vm<0>();
    // ... and everybody's; all waves are also past their reads of the slot issued next.  (The
    // fence-less barrier: __syncthreads() would drain the younger k-steps' DMAs as well.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + NST - 1 < nk_loc) issue(islot);
    islot = islot + 1 == NST ? 0 : islot + 1;
    const char* as = ring + slot * ST;
    const char* bs = as + A_ST;
    slot = slot + 1 == NST ? 0 : slot + 1;
    bf16x8 av[TM][NS], bv[TN][NS];
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
#define SNAP_PS_PRODUCT(PA, PB)                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA], bv[j][PB], acc[i][j], 0, 0, 0);
    SNAP_PS_PRODUCT(1, 0)
    SNAP_PS_PRODUCT(0, 1)
    SNAP_PS_PRODUCT(0, 0)
#undef SNAP_PS_PRODUCT
  }
  __syncthreads();              // the last stage is read: the ring becomes the epilogue's buffer
  conv_epilogue<BM, BN, DUAL, NT, RES_INIT>(a, acc, smem, m0, n0, Meff, row_t, split);
}

// registers: 64 accumulators + 32 fragment registers + addressing; three 48 KB workgroups
// (256 threads) or two 72 KB workgroups (512 threads) per CU
template <int BM, int BN, int NT, int NST, bool RES_INIT, bool DUAL>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 3) void conv_ps_kernel(const ConvArgs a) {
  conv_ps_body<BM, BN, NT, NST, RES_INIT, DUAL>(a);
}

template <int BM, int BN, int NT, int NST>
int launch_variant(const ConvArgs& a, bool res_init, bool dual, dim3 grid, hipStream_t s) {
  if constexpr (BN == 128) {
    if (res_init && dual)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, NST, true, true>), grid, dim3(NT), 0, s, a);
    else if (res_init)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, NST, true, false>), grid, dim3(NT), 0, s, a);
    else if (dual)
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, NST, false, true>), grid, dim3(NT), 0, s, a);
    else
      hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, NST, false, false>), grid, dim3(NT), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv_ps_kernel<BM, BN, NT, NST, false, false>), grid, dim3(NT), 0, s, a);
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

}  // namespace

// Tile: 64-wide column tiles only for Cout <= 64 (the 3x3 convolutions of the first ResNet
// stage); 256-row tiles (512 threads, two workgroups per CU) once they still give two full rounds
// of the 512 slots; force: 1 = 128 rows, 2 = 256 rows (SnapConvExtras.ps_tile, tuning).
snapconv::PsTile snapconv::ps_choose_tile(int64_t M, int64_t N, int force) {
  if (N <= 64) return {128, 64, 256};
  const int64_t t256 = snap_cdiv(M, 256) * snap_cdiv(N, 128);
  if (force == 2 || (force == 0 && t256 >= 1024)) return {256, 128, 512};
  return {128, 128, 256};
}

// split-K factor of a launch with few output tiles (as conv_common.h's rule: about 768
// workgroups, at least 8 k-steps per split, bounded by the workspace); 1 = none
int snapconv::ps_ksplit(int64_t M, int Cout, int64_t nk, int bm, int bn, size_t kpartial_bytes) {
  const int64_t tiles = snap_cdiv(M, bm) * snap_cdiv((int64_t)Cout, bn);
  const int64_t target = bm == 256 ? 1024 : 768;
  if (tiles > target / 2 || nk < 16) return 1;
  int64_t S = (target + tiles - 1) / tiles;
  S = S < nk / 8 ? S : nk / 8;
  const int64_t fit = (int64_t)(kpartial_bytes / ((size_t)M * Cout * sizeof(float)));
  S = S < fit ? S : fit;
  return S >= 2 ? (int)S : 1;
}

int snapconv::launch_ps(ConvArgs a, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  if (!a.x_ps || !a.w_bf16) return SNAP_ERR_NULL;
  if (d.prologue != SNAP_PRO_NONE || d.Cin % 16 != 0 || a.rows_in || a.rows_out || a.row_count)
    return SNAP_ERR_UNSUPPORTED;
  if ((int64_t)d.N * d.H * d.W * (d.Cin / 16) * 64 >= ((int64_t)1 << 40)) return SNAP_ERR_BAD_SHAPE;
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
  if (t.bm == 256)
    st = launch_variant<256, 128, 512, 3>(a, res_init, dual, grid, s);
  else if (t.bn == 128)
    st = launch_variant<128, 128, 256, 3>(a, res_init, dual, grid, s);
  else
    st = launch_variant<128, 64, 256, 4>(a, false, false, grid, s);
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
