// 1 x 1 convolution / Dense with the ACTIVATION TILE STATIONARY IN REGISTERS: the f32-grade
// split-bf16 arithmetic of conv_split.hip at NS = 2 ("bf16x3": a_lo b_hi + a_hi b_lo + a_hi b_hi
// per MAC on v_mfma_f32_32x32x16_bf16, f32 accumulate, the same slab order and the same product
// order per accumulator: the output is BIT-IDENTICAL to conv_split's), for the shape of the
// closing / projection convolution of a bottleneck unit (snap/models/resnet.py:112-132: 1 x 1,
// Cin = 64 / 128 / 256 -> Cout >= 2 Cin, GroupNorm + ReLU in front, the unit's residual behind).
//
// conv_split.hip tiles the output 128 x 128: every column tile fetches, normalises and splits its
// 128 x Cin activation tile again (N / 128 = 2 ... 8 times per element) and streams its own copy
// of the weight panel.  Here a workgroup (four waves) owns 128 rows for ALL (or 1 / nsplit of)
// the output columns:
//   * prologue: every wave fetches ITS 32 rows x Cin straight into the MFMA A-fragment layout
//     (lane = row, 8 consecutive channels per 16-k slab), applies GroupNorm + ReLU (operands from
//     an LDS table of the tile's two images) and splits ONCE; the hi / lo fragments of all
//     Cin / 16 slabs stay in registers (Cin / 2 VGPRs);
//   * the column loop streams the split weight image (the engine's own, conv_split.hip) through
//     a three-stage LDS ring by LDS-DMA, 16 KB per stage, two stages in flight; all four waves
//     read the same B fragments and multiply them with their own rows: no A traffic and no VALU
//     in the k loop;
//   * epilogue per column tile: accumulators -> a wave-private LDS transpose -> rows of float4;
//     the residual was prefetched one column tile ahead; every global access of the loop is a
//     BUFFER access (rows beyond M read zeros / are dropped by the range check, statistics that
//     have no entry are stored out of range), so every wave issues the same number of
//     vector-memory instructions whatever the data and the s_waitcnt counts of the ring are
//     compile-time constants (rs_behind); the GroupNorm statistics of the output (also of
//     relu(y): DUAL) leave in conv_epilogue's layout, summed over the four waves in a fixed order.
//
// Measured (round 3, tools/rs_bench.py, C2 StreetView shapes with residual + statistics; isolated
// launches): 0.435 -> 0.377 ms (M = 739840, 64 -> 256), 0.268 -> 0.229 (M = 184960, 128 -> 512),
// 0.183 -> 0.161 (M = 46240, 256 -> 1024), i.e. 12-15 % over the tiled body; level at M = 147968,
// 15-25 % SLOWER on the aerial encoder's M <= 36992 layers (too few row tiles: stationary_kind
// leaves those to the tiled body).  Inside the C2 step the same layers gain 2-8 %.
// What the ablations (SNAP_RS_ABLATE alt builds, scripts/gpu_rs_ablate.sh) say about these layers:
// the time of all three shapes is (bytes through the CU boundary) / 5.0-5.5 TB/s, L2-resident
// tile traffic counted like HBM traffic -- x + weight panel per row tile + residual + output:
// 2.07 GB -> 0.377 ms, 1.22 GB -> 0.222, 0.83 GB -> 0.164, and the tiled body's larger byte
// counts (activation tile per column tile) predict ITS times the same way.  MFMAs, fragment
// fetches and the DMA issue together are 10-25 % of the kernel (removing all three: 0.340 /
// 0.178 / 0.119 ms); the epilogue alone is 0.30 of the 0.377 ms at Cin = 64.  Skewing the two
// workgroups of a CU against each other, 512-thread workgroups of 256 rows (half the weight
// stream, but one barrier chain per CU: 0.44 / 0.246 / 0.178 ms) and other column splits were
// all measured and are not faster.
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Timing-only ablations (alt builds: scripts/build_alt.sh; WRONG results): bit0 no DMA inside the
// loop, bit1 no fragment fetches, bit2 no MFMAs, bit3 no epilogue, bit4 no column loop at all
#ifndef SNAP_RS_ABLATE
#define SNAP_RS_ABLATE 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// four f32 -> hi / lo bf16 pairs (conv_split.hip's split_bf16<2>: RNE, exact residual)
__device__ __forceinline__ void split2(const f32x4& v, u32x2& hi, u32x2& lo) {
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

__device__ __forceinline__ float xor_sum3(float v) {     // over lane bits 3, 4, 5
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// s_waitcnt vmcnt(n) for an n the optimiser knows (the unrolled stage loop): one case survives
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
#define SNAP_W(N) case N: wait_vm<N>(); break;
#define SNAP_W8(B) SNAP_W(B) SNAP_W(B + 1) SNAP_W(B + 2) SNAP_W(B + 3) SNAP_W(B + 4) SNAP_W(B + 5) SNAP_W(B + 6) SNAP_W(B + 7)
    SNAP_W8(0) SNAP_W8(8) SNAP_W8(16) SNAP_W8(24) SNAP_W8(32) SNAP_W8(40) SNAP_W8(48) SNAP_W8(56)
#undef SNAP_W8
#undef SNAP_W
    default: wait_vm<0>(); break;
  }
}

// Vector-memory instructions a wave issues BEHIND the DMA of the stage it is about to read (ring of
// three: that DMA went out two stages ago), by the stage's position c in its column tile.  A stage
// runs [wait, barrier, DMA of stage + 2 (P), statistics of the previous tile if c == 0 (SR),
// MFMAs, epilogue if c == SPT - 1 (XE)].
constexpr int rs_behind(int c, bool first_tile, int SPT, int P, int SR, int XE) {
  if (first_tile && c == 0) return 0;          // (the prologue is drained as a whole)
  int x = P;
  for (int back = 2; back >= 1; --back) {
    int cc = c - back;
    if (cc < 0) {
      if (first_tile) continue;                // no such stage
      cc += SPT;
    }
    if (cc == 0) x += SR;
    if (cc == SPT - 1) x += XE;
  }
  return x;
}

// KS = Cin / 16 slabs, TN = 32-column MFMA tiles per column tile (BN = 32 TN); a wave = 32 rows
#ifndef SNAP_RS_NT
#define SNAP_RS_NT 256        // threads per workgroup: 256 (128 rows, two workgroups per CU) | 512 (256 rows, one)
#endif
template <int KS, int TN, int PRO, bool RES, bool DUAL>
__global__ __launch_bounds__(SNAP_RS_NT, 2) void conv1x1_rs_kernel(const ConvArgs a) {
  constexpr int NT = SNAP_RS_NT, NW = NT / 64;
  constexpr int BM = 32 * NW, BN = 32 * TN;
  constexpr int kStage = 16384;                 // bytes per ring stage
  constexpr int NST = 3;
  constexpr int CH = kStage / (BN * 64);        // slabs per stage
  constexpr int SPT = KS / CH;                  // stages per column tile
  static_assert(KS % CH == 0 && SPT >= 2, "whole stages; the statistics hand-over needs two barriers per tile");
  constexpr int P = kStage / 16 / NT;           // DMA pieces per thread and stage
  constexpr int B_PART = BN * 32, B_SLAB = 2 * B_PART;
  constexpr int kStgRow = 32;                   // floats per staged row (conflict-free for the b32 stores and the b128 row reads)
  constexpr int kStg = 32 * kStgRow;            // floats per wave
  constexpr int NSTAT = DUAL ? 8 : 4;           // [slot][sum, sum of squares] (x 2: of relu(y))
  constexpr int SR = DUAL ? 2 : 1;              // statistics stores per thread and tile
  constexpr int XE = (RES ? 8 : 4) * TN;        // vector-memory instructions of one epilogue
  __shared__ __attribute__((aligned(16))) char ring[NST * kStage];
  __shared__ __attribute__((aligned(16))) float staging[NW * kStg];
  __shared__ __attribute__((aligned(16))) float stats[NW * BN * NSTAT];

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row_t = blockIdx.x;
  const int Meff = a.M;
  const int m0 = row_t * BM;
  if (m0 >= Meff) return;
  const int tiles_wg = a.tiles_per_split;       // column tiles of this workgroup
  const int nbase = blockIdx.y * tiles_wg * BN;
  const int HoWo = d.Ho * d.Wo;

  // ---- weight ring ------------------------------------------------------------------------
  const char* const wt = static_cast<const char*>(a.w_bf16);
  const int64_t col_tile_bytes = (int64_t)KS * 8192;
  // piece q = tid + NT p of a stage: LDS offset 16 q ([slab][part][column][octet])
  int b_off[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int q = tid + NT * p;
    if constexpr (BN == 128) {
      b_off[p] = 16 * q;
    } else {
      constexpr int PS = B_SLAB / 16;           // pieces per slab
      const int sl = q / PS, within = q - sl * PS;
      const int part = within / (2 * BN), rem = within - part * (2 * BN);
      b_off[p] = sl * 8192 + part * 4096 + (rem >> 1) * 32 + (rem & 1) * 16;
    }
  }
  int i_t = 0, i_c = 0, i_slot = 0;             // issue cursor: column tile, stage in it, ring slot
  auto issue = [&]() {
    // (beyond the last stage: the first one again, into a slot nobody reads any more -- every
    //  stage issues the same number of instructions, the wait counts stay constants)
    const int t = i_t < tiles_wg ? i_t : 0;
    const int c = i_t < tiles_wg ? i_c : 0;
    const int n0 = nbase + t * BN;
    const char* src = wt + (int64_t)(n0 >> 7) * col_tile_bytes + c * (CH * 8192) + (n0 & 127) * 32;
    char* dst = ring + i_slot * kStage;
#pragma unroll
    for (int p = 0; p < P; ++p)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(src + b_off[p]),
                                       (lds_void_t*)(dst + 16 * (tid + NT * p)), 16, 0, 0);
    if (++i_c == SPT) { i_c = 0; ++i_t; }
    i_slot = i_slot + 1 == NST ? 0 : i_slot + 1;
  };

  // ---- output / residual windows (buffer addressing from the tile's first row) -------------
  const int rows_here = min(BM, Meff - m0);
  const int win = (int)min((int64_t)0x7ff00000, (int64_t)rows_here * d.Cout_stride * 4);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.y) + (int64_t)m0 * d.Cout_stride * 4, 0, win, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(RES ? a.residual : a.y)) +
          (int64_t)m0 * d.Cout_stride * 4, 0, win, 0x00020000);
  // float4 layout of the epilogue: pass `it` covers rows 8 it + (lane >> 3), columns 4 (lane & 7)
  const int q8 = lane & 7, rg = lane >> 3;
  const int eo = ((32 * wid + rg) * d.Cout_stride + 4 * q8) * 4;     // byte offset at it = 0, j = 0, n0 = 0
  const int eo_it = 8 * d.Cout_stride * 4;

  u32x4 res[RES ? 4 * TN : 1];
  auto load_res = [&](int n0) {
    if constexpr (RES) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int it = 0; it < 4; ++it)
          res[j * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_r, eo + it * eo_it + (n0 + 32 * j) * 4, 0, 0);
    }
  };

  issue();
  issue();

  // ---- A: 32 rows x Cin per wave, normalised and split once, kept in registers ---------------
  // GroupNorm operands (mean, rstd * gamma per (image, channel) of the tile's two images; beta):
  // a table in the ring's third stage (its first DMA goes out behind the first barrier)
  bf16x8 a_hi[KS], a_lo[KS];
  {
    constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);
    constexpr int Cin = 16 * KS;
    float* const tab = reinterpret_cast<float*>(ring + 2 * kStage);   // [image][mu | sc][Cin], beta [Cin]
    const int n_first = m0 / HoWo;
    if constexpr (need_gn) {
      for (int i = tid; i < 5 * Cin / 4; i += NT) {
        const int seg = i / (Cin / 4), c = 4 * (i - seg * (Cin / 4));
        const int n = min(n_first + (seg >> 1), d.N - 1);
        const float* src = seg == 4 ? a.gn_beta + c : ((seg & 1) ? a.gn_sc : a.gn_mu) + (int64_t)n * Cin + c;
        *reinterpret_cast<f32x4*>(tab + seg * Cin + c) = *reinterpret_cast<const f32x4*>(src);
      }
    }
    const int m = m0 + 32 * wid + l31;
    const bool ok = m < Meff;
    const int mm = ok ? m : m0;
    const float* const px = a.x + (int64_t)mm * d.Cin_stride + 8 * lhi;
    // raw rows in two batches (the second travels while the first is converted): the whole tile
    // raw AND converted at once would not fit the register file
    constexpr int KH1 = KS > 8 ? KS / 2 : KS;
    const float* const tmu = tab + (mm / HoWo - n_first) * 2 * Cin + 8 * lhi;
    const float* const tbe = tab + 4 * Cin + 8 * lhi;
    auto convert = [&](int s, const f32x4 (&xs)[2]) {
      u32x2 h[2], l[2];
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int c = 16 * s + 4 * hq;
        f32x4 v = xs[hq];
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = mu, be = mu;
        if constexpr (need_gn) {
          mu = *reinterpret_cast<const f32x4*>(tmu + c);
          sc = *reinterpret_cast<const f32x4*>(tmu + Cin + c);
          be = *reinterpret_cast<const f32x4*>(tbe + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = apply_pro<PRO>(v[e], mu[e], sc[e], be[e], d.in_scale, d.in_shift);
          v[e] = ok ? pv : 0.f;
        }
        split2(v, h[hq], l[hq]);
      }
      const u32x4 hh = {h[0][0], h[0][1], h[1][0], h[1][1]};
      const u32x4 ll = {l[0][0], l[0][1], l[1][0], l[1][1]};
      __builtin_memcpy(&a_hi[s], &hh, 16);
      __builtin_memcpy(&a_lo[s], &ll, 16);
      // (the conversion must END here: left alone, the table reads of ALL slabs -- 6 x 4
      //  registers each -- are issued in front of the first conversion and spilled)
      asm volatile("" : "+v"(a_hi[s]), "+v"(a_lo[s]) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 xv[KH1][2];
#pragma unroll
    for (int s = 0; s < KH1; ++s) {
      xv[s][0] = *reinterpret_cast<const f32x4*>(px + 16 * s);
      xv[s][1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
    }
    __syncthreads();
    if constexpr (KH1 < KS) {
      f32x4 xw[KS - KH1][2];
#pragma unroll
      for (int s = KH1; s < KS; ++s) {
        xw[s - KH1][0] = *reinterpret_cast<const f32x4*>(px + 16 * s);
        xw[s - KH1][1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KH1; ++s) convert(s, xv[s]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = KH1; s < KS; ++s) convert(s, xw[s - KH1]);
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) convert(s, xv[s]);
    }
  }
  load_res(nbase);

  float* const stg = staging + wid * kStg;
  // ---- GroupNorm statistics of the output: conv_epilogue's layout, one entry per (image, 128-row
  // slab, channel); this workgroup covers two slabs (waves 0..3 / 4..7).  Written by the epilogue
  // into LDS per wave, summed over the four waves of a slab in a fixed order after the next
  // barrier, ONE buffer store per thread (out of range where there is nothing to write: the
  // instruction count must not depend on the data).
  const bool want_stats = a.gn_partial != nullptr;
  const int64_t gn_bytes = want_stats ? (int64_t)d.N * a.gn_slabs * d.Cout * 8 : 0;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.gn_partial), 0, (int)gn_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g2 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(DUAL ? a.gn_partial2 : a.gn_partial), 0, (int)gn_bytes, 0x00020000);
  constexpr int kOob = (int)0x80000000u;
  // this thread's statistics entry: (half, slot, column)
  const int st_half = tid / (2 * BN), st_sl = (tid / BN) & 1, st_c = tid % BN;
  int st_off = kOob;                            // byte offset at n0 = 0
  if (want_stats && st_half < NW / 4) {
    const int m0h = m0 + 128 * st_half;
    const int nf = m0h / HoWo;
    const int msp = (nf + 1) * HoWo;
    const int n = nf + st_sl;
    const bool live = m0h < Meff && n < d.N && (st_sl == 0 || (m0h + 128 > msp && msp < Meff));
    if (live) {
      const int slab = (NW / 4) * row_t + st_half - (int)(((int64_t)n * HoWo) / 128);
      st_off = (int)((((int64_t)n * a.gn_slabs + slab) * d.Cout + st_c) * 8);
    }
  }
  auto reduce_stats = [&](int t) {              // t < 0: the instruction(s) only
    float t1 = 0.f, t2 = 0.f, u1 = 0.f, u2 = 0.f;
    const bool on = t >= 0 && st_off != kOob;
    if (on) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* s = stats + ((4 * st_half + w) * BN + st_c) * NSTAT + 2 * st_sl;
        t1 += s[0];
        t2 += s[1];
        if constexpr (DUAL) { u1 += s[4]; u2 += s[5]; }
      }
    }
    const int off = on ? st_off + (nbase + t * BN) * 8 : kOob;
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(t1), __float_as_uint(t2)}, rs_g, off, 0, 0);
    if constexpr (DUAL)
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(u1), __float_as_uint(u2)}, rs_g2, off, 0, 0);
  };

  const int relu_out = d.epilogue & SNAP_EPI_RELU;
  const int mw0 = m0 + 32 * wid;                // the wave's first row
  const int m_split = ((m0 + 128 * (wid >> 2)) / HoWo + 1) * HoWo;   // first row of the slab's second image
  // a wave's rows lie in image slot 0, slot 1, or (one wave per image boundary) both
  const bool w_straddle = mw0 < m_split && mw0 + 32 > m_split;
  const int w_slot = mw0 >= m_split ? 1 : 0;
  int slot = 0;
  for (int t = 0; t < ((SNAP_RS_ABLATE & 16) ? 0 : tiles_wg); ++t) {
    const int n0 = nbase + t * BN;
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < SPT; ++c) {
      // this stage's DMA landed (the younger stage's, and whatever else was issued behind, may travel)
      if (t == 0) wait_vm_n(rs_behind(c, true, SPT, P, SR, XE));
      else wait_vm_n(rs_behind(c, false, SPT, P, SR, XE));
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (!(SNAP_RS_ABLATE & 1)) issue();
      if (c == 0) reduce_stats(t - 1);
      const char* const stage = ring + slot * kStage;
      slot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
      for (int sl = 0; sl < CH; ++sl) {
        const int s = c * CH + sl;
        const char* bs = stage + sl * B_SLAB;
        bf16x8 bv[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int C = j * 32 + l31;
          const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
          if (SNAP_RS_ABLATE & 2) {
            asm volatile("" : "=v"(bv[j][0]));
            asm volatile("" : "=v"(bv[j][1]));
          } else {
            bv[j][0] = *reinterpret_cast<const bf16x8*>(p0);
            bv[j][1] = *reinterpret_cast<const bf16x8*>(p0 + B_PART);
          }
        }
        if (SNAP_RS_ABLATE & 4) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            asm volatile("" ::"v"(bv[j][0]));
            asm volatile("" ::"v"(bv[j][1]));
          }
          asm volatile("" ::"v"(a_lo[s]), "v"(a_hi[s]));
          continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bv[j][0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bv[j][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bv[j][0], acc[j], 0, 0, 0);
      }
    }
    // ---- epilogue of column tile t -----------------------------------------------------------
    if (SNAP_RS_ABLATE & 8) {
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[j][0]));   // (a 64-byte "v" operand silently drops the host stub)
      continue;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        stg[ri * kStgRow + l31] = acc[j][r];
      }
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      float h1[4] = {0.f, 0.f, 0.f, 0.f}, h2[4] = {0.f, 0.f, 0.f, 0.f};
      const bool count = want_stats && !w_straddle;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8);
        if constexpr (RES) {
          f32x4 rr;
          __builtin_memcpy(&rr, &res[j * 4 + it], 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
        if (relu_out) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = snap_relu(v[e]);
        }
        u32x4 vo;
        __builtin_memcpy(&vo, &v, 16);
        __builtin_amdgcn_raw_buffer_store_b128(vo, rs_y, eo + it * eo_it + (n0 + 32 * j) * 4, 0, 0);
        if (count) {
          const bool live = mw0 + 8 * it + rg < Meff;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = live ? v[e] : 0.f;
            const float tt = a.gn_relu ? snap_relu(y) : y;
            s1[e] += tt;
            s2[e] += tt * tt;
            if constexpr (DUAL) {
              const float rl = snap_relu(y);
              h1[e] += rl;
              h2[e] += rl * rl;
            }
          }
        } else if (want_stats) {
          // the one wave per image boundary: park the finished values, two masked passes below
          *reinterpret_cast<f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8) = v;
        }
      }
      if (want_stats) {
        // sums over the eight row groups of the wave (lane bits 3..5); lanes 0..7 publish them
        auto publish = [&](int slot_) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float r1 = xor_sum3(s1[e]), r2 = xor_sum3(s2[e]);
            float g1 = 0.f, g2 = 0.f;
            if constexpr (DUAL) { g1 = xor_sum3(h1[e]); g2 = xor_sum3(h2[e]); }
            if (rg == 0) {
              float* o = stats + (wid * BN + 32 * j + 4 * q8 + e) * NSTAT + 2 * slot_;
              o[0] = r1;
              o[1] = r2;
              if constexpr (DUAL) { o[4] = g1; o[5] = g2; }
            }
          }
        };
        if (!w_straddle) {
          publish(w_slot);
          if (rg == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float* o = stats + (wid * BN + 32 * j + 4 * q8 + e) * NSTAT + 2 * (1 - w_slot);
              o[0] = 0.f;
              o[1] = 0.f;
              if constexpr (DUAL) { o[4] = 0.f; o[5] = 0.f; }
            }
          }
        } else {
          for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; h1[e] = 0.f; h2[e] = 0.f; }
            for (int it = 0; it < 4; ++it) {
              const int m = mw0 + 8 * it + rg;
              const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8);
              const bool live = m < Meff && (m >= m_split) == (sl == 1);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float y = live ? v[e] : 0.f;
                const float tt = a.gn_relu ? snap_relu(y) : y;
                s1[e] += tt;
                s2[e] += tt * tt;
                if constexpr (DUAL) {
                  const float rl = snap_relu(y);
                  h1[e] += rl;
                  h2[e] += rl * rl;
                }
              }
            }
            publish(sl);
          }
        }
      }
    }
    // the next tile's residual (consumed by the next epilogue)
    load_res(n0 + BN);       // (beyond the last column tile: in range of the row, unused)
  }
  wait_vm<0>();              // (the padding DMAs still write LDS: not past the end of the workgroup)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  reduce_stats(tiles_wg - 1);
}

// ------------------------------------------------------------------------------------------------
// WEIGHTS STATIONARY (Cin = 64 / 128): the split weight panel of 256 output columns (Cin / 16 slabs x
// 2 column tiles x 8 KB = 64 / 128 KB) is loaded into LDS ONCE per workgroup; the eight waves of the
// (persistent, one per CU) workgroup then run INDEPENDENTLY of each other, each over its own
// sequence of 32-row tiles: rows -> registers (normalised, split), 2 x (Cin / 16) x 12 MFMAs
// against the resident panel, epilogue.  No DMA ring, no barrier and no wait count in the steady
// state; the waves drift apart, so one wave's memory-bound epilogue runs under another's MFMAs.
// What it buys (section 5j of DESIGN.md): the row-stationary kernel above re-streams the panel for
// every 128 rows -- 18 % (Cin = 64) / 30 % (Cin = 128) of all bytes crossing the CU boundary, on
// layers whose time IS those bytes over ~5.3 TB/s.
// The GroupNorm statistics of the output leave per 32-ROW slab (conv_epilogue's layout with a
// row tile of 32: snap_conv2d_tile_rows says so, the finalize pass takes any tile height): one wave
// writes its own sums, nothing is reduced across waves.
template <int KS, bool RES, int STATS /* 0 none, 1 one set, 2 also of relu(y) */>
__global__ __launch_bounds__(512, 2) void conv1x1_bs_kernel(const ConvArgs a) {
  constexpr int NT = 512, NW = NT / 64;
  constexpr int TN = 4, BN = 128, NCT = 2;      // two column tiles of 128 per workgroup
  constexpr int Cin = 16 * KS;
  constexpr int kPanel = KS * NCT * 8192;       // [column tile][slab][part][128 columns][32 B]
  constexpr int B_PART = 4096, B_SLAB = 8192;
  constexpr int kStgRow = 32;
  constexpr int kStg = 32 * kStgRow;            // floats per wave (also holds the wave's GroupNorm table)
  static_assert(5 * Cin <= kStg, "GroupNorm table in the staging tile");
  __shared__ __attribute__((aligned(16))) char panel[kPanel];
  __shared__ __attribute__((aligned(16))) float staging[NW * kStg];

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Meff = a.M;
  const int HoWo = d.Ho * d.Wo;
  const int nbase = blockIdx.y * (NCT * BN);

  // ---- the panel: both column tiles' contiguous Cin / 16 x 8 KB blocks of the weight image -------
  {
    const char* const wt = static_cast<const char*>(a.w_bf16);
    const int64_t col_tile_bytes = (int64_t)KS * 8192;
#pragma unroll
    for (int p = 0; p < kPanel / 16 / NT; ++p) {
      const int q = tid + NT * p;                                  // 16-byte piece
      const int ct = q / (KS * 512);
      const char* src = wt + (int64_t)((nbase >> 7) + ct) * col_tile_bytes + (q - ct * (KS * 512)) * 16;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src, (lds_void_t*)(panel + 16 * q), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  float* const stg = staging + wid * kStg;
  const int q8 = lane & 7, rg = lane >> 3;
  const int relu_out = d.epilogue & SNAP_EPI_RELU;
  const int ntile = (Meff + 31) >> 5;
  const int nwaves = gridDim.x * NW;
  for (int t = blockIdx.x * NW + wid; t < ntile; t += nwaves) {
    const int mw0 = 32 * t;
    const int n_first = mw0 / HoWo;
    const int m_split = (n_first + 1) * HoWo;                      // first row of the next image
    const bool straddle = mw0 + 32 > m_split && m_split < Meff;
    // ---- output / residual windows of this tile --------------------------------------------
    const int rows_here = min(32, Meff - mw0);
    const int win = rows_here * d.Cout_stride * 4;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.y) + (int64_t)mw0 * d.Cout_stride * 4, 0, win, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(RES ? a.residual : a.y)) +
            (int64_t)mw0 * d.Cout_stride * 4, 0, win, 0x00020000);
    const int eo = (rg * d.Cout_stride + 4 * q8) * 4;
    const int eo_it = 8 * d.Cout_stride * 4;
    u32x4 res[RES ? 4 * TN : 1];
    auto load_res = [&](int n0) {
      if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            res[j * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(
                rs_r, eo + it * eo_it + (n0 + 32 * j) * 4, 0, 0);
      }
    };

    // ---- A: this wave's 32 rows, normalised and split once, in registers ------------------------
    bf16x8 a_hi[KS], a_lo[KS];
    {
      const int m = mw0 + l31;
      const bool ok = m < Meff;
      const int mm = ok ? m : mw0;
      const float* const px = a.x + (int64_t)mm * d.Cin_stride + 8 * lhi;
      f32x4 xv[KS][2];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        xv[s][0] = *reinterpret_cast<const f32x4*>(px + 16 * s);
        xv[s][1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
      }
      // GroupNorm operands of the tile's (at most two) images -> the wave's staging tile:
      // [image][mu | sc][Cin], beta [Cin]  (wave-private: no barrier, LDS is in order per wave)
      for (int i = lane; i < 5 * Cin / 4; i += 64) {
        const int seg = i / (Cin / 4), c = 4 * (i - seg * (Cin / 4));
        const int n = min(n_first + (seg >> 1), d.N - 1);
        const float* src = seg == 4 ? a.gn_beta + c : ((seg & 1) ? a.gn_sc : a.gn_mu) + (int64_t)n * Cin + c;
        *reinterpret_cast<f32x4*>(stg + seg * Cin + c) = *reinterpret_cast<const f32x4*>(src);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float* const tmu = stg + (mm >= m_split ? 2 * Cin : 0) + 8 * lhi;
      const float* const tbe = stg + 4 * Cin + 8 * lhi;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x2 h[2], l[2];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const int c = 16 * s + 4 * hq;
          f32x4 v = xv[s][hq];
          const f32x4 mu = *reinterpret_cast<const f32x4*>(tmu + c);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(tmu + Cin + c);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tbe + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = apply_pro<SNAP_PRO_GN_RELU>(v[e], mu[e], sc[e], be[e], d.in_scale, d.in_shift);
            v[e] = ok ? pv : 0.f;
          }
          split2(v, h[hq], l[hq]);
        }
        const u32x4 hh = {h[0][0], h[0][1], h[1][0], h[1][1]};
        const u32x4 ll = {l[0][0], l[0][1], l[1][0], l[1][1]};
        __builtin_memcpy(&a_hi[s], &hh, 16);
        __builtin_memcpy(&a_lo[s], &ll, 16);
        asm volatile("" : "+v"(a_hi[s]), "+v"(a_lo[s]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    load_res(nbase);          // (consumed by the first epilogue, a column tile of MFMAs later)
#pragma unroll 1
    for (int ct = 0; ct < NCT; ++ct) {
      const int n0 = nbase + ct * BN;
      f32x16 acc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const char* bs = panel + (ct * KS + s) * B_SLAB;
        bf16x8 bv[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int C = j * 32 + l31;
          const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
          bv[j][0] = *reinterpret_cast<const bf16x8*>(p0);
          bv[j][1] = *reinterpret_cast<const bf16x8*>(p0 + B_PART);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bv[j][0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bv[j][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bv[j][0], acc[j], 0, 0, 0);
      }
      // ---- epilogue of (row tile, column tile) ---------------------------------------------------
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
          stg[ri * kStgRow + l31] = acc[j][r];
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        float h1[4] = {0.f, 0.f, 0.f, 0.f}, h2[4] = {0.f, 0.f, 0.f, 0.f};
        const bool count = STATS > 0 && !straddle;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8);
          if constexpr (RES) {
            f32x4 rr;
            __builtin_memcpy(&rr, &res[j * 4 + it], 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          }
          if (relu_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = snap_relu(v[e]);
          }
          u32x4 vo;
          __builtin_memcpy(&vo, &v, 16);
          __builtin_amdgcn_raw_buffer_store_b128(vo, rs_y, eo + it * eo_it + (n0 + 32 * j) * 4, 0, 0);
          if constexpr (STATS > 0) {
            if (count) {
              const bool live = mw0 + 8 * it + rg < Meff;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float y = live ? v[e] : 0.f;
                const float tt = a.gn_relu ? snap_relu(y) : y;
                s1[e] += tt;
                s2[e] += tt * tt;
                if constexpr (STATS == 2) {
                  const float rl = snap_relu(y);
                  h1[e] += rl;
                  h2[e] += rl * rl;
                }
              }
            } else {
              *reinterpret_cast<f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8) = v;   // parked for the two masked passes
            }
          }
        }
        if constexpr (STATS > 0) {
          // sums over the eight row groups (lane bits 3..5); lanes 0..7 write 4 columns x (sum, sum of squares)
          auto publish = [&](int sl) {
            const int n = n_first + sl;
            const bool wr = rg == 0 && n < d.N;
            const int slab = t - (int)(((int64_t)n * HoWo) >> 5);
            const int64_t off = (((int64_t)n * a.gn_slabs + slab) * d.Cout + n0 + 32 * j + 4 * q8) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float r1 = xor_sum3(s1[e]), r2 = xor_sum3(s2[e]);
              if (wr) *reinterpret_cast<float2*>(a.gn_partial + off + 2 * e) = float2{r1, r2};
              if constexpr (STATS == 2) {
                const float g1 = xor_sum3(h1[e]), g2 = xor_sum3(h2[e]);
                if (wr) *reinterpret_cast<float2*>(a.gn_partial2 + off + 2 * e) = float2{g1, g2};
              }
            }
          };
          if (!straddle) {
            publish(0);
          } else {
            for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; h1[e] = 0.f; h2[e] = 0.f; }
              for (int it = 0; it < 4; ++it) {
                const int m = mw0 + 8 * it + rg;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (8 * it + rg) * kStgRow + 4 * q8);
                const bool live = m < Meff && (m >= m_split) == (sl == 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float y = live ? v[e] : 0.f;
                  const float tt = a.gn_relu ? snap_relu(y) : y;
                  s1[e] += tt;
                  s2[e] += tt * tt;
                  if constexpr (STATS == 2) {
                    const float rl = snap_relu(y);
                    h1[e] += rl;
                    h2[e] += rl * rl;
                  }
                }
              }
              publish(sl);
            }
          }
        }
      }
      if (ct + 1 < NCT) load_res(n0 + BN);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same for the 3 x 3 / stride 1 / pad 1 convolution of the first ResNet stage (64 -> 64 channels,
// 136 pixels wide: too wide for the halo body of conv_split.hip, so the tiled engine runs it as
// im2col -- 9 x 4 slabs of weights re-streamed per 128 rows, 850 MB of L2 -> LDS traffic per launch
// for 378 MB of input + output, and every pixel normalised and split once per TAP).  Here the split
// weight panel of all nine taps (36 slabs x 4 KB = 144 KB) is resident in LDS and eight independent
// waves per CU each take row-aligned tiles of 30 output pixels: lane l31 holds pixel x0 - 1 + l31 of
// the row (lanes 0 and 31 are the halo columns, their outputs are dropped), so for a kernel row kh
// the wave fetches, normalises and splits ONE image row segment -- the fragments of tap kw = 1 --
// and the fragments of kw = 0 / kw = 2 are the same registers shifted by one lane (v_mov_b32 with
// the wave_shr:1 / wave_shl:1 DPP controls: one VALU instruction per register instead of ~30 for
// a conversion).  Three conversions per pixel instead of nine.  Slab order = tap major, channel
// tile minor, products lo-hi / hi-lo / hi-hi: the im2col body's, bit for bit.  No staging tile (LDS
// is full): the epilogue stores straight from the MFMA layout and takes the GroupNorm sums of a
// column from one lane's 16 rows + its partner half-wave; a tile never straddles images, its sums
// go to slab (y, tile of the row) of the image (the finalize pass is told to sum all slabs).
__device__ __forceinline__ bf16x8 lane_shift(const bf16x8& v, bool left) {
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = left ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)r[i], 0x130, 0xf, 0xf, false)    // lane l <- lane l + 1
                : (unsigned)__builtin_amdgcn_update_dpp(0, (int)r[i], 0x138, 0xf, 0xf, false);   // lane l <- lane l - 1
  bf16x8 o;
  __builtin_memcpy(&o, &r, 16);
  return o;
}

template <int STATS /* 0 none, 1 one set */>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64_kernel(const ConvArgs a) {
  constexpr int NT = 512, NW = NT / 64;
  constexpr int KS = 4, TAPS = 9, TN = 2, Cin = 64, TP = 30;
  constexpr int B_PART = 2048, B_SLAB = 4096;
  constexpr int kPanel = TAPS * KS * B_SLAB;      // 147456
  constexpr int kTab = 3 * Cin;                   // floats per wave: mu | sc | beta of the tile's image
  __shared__ __attribute__((aligned(16))) char panel[kPanel];
  __shared__ __attribute__((aligned(16))) float tables[NW * kTab];

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int W = d.W, H = d.H;
  const int TX = (W + TP - 1) / TP;               // tiles per image row

  {
    const char* const wt = static_cast<const char*>(a.w_bf16);
#pragma unroll
    for (int p = 0; p < kPanel / 16 / NT; ++p) {
      const int q = tid + NT * p;                 // 16-byte piece: [slab][part][64 columns][2 octets]
      const int sl = q >> 8, within = q & 255;
      const int part = within >> 7, rem = within & 127;
      const char* src = wt + (int64_t)sl * 8192 + part * 4096 + (rem >> 1) * 32 + (rem & 1) * 16;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src, (lds_void_t*)(panel + 16 * q), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  float* const tab = tables + wid * kTab;
  const int relu_out = d.epilogue & SNAP_EPI_RELU;
  const int ntile = d.N * H * TX;
  const int nwaves = gridDim.x * NW;
  int n_tab = -1;
  for (int t = blockIdx.x * NW + wid; t < ntile; t += nwaves) {
    const int n = t / (H * TX);
    const int ry = t - n * (H * TX);
    const int y = ry / TX, tx = ry - y * TX;
    const int x0 = tx * TP;
    const int xl = x0 - 1 + l31;                  // this lane's pixel column
    const bool px_ok = xl >= 0 && xl < W;
    const int xc = min(max(xl, 0), W - 1);
    if (n != n_tab) {                             // GroupNorm operands of image n -> the wave's table
      for (int i = lane; i < kTab / 4; i += 64) {
        const int seg = i / (Cin / 4), c = 4 * (i - seg * (Cin / 4));
        const float* src = seg == 2 ? a.gn_beta + c : (seg ? a.gn_sc : a.gn_mu) + (int64_t)n * Cin + c;
        *reinterpret_cast<f32x4*>(tab + seg * Cin + c) = *reinterpret_cast<const f32x4*>(src);
      }
      n_tab = n;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float* const tmu = tab + 8 * lhi;
    auto load_row = [&](int kh, f32x4 (&xs)[KS][2]) {
      const int yy = min(max(y + kh - 1, 0), H - 1);          // (a row outside the image: loaded, never used)
      const float* p = a.x + (((int64_t)n * H + yy) * W + xc) * d.Cin_stride + 8 * lhi;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        xs[s][0] = *reinterpret_cast<const f32x4*>(p + 16 * s);
        xs[s][1] = *reinterpret_cast<const f32x4*>(p + 16 * s + 4);
      }
    };
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    auto do_row = [&](int kh, const f32x4 (&xs)[KS][2]) {
      const int yy = y + kh - 1;
      if (yy < 0 || yy >= H) return;                          // wave-uniform: the row is zero padding
      bf16x8 f_hi[KS], f_lo[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x2 h[2], l[2];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const int c = 16 * s + 4 * hq;
          f32x4 v = xs[s][hq];
          const f32x4 mu = *reinterpret_cast<const f32x4*>(tmu + c);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(tmu + Cin + c);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tmu + 2 * Cin + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = apply_pro<SNAP_PRO_GN_RELU>(v[e], mu[e], sc[e], be[e], d.in_scale, d.in_shift);
            v[e] = px_ok ? pv : 0.f;
          }
          split2(v, h[hq], l[hq]);
        }
        const u32x4 hh = {h[0][0], h[0][1], h[1][0], h[1][1]};
        const u32x4 ll = {l[0][0], l[0][1], l[1][0], l[1][1]};
        __builtin_memcpy(&f_hi[s], &hh, 16);
        __builtin_memcpy(&f_lo[s], &ll, 16);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          // tap kw of output lane l = the pixel of lane l + kw - 1
          const bf16x8 a_hi = kw == 1 ? f_hi[s] : lane_shift(f_hi[s], kw == 2);
          const bf16x8 a_lo = kw == 1 ? f_lo[s] : lane_shift(f_lo[s], kw == 2);
          const char* bs = panel + ((kh * 3 + kw) * KS + s) * B_SLAB;
          bf16x8 bv[TN][2];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int C = j * 32 + l31;
            const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
            bv[j][0] = *reinterpret_cast<const bf16x8*>(p0);
            bv[j][1] = *reinterpret_cast<const bf16x8*>(p0 + B_PART);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bv[j][0], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bv[j][1], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bv[j][0], acc[j], 0, 0, 0);
        }
      }
    };
    // the rows of the next kernel row travel while this one is converted / multiplied
    f32x4 xa[KS][2], xb[KS][2];
    load_row(0, xa);
    load_row(1, xb);
    do_row(0, xa);
    load_row(2, xa);
    do_row(1, xb);
    do_row(2, xa);

    // ---- epilogue straight from the MFMA layout: lane = column 32 j + l31, tile rows 8 (r >> 2) + 4 lhi + (r & 3)
    float* const yb = a.y + (((int64_t)n * H + y) * W + x0 - 1) * d.Cout_stride + l31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;      // tile row = pixel x0 - 1 + ri
        float v = acc[j][r];
        if (relu_out) v = snap_relu(v);
        const bool live = ri >= 1 && ri <= TP && x0 - 1 + ri < W;
        if (live) yb[(int64_t)ri * d.Cout_stride + 32 * j] = v;
        if constexpr (STATS > 0) {
          const float y0 = live ? v : 0.f;
          const float tt = a.gn_relu ? snap_relu(y0) : y0;
          s1 += tt;
          s2 += tt * tt;
        }
      }
      if constexpr (STATS > 0) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lhi == 0)
          *reinterpret_cast<float2*>(a.gn_partial + (((int64_t)n * a.gn_slabs + ry) * d.Cout + 32 * j + l31) * 2) =
              float2{s1, s2};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ... and for the 7 x 7 / stride 2 / pad 3 root convolution of an RGB image stored with four floats
// per pixel (conv_split_root_kernel: K slab = 4 consecutive pixels x 4 floats of one kernel row, 14
// slabs): the 56 KB panel resident in LDS, eight independent waves, a wave = a row-aligned tile of
// 32 output pixels; per kernel row a lane fetches the 2 x 32 bytes its two k-octets cover (pixels
// 2 xo - 3 + 4 g + 2 lhi + {0, 1}), applies the affine prologue, zeroes what lies outside the
// image, splits, and multiplies against the resident slab.  Same slab and product order as the
// tiled root kernel: bit-identical.  Output straight from the MFMA layout.
template <int PRO>
__global__ __launch_bounds__(512, 2) void conv_root_ws64_kernel(const ConvArgs a) {
  constexpr int NT = 512, NW = NT / 64;
  constexpr int NSLAB = 14, TN = 2;
  constexpr int B_PART = 2048, B_SLAB = 4096;
  constexpr int kPanel = NSLAB * B_SLAB;          // 57344
  __shared__ __attribute__((aligned(16))) char panel[kPanel];

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int W = d.W, H = d.H, Wo = d.Wo, Ho = d.Ho;
  const int TX = (Wo + 31) / 32;
  {
    const char* const wt = static_cast<const char*>(a.w_bf16);
#pragma unroll
    for (int p = 0; p < kPanel / 16 / NT; ++p) {
      const int q = tid + NT * p;                 // 16-byte piece: [slab][part][64 columns][2 octets]
      const int sl = q >> 8, within = q & 255;
      const int part = within >> 7, rem = within & 127;
      const char* src = wt + (int64_t)sl * 8192 + part * 4096 + (rem >> 1) * 32 + (rem & 1) * 16;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src, (lds_void_t*)(panel + 16 * q), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int relu_out = d.epilogue & SNAP_EPI_RELU;
  const int ntile = d.N * Ho * TX;
  const int nwaves = gridDim.x * NW;
  for (int t = blockIdx.x * NW + wid; t < ntile; t += nwaves) {
    const int n = t / (Ho * TX);
    const int ry = t - n * (Ho * TX);
    const int yo = ry / TX, x0 = (ry - yo * TX) * 32;
    const int xo = x0 + l31;                      // this lane's output pixel (may lie beyond the row: dropped)
    const int xi0 = 2 * xo - 3 + 2 * lhi;         // first input pixel of the lane's octet at g = 0
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* const img = a.x + (int64_t)n * H * W * 4;
    auto load_row = [&](int kh, f32x4 (&xs)[2][2]) {
      const int yy = min(max(2 * yo - 3 + kh, 0), H - 1);     // (a row outside the image: loaded, never used)
      const float* p = img + (int64_t)yy * W * 4;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int xi = min(max(xi0 + 4 * g + q, 0), W - 1);
          xs[g][q] = *reinterpret_cast<const f32x4*>(p + 4 * xi);
        }
    };
    auto do_row = [&](int kh, const f32x4 (&xs)[2][2]) {
      const int yy = 2 * yo - 3 + kh;
      if (yy < 0 || yy >= H) return;                          // wave-uniform: zero padding
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        u32x2 h[2], l[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int xi = xi0 + 4 * g + q;
          const bool in = xi >= 0 && xi < W;
          f32x4 v = xs[g][q];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = apply_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
            v[e] = in ? pv : 0.f;
          }
          split2(v, h[q], l[q]);
        }
        const u32x4 hh = {h[0][0], h[0][1], h[1][0], h[1][1]};
        const u32x4 ll = {l[0][0], l[0][1], l[1][0], l[1][1]};
        bf16x8 a_hi, a_lo;
        __builtin_memcpy(&a_hi, &hh, 16);
        __builtin_memcpy(&a_lo, &ll, 16);
        const char* bs = panel + (kh * 2 + g) * B_SLAB;
        bf16x8 bv[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int C = j * 32 + l31;
          const char* p0 = bs + C * 32 + ((lhi ^ ((C >> 3) & 1)) * 16);
          bv[j][0] = *reinterpret_cast<const bf16x8*>(p0);
          bv[j][1] = *reinterpret_cast<const bf16x8*>(p0 + B_PART);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bv[j][0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bv[j][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bv[j][0], acc[j], 0, 0, 0);
      }
    };
    // the next kernel row's pixels travel while this one is converted / multiplied
    f32x4 xa[2][2], xb[2][2];
    load_row(0, xa);
#pragma unroll 1
    for (int kh = 0; kh < 6; kh += 2) {
      load_row(kh + 1, xb);
      do_row(kh, xa);
      load_row(kh + 2, xa);
      do_row(kh + 1, xb);
    }
    do_row(6, xa);

    float* const yb = a.y + (((int64_t)n * Ho + yo) * Wo + x0) * d.Cout_stride + l31;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float v = acc[j][r];
        if (relu_out) v = snap_relu(v);
        if (x0 + ri < Wo) yb[(int64_t)ri * d.Cout_stride + 32 * j] = v;
      }
  }
}

template <int KS, int TN, bool RES>
int launch_rs_dual(const ConvArgs& a, dim3 grid, bool dual, hipStream_t s) {
  if (dual)
    hipLaunchKernelGGL((conv1x1_rs_kernel<KS, TN, SNAP_PRO_GN_RELU, RES, true>), grid, dim3(SNAP_RS_NT), 0, s, a);
  else
    hipLaunchKernelGGL((conv1x1_rs_kernel<KS, TN, SNAP_PRO_GN_RELU, RES, false>), grid, dim3(SNAP_RS_NT), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

template <int KS, int TN>
int launch_rs_res(const ConvArgs& a, dim3 grid, bool dual, hipStream_t s) {
  return (a.d.epilogue & SNAP_EPI_RESIDUAL) ? launch_rs_dual<KS, TN, true>(a, grid, dual, s)
                                            : launch_rs_dual<KS, TN, false>(a, grid, dual, s);
}

template <int KS, bool RES>
int launch_bs_stats(const ConvArgs& a, dim3 grid, hipStream_t s) {
  const int stats = !a.gn_partial ? 0 : (a.gn_partial2 ? 2 : 1);
  if (stats == 2)
    hipLaunchKernelGGL((conv1x1_bs_kernel<KS, RES, 2>), grid, dim3(512), 0, s, a);
  else if (stats == 1)
    hipLaunchKernelGGL((conv1x1_bs_kernel<KS, RES, 1>), grid, dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((conv1x1_bs_kernel<KS, RES, 0>), grid, dim3(512), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

}  // namespace

int snapconv::stationary_kind(const SnapConvDesc& d, int parts, bool row_lists) {
  const int mode = hint_mode(d.tile_hint);
  if (parts != 2 || mode == 1 || row_lists) return 0;
  // 3: the 3 x 3 of the first ResNet stage (64 -> 64 channels, images too wide for the halo body)
  if (d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 && d.H == d.Ho && d.W == d.Wo &&
      d.Cin == 64 && d.Cout == 64 && d.Cout_stride == 64 && (d.Cin_stride & 3) == 0 &&
      d.prologue == SNAP_PRO_GN_RELU && !(d.epilogue & ~SNAP_EPI_RELU) && mode != 3 && mode != 4) {
    const int64_t HW3 = (int64_t)d.H * d.W, M3 = (int64_t)d.N * HW3;
    // measured in the C2 step: 0.848 -> 0.623 ms over three launches at M = 739 840, 0.212 -> 0.159 at
    // M = 147 968 (a first version that converted every pixel once per TAP, as the im2col body does,
    // gained 5 %: the conversion, not the weight stream, bounds this layer)
    if (128 + 2 * d.W + 2 > 288 && HW3 >= 32 && M3 <= 0x7fffffffLL && (M3 >= 40000 || mode == 2) &&
        (int64_t)d.N * d.H * ((d.W + 29) / 30) * d.Cout * 8 < 0x7ff00000LL)
      return 3;
    return 0;
  }
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad_t || d.pad_l || d.H != d.Ho || d.W != d.Wo)
    return 0;
  if (d.prologue != SNAP_PRO_GN_RELU) return 0;
  if (d.Cin != 64 && d.Cin != 128 && d.Cin != 256) return 0;
  if ((d.Cin_stride & 3) || (d.Cout_stride & 3)) return 0;
  const int bn = d.Cin == 256 ? 64 : 128;
  if (d.Cout % bn != 0 || d.Cout < 256 || d.Cout < 2 * d.Cin) return 0;   // (expansions: narrower outputs measured slower)
  if (d.epilogue & ~(SNAP_EPI_RESIDUAL | SNAP_EPI_RELU)) return 0;
  const int64_t HoWo = (int64_t)d.Ho * d.Wo, M = (int64_t)d.N * HoWo;
  if (M > 0x7fffffffLL) return 0;
  if ((int64_t)256 * d.Cout_stride * 4 >= 0x7ff00000LL) return 0;
  // measured (tools/rs_bench.py): 12-15 % faster than the tiled body at M = 46240 ... 739840, level
  // at M = 147968 (K = 64), 15-25 % slower on the aerial encoder's M <= 36992 layers
  if (M < 40000 && mode != 2 && mode != 4) return 0;
  // weights-stationary: the panel of 256 columns must fit LDS next to the staging tiles
  if (mode != 3 && mode != 4 && d.Cin <= 128 && d.Cout % 256 == 0 && HoWo >= 32 &&
      d.N * (HoWo / 32 + 2) * (int64_t)d.Cout * 8 < 0x7ff00000LL)
    return 2;
  if (HoWo < SNAP_RS_NT / 2) return 0;                           // at most two images per row tile
  if (d.N * (HoWo / 128 + 2) * (int64_t)d.Cout * 8 >= 0x7ff00000LL) return 0;
  // the row tile (and with it the statistics layout) must be the tiled engine's
  if (choose_tile(M, d.Cout, d.tile_hint, desc_k(d)).bm != 128) return 0;
  return 1;
}

// the RGB root convolution on the weights-stationary kernel (called by launch_split_root where it applies)
int snapconv::launch_root_ws(const ConvArgs& a, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  const int64_t nt = (int64_t)d.N * d.Ho * ((d.Wo + 31) / 32);
  const dim3 grid((unsigned)(nt / 8 < 256 ? (nt + 7) / 8 : 256));
  if (d.prologue == SNAP_PRO_AFFINE)
    hipLaunchKernelGGL((conv_root_ws64_kernel<SNAP_PRO_AFFINE>), grid, dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((conv_root_ws64_kernel<SNAP_PRO_NONE>), grid, dim3(512), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

int snapconv::launch_bs(ConvArgs a, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  a.gn_slabs = (d.Ho * d.Wo) / 32 + 2;
  a.ksplit = 1;
  if (d.KH == 3) {                               // the first stage's 3 x 3 (stationary_kind == 3)
    if (a.gn_partial2_done) *a.gn_partial2_done = 0;
    a.gn_slabs = d.H * ((d.W + 29) / 30);         // one slab per tile of an image (snap_conv2d_tile_rows_ex < 0)
    const int64_t nt = (int64_t)d.N * a.gn_slabs;
    const dim3 g3((unsigned)(nt / 8 < 256 ? (nt + 7) / 8 : 256));
    if (a.gn_partial)
      hipLaunchKernelGGL((conv3x3_ws64_kernel<1>), g3, dim3(512), 0, s, a);
    else
      hipLaunchKernelGGL((conv3x3_ws64_kernel<0>), g3, dim3(512), 0, s, a);
    SNAP_CHECK_LAUNCH();
    return SNAP_OK;
  }
  if (a.gn_partial2_done) *a.gn_partial2_done = (a.gn_partial2 && a.gn_partial) ? 1 : 0;
  const int64_t ntile = snap_cdiv(a.M, 32);
  // one persistent workgroup per CU (the panel takes most of its LDS); fewer where the rows run out
  const int wgs = (int)(ntile / 8 < 256 ? (ntile + 7) / 8 : 256);
  const dim3 grid((unsigned)wgs, (unsigned)(d.Cout / 256));
  const bool res = (d.epilogue & SNAP_EPI_RESIDUAL) != 0;
  if (d.Cin == 64) return res ? launch_bs_stats<4, true>(a, grid, s) : launch_bs_stats<4, false>(a, grid, s);
  return res ? launch_bs_stats<8, true>(a, grid, s) : launch_bs_stats<8, false>(a, grid, s);
}

int snapconv::launch_rs(ConvArgs a, hipStream_t s) {
  const SnapConvDesc& d = a.d;
  const int bn = d.Cin == 256 ? 64 : 128;
  const int tiles = d.Cout / bn;
  const int64_t nrow = snap_cdiv(a.M, SNAP_RS_NT / 2);
  if (nrow > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  // Column split: the fewest workgroup "rounds" over the 256 CUs, counting the activation prologue
  // as half a column tile of work
  int best = 1;
  double best_cost = 1e30;
  for (int ns = 1; ns <= tiles; ns *= 2) {
    if (tiles % ns) break;
    const double rounds = (double)snap_cdiv(nrow * ns, (int64_t)(SNAP_RS_NT == 512 ? 256 : 512));
    const double cost = rounds * (tiles / ns + 0.5);
    if (cost < best_cost * 0.97) { best_cost = cost; best = ns; }
  }
  if (a.rs_nsplit > 0 && tiles % a.rs_nsplit == 0) best = a.rs_nsplit;
  a.tiles_per_split = tiles / best;
  a.gn_slabs = (d.Ho * d.Wo) / 128 + 2;
  a.ksplit = 1;
  const bool dual = a.gn_partial2 != nullptr && a.gn_partial != nullptr;
  if (a.gn_partial2_done) *a.gn_partial2_done = dual ? 1 : 0;
  const dim3 grid((unsigned)nrow, (unsigned)best);
  if (d.Cin == 64) return launch_rs_res<4, 4>(a, grid, dual, s);
  if (d.Cin == 128) return launch_rs_res<8, 4>(a, grid, dual, s);
  return launch_rs_res<16, 2>(a, grid, dual, s);
}
