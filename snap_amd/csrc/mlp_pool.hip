// Fusion MLP + vertical max pooling of the StreetView encoder in ONE kernel: the two Dense layers
// of StreetViewEncoder.fusion_mlp (snap/models/streetview_encoder.py:279-286, layers.py:55-78)
// followed by VerticalPooling('max') (snap/models/bev_mapper.py:78-88), for the observed voxels
// only.  Neither the hidden activations ([rows, 256] f32, written and re-read by the unfused
// chain) nor the dense feature volume ([B, X, Y, Z, D], written, zero-filled, re-read by the
// pooling) touch memory: the kernel reads the lift's `pooled` rows once and writes the plane.
//
// Arithmetic = the split-bf16 engine at NS = 2 ("bf16x3", conv_split.hip): f32 operands split
// into hi + lo bf16 parts, three part products per MAC on v_mfma_f32_32x32x16_bf16, f32
// accumulation, the same slab order and the same product order per accumulator -- the plane is
// bitwise what conv_split + fill_masked_rows + vertical_pool produce.
//
// One workgroup (4 waves) = 128 rows of the compacted row list (snap_compact_rows_u8: observed
// voxels in (b, x, y, z) order, so a column's levels are consecutive rows).  Each WAVE owns 32
// rows and ALL hidden columns, and both GEMMs are computed TRANSPOSED (weights as the first MFMA
// operand, rows as the second), so that an accumulator lane holds ONE row and 16 hidden columns:
//   GEMM0  hidden^T[256 x 32 rows] = W0^T x^T      x: global -> registers -> split -> LDS
//                                                  W0: packed image -> LDS by LDS-DMA (as conv_split)
//   GEMM1  out^T[128 x 32 rows]    = W1^T relu(hidden + b0)^T
//          the second operand comes STRAIGHT FROM THE GEMM0 ACCUMULATORS (bias, ReLU, split in
//          registers): lane (row, half) of accumulator tile t holds hidden columns
//          32t + 8q + 4 half + {0..3}, q = 0..3; k-step s of the tile takes q = 2s, 2s+1, and one
//          v_permlane32_swap per register exchanges the middle quads of the two half-waves so
//          that half 0 holds k = 0..7 and half 1 holds k = 8..15 of the slab -- the MFMA operand
//          order of the unfused engine.  The fragments are built in place of the accumulators
//          before the loop; W1 (the same packed image) streams through an eight-slot LDS ring,
//          one k-step per slot, up to seven in flight.
//   max    out (+ b1) -> LDS, a scan down the rows per output channel with a flush at every
//          column change -> float atomic max into the plane (prefilled with -inf; max is
//          order-independent, so the result is deterministic); a finalize pass turns untouched
//          columns into zeros / valid = 0 (the reference's where(any valid, max, 0)).
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct MlpPoolArgs {
  const float* x;          // [*, x_stride] f32 rows (the lift's pooled statistics)
  int64_t x_stride;
  int Cin;                 // logical input width (257)
  const int32_t* rows;     // compacted row list
  const int32_t* row_count;
  int M;                   // upper bound of the row count
  const char* w0;          // split image of W0 [Cin, H] (snap_conv2d_pack_weights_split_bf16, parts 2)
  int ctiles0;
  const float* b0;
  int H;
  const char* w1;          // split image of W1 [H, D] (same packing, D <= 128: one column tile)
  const float* b1;
  int D;
  int Z;                   // levels per column: column id = row / Z
  float* plane;            // [ncols, D]
  // GEMM0 slabs [skip_lo, skip_lo + skip_n) are not visited: the listed rows hold exact zeros
  // there (and may not have written them): single-observation rows, SnapLiftDesc.class_rows
  int skip_lo, skip_n;
  // GATHER (the lift inside the consumer): the listed rows are voxels with ONE visible observation
  // and `recs` holds their tap records (lift.hip: tap byte offset | clamp flags | wi1 | wj1 | score);
  // slab s < nmean of such a row = the bilinear blend of 16 channels of four image taps (mean = the
  // observation, weight e / e == 1), slab 2 nmean = (score, 0, ...), the variance slabs are zero
  const char* fimg;        // f_images [B, V, h, w, C] f32 (< 4 GB: 32-bit byte offsets)
  const uint32_t* recs;    // [*, 8]
  uint32_t Cb, Wb;         // bytes per pixel / per image row
  int nmean;               // feature_dim / 16
  // consecutive tiles owned by one XCD (0: dispatch order): a tile's taps then stay in ITS L2
  int xcd_group;
};

// hi / lo bf16 parts of four f32 (as conv_split.hip: one v_cvt_pk per pair, exact residual)
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
    const f32x2 rs = {pr[0] - __uint_as_float(u << 16), pr[1] - __uint_as_float(u & 0xffff0000u)};
    const bf16x2 bl = __builtin_convertvector(rs, bf16x2);
    __builtin_memcpy(&u, &bl, 4);
    lo[h] = u;
  }
}

__device__ __forceinline__ void atomic_max_f32(float* p, float v) {
  const unsigned u = __float_as_uint(v);
  if (u >> 31)
    atomicMin(reinterpret_cast<unsigned*>(p), u);        // negative: larger float = smaller bits
  else
    atomicMax(reinterpret_cast<int*>(p), (int)u);
}

#ifndef SNAP_MLP_POOL_PAIRS
#define SNAP_MLP_POOL_PAIRS 1      // 1: GEMM1 takes TWO k-steps per barrier (four 16 KB ring slots); 0: one (eight 8 KB slots)
#endif
// NT = 256: 128 rows per workgroup, two workgroups per CU.  NT = 512: 256 rows (eight waves), one
// workgroup per CU -- the same 8 waves per CU, and W0 / W1 (426 KB per workgroup, 25 GB per C2 step
// at 128 rows) stream from L2 once per 256 rows.  Measured at C2 (scripts/gpu_mlp_nt.sh, round 3):
// 3.64 ms per step against 3.30 ms at NT = 256 -- halving the weight stream buys nothing, eight
// waves behind ONE barrier chain lose more than two independent workgroups of four: the kernel is
// bound by its phase structure, not by the L2 -> LDS stream.  NT = 256 stays.
#ifndef SNAP_MLP_POOL_NT
#define SNAP_MLP_POOL_NT 256
#endif
template <int N0, bool RELU_IN, bool XSPLIT, int NT, int NST = 2, bool GATHER = false>
__global__ __launch_bounds__(NT, 2) void mlp2_pool_kernel(const MlpPoolArgs a) {
  // NST = 3 (pre-split rows only, the default for them): the GEMM0 slabs (rows + W0, 24 KB) travel TWO
  // ahead through a three-stage ring -- one barrier per slab as before, but a slab's 24 MFMAs per wave
  // (0.3 us) no longer wait for a memory round trip (~1 us) that started one slab earlier.  The ring
  // takes 72 KB, so W1's first pair cannot travel during GEMM0 any more; all four ring slots are
  // issued after GEMM0 and arrive under the ReLU / split conversion.  C2 map, 6.8 M rows:
  // 3.09-3.10 ms against 3.21-3.39 ms (tools/mlp_pool_bench.py), same bits.
  static_assert(NST == 2 || (NST == 3 && XSPLIT && NT == 256), "the ring is the LDS-DMA path's");
  static_assert(!GATHER || (!XSPLIT && !RELU_IN && NST == 2), "the gather stages through registers");
  constexpr int BM = NT / 2, N1 = 128;
  constexpr int RPP = NT / 4;                                 // rows staged per pass (4 threads per row)
  constexpr int T0 = N0 / 32, T1 = N1 / 32;
  constexpr int A_PART = BM * 32, A_ST = 2 * A_PART;          // 8 KB per stage at 128 rows
  constexpr int B0_ST = (N0 / 128) * 8192;                    // 16 KB per stage at N0 = 256
  constexpr int kB0 = NST * A_ST;
  // [GEMM0 stages | W1 ring] share the front of the buffer with the [BM][128] f32 tile of the max
  // scan; 128 rows: the ring lies over the GEMM0 stages (64 KB in all), 256 rows: behind them
  constexpr int kRing = NT == 256 ? 0 : 65536;
  constexpr int kBias = NT == 256 ? (NST == 3 ? 73728 : 65536) : 131072;   // b0 [N0] | b1 [N1] behind
  static_assert(kB0 + NST * B0_ST <= (NST == 3 ? 73728 : 65536), "GEMM0 stages overlap the ring / the bias table");
  static_assert(BM * N1 * 4 <= kBias, "scan tile overlaps the bias table");
  __shared__ __attribute__((aligned(16))) float smem[kBias / 4 + N0 + N1];   // 65.5 KB (two per CU) / 129.5 KB
  char* const sm = reinterpret_cast<char*>(smem);
  float* const bias0 = reinterpret_cast<float*>(sm + kBias);
  float* const bias1 = bias0 + N0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Meff = min(*a.row_count, a.M);
  int bid = blockIdx.x;
  if (a.xcd_group > 0) {               // runs of G tiles per XCD (workgroups go round-robin over the XCDs)
    const int G = a.xcd_group, xc = bid & 7, sq = bid >> 3;
    bid = ((sq / G) * 8 + xc) * G + (sq % G);
  }
  const int m0 = bid * BM;
  if (m0 >= Meff) return;

  for (int i = tid; i < N0 + N1; i += NT)
    bias0[i] = i < N0 ? (i < a.H ? a.b0[i] : 0.f) : (i - N0 < a.D ? a.b1[i - N0] : 0.f);

  // ---- GEMM0: A staging as conv_split_body (4 threads per row, 2 rows per thread) --------------
  const int akq = tid & 3;
  const float* r_px[2];
  bool r_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (tid >> 2) + RPP * i;
    r_ok[i] = m < Meff;
    r_px[i] = a.x + (int64_t)a.rows[r_ok[i] ? m : m0] * a.x_stride;
  }
  f32x4 xa[2];
  bool xin[2];
  int cur_c = 0;
  // GATHER: per row the four tap offsets (this thread's 16-byte quad folded in), the four bilinear
  // weights and the score; ta = the taps of the slab in flight
  uint32_t g_o[2][4];
  float g_w[2][4], g_s[2];
  f32x4 ta[2][4];
  if constexpr (GATHER) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (tid >> 2) + RPP * i;
      const uint4* rp = reinterpret_cast<const uint4*>(a.recs + (int64_t)a.rows[r_ok[i] ? m : m0] * 8);
      const uint4 r0 = rp[0];
      g_s[i] = __uint_as_float(rp[1].x);
      const float wi1 = __uint_as_float(r0.z), wj1 = __uint_as_float(r0.w);
      const float wi0 = 1.f - wi1, wj0 = 1.f - wj1;                 // (lift.hip phase B, the same products)
      g_w[i][0] = wi0 * wj0; g_w[i][1] = wi0 * wj1; g_w[i][2] = wi1 * wj0; g_w[i][3] = wi1 * wj1;
      const uint32_t o00 = r0.x + 16u * (tid & 3);
      const uint32_t o01 = o00 + ((r0.y >> 9) & 1u ? a.Cb : 0u);
      const uint32_t o10 = o00 + ((r0.y >> 8) & 1u ? a.Wb : 0u);
      g_o[i][0] = o00; g_o[i][1] = o01; g_o[i][2] = o10; g_o[i][3] = o10 + (o01 - o00);
    }
  }
#ifndef SNAP_MLP_POOL_ABLATE
#define SNAP_MLP_POOL_ABLATE 0     // timing experiments only (wrong results): 1 = no A loads,
#endif                             // 2 = no A loads / split / LDS stores, 4 = no GEMM1, 8 = no max scan
  auto load_a = [&](int ct) {
    if constexpr (GATHER) {
      cur_c = ct;
      if (ct < a.nmean) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            ta[i][t] = *reinterpret_cast<const f32x4*>(a.fimg + (g_o[i][t] + 64u * (uint32_t)ct));
      }
      return;
    }
    if (SNAP_MLP_POOL_ABLATE & 3) { cur_c = 0; xa[0] = xa[1] = f32x4{1.f, 1.f, 1.f, 1.f}; xin[0] = xin[1] = true; return; }
    const int c = ct * 16 + 4 * akq;
    cur_c = c;
    const bool cvalid = c < a.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xin[i] = r_ok[i] && cvalid;
      xa[i] = *reinterpret_cast<const f32x4*>(xin[i] ? r_px[i] + c : a.x);
    }
  };
  auto store_a = [&](int buf) {
    if (SNAP_MLP_POOL_ABLATE & 2) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 2) + RPP * i;
      f32x4 v;
      if constexpr (GATHER) {
        if (cur_c < a.nmean) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = ((g_w[i][0] * ta[i][0][e] + g_w[i][1] * ta[i][1][e]) + g_w[i][2] * ta[i][2][e]) + g_w[i][3] * ta[i][3][e];
        } else {
          v = f32x4{akq == 0 ? g_s[i] : 0.f, 0.f, 0.f, 0.f};
        }
        if (!r_ok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        v = xa[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = RELU_IN ? snap_relu(v[e]) : v[e];
          v[e] = (xin[i] && cur_c + e < a.Cin) ? pv : 0.f;
        }
      }
      u32x2 hi, lo;
      split2(v, hi, lo);
      const int oct = (akq >> 1) ^ ((row >> 3) & 1);
      char* dst = sm + buf * A_ST + row * 32 + oct * 16 + (akq & 1) * 8;
      *reinterpret_cast<u32x2*>(dst) = hi;
      *reinterpret_cast<u32x2*>(dst + A_PART) = lo;
    }
  };
  // W0 slab s: per column tile of 128 one contiguous 8 KB block [part][column][32 B]
  constexpr int B0_PIECES = B0_ST / 16 / NT;
  auto issue_b0 = [&](int buf, int s) {
#pragma unroll
    for (int p = 0; p < B0_PIECES; ++p) {
      const int slot = tid + NT * p;
      const int j = slot >> 9;
      const char* src = a.w0 + ((int64_t)j * a.ctiles0 + s) * 8192 + (slot & 511) * 16;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                       (lds_void_t*)(sm + kB0 + buf * B0_ST + 16 * slot), 16, 0, 0);
    }
  };

  // W1 k-step ks (8 KB: [part][column][32 B]) -> slot (ks + 6) & 7 of an eight-slot ring over the
  // first 64 KB.  A stage is consumed in 12 MFMAs (~0.2 us) but takes ~1 us to arrive, so up to
  // seven k-steps are kept in flight; slots 6, 7 lie behind the GEMM0 stages: k-steps 0 and 1
  // travel while GEMM0 runs.
  auto issue_b1 = [&](int ks) {
    static_assert(SNAP_MLP_POOL_PAIRS || NT == 256, "the one-k-step ring is the 256-thread layout");
    const char* src = a.w1 + (int64_t)ks * 8192 + tid * 16;
    char* dst = sm + ((ks + 6) & 7) * 8192 + tid * 16;
    __builtin_amdgcn_global_load_lds((cglobal_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((cglobal_void_t*)(src + 4096), (lds_void_t*)(dst + 4096), 16, 0, 0);
  };
  // pair p = k-steps 2p, 2p + 1 (16 KB) -> slot (p + 3) & 3 of a four-slot ring over the same 64 KB:
  // pair 0 lands behind the GEMM0 stages (it travels while GEMM0 runs), as k-steps 0 and 1 did
  constexpr int PP = 16384 / 16 / NT;                             // DMA instructions per thread and pair
  auto issue_b1_pair = [&](int p) {
    const char* src = a.w1 + (int64_t)p * 16384 + tid * 16;
    char* dst = sm + kRing + ((p + 3) & 3) * 16384 + tid * 16;
#pragma unroll
    for (int q = 0; q < PP; ++q)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(src + NT * 16 * q), (lds_void_t*)(dst + NT * 16 * q), 16, 0, 0);
  };
  if (!(SNAP_MLP_POOL_ABLATE & 4) && NST == 2) {
    if (SNAP_MLP_POOL_PAIRS) {
      issue_b1_pair(0);
    } else {
      issue_b1(0);
      issue_b1(1);                                                // (H >= 32: two k-steps exist)
    }
  }

  f32x16 acc0[T0];
#pragma unroll
  for (int t = 0; t < T0; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[t][r] = 0.f;

  // XSPLIT: the rows arrive pre-split ([slab][hi | lo][16] bf16, 64 B per row and slab: the lift's
  // out_split format) and go global -> LDS by LDS-DMA: no staging registers, no VALU, no LDS stores.
  // Stage layout [row][4 x 16 B] = hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15, the chunk index XOR-ed
  // with (row >> 1) & 3 (eight consecutive rows then hit eight distinct 16-byte bank groups); the
  // DMA writes LDS lane-contiguously, so each lane FETCHES the global chunk its slot holds.
  const char* xs_px[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 2) + RPP * i;
    const int c = (tid & 3) ^ ((row >> 1) & 3);
    xs_px[i] = r_ok[i] ? reinterpret_cast<const char*>(r_px[i]) + c * 16
                       : reinterpret_cast<const char*>(kZeroChunk);
  }
  auto issue_a = [&](int buf, int s_) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(xs_px[i] + (r_ok[i] ? s_ * 64 : 0)),
                                       (lds_void_t*)(sm + buf * A_ST + (tid + NT * i) * 16), 16, 0, 0);
  };
  const int s_first = a.skip_lo > 0 ? 0 : a.skip_n;
  const int nk0 = a.ctiles0 - a.skip_n;
  auto slab_of = [&](int kt) { return kt < a.skip_lo ? kt : kt + a.skip_n; };   // the kt-th visited slab
  if constexpr (NST == 3) {
    issue_a(0, s_first);
    issue_b0(0, s_first);
    if (nk0 > 1) {
      issue_a(1, slab_of(1));
      issue_b0(1, slab_of(1));
    }
  } else {
    if constexpr (XSPLIT) {
      issue_a(0, s_first);
      issue_b0(0, s_first);
    } else {
      load_a(s_first);
      issue_b0(0, s_first);
      store_a(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int R = 32 * wid + l31;                                   // this lane's row of the tile
  const int a_off = XSPLIT ? R * 64 + ((lhi ^ ((R >> 1) & 3)) * 16)
                           : R * 32 + ((lhi ^ ((R >> 3) & 1)) * 16);
  const int a_lo_off = XSPLIT ? R * 64 + (((2 + lhi) ^ ((R >> 1) & 3)) * 16) : a_off + A_PART;
  const int w_off = l31 * 32 + ((lhi ^ ((l31 >> 3) & 1)) * 16);   // column 32 t' + l31 of a 128-tile
  constexpr int kSlabOps = 2 + B0_PIECES;                         // DMA instructions per thread and slab (pre-split rows)
  for (int kt = 0; kt < nk0; ++kt) {
    const int cur = NST == 3 ? kt % 3 : (kt & 1);
    const bool more = kt + 1 < nk0;
    if constexpr (NST == 3) {
      // slabs <= kt + 1 are issued: slab kt must have landed, the next one may travel on
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSlabOps) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // every wave is past slab kt - 1: its stage takes slab kt + 2
      if (kt + 2 < nk0) {
        issue_a((kt + 2) % 3, slab_of(kt + 2));
        issue_b0((kt + 2) % 3, slab_of(kt + 2));
      }
    } else if (more) {
      const int sn = kt + 1 < a.skip_lo ? kt + 1 : kt + 1 + a.skip_n;    // the next visited slab
      if constexpr (XSPLIT) issue_a(cur ^ 1, sn); else load_a(sn);
      issue_b0(cur ^ 1, sn);
    }
    const char* bs = sm + kB0 + cur * B0_ST + w_off;
    const bf16x8 x_hi = *reinterpret_cast<const bf16x8*>(sm + cur * A_ST + a_off);
    const bf16x8 x_lo = *reinterpret_cast<const bf16x8*>(sm + cur * A_ST + a_lo_off);
#pragma unroll
    for (int g = 0; g < T0 / 4; ++g) {
      bf16x8 w_hi[4], w_lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * g + j;
        const char* p0 = bs + (t >> 2) * 8192 + (t & 3) * 1024;
        w_hi[j] = *reinterpret_cast<const bf16x8*>(p0);
        w_lo[j] = *reinterpret_cast<const bf16x8*>(p0 + 4096);
      }
      // per accumulator: x_lo w_hi, x_hi w_lo, x_hi w_hi (conv_split's order at NS = 2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc0[4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], x_lo, acc0[4 * g + j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc0[4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[j], x_hi, acc0[4 * g + j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc0[4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], x_hi, acc0[4 * g + j], 0, 0, 0);
    }
    if constexpr (!XSPLIT) {
      if (more) store_a(cur ^ 1);
    }
    if constexpr (NST == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if constexpr (NST == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring is free: W1 takes its place

  // ---- hidden = relu(acc0 + b0) -> the GEMM1 operand fragments, IN PLACE (16 accumulator
  // registers of a tile become 2 k-steps x (hi, lo) x 4 registers), while the first W1 stages
  // travel.  Keeping this VALU chain out of the GEMM1 loop leaves that loop LDS reads + MFMAs only.
  const int nks = a.H >> 4;                                       // 16-k steps of GEMM1
  const bool run1 = !(SNAP_MLP_POOL_ABLATE & 4);
  const int npairs = nks >> 1;                                    // (H % 32 == 0)
  if (run1) {
    if (SNAP_MLP_POOL_PAIRS) {
#pragma unroll
      for (int p = NST == 3 ? 0 : 1; p < 4; ++p)
        if (p < npairs) issue_b1_pair(p);                         // (NST = 2: pair 0 was issued at the start)
    } else {
#pragma unroll
      for (int k = 2; k < 8; ++k)
        if (k < nks) issue_b1(k);                                 // (k-steps 0, 1: issued at the start)
    }
  }
  u32x4 f_hi[T0][2], f_lo[T0][2];
#pragma unroll
  for (int t = 0; t < T0; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // this lane's row, hidden columns 32 t + 16 s + {4 lhi + 0..3, 8 + 4 lhi + 0..3}
      f32x4 v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias0 + 32 * t + 16 * s + 8 * q + 4 * lhi);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][e] = snap_relu(acc0[t][8 * s + 4 * q + e] + bb[e]);
      }
      // half 0: columns 0-3 | 8-11, half 1: 4-7 | 12-15  ->  half 0: 0-7, half 1: 8-15
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0][e]),
                                                         __float_as_uint(v[1][e]), false, false);
        v[0][e] = __uint_as_float(sw[0]);
        v[1][e] = __uint_as_float(sw[1]);
      }
      u32x2 h0, l0, h1, l1;
      split2(v[0], h0, l0);
      split2(v[1], h1, l1);
      f_hi[t][s] = u32x4{h0[0], h0[1], h1[0], h1[1]};
      f_lo[t][s] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }

  // ---- GEMM1: one k-step per ring slot -----------------------------------------------------------
  f32x16 acc1[T1];
#pragma unroll
  for (int t = 0; t < T1; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;

  // One barrier per k-step leaves 12 MFMAs (0.16 us) between two barriers, and the ablation puts
  // GEMM1 at 1.3 of the kernel's 3.3 ms for 192 of a single-observation tile's 408 MFMAs: the
  // loop pays a barrier + the LDS latency of the operand fetch per k-step.  Two k-steps per
  // barrier (the same 32 operand registers, reloaded between the two) halve that.
#pragma unroll
  for (int pr = 0; pr < T0; ++pr) {
    if (!SNAP_MLP_POOL_PAIRS) break;
    if (pr < npairs && run1) {
      // wait for pair pr; issued so far: 1 .. pr + 2, so pr + 1 .. min(pr + 2, npairs - 1) may stay
      // in flight, four DMA instructions each (pair 0 was drained by GEMM0's waits)
      if (pr >= 1 || NST == 3) {
        const int younger = min(pr == 0 ? 3 : pr + 2, npairs - 1) - pr;
        switch (younger) {
          case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // every wave is past pair pr - 1: its slot takes pair pr + 3
      if (pr >= 1 && pr + 3 < npairs) issue_b1_pair(pr + 3);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const char* ws = sm + kRing + ((pr + 3) & 3) * 16384 + sub * 8192 + w_off;
        bf16x8 h_hi, h_lo;
        __builtin_memcpy(&h_hi, &f_hi[pr][sub], 16);
        __builtin_memcpy(&h_lo, &f_lo[pr][sub], 16);
        bf16x8 w_hi[T1], w_lo[T1];
#pragma unroll
        for (int j = 0; j < T1; ++j) {
          const char* p0 = ws + j * 1024;
          w_hi[j] = *reinterpret_cast<const bf16x8*>(p0);
          w_lo[j] = *reinterpret_cast<const bf16x8*>(p0 + 4096);
        }
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_lo, acc1[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[j], h_hi, acc1[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_hi, acc1[j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int ks = 0; ks < 2 * T0; ++ks) {
    if (SNAP_MLP_POOL_PAIRS) break;
    if (ks < nks && run1) {
      // wait for k-step ks; issued so far: 2 .. ks + 6, so ks + 1 .. min(ks + 6, nks - 1) may
      // stay in flight, two DMA instructions each (k-steps 0, 1 were drained by GEMM0's waits)
      if (ks >= 2) {
        const int younger = min(ks + 6, nks - 1) - ks;
        switch (younger) {
          case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // every wave is past k-step ks - 1: its slot takes k-step ks + 7
      if (ks >= 1 && ks + 7 < nks) issue_b1(ks + 7);
      const char* ws = sm + ((ks + 6) & 7) * 8192 + w_off;
      bf16x8 h_hi, h_lo;
      __builtin_memcpy(&h_hi, &f_hi[ks >> 1][ks & 1], 16);
      __builtin_memcpy(&h_lo, &f_lo[ks >> 1][ks & 1], 16);
      bf16x8 w_hi[T1], w_lo[T1];
#pragma unroll
      for (int j = 0; j < T1; ++j) {
        const char* p0 = ws + j * 1024;
        w_hi[j] = *reinterpret_cast<const bf16x8*>(p0);
        w_lo[j] = *reinterpret_cast<const bf16x8*>(p0 + 4096);
      }
#pragma unroll
      for (int j = 0; j < T1; ++j)
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_lo, acc1[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < T1; ++j)
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[j], h_hi, acc1[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < T1; ++j)
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_hi, acc1[j], 0, 0, 0);
    }
  }

  // ---- + b1, stage [128 rows][128 channels] (float4 quads XOR-swizzled by the row) -------------
#pragma unroll
  for (int j = 0; j < T1; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias1 + 32 * j + 8 * q + 4 * lhi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc1[j][4 * q + e] += bb[e];
    }
  __syncthreads();                                                // ring and bias table drained
#pragma unroll
  for (int j = 0; j < T1; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int quad = (8 * j + 2 * q + lhi) ^ (R & 31);
      *reinterpret_cast<f32x4*>(smem + R * N1 + 4 * quad) =
          f32x4{acc1[j][4 * q], acc1[j][4 * q + 1], acc1[j][4 * q + 2], acc1[j][4 * q + 3]};
    }
  __syncthreads();

  // ---- segmented max down the rows: thread = (channel, half of the tile) ------------------------
  const int c = tid & 127;
  const int h = tid >> 7;                                         // wave-uniform (two waves per half)
  const int my = m0 + 64 * h + lane;
  const int cid = my < Meff ? a.rows[my] / a.Z : -1;
  int cur = -1;
  float run = -INFINITY;
  const bool live = c < a.D && !(SNAP_MLP_POOL_ABLATE & 8);
  // (snap_max_nan: a NaN of an observed voxel makes the column's maximum NaN, as jnp.max does
  //  (bev_mapper.py:63-78) -- the same in vertical_pool_kernel; the canonical positive NaN wins the
  //  integer atomic max below.  The MLP's ReLUs propagate it too (snap_relu).)
  // the thread's 64 values first (the accumulators are dead: 64 free registers): read inside the
  // loop, every ds_read sat behind the row's scalar branch and was waited for at once -- 64 dependent
  // LDS round trips (~3.4 us of a tile's ~18)
  float vals[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int row = 64 * h + r;
    vals[r] = smem[row * N1 + ((((c >> 2) ^ (row & 31)) << 2) | (c & 3))];
  }
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int cr = __builtin_amdgcn_readlane(cid, r);
    if (cr != cur) {
      if (cur >= 0 && live) atomic_max_f32(a.plane + (int64_t)cur * a.D + c, run);
      cur = cr;
      run = -INFINITY;
    }
    run = snap_max_nan(run, vals[r]);
  }
  if (cur >= 0 && live) atomic_max_f32(a.plane + (int64_t)cur * a.D + c, run);
}

// ------------------------------------------------------------------------------------------------
// The same computation on 256-ROW tiles, for pre-split rows and H = 256: one workgroup of four waves
// per CU (launch bound: one wave per SIMD, so a wave may hold 512 registers -- the accumulators live
// in the AccVGPR half), each wave owns 64 rows = TWO 32-row blocks.  Why: the 128-row kernel moves
// W0 + W1 (up to 400 KB) from L2 into LDS once per 128 rows -- 18 GB of its 24 GB per C2 step, at
// the ~7 TB/s the CUs take from L2 that IS its run time -- and every weight fragment a wave reads
// from LDS feeds one MFMA.  Here the weight stream is paid once per 256 rows and a fragment feeds
// two MFMAs; the eight-wave variant of the old kernel (SNAP_MLP_POOL_NT = 512) had the same stream
// but twice the waves behind every barrier and measured slower.  With one workgroup per CU nothing
// else hides a memory round trip, so the GEMM0 slabs travel THREE ahead through a four-stage ring
// (4 x 32 KB), all of W1 (128 KB) is resident before GEMM1 needs it (pair 0 arrives while GEMM0
// runs, pairs 1-7 while the hidden activations are converted), and the scan tile ([256][128] f32)
// takes the ring's place at the end.  Slab order and product order per accumulator are the
// 128-row kernel's: the plane is bit-identical (tests/test_gpu_kernels.py).
// MEASURED (tools/mlp_pool_bench.py, the C2 map: 6.8 M rows in two classes): 4.11-4.18 ms against
// 3.30-3.43 ms for the 128-row kernel.  One wave per SIMD cannot hide its own non-matrix phases --
// row-list fetch and first slab (~4 us), the ReLU / split conversion of 256 accumulator registers
// (~3 us), the scan (~4 us), the LDS latency after every barrier -- which two co-resident 128-row
// workgroups hide for each other: 39 us per 256-row tile for 11-16 us of MFMA time.  Halving the
// weight stream is worth less than that overlap.  Kept as an opt-in (x_split = 3) with its test.
template <int N0>
__global__ __launch_bounds__(256, 1) void mlp2_pool_wide_kernel(const MlpPoolArgs a) {
  constexpr int NT = 256, BM = 256, N1 = 128, RB = 2;
  constexpr int T0 = N0 / 32, T1 = N1 / 32;
  constexpr int A_ST = BM * 64;                               // 16 KB: [row][4 x 16 B] (hi k0-7 | hi k8-15 | lo | lo)
  constexpr int B0_ST = (N0 / 128) * 8192;                    // 16 KB
  constexpr int kStage = A_ST + B0_ST;
  constexpr int NSTG = 4;
  constexpr int kPair0 = NSTG * kStage;                       // W1 pair 0 behind the ring
  constexpr int kBias = kPair0 + 16384;
  static_assert(BM * N1 * 4 <= kPair0, "scan tile overlaps pair 0 / the bias table");
  static_assert(7 * 16384 <= kPair0, "W1 pairs 1..7 take the ring's place");
  __shared__ __attribute__((aligned(16))) float smem[kBias / 4 + N0 + N1];   // 145.5 KB
  char* const sm = reinterpret_cast<char*>(smem);
  float* const bias0 = reinterpret_cast<float*>(sm + kBias);
  float* const bias1 = bias0 + N0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Meff = min(*a.row_count, a.M);
  const int m0 = blockIdx.x * BM;
  if (m0 >= Meff) return;

  for (int i = tid; i < N0 + N1; i += NT)
    bias0[i] = i < N0 ? (i < a.H ? a.b0[i] : 0.f) : (i - N0 < a.D ? a.b1[i - N0] : 0.f);

  // W1 pair p = k-steps 2p, 2p + 1 (16 KB)
  auto issue_b1_pair = [&](int p, int dst_off) {
    const char* src = a.w1 + (int64_t)p * 16384 + tid * 16;
    char* dst = sm + dst_off + tid * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(src + NT * 16 * q), (lds_void_t*)(dst + NT * 16 * q), 16, 0, 0);
  };
  issue_b1_pair(0, kPair0);

  // rows: thread = (row tid >> 2 (+ 64 i), 16-byte chunk tid & 3); the DMA writes LDS lane-contiguously,
  // so each lane FETCHES the global chunk its (XOR-swizzled) slot holds
  const char* xs_px[4];
  bool r_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 2) + 64 * i;
    const int m = m0 + row;
    r_ok[i] = m < Meff;
    const int c = (tid & 3) ^ ((row >> 1) & 3);
    xs_px[i] = r_ok[i] ? reinterpret_cast<const char*>(a.x + (int64_t)a.rows[m] * a.x_stride) + c * 16
                       : reinterpret_cast<const char*>(kZeroChunk);
  }
  auto issue_slab = [&](int stg, int s_) {
    char* const base = sm + stg * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(xs_px[i] + (r_ok[i] ? s_ * 64 : 0)),
                                       (lds_void_t*)(base + (tid + NT * i) * 16), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < B0_ST / 16 / NT; ++p) {
      const int slot = tid + NT * p;
      const int j = slot >> 9;
      const char* src = a.w0 + ((int64_t)j * a.ctiles0 + s_) * 8192 + (slot & 511) * 16;
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)src, (lds_void_t*)(base + A_ST + 16 * slot), 16, 0, 0);
    }
  };
  constexpr int kSlabOps = 4 + B0_ST / 16 / NT;                 // DMA instructions per thread and slab
  const int nk0 = a.ctiles0 - a.skip_n;
  auto slab_of = [&](int kt) { return kt < a.skip_lo ? kt : kt + a.skip_n; };   // the kt-th visited slab
#pragma unroll
  for (int kt = 0; kt < NSTG - 1; ++kt)
    if (kt < nk0) issue_slab(kt, slab_of(kt));

  f32x16 acc0[RB][T0];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[b][t][r] = 0.f;

  int a_off[RB], a_lo_off[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int R = 64 * wid + 32 * b + l31;                      // this lane's row of the tile
    a_off[b] = R * 64 + ((lhi ^ ((R >> 1) & 3)) * 16);
    a_lo_off[b] = R * 64 + (((2 + lhi) ^ ((R >> 1) & 3)) * 16);
  }
  const int w_off = l31 * 32 + ((lhi ^ ((l31 >> 3) & 1)) * 16);   // column 32 t' + l31 of a 128-tile

  for (int kt = 0; kt < nk0; ++kt) {
    // slabs <= kt + 2 are issued; slab kt must have landed, the younger ones may travel on
    const int younger = min(kt + 2, nk0 - 1) - kt;
    if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kSlabOps) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSlabOps) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // every wave is past slab kt - 1: its stage takes slab kt + 3
    if (kt + NSTG - 1 < nk0) issue_slab((kt + NSTG - 1) & (NSTG - 1), slab_of(kt + NSTG - 1));
    const char* const st = sm + (kt & (NSTG - 1)) * kStage;
    bf16x8 x_hi[RB], x_lo[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      x_hi[b] = *reinterpret_cast<const bf16x8*>(st + a_off[b]);
      x_lo[b] = *reinterpret_cast<const bf16x8*>(st + a_lo_off[b]);
    }
    const char* const bs = st + A_ST + w_off;
#pragma unroll
    for (int g = 0; g < T0 / 4; ++g) {
      bf16x8 w_hi[4], w_lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * g + j;
        const char* p0 = bs + (t >> 2) * 8192 + (t & 3) * 1024;
        w_hi[j] = *reinterpret_cast<const bf16x8*>(p0);
        w_lo[j] = *reinterpret_cast<const bf16x8*>(p0 + 4096);
      }
      // per accumulator: x_lo w_hi, x_hi w_lo, x_hi w_hi (conv_split's order at NS = 2)
#pragma unroll
      for (int b = 0; b < RB; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc0[b][4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], x_lo[b], acc0[b][4 * g + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc0[b][4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[j], x_hi[b], acc0[b][4 * g + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc0[b][4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], x_hi[b], acc0[b][4 * g + j], 0, 0, 0);
      }
    }
  }
  // every wave is done with the ring: W1 pairs 1..7 take its place while the hidden activations
  // are converted (pair 0 has landed: the last slab's wait drained the queue)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int p = 1; p < 8; ++p) issue_b1_pair(p, (p - 1) * 16384);

  // ---- hidden = relu(acc0 + b0) -> the GEMM1 operand fragments, in place of the accumulators ----
  u32x4 f_hi[RB][T0][2], f_lo[RB][T0][2];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bias0 + 32 * t + 16 * s + 8 * q + 4 * lhi);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] = snap_relu(acc0[b][t][8 * s + 4 * q + e] + bb[e]);
        }
        // half 0: columns 0-3 | 8-11, half 1: 4-7 | 12-15  ->  half 0: 0-7, half 1: 8-15
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0][e]),
                                                           __float_as_uint(v[1][e]), false, false);
          v[0][e] = __uint_as_float(sw[0]);
          v[1][e] = __uint_as_float(sw[1]);
        }
        u32x2 h0, l0, h1, l1;
        split2(v[0], h0, l0);
        split2(v[1], h1, l1);
        f_hi[b][t][s] = u32x4{h0[0], h0[1], h1[0], h1[1]};
        f_lo[b][t][s] = u32x4{l0[0], l0[1], l1[0], l1[1]};
      }

  // ---- GEMM1: W1 resident, one barrier per pair of k-steps (the pair's arrival) ------------------
  f32x16 acc1[RB][T1];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[b][t][r] = 0.f;
#pragma unroll
  for (int pr = 0; pr < 8; ++pr) {
    if (pr >= 1) {                                              // pairs pr + 1 .. 7 may travel on (4 DMAs each)
      switch (7 - pr) {
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const char* ws = sm + (pr == 0 ? kPair0 : (pr - 1) * 16384) + sub * 8192 + w_off;
      bf16x8 w_hi[T1], w_lo[T1];
#pragma unroll
      for (int j = 0; j < T1; ++j) {
        const char* p0 = ws + j * 1024;
        w_hi[j] = *reinterpret_cast<const bf16x8*>(p0);
        w_lo[j] = *reinterpret_cast<const bf16x8*>(p0 + 4096);
      }
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        bf16x8 h_hi, h_lo;
        __builtin_memcpy(&h_hi, &f_hi[b][pr][sub], 16);
        __builtin_memcpy(&h_lo, &f_lo[b][pr][sub], 16);
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_lo, acc1[b][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[j], h_hi, acc1[b][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T1; ++j)
          acc1[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[j], h_hi, acc1[b][j], 0, 0, 0);
      }
    }
  }

  // ---- + b1, stage [256 rows][128 channels] (float4 quads XOR-swizzled by the row) -------------
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int j = 0; j < T1; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias1 + 32 * j + 8 * q + 4 * lhi);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[b][j][4 * q + e] += bb[e];
      }
  __syncthreads();                                                // W1 drained
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int R = 64 * wid + 32 * b + l31;
#pragma unroll
    for (int j = 0; j < T1; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int quad = (8 * j + 2 * q + lhi) ^ (R & 31);
        *reinterpret_cast<f32x4*>(smem + R * N1 + 4 * quad) =
            f32x4{acc1[b][j][4 * q], acc1[b][j][4 * q + 1], acc1[b][j][4 * q + 2], acc1[b][j][4 * q + 3]};
      }
  }
  __syncthreads();

  // ---- segmented max down the rows: thread = (channel, half of the tile), 2 x 64 rows ------------
  const int c = tid & 127;
  const int h = tid >> 7;                                         // wave-uniform (two waves per half)
  int cur = -1;
  float run = -INFINITY;
  const bool live = c < a.D;
  for (int q = 0; q < 2; ++q) {
    const int my = m0 + 128 * h + 64 * q + lane;
    const int cid = my < Meff ? a.rows[my] / a.Z : -1;
    float vals[64];                          // (read ahead of the branches, as in the 128-row kernel)
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int row = 128 * h + 64 * q + r;
      vals[r] = smem[row * N1 + ((((c >> 2) ^ (row & 31)) << 2) | (c & 3))];
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int cr = __builtin_amdgcn_readlane(cid, r);
      if (cr != cur) {
        if (cur >= 0 && live) atomic_max_f32(a.plane + (int64_t)cur * a.D + c, run);
        cur = cr;
        run = -INFINITY;
      }
      run = snap_max_nan(run, vals[r]);
    }
  }
  if (cur >= 0 && live) atomic_max_f32(a.plane + (int64_t)cur * a.D + c, run);
}

// plane prefilled with -inf -> where(any level valid, max, 0) + the validity byte
__global__ __launch_bounds__(256) void mlp2_pool_finalize_kernel(float* __restrict__ plane,
                                                                 uint8_t* __restrict__ pvalid,
                                                                 int64_t ncols, int D) {
  const int Q = D >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncols * Q) return;
  const int64_t col = i / Q;
  f32x4* p = reinterpret_cast<f32x4*>(plane) + i;
  f32x4 v = *p;
  // (the quad is observed if ANY of its channels was written; a NaN maximum counts as written)
  const bool any = v[0] != -INFINITY || v[1] != -INFINITY || v[2] != -INFINITY || v[3] != -INFINITY;
  if (!any) *p = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i - col * Q == 0) pvalid[col] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, int64_t n4, float v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{v, v, v, v};
}

}  // namespace

static int mlp2_pool_check(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                           const void* w0_split, size_t w0_bytes, int32_t H, const void* w1_split,
                           size_t w1_bytes, int32_t D, int32_t relu_in, int32_t x_split, int32_t Z,
                           int64_t ncols, const float* plane) {
  if (M <= 0 || M > 0x7fffffffLL || Cin <= 0 || x_stride < Cin || x_stride % 4 != 0 || Z <= 0 ||
      ncols <= 0 || ncols * Z > 0x7fffffffLL)
    return SNAP_ERR_BAD_SHAPE;
  if (H <= 0 || H % 32 != 0 || H > 256 || D <= 0 || D % 4 != 0 || D > 128) return SNAP_ERR_UNSUPPORTED;
  // x_split: 0 f32 rows | 1 pre-split rows (three-stage ring) | 3 / 5 its measured alternatives; 7 (tap records) is
  // internal to snap_mlp2_pool_max_gather_f32, which brings the records and the image with it
  if (x_split != 0 && x_split != 1 && x_split != 3 && x_split != 5) return SNAP_ERR_UNSUPPORTED;
  if (x_split && (relu_in || x_stride < ((Cin + 15) / 16) * 16)) return SNAP_ERR_UNSUPPORTED;
  if (w0_bytes < snap_conv2d_packed_weights_split_bytes(1, Cin, H, 2) ||
      w1_bytes < snap_conv2d_packed_weights_split_bytes(1, H, D, 2))
    return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w0_split) |
       reinterpret_cast<uintptr_t>(w1_split) | reinterpret_cast<uintptr_t>(plane)) & 15)
    return SNAP_ERR_BAD_SHAPE;
  return SNAP_OK;
}

static void mlp2_pool_launch(const MlpPoolArgs& a, int relu_in, int x_split, hipStream_t s) {
  constexpr int NT = SNAP_MLP_POOL_NT;
  if (x_split == 7) {                                // tap records: the gather inside the kernel
    int64_t nb = snap_cdiv(a.M, 128);
    if (a.xcd_group > 0) nb = snap_cdiv(nb, 8LL * a.xcd_group) * 8LL * a.xcd_group;
    if (a.H <= 128) hipLaunchKernelGGL((mlp2_pool_kernel<128, false, false, 256, 2, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mlp2_pool_kernel<256, false, false, 256, 2, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
    return;
  }
  if (x_split == 3 && !relu_in && a.H == 256) {      // 256-row tiles: measured slower (see the kernel), opt-in
    hipLaunchKernelGGL((mlp2_pool_wide_kernel<256>), dim3((unsigned)snap_cdiv(a.M, 256)), dim3(256), 0, s, a);
    return;
  }
  const dim3 grid((unsigned)snap_cdiv(a.M, NT / 2));
  // pre-split rows: the three-stage ring (x_split = 5: the two-stage loop, tuning / tests)
  constexpr bool kRingOk = NT == 256;
  if (a.H <= 128) {
    if (x_split && x_split != 5 && kRingOk) hipLaunchKernelGGL((mlp2_pool_kernel<128, false, true, 256, 3>), grid, dim3(NT), 0, s, a);
    else if (x_split) hipLaunchKernelGGL((mlp2_pool_kernel<128, false, true, NT>), grid, dim3(NT), 0, s, a);
    else if (relu_in) hipLaunchKernelGGL((mlp2_pool_kernel<128, true, false, NT>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((mlp2_pool_kernel<128, false, false, NT>), grid, dim3(NT), 0, s, a);
  } else {
    if (x_split && x_split != 5 && kRingOk) hipLaunchKernelGGL((mlp2_pool_kernel<256, false, true, 256, 3>), grid, dim3(NT), 0, s, a);
    else if (x_split) hipLaunchKernelGGL((mlp2_pool_kernel<256, false, true, NT>), grid, dim3(NT), 0, s, a);
    else if (relu_in) hipLaunchKernelGGL((mlp2_pool_kernel<256, true, false, NT>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((mlp2_pool_kernel<256, false, false, NT>), grid, dim3(NT), 0, s, a);
  }
}

extern "C" int snap_mlp2_pool_max_classes_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                                              const int32_t* rows, const int32_t* row_count,
                                              const int32_t* rows_z, const int32_t* row_count_z,
                                              int32_t zero_slab_lo, int32_t zero_slabs,
                                              const void* w0_split, size_t w0_bytes, const float* b0,
                                              int32_t H, const void* w1_split, size_t w1_bytes,
                                              const float* b1, int32_t D, int32_t relu_in,
                                              int32_t x_split, int32_t Z, int64_t ncols, float* plane,
                                              uint8_t* pvalid, void* stream) {
  if (!x || !rows || !row_count || !w0_split || !b0 || !w1_split || !b1 || !plane || !pvalid)
    return SNAP_ERR_NULL;
  if ((rows_z == nullptr) != (row_count_z == nullptr)) return SNAP_ERR_NULL;
  const int st = mlp2_pool_check(x, M, Cin, x_stride, w0_split, w0_bytes, H, w1_split, w1_bytes, D,
                                 relu_in, x_split, Z, ncols, plane);
  if (st != SNAP_OK) return st;
  const int ctiles0 = (Cin + 15) / 16;
  if (rows_z && (zero_slab_lo < 0 || zero_slabs <= 0 || zero_slab_lo + zero_slabs > ctiles0 ||
                 zero_slabs >= ctiles0))
    return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n4 = ncols * (D / 4);
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)snap_cdiv(n4, 256)), dim3(256), 0, s, plane, n4,
                     -INFINITY);
  SNAP_CHECK_LAUNCH();
  MlpPoolArgs a;
  a.x = x; a.x_stride = x_stride; a.Cin = Cin;
  a.rows = rows; a.row_count = row_count; a.M = (int)M;
  a.w0 = static_cast<const char*>(w0_split); a.ctiles0 = ctiles0; a.b0 = b0; a.H = H;
  a.w1 = static_cast<const char*>(w1_split); a.b1 = b1; a.D = D;
  a.Z = Z; a.plane = plane;
  a.skip_lo = ctiles0; a.skip_n = 0;
  a.fimg = nullptr; a.recs = nullptr; a.Cb = a.Wb = 0; a.nmean = 0; a.xcd_group = 0;
  mlp2_pool_launch(a, relu_in, x_split, s);
  SNAP_CHECK_LAUNCH();
  if (rows_z) {           // the rows that are zero over [zero_slab_lo, +zero_slabs): same plane (max)
    a.rows = rows_z; a.row_count = row_count_z;
    a.skip_lo = zero_slab_lo; a.skip_n = zero_slabs;
    mlp2_pool_launch(a, relu_in, x_split, s);
    SNAP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(mlp2_pool_finalize_kernel, dim3((unsigned)snap_cdiv(n4, 256)), dim3(256), 0, s,
                     plane, pvalid, ncols, D);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// The lift inside the consumer: rows_g lists voxels with ONE visible observation whose `pooled` row
// does not exist -- the kernel blends their four image taps from the tap records
// (snap_lift_pool_records_f32) while it stages GEMM0's operand; `rows` (several observations) are
// read pre-split from x as snap_mlp2_pool_max_classes_f32 does.  One plane, the same bits.
extern "C" int snap_mlp2_pool_max_gather_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                                             const int32_t* rows, const int32_t* row_count,
                                             const int32_t* rows_g, const int32_t* row_count_g,
                                             const float* f_images, int64_t f_bytes, int32_t img_w,
                                             int32_t img_C, int32_t feature_dim,
                                             const uint32_t* tap_records, int32_t xcd_group,
                                             const void* w0_split, size_t w0_bytes, const float* b0,
                                             int32_t H, const void* w1_split, size_t w1_bytes,
                                             const float* b1, int32_t D, int32_t Z, int64_t ncols,
                                             float* plane, uint8_t* pvalid, void* stream) {
  if (!x || !rows || !row_count || !rows_g || !row_count_g || !f_images || !tap_records || !w0_split ||
      !b0 || !w1_split || !b1 || !plane || !pvalid)
    return SNAP_ERR_NULL;
  const int st = mlp2_pool_check(x, M, Cin, x_stride, w0_split, w0_bytes, H, w1_split, w1_bytes, D,
                                 0, 1, Z, ncols, plane);
  if (st != SNAP_OK) return st;
  // rows = mean(fd) | var(fd) | score: whole 16-channel slabs, the score opens a slab of its own
  if (feature_dim <= 0 || feature_dim % 16 != 0 || Cin != 2 * feature_dim + 1 || img_C < feature_dim ||
      img_C % 4 != 0 || img_w <= 0 || xcd_group < 0)
    return SNAP_ERR_BAD_SHAPE;
  if (f_bytes <= 0 || f_bytes >= (1LL << 32)) return SNAP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(f_images) | reinterpret_cast<uintptr_t>(tap_records)) & 15)
    return SNAP_ERR_BAD_SHAPE;
  const int ctiles0 = (Cin + 15) / 16;
  const int nmean = feature_dim / 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n4 = ncols * (D / 4);
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)snap_cdiv(n4, 256)), dim3(256), 0, s, plane, n4,
                     -INFINITY);
  SNAP_CHECK_LAUNCH();
  MlpPoolArgs a;
  a.x = x; a.x_stride = x_stride; a.Cin = Cin;
  a.rows = rows; a.row_count = row_count; a.M = (int)M;
  a.w0 = static_cast<const char*>(w0_split); a.ctiles0 = ctiles0; a.b0 = b0; a.H = H;
  a.w1 = static_cast<const char*>(w1_split); a.b1 = b1; a.D = D;
  a.Z = Z; a.plane = plane;
  a.skip_lo = ctiles0; a.skip_n = 0;
  a.fimg = reinterpret_cast<const char*>(f_images); a.recs = tap_records;
  a.Cb = (uint32_t)img_C * 4u; a.Wb = (uint32_t)img_w * a.Cb; a.nmean = nmean; a.xcd_group = 0;
  mlp2_pool_launch(a, 0, 1, s);
  SNAP_CHECK_LAUNCH();
  a.rows = rows_g; a.row_count = row_count_g;
  a.skip_lo = nmean; a.skip_n = nmean;
  a.xcd_group = xcd_group;
  mlp2_pool_launch(a, 0, 7, s);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(mlp2_pool_finalize_kernel, dim3((unsigned)snap_cdiv(n4, 256)), dim3(256), 0, s,
                     plane, pvalid, ncols, D);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_mlp2_pool_max_f32(const float* x, int64_t M, int32_t Cin, int32_t x_stride,
                                      const int32_t* rows, const int32_t* row_count,
                                      const void* w0_split, size_t w0_bytes, const float* b0,
                                      int32_t H, const void* w1_split, size_t w1_bytes,
                                      const float* b1, int32_t D, int32_t relu_in, int32_t x_split,
                                      int32_t Z, int64_t ncols, float* plane, uint8_t* pvalid,
                                      void* stream) {
  return snap_mlp2_pool_max_classes_f32(x, M, Cin, x_stride, rows, row_count, nullptr, nullptr, 0, 0,
                                        w0_split, w0_bytes, b0, H, w1_split, w1_bytes, b1, D, relu_in,
                                        x_split, Z, ncols, plane, pvalid, stream);
}
