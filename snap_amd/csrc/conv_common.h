// Shared pieces of the implicit-GEMM conv engines (conv_igemm.hip: f32 matrix cores, exact;
// conv_bf16.hip: bf16 operands / f32 accumulate): launch arguments, the fused prologues, the
// output epilogue, the split-K reduce and the tile / split-K heuristics.
#ifndef SNAP_CSRC_CONV_COMMON_H_
#define SNAP_CSRC_CONV_COMMON_H_

#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace snapconv {
struct ConvArgs {
  SnapConvDesc d;
  const float* x;
  const float* w;
  float* y;
  const float* gn_mu;
  const float* gn_sc;
  const float* gn_beta;
  const float* bias;
  const float* residual;
  const float* up_prev;
  const uint8_t* row_mask;
  const int32_t* rows_in;    // optional: GEMM row m reads output pixel rows_in[m]
  const int32_t* rows_out;   // optional: GEMM row m is written to y row rows_out[m]
  const int32_t* row_count;  // optional device scalar: only the first *row_count rows exist
  float* kpartial;           // split-K: [ksplit][M][Cout] raw partial tiles (workspace)
  size_t kpartial_bytes;
  int ksplit;                // number of K splits (1 = none)
  int slabs_per_split;
  int tiles_per_split;       // workgroups of one split (multiple of 8)
  float* gn_partial;         // optional: per-(image, row tile, channel) sums of y and y^2
  float* gn_partial2;        // optional, with gn_partial (gn_relu = 0): the same sums of relu(y)
  int32_t* gn_partial2_done; // HOST pointer (launcher only): 1 when the launch emits gn_partial2
  int gn_relu;               // ... of relu(y) (FPN order)
  int gn_rows32;             // gn_partial is laid out per 32-row slab (split-K launches: the reduce pass emits it)
  int gn_slabs;              // row tiles per image in gn_partial (= HoWo / BM + 2)
  int M;       // N*Ho*Wo  (upper bound of the row count when row_count is set)
  int K;       // KH*KW*Cin
  int ctiles;  // ceil(Cin/16)   (VEC path)
  int nk;      // number of K slabs
  int prio;    // experiment knob: s_setprio(1) around the MFMA block
  int ncol;    // number of column tiles (set in launch<>)
  int ablate;  // alt builds only (-DSNAP_CONV_SPLIT_ABLATE=1, env SNAP_ALT_ABLATE): bit0 skip global loads, bit1 skip LDS stores+barrier, bit2 skip LDS reads, bit3 skip the barrier; 0 in the product build
  int bk;      // f32 engine: K-slab depth of the large tiles (16 | 32)
  int no_halo; // split engine: 1 = im2col body for every 3x3
  int no_plain;  // split engine: 1 = the general loader also for 1 x 1 / stride 1 / unpadded layers
  int use_raw;   // split engine: 1 = the raw-row LDS-DMA ring body (conv_raw.hip) for the K >= 256 1 x 1 layers (opt-in: measured level / slower)
  int rs_nsplit;  // split engine: forced column split of conv_rs.hip (0 = automatic)
  const void* w_bf16;  // bf16 engine: weights packed by snap_conv2d_pack_weights_bf16 ([Cout][taps][cin8])
  int cin8;            // ... channel count rounded up to 8
  int half;            // ... 1: the image and the A operand are IEEE half (SnapConvExtras.w_half)
  const void* x_half;  // ... non-NULL: the input ALREADY in the engine's element type, [N,H,W,Cin_stride]
                       //     (SnapConvExtras.x_half; prologue NONE): both operands by LDS-DMA
  void* y_half;        // ... non-NULL: the output (also / only, when y is NULL) rounded to the engine's
                       //     element type, same row / column indexing as y (SnapConvExtras.y_half)
  const void* x_ps;    // pre-split engine (conv_ps.hip): the input as [pixel][Cin/16][hi 16 | lo 16] bf16
  int ps_tile;         // ... 0 = automatic tile, 1 = 128 rows, 2 = 256 rows
  int ps_res_init;     // ... 1 = the residual is loaded into the accumulators before the K loop
  int ps_parts;        // ... 2 = the two-part image (bf16x3); 1 = x_ps is a plain bf16 matrix, w_bf16 the one-part image (bf16)
  // GroupNorm-VJP statistics in the epilogue (SnapConvExtras.gnb_*; conv_epilogue<..., GNB = true> only)
  const float* gnb_x;
  const float* gnb_mu;
  const float* gnb_rstd;
  const float* gnb_gamma;
  const float* gnb_beta;
  int gnb_mode;
};

// bf16-operand engine (conv_bf16.hip); `a` validated by snap_conv2d_nhwc_ex_f32
int launch_bf16(ConvArgs a, hipStream_t s);
// split-bf16 engine (conv_split.hip): `parts` = 2 (three products) or 3 (six products);
// a.w_bf16 then holds [parts][Cout][taps][cin8]
int launch_split(ConvArgs a, int parts, hipStream_t s);
// ... the 7 x 7 / stride 2 root convolution of a 4-floats-per-pixel RGB image (root weight image)
int launch_split_root(ConvArgs a, int parts, hipStream_t s);
// pre-split engine (conv_ps.hip): a.x_ps = the input already normalised and split in two bf16
// parts (snap_gn_norm_split_f32 / snap_presplit_f32), a.w_bf16 = the split weight image (parts = 2)
int launch_ps(ConvArgs a, hipStream_t s);
// stationary-operand 1 x 1 kernels (conv_rs.hip) for Cin = 64 / 128 / 256 -> Cout >= 256, GroupNorm
// + ReLU prologue, optional residual; bit-identical to launch_split(a, 2, s) where they apply.
// stationary_kind: 0 = tiled body, 1 = row-stationary (activation tile in registers, weights
// streamed per 128 rows), 2 = weights-stationary (panel resident in LDS, Cin <= 128; GroupNorm
// partials per 32-row slab).  A pure function of the descriptor (+ the launch's row lists): the
// workspace / statistics queries must agree with the launch.
int stationary_kind(const SnapConvDesc& d, int parts, bool row_lists);
int launch_rs(ConvArgs a, hipStream_t s);
int launch_bs(ConvArgs a, hipStream_t s);
int launch_root_ws(const ConvArgs& a, hipStream_t s);   // the RGB root convolution, 64 output channels
// raw-row LDS-DMA body for 1 x 1 / stride 1 layers with a GroupNorm prologue and K >= 256 (conv_raw.hip):
// bit-identical to the tiled body; `a` as launch<128, bn, pro, 2> of conv_split.hip has set it up
bool raw_ok(const ConvArgs& a, int bm, int bn, int pro);
int launch_raw(const ConvArgs& a, int bn, int pro, dim3 grid, hipStream_t s);
struct PsTile { int bm, bn, nt; };
PsTile ps_choose_tile(int64_t M, int64_t N, int force);
int ps_ksplit(int64_t M, int Cout, int64_t nk, int bm, int bn, size_t kpartial_bytes);

}  // namespace snapconv

namespace {
using snapconv::ConvArgs;

// Timing ablations of the K loops (WRONG results; tools/conv_ablate*.py) exist only in an alt build
// (-DSNAP_CONV_SPLIT_ABLATE=1; the bits come from the environment variable SNAP_ALT_ABLATE): the
// product build has neither a switch in the ABI nor a run-time branch.
#if defined(SNAP_CONV_SPLIT_ABLATE) && SNAP_CONV_SPLIT_ABLATE
inline int snap_alt_ablate_bits() {
  const char* e = getenv("SNAP_ALT_ABLATE");
  return e ? atoi(e) : 0;
}
#define SNAP_IGEMM_ABL(bit) (a.ablate & (bit))
#else
inline int snap_alt_ablate_bits() { return 0; }
#define SNAP_IGEMM_ABL(bit) false
#endif


// The prologue is a COMPILE-TIME parameter: a run-time switch here is lowered to a
// branch tree per staged element and wrecks the schedule of the whole main loop.
template <int PRO>
__device__ __forceinline__ float apply_pro(float v, float mu, float sc, float beta, float s,
                                           float t) {
  if constexpr (PRO == SNAP_PRO_AFFINE) return v * s + t;
  if constexpr (PRO == SNAP_PRO_GN_RELU) return snap_relu((v - mu) * sc + beta);
  if constexpr (PRO == SNAP_PRO_RELU_GN) return (snap_relu(v) - mu) * sc + beta;
  if constexpr (PRO == SNAP_PRO_RELU) return snap_relu(v);
  return v;
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void cglobal_void_t;

// zero source for LDS-DMA lanes whose tap / channel / row is out of range
__device__ __attribute__((aligned(64))) const float kZeroChunk[16] = {0.f};

// tanh-approximated GELU (flax.linen.gelu default): 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),
// u = sqrt(2/pi) (x + 0.044715 x^3)
__device__ __forceinline__ float snap_gelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x / (1.0f + __expf(-2.0f * u));
}

// ---- epilogue ------------------------------------------------------------
// NT threads = NT / 64 waves laid out (NT / 128) x 2; SKIP_RES: the residual already sits in the
// accumulators (conv_ps.hip loads it there before the K loop)
// OUTH (training-precision engine only): 1 / 2 = the stored values also go, rounded (RNE) to bf16 /
// IEEE half, to a.y_half; a.y may then be NULL (half-only output: the hidden activations and
// inter-layer gradients of the masked MLP, which every consumer rounds to that type anyway)
// GNB (the half-input data-gradient kernels only): with a.gnb_mode the statistics are those of the GroupNorm VJP
// this gradient feeds (sums of dyp and dyp * xh, see SnapConvExtras.gnb_*) instead of y / y^2.
template <int BM, int BN, bool DUAL = false, int NT = 256, bool SKIP_RES = false, int OUTH = 0, bool GNB = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a,
                                              f32x16 (&acc)[BM / (NT / 4)][BN / 64],
                                              float* smem, int m0, int n0, int Meff, int row_t,
                                              int split) {
  constexpr int WR = NT / 128;                // wave rows
  constexpr int TM = BM / (32 * WR), TN = BN / 64;
  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HoWo = d.Ho * d.Wo;
  // The MFMA C layout gives every lane one column of 16 rows: direct stores would be
  // 4-byte scattered.  Instead each pass stages 64 rows x BN columns through LDS and
  // writes them back as float4 rows (512 B contiguous per row for BN = 128); bias,
  // residual, FPN up-sample-add, ReLU and the row mask are applied on the way out
  // with float4 loads.  The final __syncthreads of the main loop already fenced the
  // slab ring, so the buffer can be reused at once.
  const int epi = d.epilogue;
  const int Hp = d.Ho >> 1, Wp = d.Wo >> 1;
  constexpr int Q = BN / 4;               // float4 per staged row
  constexpr int PER_THREAD = (32 * WR * Q) / NT;
  // GroupNorm statistics of the OUTPUT (consumed by the next layer's fused GN prologue):
  // every thread owns 4 fixed columns (NT % Q == 0), accumulates sum / sum of squares of
  // what it stores, split by image (a tile of BM <= HoWo rows touches at most two).
  const bool want_stats = a.gn_partial != nullptr && a.ksplit == 1;   // (split-K: the reduce pass emits them)
  const int n_first = m0 / HoWo;
  const int m_split = (n_first + 1) * HoWo;   // first row of the second image
  float gs1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float gs2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // a tensor read by a GroupNorm -> ReLU layer AND a ReLU -> GroupNorm layer (the last unit of a
  // ResNet stage: next stage / FPN level) gets both statistics out of the one epilogue
  // (DUAL is a template flag of the one kernel variant that is launched for such layers: as a
  //  run-time option it cost every conv kernel 20 registers)
  const bool want_relu_too = DUAL && want_stats;
  float hs1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float hs2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // GNB: the thread's four columns are fixed (NT % Q == 0): gamma / beta once, mean / rstd of the two images
  f32x4 vb_mu[2], vb_rs[2], vb_g = {0.f, 0.f, 0.f, 0.f}, vb_b = {0.f, 0.f, 0.f, 0.f};
  const bool gnb = GNB && want_stats && a.gnb_mode != 0;
  if constexpr (GNB) {
    vb_mu[0] = vb_mu[1] = vb_rs[0] = vb_rs[1] = vb_g;
    const int colq = n0 + 4 * (tid % Q);
    if (gnb && colq < d.Cout) {
      vb_g = *reinterpret_cast<const f32x4*>(a.gnb_gamma + colq);
      vb_b = *reinterpret_cast<const f32x4*>(a.gnb_beta + colq);
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int64_t so = (int64_t)min(n_first + sl, d.N - 1) * d.Cout + colq;
        vb_mu[sl] = *reinterpret_cast<const f32x4*>(a.gnb_mu + so);
        vb_rs[sl] = *reinterpret_cast<const f32x4*>(a.gnb_rstd + so);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < TM; ++h) {
    if (h > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        smem[(wr * 32 + ri) * BN + wc * (BN / 2) + j * 32 + l31] = acc[h][j][r];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PER_THREAD; ++it) {
      const int idx = tid + NT * it;
      const int row = idx / Q, q = idx - row * Q;
      const int m = m0 + (row >> 5) * (BM / WR) + h * 32 + (row & 31);
      const int col = n0 + 4 * q;
      if (m >= Meff || col >= d.Cout) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * BN + 4 * q);
      if (a.ksplit > 1) {  // raw partial tile; the epilogue runs in splitk_reduce_kernel
        *reinterpret_cast<f32x4*>(a.kpartial + ((int64_t)split * a.M + m) * d.Cout + col) = v;
        continue;
      }
      const int64_t o = (int64_t)(a.rows_out ? a.rows_out[m] : m) * d.Cout_stride + col;
      if (epi & SNAP_EPI_BIAS) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bb[e];
      }
      if (!SKIP_RES && (epi & SNAP_EPI_RESIDUAL)) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(a.residual + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rr[e];
      }
      if (epi & SNAP_EPI_UPSAMPLE2X_ADD) {
        // bilinear x2 of the coarser level (half-pixel centres, edge clamp)
        const int n = m / HoWo;
        const int rr = m - n * HoWo;
        const int ho = rr / d.Wo;
        const int wo = rr - ho * d.Wo;
        const float sh = (ho + 0.5f) * 0.5f - 0.5f;
        const float sw = (wo + 0.5f) * 0.5f - 0.5f;
        const float fh = floorf(sh), fw = floorf(sw);
        const float wh_hi = sh - fh, wh_lo = 1.f - wh_hi;
        const float ww_hi = sw - fw, ww_lo = 1.f - ww_hi;
        const int h0 = min(max((int)fh, 0), Hp - 1), h1 = min(max((int)fh + 1, 0), Hp - 1);
        const int w0 = min(max((int)fw, 0), Wp - 1), w1 = min(max((int)fw + 1, 0), Wp - 1);
        const int64_t base = (int64_t)n * Hp * Wp;
        const f32x4 p00 = *reinterpret_cast<const f32x4*>(
            a.up_prev + (base + (int64_t)h0 * Wp + w0) * d.Cout_stride + col);
        const f32x4 p01 = *reinterpret_cast<const f32x4*>(
            a.up_prev + (base + (int64_t)h0 * Wp + w1) * d.Cout_stride + col);
        const f32x4 p10 = *reinterpret_cast<const f32x4*>(
            a.up_prev + (base + (int64_t)h1 * Wp + w0) * d.Cout_stride + col);
        const f32x4 p11 = *reinterpret_cast<const f32x4*>(
            a.up_prev + (base + (int64_t)h1 * Wp + w1) * d.Cout_stride + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c0 = p00[e] * wh_lo + p10[e] * wh_hi;
          const float c1 = p01[e] * wh_lo + p11[e] * wh_hi;
          v[e] += c0 * ww_lo + c1 * ww_hi;
        }
      }
      if (epi & SNAP_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = snap_relu(v[e]);
      }
      if (epi & SNAP_EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = snap_gelu_tanh(v[e]);
      }
      if ((epi & SNAP_EPI_ROWMASK) && a.row_mask[m] == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (OUTH == 0) {
        *reinterpret_cast<f32x4*>(a.y + o) = v;
      } else {
        if (a.y) *reinterpret_cast<f32x4*>(a.y + o) = v;
        if constexpr (OUTH == 1) {
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          *reinterpret_cast<bf16x4_t*>(static_cast<__bf16*>(a.y_half) + o) = __builtin_convertvector(v, bf16x4_t);
        } else {
          typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
          *reinterpret_cast<f16x4_t*>(static_cast<_Float16*>(a.y_half) + o) = __builtin_convertvector(v, f16x4_t);
        }
      }
      if (GNB && gnb) {
        const int sl = m >= m_split ? 1 : 0;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.gnb_x + (int64_t)m * d.Cout + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (the arithmetic of gn_bwd_partial_kernel / gn_bwd_apply_kernel: the same gate, bit for bit)
          float xh, dyp;
          if (a.gnb_mode == SNAP_PRO_GN_RELU) {
            xh = (xv[e] - vb_mu[sl][e]) * vb_rs[sl][e];
            dyp = (xh * vb_g[e] + vb_b[e] > 0.f) ? v[e] : 0.f;
          } else {
            xh = (fmaxf(xv[e], 0.f) - vb_mu[sl][e]) * vb_rs[sl][e];
            dyp = v[e];
          }
          gs1[sl][e] += dyp;
          gs2[sl][e] += dyp * xh;
        }
      } else if (want_stats) {
        const int sl = m >= m_split ? 1 : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = a.gn_relu ? snap_relu(v[e]) : v[e];
          gs1[sl][e] += t;
          gs2[sl][e] += t * t;
          if constexpr (DUAL) {
            const float r = snap_relu(v[e]);
            hs1[sl][e] += r;
            hs2[sl][e] += r * r;
          }
        }
      }
    }
  }
  if (want_stats) {
    // fixed-order reduction over the 256/Q row groups through LDS, then one writer per
    // (image slot, column): deterministic.
    constexpr int RG = NT / Q;
    const int q = tid % Q, rg = tid / Q;
    for (int pass = 0; pass < (want_relu_too ? 2 : 1); ++pass) {
    float* const dst_partial = pass ? a.gn_partial2 : a.gn_partial;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        smem[((rg * 2 + sl) * BN + 4 * q + e) * 2 + 0] = pass ? hs1[sl][e] : gs1[sl][e];
        smem[((rg * 2 + sl) * BN + 4 * q + e) * 2 + 1] = pass ? hs2[sl][e] : gs2[sl][e];
      }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NT) {
      const int sl = i / BN, c = i - sl * BN;
      const int col = n0 + c;
      const int n = n_first + sl;
      // slot 1 exists only if the tile reaches into the next image
      const bool live = col < d.Cout && n < d.N && (sl == 0 || (m0 + BM > m_split && m_split < Meff));
      if (!live) continue;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        t1 += smem[((r * 2 + sl) * BN + c) * 2 + 0];
        t2 += smem[((r * 2 + sl) * BN + c) * 2 + 1];
      }
      const int slab = row_t - (int)(((int64_t)n * HoWo) / BM);
      float* o = dst_partial + (((int64_t)n * a.gn_slabs + slab) * d.Cout + col) * 2;
      o[0] = t1;
      o[1] = t2;
    }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    const float* __restrict__ partial, int S, int64_t M, int Cout, int Cout_stride, int epi,
    const float* __restrict__ bias, const float* __restrict__ residual,
    const uint8_t* __restrict__ row_mask, float* __restrict__ y) {
  const int Q = Cout >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * Q) return;
  const int64_t m = i / Q;
  const int col = 4 * (int)(i - m * Q);
  f32x4 v = *reinterpret_cast<const f32x4*>(partial + m * Cout + col);
  int s = 1;
  for (; s + 3 < S; s += 4) {        // four splits in flight (added in ascending order: the same sum)
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      t[u] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(s + u) * M + m) * Cout + col);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += t[u][e];
  }
  for (; s < S; ++s) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(partial + ((int64_t)s * M + m) * Cout + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += t[e];
  }
  const int64_t o = m * Cout_stride + col;
  if (epi & SNAP_EPI_BIAS) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bb[e];
  }
  if (epi & SNAP_EPI_RESIDUAL) {
    const f32x4 rr = *reinterpret_cast<const f32x4*>(residual + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += rr[e];
  }
  if (epi & SNAP_EPI_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = snap_relu(v[e]);
  }
  if (epi & SNAP_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = snap_gelu_tanh(v[e]);
  }
  if ((epi & SNAP_EPI_ROWMASK) && row_mask[m] == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
  *reinterpret_cast<f32x4*>(y + o) = v;
}

// The same reduce for a launch that also owes GroupNorm partial sums (a split-K launch has no
// epilogue that sees finished outputs): one workgroup = 32 consecutive rows x all columns, a thread =
// a column quad and every PW-th row; the sums of what is stored leave per (image, 32-ROW slab,
// channel) in conv_epilogue's layout (tile_rows = 32), reduced over the row lanes in a fixed order.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(
    const float* __restrict__ partial, int S, int64_t M, int Cout, int epi,
    const float* __restrict__ bias, const float* __restrict__ residual,
    const uint8_t* __restrict__ row_mask, float* __restrict__ y, int HoWo, int N, int gn_slabs,
    int gn_relu, float* __restrict__ gn_partial) {
  __shared__ float red[256 * 16];
  const int Q = Cout >> 2;
  const int QW = Q < 256 ? Q : 256;                   // column quads in flight (Cout % 4 == 0; Q | 256 or Q >= 256)
  const int PW = 256 / QW;                            // row lanes
  const int tq = threadIdx.x % QW, tp = threadIdx.x / QW;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  const int n_first = (int)(m0 / HoWo);
  const int64_t m_split = (int64_t)(n_first + 1) * HoWo;
  for (int q0 = 0; q0 < Q; q0 += QW) {
    const int q = q0 + tq;
    const int col = 4 * q;
    float s1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float s2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (q < Q && tp < PW) {
      // RB rows at a time: the loads of a split are issued for all of them before the first is added (a row
      // at a time, the S partial tiles of its quad were S dependent round trips, times 8-32 rows per thread);
      // every element still adds its splits in ascending order: the bits of splitk_reduce_kernel
      constexpr int RB = 8;
      for (int r0 = tp; r0 < 32; r0 += RB * PW) {
        f32x4 v[RB];
        bool ok[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          const int64_t m = m0 + r0 + k * PW;
          ok[k] = r0 + k * PW < 32 && m < M;
          v[k] = *reinterpret_cast<const f32x4*>(partial + (ok[k] ? m : m0) * Cout + col);
        }
        for (int sp = 1; sp < S; ++sp) {
          f32x4 t[RB];
#pragma unroll
          for (int k = 0; k < RB; ++k)
            t[k] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)sp * M + (ok[k] ? m0 + r0 + k * PW : m0)) * Cout + col);
#pragma unroll
          for (int k = 0; k < RB; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += t[k][e];
        }
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (epi & SNAP_EPI_BIAS) bb = *reinterpret_cast<const f32x4*>(bias + col);
        f32x4 rr[RB];
        if (epi & SNAP_EPI_RESIDUAL) {
#pragma unroll
          for (int k = 0; k < RB; ++k)
            rr[k] = *reinterpret_cast<const f32x4*>(residual + (ok[k] ? m0 + r0 + k * PW : m0) * Cout + col);
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          if (!ok[k]) continue;
          const int64_t m = m0 + r0 + k * PW;
          const int64_t o = m * Cout + col;
          if (epi & SNAP_EPI_BIAS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += bb[e];
          }
          if (epi & SNAP_EPI_RESIDUAL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += rr[k][e];
          }
          if (epi & SNAP_EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = snap_relu(v[k][e]);
          }
          if (epi & SNAP_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = snap_gelu_tanh(v[k][e]);
          }
          if ((epi & SNAP_EPI_ROWMASK) && row_mask[m] == 0) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(y + o) = v[k];
          const int sl = m >= m_split ? 1 : 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = gn_relu ? snap_relu(v[k][e]) : v[k][e];
            s1[sl][e] += t;
            s2[sl][e] += t * t;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[threadIdx.x * 16 + sl * 8 + e] = s1[sl][e];
        red[threadIdx.x * 16 + sl * 8 + 4 + e] = s2[sl][e];
      }
    __syncthreads();
    // one writer per (slot, column quad): fixed order over the row lanes
    if (tp == 0 && q < Q) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int n = n_first + sl;
        const bool live = n < N && (sl == 0 || (m0 + 32 > m_split && m_split < M));
        if (!live) continue;
        const int slab = (int)(blockIdx.x - (((int64_t)n * HoWo) >> 5));
        float* o = gn_partial + (((int64_t)n * gn_slabs + slab) * Cout + col) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t1 = 0.f, t2 = 0.f;
          for (int pp = 0; pp < PW; ++pp) {
            t1 += red[(pp * QW + tq) * 16 + sl * 8 + e];
            t2 += red[(pp * QW + tq) * 16 + sl * 8 + 4 + e];
          }
          o[2 * e] = t1;
          o[2 * e + 1] = t2;
        }
      }
    }
  }
}

// the closing pass of a split-K launch (with the GroupNorm partial sums where the launch owes them)
inline int launch_splitk_reduce(const ConvArgs& a, hipStream_t s) {
  if (a.gn_partial) {
    if (!a.gn_rows32 || (a.d.Cout & 3) || (256 % ((a.d.Cout >> 2) < 256 ? (a.d.Cout >> 2) : 256)) != 0)
      return SNAP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)snap_cdiv((int64_t)a.M, 32)), dim3(256), 0, s,
                       (const float*)a.kpartial, a.ksplit, (int64_t)a.M, a.d.Cout, a.d.epilogue, a.bias,
                       a.residual, a.row_mask, a.y, a.d.Ho * a.d.Wo, a.d.N, (a.d.Ho * a.d.Wo) / 32 + 2,
                       a.gn_relu, a.gn_partial);
  } else {
    const int64_t total4 = (int64_t)a.M * (a.d.Cout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0, s,
                       (const float*)a.kpartial, a.ksplit, (int64_t)a.M, a.d.Cout, a.d.Cout_stride,
                       a.d.epilogue, a.bias, a.residual, a.row_mask, a.y);
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// split-K heuristic: launches with at most splitk_max_tiles() output tiles (1.5 per CU)
// are split into about splitk_target() workgroups.  Measured in one box (C2 inference /
// C3 train step): off 59.75 / 162.5 ms; tiles<=128 59.96 / 159.9; tiles<=384, target 768
// 59.34 / 157.8; tiles<=256, target 1024 59.55 / 158.4.  (A caller that wants no split-K
// passes no workspace.)
#ifndef SNAP_SPLITK_MAX_TILES
#define SNAP_SPLITK_MAX_TILES 384
#endif
#ifndef SNAP_SPLITK_TARGET
#define SNAP_SPLITK_TARGET 768
#endif
constexpr int64_t splitk_max_tiles() { return SNAP_SPLITK_MAX_TILES; }
constexpr int splitk_target() { return SNAP_SPLITK_TARGET; }

// Tile choice: the largest tile that still yields >= 2 workgroups per CU; small-M
// layers (deep stages, few images) fall back to 64x64 tiles to fill the 256 CUs.
struct TileChoice { int bm, bn; };
// SnapConvDesc.tile_hint = tile code (bm * 1000 + bn; 0 = automatic) + 1 000 000 x mode (tests and
// tools; 0 = automatic): 1 = tiled body only, 2 = the stationary kernels also below their row-count
// threshold, 3 = no weights-stationary kernel, 4 = 2 and 3
inline int hint_tile(int h) { return h % 1000000; }
inline int hint_mode(int h) { return h / 1000000; }
// forced = SnapConvDesc.tile_hint; K = KH * KW * Cin.  Deep reductions over few rows (the 3 x 3
// layers of the aerial encoder and of the last StreetView stage: K >= 2304) keep the 128 x 128 tile
// and fill the machine by split-K instead of shrinking the tile: measured 0.098 -> 0.068 ms
// (8 x 34 x 34, 3x3 256 -> 256), 0.087 -> 0.076 (8 x 17 x 17, 3x3 512 -> 512), 0.234 -> 0.208
// (40 x 17 x 17), tools/conv3x3_small_bench.py; in the C2 step these layers 1.65 -> 1.24 ms.  (From
// K = 1024 the 1 x 1 layers would switch too: level in time, and a split-K launch cannot emit the
// GroupNorm partial sums -- the stand-alone statistics pass then costs what the 3 x 3 layers gain.)
inline int64_t desc_k(const SnapConvDesc& d) { return (int64_t)d.KH * d.KW * d.Cin; }
inline TileChoice choose_tile(int64_t M, int64_t N, int forced, int64_t K) {
  forced = hint_tile(forced);
  const int64_t kMin = 512;
  const bool n_wide_ok = N > 64 && !(N % 128 != 0 && N % 128 <= 64);
  if (forced != 128128 && forced != 128064 && forced != 64128 && forced != 64064) forced = 0;
  if (forced == 128128 || (!forced && n_wide_ok && snap_cdiv(M, 128) * snap_cdiv(N, 128) >= kMin))
    return {128, 128};
  if (!forced && n_wide_ok && K >= 2304 && M >= 128) return {128, 128};
  // deep 1 x 1 reductions: a row's K operands are fetched / normalised / split once per COLUMN tile, so the
  // 128-wide tile halves that work per MFMA (tools/conv_tile_sweep.py: 2048 -> 512 @ 40 x 17 x 17 119 -> 100 us,
  // 1024 -> 256 @ 8 x 34 x 34 42.7 -> 33.3 us on one workgroup per CU)
  const int64_t t64128 = snap_cdiv(M, 64) * snap_cdiv(N, 128);
  if (!forced && n_wide_ok && N >= 512 && K >= 1024 && t64128 >= kMin) return {64, 128};
  if (forced == 128064 || (!forced && snap_cdiv(M, 128) * snap_cdiv(N, 64) >= kMin)) return {128, 64};
  if (forced == 64128 || (!forced && n_wide_ok && N >= 512 && t64128 >= kMin)) return {64, 128};
  if (!forced && n_wide_ok && K >= 1024 && t64128 >= 256) return {64, 128};
  return {64, 64};   // (forced == 64064 included)
}

}  // namespace

#endif  // SNAP_CSRC_CONV_COMMON_H_
