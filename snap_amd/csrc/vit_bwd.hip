// Backward kernels of the ViT encoder pieces of vit_ops.hip (training with
// encoder_name='vit'; the reference has no ViT, see vit_ops.hip).
//   * LayerNorm VJP: dx per token (one wave per token, row in registers) + per-workgroup
//     column partials of dgamma / dbeta, summed in fixed order by a second kernel.
//   * GELU (tanh form) forward / VJP as plain element-wise kernels (the training path keeps
//     the pre-activation, so the epilogue-fused GELU of the inference path is not used).
//   * attention VJP, two kernels on v_mfma_f32_32x32x16_bf16, no atomics:
//       dq kernel   -- a workgroup owns 128 queries and walks the key blocks (mirror of the
//                      forward: transposed scores, one query per lane);
//       dk/dv kernel-- a workgroup owns 128 keys and walks the query blocks (one key per lane).
//     Both recompute P = exp2(c s - lse) from the saved log-sum-exp; as in the forward, the
//     result of the first MFMA (P or dS, one query / key per lane) is already the B fragment
//     of the next one, so neither P nor dS ever touches LDS.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- LayerNorm backward -------------------------------------------------------------------
constexpr int LN_MAXQ = 4;
constexpr int LNB_ROWS = 8;    // rows per workgroup: 2 per wave, both in flight together

__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
    float* __restrict__ dx, float* __restrict__ partial /* [blocks][2][C] */, int64_t M, int C,
    float eps) {
  __shared__ float red[4][2][LN_MAXQ * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Q = C >> 2;
  constexpr int R = LNB_ROWS / 4;
  f32x4 g[LN_MAXQ], ag[LN_MAXQ], ab[LN_MAXQ];
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    g[i] = q < Q ? *reinterpret_cast<const f32x4*>(gamma + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // all loads of the wave's rows are issued before the first reduction (a row costs three
  // dependent wave reductions: one row at a time would expose the full memory latency per row)
  f32x4 v[R][LN_MAXQ], d[R][LN_MAXQ];
  bool rok[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int64_t m = (int64_t)blockIdx.x * LNB_ROWS + wid * R + rr;
    rok[rr] = m < M;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      const bool ok = rok[rr] && q < Q;
      v[rr][i] = ok ? *reinterpret_cast<const f32x4*>(x + m * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      d[rr][i] = ok ? *reinterpret_cast<const f32x4*>(dy + m * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int64_t m = (int64_t)blockIdx.x * LNB_ROWS + wid * R + rr;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) s += (v[rr][i][0] + v[rr][i][1]) + (v[rr][i][2] + v[rr][i][3]);
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i)
      if (lane + 64 * i < Q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = v[rr][i][e] - mean;
          s2 += dl * dl;
        }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)C + eps);
    // xhat, g*dy and their row means (rows past M hold zeros: they add nothing)
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i)
      if (lane + 64 * i < Q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (v[rr][i][e] - mean) * rstd;
          const float gd = g[i][e] * d[rr][i][e];
          a1 += gd;
          a2 += gd * xh;
          ag[i][e] += d[rr][i][e] * xh;
          ab[i][e] += d[rr][i][e];
          v[rr][i][e] = xh;
          d[rr][i][e] = gd;
        }
      }
    const float m1 = wave_sum(a1) / (float)C, m2 = wave_sum(a2) / (float)C;
    if (rok[rr]) {
#pragma unroll
      for (int i = 0; i < LN_MAXQ; ++i) {
        const int q = lane + 64 * i;
        if (q < Q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd * (d[rr][i][e] - m1 - v[rr][i][e] * m2);
          *reinterpret_cast<f32x4*>(dx + m * C + 4 * q) = o;
        }
      }
    }
  }
  // column partials: the 4 waves own the same columns -> fixed-order sum through LDS
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wid][0][4 * (lane + 64 * i) + e] = ag[i][e];
      red[wid][1][4 * (lane + 64 * i) + e] = ab[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      t0 += red[w][0][c];
      t1 += red[w][1][c];
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = t0;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = t1;
  }
}

// 32 columns x 8 block groups per workgroup; group g sums blocks g, g + 8, ... and the eight
// group sums are added in fixed order: deterministic, and wide enough to stream the partials.
__global__ __launch_bounds__(256) void layer_norm_bwd_reduce_kernel(
    const float* __restrict__ partial, int nblocks, int C, float* __restrict__ dgamma,
    float* __restrict__ dbeta) {
  __shared__ float red[8][2][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + col;
  float t0 = 0.f, t1 = 0.f;
  if (c < C) {
    for (int b = grp; b < nblocks; b += 8) {
      t0 += partial[((int64_t)b * 2 + 0) * C + c];
      t1 += partial[((int64_t)b * 2 + 1) * C + c];
    }
  }
  red[grp][0][col] = t0;
  red[grp][1][col] = t1;
  __syncthreads();
  if (grp == 0 && c < C) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      s0 += red[g][0][col];
      s1 += red[g][1][col];
    }
    dgamma[c] = s0;
    dbeta[c] = s1;
  }
}

// ---- GELU ------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_fwd(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x / (1.0f + __expf(-2.0f * u));
}
__device__ __forceinline__ float gelu_grad(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float sg = 1.0f / (1.0f + __expf(-2.0f * u));         // 0.5 (1 + tanh u)
  const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
  return sg + x * 2.0f * sg * (1.0f - sg) * du;               // d/dx [x sg(2u)]
}
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  reinterpret_cast<f32x4*>(y)[i] = f32x4{gelu_fwd(v[0]), gelu_fwd(v[1]), gelu_fwd(v[2]), gelu_fwd(v[3])};
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                float* __restrict__ dx, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
  reinterpret_cast<f32x4*>(dx)[i] = f32x4{g[0] * gelu_grad(v[0]), g[1] * gelu_grad(v[1]),
                                          g[2] * gelu_grad(v[2]), g[3] * gelu_grad(v[3])};
}

// ---- attention backward ------------------------------------------------------------------------
constexpr int AT_D = 64;
constexpr int AT_BLK = 64;      // rows of the streamed operand per block
constexpr int AT_RS = 144;      // row-major [row][d] image: row stride in bytes (b128 fragment reads)
constexpr int AT_TS = 136;      // transposed [d][row] image: row stride in bytes (b64 fragment reads)
constexpr int AT_RM = AT_BLK * AT_RS, AT_TM = AT_D * AT_TS;

struct AttnBwdArgs {
  const float* qkv;    // [B, N, 3, H, 64]
  const float* out;    // [B, N, H*64]   forward result
  const float* dout;   // [B, N, H*64]
  const float* lse;    // [B, H, N]      base-2 log-sum-exp (forward)
  float* delta;        // [B, H, N]      sum_d dout * out (written by the dq kernel)
  float* dqkv;         // [B, N, 3, H, 64]
  int B, N, H;
  float scale, scale_log2e;
};

__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
  const bf16x4 bl = __builtin_convertvector(lo, bf16x4), bh = __builtin_convertvector(hi, bf16x4);
  return bf16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
}
// rows of an acc tile held by this lane: rmap(r) = (r & 3) + 8 (r >> 2) + 4 lhi
__device__ __forceinline__ int rmap(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }

// B-fragment of a row-major global [*, 64] row: 8 consecutive d at 16 s + 8 lhi, times `mul`
__device__ __forceinline__ bf16x8 row_frag(const float* row, int s, int lhi, float mul) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(row + 16 * s + 8 * lhi);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 16 * s + 8 * lhi + 4);
  return pack8(f32x4{lo[0] * mul, lo[1] * mul, lo[2] * mul, lo[3] * mul},
               f32x4{hi[0] * mul, hi[1] * mul, hi[2] * mul, hi[3] * mul});
}
// A-fragment from a transposed image [d][row]: the 8 rows this lane half pairs with the
// acc-layout B fragment of (tile, hs): rows 32 tile + 16 hs + 4 lhi + {0..3} and the same + 8
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int drow, int tile, int hs, int lhi) {
  const char* p = img + drow * AT_TS + (32 * tile + 16 * hs + 4 * lhi) * 2;
  const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(p + 16);
  return bf16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}
// acc regs 8 hs .. 8 hs + 7 -> B fragment
__device__ __forceinline__ bf16x8 acc_frag(const float* p, int hs) {
  return pack8(f32x4{p[8 * hs + 0], p[8 * hs + 1], p[8 * hs + 2], p[8 * hs + 3]},
               f32x4{p[8 * hs + 4], p[8 * hs + 5], p[8 * hs + 6], p[8 * hs + 7]});
}
// [d (acc rows)][lane column] accumulators of one wave -> 32 rows x 64 d through LDS, 256-byte
// row stores to dst + row * row_stride
__device__ __forceinline__ void store_transposed(const f32x16 (&acc)[2], float mul, float* stage,
                                                 float* dst, int64_t row_stride, int row0, int N,
                                                 int lane) {
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[l31 * 68 + 32 * t + rmap(r, lhi)] = acc[t][r] * mul;
  __syncthreads();                       // (every wave of the workgroup stores: uniform call)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    const int row = idx >> 4, qd = idx & 15;
    if (row0 + row < N)
      *reinterpret_cast<f32x4*>(dst + (int64_t)(row0 + row) * row_stride + 4 * qd) =
          *reinterpret_cast<const f32x4*>(stage + row * 68 + 4 * qd);
  }
  __syncthreads();
}

// dq: one query per lane; streams K (row-major + transposed) and V (row-major)
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (2 * AT_RM + AT_TM)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int64_t tok = (int64_t)3 * a.H * AT_D;
  const int64_t orow = (int64_t)a.H * AT_D;
  const float* base = a.qkv + (int64_t)b * a.N * tok + (int64_t)h * AT_D;
  const int q = min(q0 + l31, a.N - 1);
  const bool q_ok = q0 + l31 < a.N;
  const float* qp = base + (int64_t)q * tok;
  const float* dop = a.dout + ((int64_t)b * a.N + q) * orow + h * AT_D;
  const float* op = a.out + ((int64_t)b * a.N + q) * orow + h * AT_D;
  bf16x8 qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = row_frag(qp, s, lhi, a.scale_log2e);
    dof[s] = row_frag(dop, s, lhi, 1.0f);
#pragma unroll
    for (int e = 0; e < 8; ++e) dpart += dop[16 * s + 8 * lhi + e] * op[16 * s + 8 * lhi + e];
  }
  const float delta = dpart + __shfl_xor(dpart, 32);
  const int64_t stat = ((int64_t)b * a.H + h) * a.N;
  if (q_ok && lhi == 0) a.delta[stat + q] = delta;
  const float lse = a.lse[stat + q];

  const int quad = tid & 15, grp = tid >> 4;     // loader: d quad, group of 4 consecutive keys
  f32x4 kr[4], vr[4];
  const float* kbase = base + (int64_t)a.H * AT_D;
  const float* vbase = base + (int64_t)2 * a.H * AT_D;
  auto load_block = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(k0 + 4 * grp + i, a.N - 1);
      kr[i] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * tok + 4 * quad);
      vr[i] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * tok + 4 * quad);
    }
  };
  auto store_block = [&](int buf) {
    char* ks = smem + buf * (2 * AT_RM + AT_TM);
    char* vs = ks + AT_RM;
    char* kt = vs + AT_RM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<bf16x4*>(ks + (4 * grp + i) * AT_RS + quad * 8) = __builtin_convertvector(kr[i], bf16x4);
      *reinterpret_cast<bf16x4*>(vs + (4 * grp + i) * AT_RS + quad * 8) = __builtin_convertvector(vr[i], bf16x4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 t = {kr[0][e], kr[1][e], kr[2][e], kr[3][e]};
      *reinterpret_cast<bf16x4*>(kt + (4 * quad + e) * AT_TS + grp * 8) = __builtin_convertvector(t, bf16x4);
    }
  };

  f32x16 dqt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[t][r] = 0.f;

  const int nblk = (a.N + AT_BLK - 1) / AT_BLK;
  load_block(0);
  store_block(0);
  __syncthreads();
  for (int kb = 0; kb < nblk; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nblk;
    if (more) load_block((kb + 1) * AT_BLK);
    const char* ks = smem + cur * (2 * AT_RM + AT_TM);
    const char* vs = ks + AT_RM;
    const char* kt = vs + AT_RM;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {          // key tile
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (32 * t2 + l31) * AT_RS + (2 * s + lhi) * 16);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs + (32 * t2 + l31) * AT_RS + (2 * s + lhi) * 16);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * AT_BLK + 32 * t2 + rmap(r, lhi);
        const float p = key < a.N ? exp2f(st[r] - lse) : 0.f;
        ds[r] = p * (dp[r] - delta);
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const bf16x8 dsf = acc_frag(ds, hs);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          dqt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(kt, 32 * t + l31, t2, hs, lhi), dsf,
                                                          dqt[t], 0, 0, 0);
      }
    }
    if (more) store_block(cur ^ 1);
    __syncthreads();
  }
  float* stage = reinterpret_cast<float*>(smem) + wid * (32 * 68);
  store_transposed(dqt, a.scale, stage, a.dqkv + (int64_t)b * a.N * tok + (int64_t)h * AT_D, tok, q0,
                   a.N, lane);
}

// dk, dv: one key per lane; streams Q~ = q * c and dO, each row-major + transposed
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const AttnBwdArgs a) {
  constexpr int STAGE = 2 * AT_RM + 2 * AT_TM + 2 * AT_BLK * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * 128 + wid * 32;
  const int64_t tok = (int64_t)3 * a.H * AT_D;
  const int64_t orow = (int64_t)a.H * AT_D;
  const float* base = a.qkv + (int64_t)b * a.N * tok + (int64_t)h * AT_D;
  const int key = min(k0 + l31, a.N - 1);
  const bool key_ok = k0 + l31 < a.N;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = row_frag(base + (int64_t)a.H * AT_D + (int64_t)key * tok, s, lhi, 1.0f);
    vf[s] = row_frag(base + (int64_t)2 * a.H * AT_D + (int64_t)key * tok, s, lhi, 1.0f);
  }
  const int64_t stat = ((int64_t)b * a.H + h) * a.N;

  const int quad = tid & 15, grp = tid >> 4;     // loader: d quad, group of 4 consecutive queries
  f32x4 qr[4], gr[4];
  float lse_r = 0.f, del_r = 0.f;
  const float* dobase = a.dout + (int64_t)b * a.N * orow + h * AT_D;
  auto load_block = [&](int qb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = min(qb + 4 * grp + i, a.N - 1);
      qr[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)qq * tok + 4 * quad);
      gr[i] = *reinterpret_cast<const f32x4*>(dobase + (int64_t)qq * orow + 4 * quad);
    }
    if (tid < AT_BLK) {
      const int qq = min(qb + tid, a.N - 1);
      lse_r = a.lse[stat + qq];
      del_r = a.delta[stat + qq];
    }
  };
  auto store_block = [&](int buf) {
    char* qs = smem + buf * STAGE;
    char* gs = qs + AT_RM;
    char* qt = gs + AT_RM;
    char* gt = qt + AT_TM;
    float* ls = reinterpret_cast<float*>(gt + AT_TM);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) qr[i][e] *= a.scale_log2e;
      *reinterpret_cast<bf16x4*>(qs + (4 * grp + i) * AT_RS + quad * 8) = __builtin_convertvector(qr[i], bf16x4);
      *reinterpret_cast<bf16x4*>(gs + (4 * grp + i) * AT_RS + quad * 8) = __builtin_convertvector(gr[i], bf16x4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 tq = {qr[0][e], qr[1][e], qr[2][e], qr[3][e]};
      const f32x4 tg = {gr[0][e], gr[1][e], gr[2][e], gr[3][e]};
      *reinterpret_cast<bf16x4*>(qt + (4 * quad + e) * AT_TS + grp * 8) = __builtin_convertvector(tq, bf16x4);
      *reinterpret_cast<bf16x4*>(gt + (4 * quad + e) * AT_TS + grp * 8) = __builtin_convertvector(tg, bf16x4);
    }
    if (tid < AT_BLK) {
      ls[tid] = lse_r;
      ls[AT_BLK + tid] = del_r;
    }
  };

  f32x16 dvt[2], dkt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[t][r] = 0.f; dkt[t][r] = 0.f; }

  const int nblk = (a.N + AT_BLK - 1) / AT_BLK;
  load_block(0);
  store_block(0);
  __syncthreads();
  for (int qb = 0; qb < nblk; ++qb) {
    const int cur = qb & 1;
    const bool more = qb + 1 < nblk;
    if (more) load_block((qb + 1) * AT_BLK);
    const char* qs = smem + cur * STAGE;
    const char* gs = qs + AT_RM;
    const char* qt = gs + AT_RM;
    const char* gt = qt + AT_TM;
    const float* ls = reinterpret_cast<const float*>(gt + AT_TM);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {          // query tile: rows of the acc tile = queries
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(qs + (32 * t2 + l31) * AT_RS + (2 * s + lhi) * 16);
        const bf16x8 gfr = *reinterpret_cast<const bf16x8*>(gs + (32 * t2 + l31) * AT_RS + (2 * s + lhi) * 16);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, vf[s], dp, 0, 0, 0);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = 32 * t2 + rmap(r, lhi);
        const bool ok = key_ok && (qb * AT_BLK + ql) < a.N;
        p[r] = ok ? exp2f(st[r] - ls[ql]) : 0.f;
        ds[r] = p[r] * (dp[r] - ls[AT_BLK + ql]);
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const bf16x8 pf = acc_frag(p, hs), dsf = acc_frag(ds, hs);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          dvt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(gt, 32 * t + l31, t2, hs, lhi), pf,
                                                          dvt[t], 0, 0, 0);
          dkt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qt, 32 * t + l31, t2, hs, lhi), dsf,
                                                          dkt[t], 0, 0, 0);
        }
      }
    }
    if (more) store_block(cur ^ 1);
    __syncthreads();
  }
  float* stage = reinterpret_cast<float*>(smem) + wid * (32 * 68);
  float* dbase = a.dqkv + (int64_t)b * a.N * tok + (int64_t)h * AT_D;
  // dK = scale * dS^T q = ln2 * dS^T (q c)
  store_transposed(dkt, 0.6931471805599453f, stage, dbase + (int64_t)a.H * AT_D, tok, k0, a.N, lane);
  store_transposed(dvt, 1.0f, stage, dbase + (int64_t)2 * a.H * AT_D, tok, k0, a.N, lane);
}

}  // namespace

extern "C" size_t snap_layer_norm_bwd_workspace_bytes(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)snap_cdiv(M, LNB_ROWS) * 2 * C * sizeof(float);
}

extern "C" int snap_layer_norm_bwd_f32(const float* x, const float* dy, const float* gamma,
                                       float* dx, float* dgamma, float* dbeta, int64_t M,
                                       int32_t C, float eps, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  if (!x || !dy || !gamma || !dx || !dgamma || !dbeta || !workspace) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0 || C > 256 * LN_MAXQ) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_layer_norm_bwd_workspace_bytes(M, C)) return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15) ||
      (reinterpret_cast<uintptr_t>(dx) & 15) || (reinterpret_cast<uintptr_t>(gamma) & 15))
    return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = (int)snap_cdiv(M, LNB_ROWS);
  hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, dy, gamma, dx,
                     static_cast<float*>(workspace), M, C, eps);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(layer_norm_bwd_reduce_kernel, dim3((unsigned)snap_cdiv(C, 32)), dim3(256), 0, s,
                     static_cast<const float*>(workspace), nb, C, dgamma, dbeta);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_gelu_f32(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y) return SNAP_ERR_NULL;
  if (n <= 0 || n % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)snap_cdiv(n / 4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, n / 4);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx) return SNAP_ERR_NULL;
  if (n <= 0 || n % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15) ||
      (reinterpret_cast<uintptr_t>(dx) & 15))
    return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)snap_cdiv(n / 4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, dy, dx, n / 4);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_attention_bwd_bf16_f32(const float* qkv, const float* out, const float* dout,
                                           const float* lse, float* delta, float* dqkv, int32_t B,
                                           int32_t N, int32_t H, int32_t D, float scale,
                                           void* stream) {
  if (!qkv || !out || !dout || !lse || !delta || !dqkv) return SNAP_ERR_NULL;
  if (B <= 0 || N <= 0 || H <= 0 || B > 65535 || H > 65535) return SNAP_ERR_BAD_SHAPE;
  if (D != AT_D) return SNAP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(dout) & 15) || (reinterpret_cast<uintptr_t>(dqkv) & 15))
    return SNAP_ERR_BAD_SHAPE;
  AttnBwdArgs a;
  a.qkv = qkv; a.out = out; a.dout = dout; a.lse = lse; a.delta = delta; a.dqkv = dqkv;
  a.B = B; a.N = N; a.H = H;
  a.scale = scale;
  a.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)snap_cdiv(N, 128), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attention_bwd_dq_kernel, grid, dim3(256), 0, s, a);     // also writes delta
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(attention_bwd_dkv_kernel, grid, dim3(256), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
