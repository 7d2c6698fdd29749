// Producers of the pre-split activation format of conv_ps.hip:
//   out [row][C / 16][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]  bf16 (64 B per row and 16 channels),
//   hi = bf16(v) (round to nearest even), lo = bf16(v - hi) (the subtraction is exact in f32):
// the two-part split conv_split.hip performs on the fly, done ONCE per activation.
//
// gn_norm_split_kernel: v = relu((y - mean) * rstd * gamma + beta), i.e. GroupNorm -> ReLU of
// snap/models/resnet.py:34-60,117-130 applied to a conv output whose per-(image, row tile,
// channel) partial sums came out of the producing conv's epilogue.  It REPLACES the stand-alone
// statistics finalize launch of that tensor (snap_group_norm_stats_from_partial_f32): every
// workgroup first reduces the partial sums of its image in fp64 (fixed order; the same mean /
// mean((v - mean)^2) / x / sqrt(var + eps) formulas), then streams its share of the image's rows.
// The arithmetic per element is the fused prologue's (apply_pro<SNAP_PRO_GN_RELU>), so a
// conv_ps launch over the result multiplies the very operands conv_split would have built.
//
// presplit_kernel: the plain split of an f32 tensor (the edge-padded map of the exhaustive voting).
#include "conv_common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// hi / lo bf16 parts of four f32 (one v_cvt_pk per pair, exact residual: as conv_split.hip)
__device__ __forceinline__ void split2x4(const f32x4& v, u32x2& hi, u32x2& lo) {
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

// lanes 2j (channel quad q even) and 2j + 1 hold the two halves of one k-octet: after one quad-
// permuted exchange the even lane owns the octet's 16 hi bytes, the odd lane its 16 lo bytes
__device__ __forceinline__ u32x4 pair_chunks(const u32x2& hi, const u32x2& lo, bool odd) {
  const u32x2 send = odd ? hi : lo;
  u32x2 recv;
#pragma unroll
  for (int e = 0; e < 2; ++e)
    recv[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[e], 0xB1 /* quad_perm [1,0,3,2] */,
                                                    0xF, 0xF, false);
  return odd ? u32x4{recv[0], recv[1], lo[0], lo[1]} : u32x4{hi[0], hi[1], recv[0], recv[1]};
}

struct NormSplitArgs {
  const float* y;        // [N, HW, C]
  const float* partial;  // [N, S, C, 2] sums / sums of squares per row tile of `tile_rows` pixels
  const float* gamma;    // [C]
  const float* beta;     // [C]
  void* out;             // [N * HW][C / 16][64 B]
  float* mu;             // optional [N, C]: the statistics, as the finalize launch would write them
  float* sc;
  int N, HW, C, groups, tile_rows, S;
  float eps;
  int rows_per_wg;
};

constexpr int kNsMaxC = 2048;

__global__ __launch_bounds__(256) void gn_norm_split_kernel(const NormSplitArgs a) {
  __shared__ float t_mu[kNsMaxC], t_sc[kNsMaxC], t_beta[kNsMaxC];
  __shared__ double red[256][2];
  __shared__ float g_mean[64], g_rstd[64];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int C = a.C, HW = a.HW;
  const int cpg = C / a.groups;
  // ---- statistics of image n: groups x tpg threads, fp64, fixed order ------------------------
  const int tpg = 256 / a.groups;                    // threads per group (groups <= 64, power of 2)
  const int g = tid / tpg, sub = tid - g * tpg;
  const int live = (int)((((int64_t)(n + 1) * HW - 1) / a.tile_rows) - (((int64_t)n * HW) / a.tile_rows)) + 1;
  {
    const int count = live * cpg;
    const float* pb = a.partial + ((int64_t)n * a.S * C + g * cpg) * 2;
    double t1 = 0.0, t2 = 0.0;
    for (int e = sub; e < count; e += tpg) {
      const int s = e / cpg, cc = e - s * cpg;
      const float* pp = pb + ((int64_t)s * C + cc) * 2;
      t1 += (double)pp[0];
      t2 += (double)pp[1];
    }
    red[tid][0] = t1;
    red[tid][1] = t2;
  }
  __syncthreads();
  if (sub == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < tpg; ++i) { t1 += red[tid + i][0]; t2 += red[tid + i][1]; }
    const double cnt = (double)HW;
    const double mean = t1 / (cnt * cpg);
    const double m2 = t2 - 2.0 * mean * t1 + cnt * (cpg * mean * mean);
    const float var = (float)(m2 / (cnt * cpg));
    g_mean[g] = (float)mean;
    g_rstd[g] = 1.0f / sqrtf(snap_relu(var) + a.eps);   // x / sqrt(var + eps): resnet.py:40
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int gg = c / cpg;
    const float m = g_mean[gg], s = g_rstd[gg] * a.gamma[c];
    t_mu[c] = m;
    t_sc[c] = s;
    t_beta[c] = a.beta[c];
    if (a.mu && blockIdx.x == 0) {
      a.mu[(int64_t)n * C + c] = m;
      a.sc[(int64_t)n * C + c] = s;
    }
  }
  __syncthreads();
  // ---- rows [r0, r1) of image n --------------------------------------------------------------
  const int r0 = blockIdx.x * a.rows_per_wg;
  const int r1 = min(r0 + a.rows_per_wg, HW);
  const int C4 = C >> 2;
  const int64_t total = (int64_t)(r1 - r0) * C4;
  const float* const yb = a.y + ((int64_t)n * HW + r0) * C;
  char* const ob = static_cast<char*>(a.out) + ((int64_t)n * HW + r0) * (int64_t)C * 4;
  constexpr int UN = 8;
  for (int64_t base = 0; base < total; base += 256 * UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = base + tid + 256 * u;
      v[u] = i < total ? *reinterpret_cast<const f32x4*>(yb + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = base + tid + 256 * u;
      const int q4 = (int)(i % C4);
      const int c = 4 * q4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(t_mu + c);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(t_sc + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(t_beta + c);
      f32x4 p;
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = apply_pro<SNAP_PRO_GN_RELU>(v[u][e], mu[e], sc[e], be[e], 1.f, 0.f);
      u32x2 hi, lo;
      split2x4(p, hi, lo);
      const bool odd = q4 & 1;
      const u32x4 ch = pair_chunks(hi, lo, odd);     // (C4 is even: a pair never straddles rows)
      if (i < total) {
        // byte offset of the row-major f32 quad i = 16 i; its 64-byte (row, channel tile) block
        // starts at 64 (i / 4) and holds hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15
        char* o = ob + 64 * (i >> 2) + (odd ? 32 : 0) + ((q4 >> 1) & 1) * 16;
        *reinterpret_cast<u32x4*>(o) = ch;
      }
    }
  }
}

__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ x, int64_t total4,
                                                       void* __restrict__ out) {
  // x [rows, C] f32 with C % 16 == 0 -> the same bytes re-arranged per 16-channel block
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const f32x4 v = i < total4 ? reinterpret_cast<const f32x4*>(x)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  u32x2 hi, lo;
  split2x4(v, hi, lo);
  const bool odd = i & 1;
  const u32x4 ch = pair_chunks(hi, lo, odd);
  if (i < total4) {
    char* o = static_cast<char*>(out) + 64 * (i >> 2) + (odd ? 32 : 0) + ((i >> 1) & 1) * 16;
    *reinterpret_cast<u32x4*>(o) = ch;
  }
}

}  // namespace

extern "C" int snap_gn_norm_split_f32(const float* y, const float* partial, int32_t N, int32_t HW,
                                      int32_t C, int32_t groups, float eps, int32_t tile_rows,
                                      const float* gamma, const float* beta, void* out,
                                      float* mu, float* sc, void* stream) {
  if (!y || !partial || !gamma || !beta || !out) return SNAP_ERR_NULL;
  if ((mu == nullptr) != (sc == nullptr)) return SNAP_ERR_NULL;
  if (N <= 0 || HW <= 0 || C <= 0 || C % 16 != 0 || C > kNsMaxC) return SNAP_ERR_BAD_SHAPE;
  if (groups <= 0 || groups > 64 || (groups & (groups - 1)) || C % groups != 0) return SNAP_ERR_BAD_SHAPE;
  if (tile_rows <= 0 || HW < tile_rows) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) return SNAP_ERR_BAD_SHAPE;
  NormSplitArgs a;
  a.y = y; a.partial = partial; a.gamma = gamma; a.beta = beta; a.out = out; a.mu = mu; a.sc = sc;
  a.N = N; a.HW = HW; a.C = C; a.groups = groups; a.tile_rows = tile_rows;
  a.S = HW / tile_rows + 2;
  a.eps = eps;
  // about 512 KB of rows per workgroup (the statistics prefix re-reads the image's partial sums:
  // S x C x 8 bytes from L2), at least ~512 workgroups in the launch when the tensor allows
  int64_t rows = (512 * 1024) / ((int64_t)C * 4);
  const int64_t cap = ((int64_t)N * HW + 511) / 512;
  if (rows > cap) rows = cap;
  if (rows < 16) rows = 16;
  if (rows > HW) rows = HW;
  a.rows_per_wg = (int)rows;
  const dim3 grid((unsigned)snap_cdiv(HW, rows), (unsigned)N);
  hipLaunchKernelGGL(gn_norm_split_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_presplit_f32(const float* x, int64_t rows, int32_t C, void* out, void* stream) {
  if (!x || !out) return SNAP_ERR_NULL;
  if (rows <= 0 || C <= 0 || C % 16 != 0) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return SNAP_ERR_BAD_SHAPE;
  const int64_t total4 = rows * (C / 4);
  if (snap_cdiv(total4, 256) > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, total4, out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
