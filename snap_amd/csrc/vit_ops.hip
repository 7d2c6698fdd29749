// ViT encoder kernels (BASELINE.json configs[4]: ViT-B/16 image encoder, bf16 attention GEMMs).
// The reference has no ViT (snap/models/image_encoder.py:103 accepts only 'resnet'); these
// kernels implement the published ViT block (Dosovitskiy et al. 2021; parameter layout of the
// big_vision / scenic `vit.py` encoders the reference's BiT loader comes from) so that
// `encoder_name='vit'` can feed the same lift / BEV / pose path.  Dense layers run on the conv
// engines; this file holds what is not a GEMM-with-epilogue:
//   * LayerNorm over the channel axis (one wave per token, two-pass statistics in registers);
//   * multi-head self-attention, flash style: S^T = K Q^T and O^T = V^T P^T on
//     v_mfma_f32_32x32x16_bf16 with the softmax kept in registers.  Computing the TRANSPOSED
//     scores puts one query per lane (column of the MFMA result), so the row statistics of the
//     softmax are per-lane scalars (one cross-half exchange per block) and the probabilities
//     come out of the first MFMA already in the B-fragment layout of the second one -- P never
//     touches LDS.  K is staged [key][d] and V transposed [d][key] in LDS (bf16, padded rows).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- LayerNorm ------------------------------------------------------------------------------
// y[m, :] = (x[m, :] - mean) * rsqrt(var + eps) * gamma + beta   (biased variance, as
// flax.linen.LayerNorm).  C % 4 == 0, C <= 64 * 4 * LN_MAXQ.
constexpr int LN_MAXQ = 4;   // float4 per lane: C <= 1024

// HALF: the row goes out rounded (RNE) to bf16 into y_half INSTEAD of y -- the operand of the dense layer that
// follows, which rounds it to that type anyway (snap_layer_norm_bf16out_f32)
template <bool HALF>
__global__ __launch_bounds__(256) void layer_norm_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, int64_t M, int C, float eps, __bf16* __restrict__ y_half = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int Q = C >> 2;
  f32x4 v[LN_MAXQ];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    v[i] = q < Q ? *reinterpret_cast<const f32x4*>(x + m * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < Q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dlt = v[i][e] - mean;
        s2 += dlt * dlt;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < Q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * q);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if constexpr (HALF) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4*>(y_half + m * C + 4 * q) = __builtin_convertvector(o, bf16x4);
      } else {
        *reinterpret_cast<f32x4*>(y + m * C + 4 * q) = o;
      }
    }
  }
}

// ---- attention ------------------------------------------------------------------------------
constexpr int AT_D = 64;        // head dimension
constexpr int AT_KB = 64;       // keys per block
constexpr int AT_QW = 32;       // queries per wave
constexpr int AT_KS = 144;      // K row stride in LDS, bytes (128 + 16: b128 reads conflict-free)
constexpr int AT_VS = 136;      // V^T row stride in LDS, bytes (128 + 8: b64 reads conflict-free)

struct AttnArgs {
  const float* qkv;   // [B, N, 3, H, 64]
  const __bf16* qkv_h;  // HIN kernels: the same tensor in bf16 (the fused QKV projection's bf16-only output)
  float* out;         // [B, N, H * 64]
  __bf16* out_half;   // non-null: the output rounded (RNE) to bf16 goes HERE instead of `out` (inference: the operand
                      // of the output projection)
  float* lse;         // optional [B, H, N]: base-2 log-sum-exp of the scaled scores (for the VJP)
  int B, N, H;
  float scale_log2e;  // softmax scale * log2(e)
};

// HIN: qkv arrives in bf16 (a.qkv_h).  K and V are the values the f32 kernel rounds to on its way into LDS -- they
// travel at half the bytes (a (b, h) panel is re-read by every 128-query workgroup: the kernel is bound by that L2
// traffic); Q is scaled AFTER its rounding to bf16 (one more rounding than the f32 kernel's bf16(q * scale)).
template <bool HIN>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (AT_KB * AT_KS + AT_D * AT_VS)];
  char* const Ks0 = smem;
  char* const Vs0 = smem + 2 * AT_KB * AT_KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (4 * AT_QW) + wid * AT_QW;
  const int64_t tok_stride = (int64_t)3 * a.H * AT_D;          // floats per token in qkv
  const float* base = a.qkv + (int64_t)b * a.N * tok_stride + (int64_t)h * AT_D;

  // Q^T fragments (B operand): lane = query column, 8 consecutive d per k-step
  bf16x8 qf[4];
  {
    const int q = min(q0 + l31, a.N - 1);
    const float* qp = base + (int64_t)q * tok_stride;
    const __bf16* qph = HIN ? a.qkv_h + ((int64_t)b * a.N + q) * tok_stride + (int64_t)h * AT_D : nullptr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 lo, hi;
      if constexpr (HIN) {
        const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qph + 16 * s + 8 * lhi);
        lo = f32x4{(float)qb[0], (float)qb[1], (float)qb[2], (float)qb[3]};
        hi = f32x4{(float)qb[4], (float)qb[5], (float)qb[6], (float)qb[7]};
      } else {
        lo = *reinterpret_cast<const f32x4*>(qp + 16 * s + 8 * lhi);
        hi = *reinterpret_cast<const f32x4*>(qp + 16 * s + 8 * lhi + 4);
      }
      const f32x4 l2 = {lo[0] * a.scale_log2e, lo[1] * a.scale_log2e, lo[2] * a.scale_log2e, lo[3] * a.scale_log2e};
      const f32x4 h2 = {hi[0] * a.scale_log2e, hi[1] * a.scale_log2e, hi[2] * a.scale_log2e, hi[3] * a.scale_log2e};
      const bf16x4 bl = __builtin_convertvector(l2, bf16x4), bh = __builtin_convertvector(h2, bf16x4);
      qf[s] = bf16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
    }
  }

  // loader coordinates
  const int kq = tid & 15, krow = tid >> 4;        // K: float4 quad of a key row, 16 rows per pass
  const int vq = tid & 15, vg = tid >> 4;          // V: d quad, group of 4 consecutive keys
  f32x4 kr[4], vr[4];
  bf16x4 krh[4], vrh[4];
  const float* kbase = base + (int64_t)a.H * AT_D;        // K plane
  const float* vbase = base + (int64_t)2 * a.H * AT_D;    // V plane
  const __bf16* kbase_h = HIN ? a.qkv_h + (int64_t)b * a.N * tok_stride + (int64_t)(a.H + h) * AT_D : nullptr;
  const __bf16* vbase_h = HIN ? a.qkv_h + (int64_t)b * a.N * tok_stride + (int64_t)(2 * a.H + h) * AT_D : nullptr;
  auto load_block = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(k0 + krow + 16 * i, a.N - 1);
      const int vkey = min(k0 + 4 * vg + i, a.N - 1);
      if constexpr (HIN) {
        krh[i] = *reinterpret_cast<const bf16x4*>(kbase_h + (int64_t)key * tok_stride + 4 * kq);
        vrh[i] = *reinterpret_cast<const bf16x4*>(vbase_h + (int64_t)vkey * tok_stride + 4 * vq);
      } else {
        kr[i] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * tok_stride + 4 * kq);
        vr[i] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)vkey * tok_stride + 4 * vq);
      }
    }
  };
  auto store_block = [&](int buf) {
    char* ks = Ks0 + buf * (AT_KB * AT_KS);
    char* vs = Vs0 + buf * (AT_D * AT_VS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (HIN)
        *reinterpret_cast<bf16x4*>(ks + (krow + 16 * i) * AT_KS + kq * 8) = krh[i];
      else
        *reinterpret_cast<bf16x4*>(ks + (krow + 16 * i) * AT_KS + kq * 8) = __builtin_convertvector(kr[i], bf16x4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (HIN) {
        const bf16x4 t = {vrh[0][e], vrh[1][e], vrh[2][e], vrh[3][e]};   // 4 consecutive keys of channel e
        *reinterpret_cast<bf16x4*>(vs + (4 * vq + e) * AT_VS + vg * 8) = t;
      } else {
        const f32x4 t = {vr[0][e], vr[1][e], vr[2][e], vr[3][e]};   // 4 consecutive keys of channel e
        *reinterpret_cast<bf16x4*>(vs + (4 * vq + e) * AT_VS + vg * 8) = __builtin_convertvector(t, bf16x4);
      }
    }
  };

  f32x16 ot[2];     // O^T: [d tile][16 rows of d] x query column
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nblk = (a.N + AT_KB - 1) / AT_KB;
  load_block(0);
  store_block(0);
  __syncthreads();
  for (int kb = 0; kb < nblk; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nblk;
    if (more) load_block((kb + 1) * AT_KB);
    const char* ks = Ks0 + cur * (AT_KB * AT_KS);
    const char* vs = Vs0 + cur * (AT_D * AT_VS);
    // S^T tiles: rows = keys (32 per tile), column = this lane's query
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (32 * kt + l31) * AT_KS + (2 * s + lhi) * 16);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
    }
    // online softmax (base 2); keys past N are masked out
    const int kbase_i = kb * AT_KB;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase_i + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (key >= a.N) st[kt][r] = -INFINITY;
        mx = fmaxf(mx, st[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = exp2f(st[kt][r] - m_new);
        ps += p[r];
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const f32x4 lo = {p[8 * hs + 0], p[8 * hs + 1], p[8 * hs + 2], p[8 * hs + 3]};
        const f32x4 hi = {p[8 * hs + 4], p[8 * hs + 5], p[8 * hs + 6], p[8 * hs + 7]};
        const bf16x4 bl = __builtin_convertvector(lo, bf16x4), bh = __builtin_convertvector(hi, bf16x4);
        pf[kt][hs] = bf16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
      }
    }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
    // O^T += V^T P^T : A = V^T rows (d), keys in the order the probabilities sit in pf
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int hs = 0; hs < 2; ++hs)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* vrow = vs + (32 * t + l31) * AT_VS + (32 * kt + 16 * hs + 4 * lhi) * 2;
          const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow);
          const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + 16);
          const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][hs], ot[t], 0, 0, 0);
        }
    if (more) store_block(cur ^ 1);
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (a.lse && lhi == 0 && q0 + l31 < a.N)
    a.lse[((int64_t)b * a.H + h) * a.N + q0 + l31] = m_run + log2f(l_tot);
  // transpose through LDS (each wave its own [32 queries][64 d] patch, 68-float rows), then
  // 256-byte row stores
  float* stage = reinterpret_cast<float*>(smem) + wid * (AT_QW * 68);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      stage[l31 * 68 + dd] = ot[t][r] * inv;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;          // 32 rows x 16 quads
    const int row = idx >> 4, qd = idx & 15;
    const int q = q0 + row;
    if (q < a.N) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 68 + 4 * qd);
      if (a.out_half) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4*>(a.out_half + ((int64_t)b * a.N + q) * (a.H * AT_D) + h * AT_D + 4 * qd) =
            __builtin_convertvector(v, bf16x4);
      } else {
        *reinterpret_cast<f32x4*>(a.out + ((int64_t)b * a.N + q) * (a.H * AT_D) + h * AT_D + 4 * qd) = v;
      }
    }
  }
}

}  // namespace

extern "C" int snap_layer_norm_f32(const float* x, const float* gamma, const float* beta, float* y,
                                   int64_t M, int32_t C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0 || C > 256 * LN_MAXQ) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(gamma) & 15) || (reinterpret_cast<uintptr_t>(beta) & 15))
    return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(layer_norm_kernel<false>, dim3((unsigned)snap_cdiv(M, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, gamma, beta, y, M, C, eps, (__bf16*)nullptr);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_layer_norm_bf16out_f32(const float* x, const float* gamma, const float* beta, void* y_bf16,
                                           int64_t M, int32_t C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y_bf16) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0 || C > 256 * LN_MAXQ) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y_bf16) & 7) ||
      (reinterpret_cast<uintptr_t>(gamma) & 15) || (reinterpret_cast<uintptr_t>(beta) & 15))
    return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(layer_norm_kernel<true>, dim3((unsigned)snap_cdiv(M, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, gamma, beta, (float*)nullptr, M, C, eps,
                     static_cast<__bf16*>(y_bf16));
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_attention_bf16_f32(const float* qkv, float* out, int32_t B, int32_t N,
                                       int32_t H, int32_t D, float scale, void* stream) {
  return snap_attention_lse_bf16_f32(qkv, out, nullptr, B, N, H, D, scale, stream);
}

static int attention_launch(const float* qkv, float* out, void* out_bf16, float* lse, int32_t B, int32_t N, int32_t H,
                            int32_t D, float scale, void* stream);

extern "C" int snap_attention_lse_bf16_f32(const float* qkv, float* out, float* lse, int32_t B,
                                           int32_t N, int32_t H, int32_t D, float scale,
                                           void* stream) {
  if (!out) return SNAP_ERR_NULL;
  return attention_launch(qkv, out, nullptr, lse, B, N, H, D, scale, stream);
}

extern "C" int snap_attention_bf16out_f32(const float* qkv, void* out_bf16, int32_t B, int32_t N, int32_t H,
                                          int32_t D, float scale, void* stream) {
  if (!out_bf16 || (reinterpret_cast<uintptr_t>(out_bf16) & 7)) return out_bf16 ? SNAP_ERR_BAD_SHAPE : SNAP_ERR_NULL;
  return attention_launch(qkv, nullptr, out_bf16, nullptr, B, N, H, D, scale, stream);
}

extern "C" int snap_attention_bf16io(const void* qkv_bf16, void* out_bf16, int32_t B, int32_t N, int32_t H,
                                     int32_t D, float scale, void* stream) {
  if (!qkv_bf16 || !out_bf16) return SNAP_ERR_NULL;
  if (B <= 0 || N <= 0 || H <= 0 || B > 65535 || H > 65535) return SNAP_ERR_BAD_SHAPE;
  if (D != AT_D) return SNAP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv_bf16) & 15) || (reinterpret_cast<uintptr_t>(out_bf16) & 7)) return SNAP_ERR_BAD_SHAPE;
  AttnArgs a;
  a.qkv = nullptr; a.qkv_h = static_cast<const __bf16*>(qkv_bf16); a.out = nullptr;
  a.out_half = static_cast<__bf16*>(out_bf16); a.lse = nullptr; a.B = B; a.N = N; a.H = H;
  a.scale_log2e = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)snap_cdiv(N, 4 * AT_QW), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

static int attention_launch(const float* qkv, float* out, void* out_bf16, float* lse, int32_t B, int32_t N, int32_t H,
                            int32_t D, float scale, void* stream) {
  if (!qkv || (!out && !out_bf16)) return SNAP_ERR_NULL;
  if (B <= 0 || N <= 0 || H <= 0) return SNAP_ERR_BAD_SHAPE;
  if (D != AT_D) return SNAP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return SNAP_ERR_BAD_SHAPE;
  if (B > 65535 || H > 65535) return SNAP_ERR_BAD_SHAPE;
  AttnArgs a;
  a.qkv = qkv; a.qkv_h = nullptr; a.out = out; a.out_half = static_cast<__bf16*>(out_bf16); a.lse = lse; a.B = B; a.N = N; a.H = H;
  a.scale_log2e = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)snap_cdiv(N, 4 * AT_QW), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
