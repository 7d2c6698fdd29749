// Exhaustive (x, y, theta) voting: rotated query templates, edge-padded map and
// the -inf / normalisation pass around the direct correlation, which itself runs
// on the MFMA conv engine (conv_igemm.hip) with the templates as an
// (H x W x D) -> R filter bank.
//
// Replaces snap/models/pose_exhaustive_voting.py:37-69 (sample_query_templates)
// and the non-GEMM parts of :72-104 (template_matching).
#include "common.h"
#include "rotate_sample.h"

namespace {

// thread <-> (r0, i, j, channel quad); r0 < R/4 (first quadrant of rotations);
// the other three quadrants are written as rot90 copies, exactly as the
// reference completes them with jnp.rot90(quarter, k, axes=(2, 1)).
__global__ void rotate_templates_kernel(const float* __restrict__ feat,
                                        const uint8_t* __restrict__ valid,
                                        const float* __restrict__ tfm, int H, int W, int D, int R,
                                        int Rp, float cell, float* __restrict__ templates,
                                        uint8_t* __restrict__ tvalid, float* __restrict__ tw,
                                        float* __restrict__ cw, float* __restrict__ tcount) {
  const int D4 = D >> 2;
  const int RQ = R >> 2;
  const int64_t total = (int64_t)RQ * H * W * D4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % D4);
  int64_t r = idx / D4;
  const int sj = (int)(r % W); r /= W;
  const int si = (int)(r % H);
  const int r0 = (int)(r / H);
  const SnapRotSample rs = snap_rot_sample(tfm + r0 * 4, si, sj, H, W, cell, valid);
  const bool ok = rs.ok;
  const int i0 = rs.i0, i1 = rs.i1, j0 = rs.j0, j1 = rs.j1;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const f32x4 a00 = *reinterpret_cast<const f32x4*>(feat + ((int64_t)i0 * W + j0) * D + 4 * q);
    const f32x4 a01 = *reinterpret_cast<const f32x4*>(feat + ((int64_t)i0 * W + j1) * D + 4 * q);
    const f32x4 a10 = *reinterpret_cast<const f32x4*>(feat + ((int64_t)i1 * W + j0) * D + 4 * q);
    const f32x4 a11 = *reinterpret_cast<const f32x4*>(feat + ((int64_t)i1 * W + j1) * D + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = snap_rot_mix(rs, a00[e], a01[e], a10[e], a11[e]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // destination of source cell (si, sj) under rot90(., k, axes=(2,1)); H == W.
    int di, dj;
    if (k == 0) { di = si; dj = sj; }
    else if (k == 1) { di = sj; dj = H - 1 - si; }
    else if (k == 2) { di = H - 1 - si; dj = W - 1 - sj; }
    else { di = H - 1 - sj; dj = si; }
    const int rr = k * RQ + r0;
    *reinterpret_cast<f32x4*>(templates + (((int64_t)rr * H + di) * W + dj) * D + 4 * q) = o;
    if (tw) {     // HWIO filter bank of the plain path (r fastest: 4-byte scattered stores)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        tw[(((int64_t)di * W + dj) * D + 4 * q + e) * Rp + rr] = o[e];
    }
    if (q == 0) {
      tvalid[((int64_t)rr * H + di) * W + dj] = ok ? 1 : 0;
      // count filter = 180-degree rotated mask (un-flipped true convolution).
      cw[((int64_t)(H - 1 - di) * W + (W - 1 - dj)) * Rp + rr] = ok ? 1.f : 0.f;
    }
  }
  // tcount[r] = number of valid cells of template r (integer-valued: order independent).  One
  // atomic per WAVE and quadrant instead of one per cell: 36 addresses took 2 M serialised
  // atomics at 256^2 (9 ms).  r0 is wave-uniform except where a wave straddles two rotations.
  const bool mine = (q == 0) && ok;
  const int r0_first = __builtin_amdgcn_readfirstlane(r0);
  const bool uniform = __all(r0 == r0_first);
  if (uniform) {
    const int n = __popcll(__ballot(mine));
    if ((threadIdx.x & 63) == 0 && n > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(tcount + k * RQ + r0_first, (float)n);
    }
  } else if (mine) {
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(tcount + k * RQ + r0, 1.f);
  }
}

__global__ void pad_map_kernel(const float* __restrict__ map, const uint8_t* __restrict__ mvalid,
                               int H, int W, int D, float* __restrict__ map_pad,
                               float* __restrict__ mvalid_pad) {
  const int D4 = D >> 2;
  const int Hp = 3 * H - 2, Wp = 3 * W - 2;
  const int64_t total = (int64_t)Hp * Wp * D4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % D4);
  int64_t r = idx / D4;
  const int b = (int)(r % Wp);
  const int a = (int)(r / Wp);
  const int si = a - (H - 1), sj = b - (W - 1);
  const int ci = min(max(si, 0), H - 1), cj = min(max(sj, 0), W - 1);
  reinterpret_cast<f32x4*>(map_pad)[idx] =
      *reinterpret_cast<const f32x4*>(map + ((int64_t)ci * W + cj) * D + 4 * q);
  if (q == 0) {
    const bool inside = si >= 0 && si < H && sj >= 0 && sj < W;
    mvalid_pad[(int64_t)a * Wp + b] = (inside && mvalid[ci * W + cj]) ? 1.f : 0.f;
  }
}

__global__ void template_finalize_kernel(const float* __restrict__ raw,
                                         const float* __restrict__ cnt,
                                         const float* __restrict__ tcount, int Ho, int Wo, int R,
                                         int Rp, float thr, int use_overlap,
                                         float* __restrict__ scores) {
  const int64_t total = (int64_t)R * Ho * Wo;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t ab = idx % ((int64_t)Ho * Wo);
  const int r = (int)(idx / ((int64_t)Ho * Wo));
  float v = raw[ab * Rp + r];
  if (use_overlap && !(cnt[ab * Rp + r] > thr)) v = -INFINITY;
  scores[idx] = v / tcount[r];
}

}  // namespace

extern "C" int snap_rotate_templates_f32(const float* feat, const uint8_t* valid,
                                         const float* tfm, int32_t H, int32_t W, int32_t D,
                                         int32_t R, float cell_size, float* templates,
                                         uint8_t* tvalid, float* tw, float* cw, float* tcount,
                                         void* stream) {
  if (!feat || !valid || !tfm || !templates || !tvalid || !cw || !tcount)   // (tw may be NULL)
    return SNAP_ERR_NULL;
  if (H <= 0 || H != W || D <= 0 || D % 4 != 0 || R <= 0 || R % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int Rp = R;  // R % 4 == 0 already
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(tcount, 0, sizeof(float) * R, s) != hipSuccess) return SNAP_ERR_LAUNCH;
  const int64_t total = (int64_t)(R / 4) * H * W * (D / 4);
  hipLaunchKernelGGL(rotate_templates_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     s, feat, valid, tfm, H, W, D, R, Rp, cell_size, templates, tvalid, tw, cw,
                     tcount);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// Shift-stacked filter bank: tws[i', j', d, (r*S + sa)*S + sb] = tw[i'-sa, j'-sb, d, r] (0 outside).
// A stride-S correlation with it yields, per output pixel (a4, b4), the S x S block of direct-form
// outputs (S a4 + sa, S b4 + sb) of every template: the GEMM's N grows from R (36: 56 % of a
// 64-wide tile) to R S^2 (576 = 9 full tiles), M shrinks by S^2, K grows by ((H+S-1)(W+S-1))/(HW).
__global__ __launch_bounds__(256) void stack_templates_kernel(
    const float* __restrict__ tw, float* __restrict__ tws, int H, int W, int D, int R, int S) {
  const int RS = R * S * S;
  const int64_t total = (int64_t)(H + S - 1) * (W + S - 1) * D * RS;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % RS);
  int64_t t = i / RS;
  const int d = (int)(t % D); t /= D;
  const int jp = (int)(t % (W + S - 1));
  const int ip = (int)(t / (W + S - 1));
  const int sb = n % S, sa = (n / S) % S, r = n / (S * S);
  const int ii = ip - sa, jj = jp - sb;
  tws[i] = (ii >= 0 && ii < H && jj >= 0 && jj < W)
               ? tw[(((int64_t)ii * W + jj) * D + d) * R + r] : 0.f;
}

// the same from the [R, H, W, D] template tensor (what rotate_templates writes coalesced), so
// the large-map path never materialises the r-fastest HWIO bank
__global__ __launch_bounds__(256) void stack_templates_rhwd_kernel(
    const float* __restrict__ templates, float* __restrict__ tws, int H, int W, int D, int R, int S) {
  const int RS = R * S * S;
  const int64_t total = (int64_t)(H + S - 1) * (W + S - 1) * D * RS;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % RS);
  int64_t t = i / RS;
  const int d = (int)(t % D); t /= D;
  const int jp = (int)(t % (W + S - 1));
  const int ip = (int)(t / (W + S - 1));
  const int sb = n % S, sa = (n / S) % S, r = n / (S * S);
  const int ii = ip - sa, jj = jp - sb;
  tws[i] = (ii >= 0 && ii < H && jj >= 0 && jj < W)
               ? templates[(((int64_t)r * H + ii) * W + jj) * D + d] : 0.f;
}

extern "C" int snap_stack_templates_rhwd_f32(const float* templates, float* tws, int32_t H,
                                             int32_t W, int32_t D, int32_t R, int32_t S,
                                             void* stream) {
  if (!templates || !tws) return SNAP_ERR_NULL;
  if (H <= 0 || W <= 0 || D <= 0 || R <= 0 || S < 1 || S > 8) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)(H + S - 1) * (W + S - 1) * D * R * S * S;
  if (snap_cdiv(total, 256) > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(stack_templates_rhwd_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), templates, tws, H, W, D, R, S);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// The shift-stacked bank written DIRECTLY as the split engine's two-part weight image
//   out[column tile of 128][tap (i', j')][channel tile of 16][part][column 0..127][16 k] bf16
// (conv_split.hip's layout, octet swizzle applied at rest, columns >= R S^2 zero): the f32 bank
// [H+S-1, W+S-1, D, R S^2] (4.9 GB at C4) is never materialised and the separate pack pass (read
// 4.9 GB, write 4.9 GB) disappears -- one pass that reads the 0.3 GB of templates and writes the
// image.  A thread = one 8-channel octet of one (tap, column): 32 B in, 2 x 16 B out.
__global__ __launch_bounds__(256) void pack_stacked_templates_split_kernel(
    const float* __restrict__ templates, __bf16* __restrict__ out, int H, int W, int D, int R, int S,
    int ctiles, int ncolpad, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int op = (int)(i & 1);                       // physical octet within the column's 16 k
  int64_t t = i >> 1;
  const int n = (int)(t % ncolpad); t /= ncolpad;    // output column (filter) incl. padding
  const int ct = (int)(t % ctiles);
  const int64_t tap = t / ctiles;                    // i' * (W + S - 1) + j'
  const int KW = W + S - 1;
  const int ip = (int)(tap / KW), jp = (int)(tap - (int64_t)ip * KW);
  const int col = n & 127;
  const int oct = op ^ ((col >> 3) & 1);             // logical octet stored in this slot
  const int d0 = 16 * ct + 8 * oct;
  const int RS = R * S * S;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (n < RS) {
    const int sb = n % S, sa = (n / S) % S, r = n / (S * S);
    const int ii = ip - sa, jj = jp - sb;
    if (ii >= 0 && ii < H && jj >= 0 && jj < W) {
      const float* src = templates + (((int64_t)r * H + ii) * W + jj) * D + d0;
      if (d0 + 8 <= D) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = d0 + e < D ? src[e] : 0.f;
      }
    }
  }
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
  const int64_t blk = (((int64_t)(n >> 7) * ((int64_t)(H + S - 1) * KW) + tap) * ctiles + ct);
  __bf16* o = out + blk * (2 * 2048) + col * 16 + op * 8;
  *reinterpret_cast<bf16x8*>(o) = hi;
  *reinterpret_cast<bf16x8*>(o + 2048) = lo;
}

extern "C" int snap_pack_stacked_templates_split_bf16(const float* templates, int32_t H, int32_t W,
                                                      int32_t D, int32_t R, int32_t S, void* out,
                                                      size_t out_bytes, void* stream) {
  if (!templates || !out) return SNAP_ERR_NULL;
  if (H <= 0 || W <= 0 || D <= 0 || R <= 0 || S < 1 || S > 8 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t taps = (int64_t)(H + S - 1) * (W + S - 1);
  const int RS = R * S * S;
  if (taps > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  const size_t need = snap_conv2d_packed_weights_split_bytes((int32_t)taps, D, RS, 2);
  if (need == 0) return SNAP_ERR_UNSUPPORTED;
  if (out_bytes < need) return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(templates)) & 15) return SNAP_ERR_BAD_SHAPE;
  const int ctiles = (D + 15) / 16;
  const int ncolpad = (RS + 127) / 128 * 128;
  const int64_t total = taps * ctiles * ncolpad * 2;
  if (snap_cdiv(total, 256) > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pack_stacked_templates_split_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), templates, static_cast<__bf16*>(out), H, W, D, R, S,
                     ctiles, ncolpad, total);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_stack_templates_f32(const float* tw, float* tws, int32_t H, int32_t W,
                                        int32_t D, int32_t R, int32_t S, void* stream) {
  if (!tw || !tws) return SNAP_ERR_NULL;
  if (H <= 0 || W <= 0 || D <= 0 || R <= 0 || S < 1 || S > 8) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)(H + S - 1) * (W + S - 1) * D * R * S * S;
  if (snap_cdiv(total, 256) > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(stack_templates_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), tw, tws, H, W, D, R, S);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_pad_map_f32(const float* map, const uint8_t* mvalid, int32_t H, int32_t W,
                                int32_t D, float* map_pad, float* mvalid_pad, void* stream) {
  if (!map || !mvalid || !map_pad || !mvalid_pad) return SNAP_ERR_NULL;
  if (H <= 0 || W <= 0 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)(3 * H - 2) * (3 * W - 2) * (D / 4);
  hipLaunchKernelGGL(pad_map_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), map, mvalid, H, W, D, map_pad, mvalid_pad);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_template_finalize_f32(const float* raw, const float* cnt, const float* tcount,
                                          int32_t Ho, int32_t Wo, int32_t R, int32_t Rp,
                                          float overlap_threshold, int32_t use_overlap,
                                          float* scores, void* stream) {
  if (!raw || !tcount || !scores) return SNAP_ERR_NULL;
  if (use_overlap && !cnt) return SNAP_ERR_NULL;
  if (Ho <= 0 || Wo <= 0 || R <= 0 || Rp < R) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)R * Ho * Wo;
  hipLaunchKernelGGL(template_finalize_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), raw, cnt, tcount, Ho, Wo, R, Rp,
                     overlap_threshold, use_overlap, scores);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
