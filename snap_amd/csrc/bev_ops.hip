// BEV plane kernels (HBM-bound):
//   * vertical pooling of the voxel volume into a BEV plane
//       snap/models/bev_mapper.py:56-88 (max / sum / mean with validity masks)
//   * modality fusion (same masked reduce over stacked planes) + matching head
//       snap/models/bev_mapper.py:225-252, :284-291 and layers.normalize
//       (snap/models/layers.py:45-52), fused in one pass over the planes.
// A 32-lane half-wave owns one BEV cell; lane q holds channels 4q..4q+3 (one
// 512-byte coalesced row per voxel for D = 128); per-column reductions run in
// registers, the L2-norm over the 32 matching channels uses xor-shuffles.
#include "common.h"

namespace {

constexpr int MAXQ = 2;  // up to D = 256 channels (MAXQ * 32 lanes * 4)

__global__ __launch_bounds__(256) void vertical_pool_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, float* __restrict__ plane,
    uint8_t* __restrict__ pvalid, int64_t M, int Z, int D, int pooling) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  const int nq = D >> 2;
  const uint8_t* vv = vvalid + m * Z;
  const float* base = vol + m * Z * D;
  f32x4 acc[MAXQ];
  const float init = pooling == SNAP_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{init, init, init, init};
  int count = 0;
  for (int z = 0; z < Z; ++z) {
    if (!vv[z]) continue;  // half-wave uniform
    ++count;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      const int q = hl + 32 * i;
      if (q < nq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)z * D + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[i][e] = pooling == SNAP_POOL_MAX ? snap_max_nan(acc[i][e], v[e]) : acc[i][e] + v[e];
      }
    }
  }
  const bool any = count > 0;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int q = hl + 32 * i;
    if (q < nq) {
      f32x4 o = acc[i];
      if (!any) o = f32x4{0.f, 0.f, 0.f, 0.f};
      if (any && pooling == SNAP_POOL_MEAN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] / (float)count;
      }
      *reinterpret_cast<f32x4*>(plane + m * D + 4 * q) = o;
    }
  }
  if (hl == 0) pvalid[m] = any ? 1 : 0;
}

// Z <= 64: ONE WAVE per column.  The validity bytes of the column become a wave-uniform 64-bit
// mask (one ballot), so the walk over the valid levels has no data-dependent branches; the two
// half-waves take alternate valid levels and each keeps two row loads in flight (four 512-byte
// rows per wave instead of one per half-wave), then the halves are combined.  max: identical
// values; sum / mean: the same terms, associated as (even levels) + (odd levels).
__global__ __launch_bounds__(256) void vertical_pool_wave_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, float* __restrict__ plane,
    uint8_t* __restrict__ pvalid, int64_t M, int Z, int D, int pooling) {
  const int lane = threadIdx.x & 63;
  const int hl = lane & 31, half = lane >> 5;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int nq = D >> 2;
  unsigned long long mask = __ballot(lane < Z && vvalid[m * Z + lane] != 0);
  const int count = __popcll(mask);
  const float* base = vol + m * Z * D;
  const bool is_max = pooling == SNAP_POOL_MAX;
  f32x4 acc[MAXQ];
  const float init = is_max ? -INFINITY : 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{init, init, init, init};
  while (mask) {
    // next four valid levels: z[0], z[2] -> half 0; z[1], z[3] -> half 1
    int z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = mask ? (int)__builtin_ctzll(mask) : -1;
      mask &= mask - 1;                    // (0 stays 0)
    }
    const int za = half ? z[1] : z[0], zb = half ? z[3] : z[2];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      const int q = hl + 32 * i;
      if (q < nq) {
        f32x4 va = f32x4{init, init, init, init}, vb = va;
        if (za >= 0) va = *reinterpret_cast<const f32x4*>(base + (int64_t)za * D + 4 * q);
        if (zb >= 0) vb = *reinterpret_cast<const f32x4*>(base + (int64_t)zb * D + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[i][e] = is_max ? snap_max_nan(acc[i][e], va[e]) : acc[i][e] + va[e];
          acc[i][e] = is_max ? snap_max_nan(acc[i][e], vb[e]) : acc[i][e] + vb[e];
        }
      }
    }
  }
  const bool any = count > 0;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int q = hl + 32 * i;
    f32x4 o = acc[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float other = __shfl_xor(o[e], 32);
      o[e] = is_max ? snap_max_nan(o[e], other) : o[e] + other;
    }
    if (q < nq && half == 0) {
      if (!any) o = f32x4{0.f, 0.f, 0.f, 0.f};
      if (any && pooling == SNAP_POOL_MEAN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] / (float)count;
      }
      *reinterpret_cast<f32x4*>(plane + m * D + 4 * q) = o;
    }
  }
  if (lane == 0) pvalid[m] = any ? 1 : 0;
}

// Max pooling of a training step (Z <= 64, D <= 128: one channel quad per lane): the plane of the kernel
// above, bit for bit, plus the level of every maximum (first of equals) and the number of levels that
// hold it -- what the VJP needs instead of a pass over the volume (vertical_pool_max_bwd_arg_kernel).
// The tracked maximum follows that VJP's comparisons (`>` then `==` over the valid levels in ascending
// order: a NaN never wins), not snap_max_nan; count << 8 | level travel in one register.
__global__ __launch_bounds__(256) void vertical_pool_max_arg_kernel(
    const float* __restrict__ vol, const uint8_t* __restrict__ vvalid, float* __restrict__ plane,
    uint8_t* __restrict__ pvalid, uint8_t* __restrict__ argz, uint8_t* __restrict__ ties, int64_t M, int Z,
    int D) {
  const int lane = threadIdx.x & 63;
  const int hl = lane & 31, half = lane >> 5;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const bool on = 4 * hl < D;
  unsigned long long mask = __ballot(lane < Z && vvalid[m * Z + lane] != 0);
  const bool any = mask != 0;
  const float* base = vol + m * Z * D + 4 * (on ? hl : 0);
  f32x4 acc = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, tb = acc;
  int tk[4] = {255, 255, 255, 255};
#pragma unroll 1
  while (mask) {
    int z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = mask ? (int)__builtin_ctzll(mask) : -1;
      mask &= mask - 1;
    }
    const int za = half ? z[1] : z[0], zb = half ? z[3] : z[2];
    f32x4 va = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, vb = va;
    if (za >= 0) va = *reinterpret_cast<const f32x4*>(base + (int64_t)za * D);
    if (zb >= 0) vb = *reinterpret_cast<const f32x4*>(base + (int64_t)zb * D);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] = snap_max_nan(snap_max_nan(acc[e], va[e]), vb[e]);
      if (za >= 0) {
        // (a first -inf EQUALS the initial bound: it becomes the recorded level, as the recompute path of the
        //  VJP shares the gradient among the -inf holders)
        tk[e] = va[e] > tb[e] ? (256 | za) : (va[e] == tb[e] ? ((tk[e] & 255) == 255 ? (256 | za) : tk[e] + 256) : tk[e]);
        tb[e] = va[e] > tb[e] ? va[e] : tb[e];
      }
      if (zb >= 0) {
        tk[e] = vb[e] > tb[e] ? (256 | zb) : (vb[e] == tb[e] ? ((tk[e] & 255) == 255 ? (256 | zb) : tk[e] + 256) : tk[e]);
        tb[e] = vb[e] > tb[e] ? vb[e] : tb[e];
      }
    }
  }
  unsigned pa = 0, pc = 0;
  f32x4 o = acc;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = snap_max_nan(o[e], __shfl_xor(o[e], 32));
    const float ob = __shfl_xor(tb[e], 32);
    const int ok = __shfl_xor(tk[e], 32);
    int a_ = tk[e] & 255, c_ = tk[e] >> 8;
    if (ob > tb[e]) { a_ = ok & 255; c_ = ok >> 8; }
    else if (ob == tb[e]) { a_ = min(a_, ok & 255); c_ += ok >> 8; }
    pa |= (unsigned)(a_ & 255) << (8 * e);
    pc |= (unsigned)min(c_, 255) << (8 * e);
  }
  if (on && half == 0) {
    if (!any) o = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(plane + m * D + 4 * hl) = o;
    *reinterpret_cast<unsigned*>(argz + m * D + 4 * hl) = pa;
    *reinterpret_cast<unsigned*>(ties + m * D + 4 * hl) = pc;
  }
  if (lane == 0) pvalid[m] = any ? 1 : 0;
}

struct FuseArgs {
  const float* planes[4];
  const uint8_t* valids[4];
  int num_planes;
  int64_t M;
  int D, pooling;
  float* fused;
  uint8_t* fvalid;
  const float* Wm;
  const float* bm;
  int Dm, normalize;
  float eps;
  float* matching;
};

__global__ __launch_bounds__(256) void plane_fuse_match_kernel(const FuseArgs a) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= a.M) return;
  const int nq = a.D >> 2;
  f32x4 acc[MAXQ];
  const float init = a.pooling == SNAP_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{init, init, init, init};
  int count = 0;
  for (int p = 0; p < a.num_planes; ++p) {
    const bool v = a.valids[p] ? (a.valids[p][m] != 0) : true;
    if (!v) continue;
    ++count;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      const int q = hl + 32 * i;
      if (q < nq) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a.planes[p] + m * a.D + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[i][e] = a.pooling == SNAP_POOL_MAX ? snap_max_nan(acc[i][e], x[e]) : acc[i][e] + x[e];
      }
    }
  }
  const bool any = count > 0;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    if (!any) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (any && a.pooling == SNAP_POOL_MEAN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e] = acc[i][e] / (float)count;
    }
    const int q = hl + 32 * i;
    if (q < nq && a.fused) *reinterpret_cast<f32x4*>(a.fused + m * a.D + 4 * q) = acc[i];
  }
  if (hl == 0 && a.fvalid) a.fvalid[m] = any ? 1 : 0;
  if (!a.matching) return;

  // matching head: lane j computes output channel j (Dm <= 32).
  const int j = hl;
  float y = (j < a.Dm) ? a.bm[j] : 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    for (int q = 0; q < 32; ++q) {
      const int cq = q + 32 * i;
      if (cq >= nq) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xv = __shfl(acc[i][e], q, 32);
        if (j < a.Dm) y += xv * a.Wm[(int64_t)(4 * cq + e) * a.Dm + j];
      }
    }
  }
  if (a.normalize) {
    float ss = (j < a.Dm) ? y * y : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 32);
    const float nrm = sqrtf(ss);
    const bool invalid = nrm < a.eps;
    // layers.normalize: y_safe = where(invalid, eps, x); z = x / ||y_safe||.
    const float denom = invalid ? sqrtf((float)a.Dm) * a.eps : nrm;
    y = invalid ? 0.f : y / denom;
  }
  if (j < a.Dm) a.matching[m * a.Dm + j] = any ? y : 0.f;
}

// The same kernel for D = 128 with a matching head (the bench's planes): the head is a [128] x [128, Dm]
// product per cell, and in the kernel above every one of its 128 terms costs a lane exchange and an L1
// load of the kernel entry.  Here a lane keeps ITS column of Wm in registers for all the cells its
// half-wave visits (a persistent grid), and the cell's fused vector comes back from LDS as 32 broadcast
// 16-byte reads.  Terms added in the same (ascending channel) order: the same bits.
__global__ __launch_bounds__(256) void plane_fuse_match_d128_kernel(const FuseArgs a) {
  __shared__ __attribute__((aligned(16))) float sv[8][128];
  const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int j = hl;
  float wreg[128];
#pragma unroll
  for (int k = 0; k < 128; ++k) wreg[k] = j < a.Dm ? a.Wm[(int64_t)k * a.Dm + j] : 0.f;
  const float bj = j < a.Dm ? a.bm[j] : 0.f;
  const float init = a.pooling == SNAP_POOL_MAX ? -INFINITY : 0.f;
  for (int64_t m = (int64_t)blockIdx.x * 8 + hw; m < a.M; m += (int64_t)gridDim.x * 8) {
    // every plane's validity byte and row are requested before the first is looked at (one memory round
    // trip per cell instead of two per plane); a row of an invalid plane is fetched and dropped
    f32x4 acc = {init, init, init, init};
    int count = 0;
    bool pv[4];
    f32x4 px[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      pv[p] = false;
      px[p] = acc;
      if (p < a.num_planes) {
        pv[p] = a.valids[p] ? (a.valids[p][m] != 0) : true;
        px[p] = *reinterpret_cast<const f32x4*>(a.planes[p] + m * 128 + 4 * hl);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!pv[p]) continue;
      ++count;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[e] = a.pooling == SNAP_POOL_MAX ? snap_max_nan(acc[e], px[p][e]) : acc[e] + px[p][e];
    }
    const bool any = count > 0;
    if (!any) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (any && a.pooling == SNAP_POOL_MEAN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] / (float)count;
    }
    if (a.fused) *reinterpret_cast<f32x4*>(a.fused + m * 128 + 4 * hl) = acc;
    if (hl == 0 && a.fvalid) a.fvalid[m] = any ? 1 : 0;
    *reinterpret_cast<f32x4*>(&sv[hw][4 * hl]) = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float y = bj;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(&sv[hw][4 * q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) y += xv[e] * wreg[4 * q + e];
    }
    __builtin_amdgcn_wave_barrier();       // (the row is rewritten by the next cell)
    if (a.normalize) {
      float ss = (j < a.Dm) ? y * y : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 32);
      const float nrm = sqrtf(ss);
      const bool invalid = nrm < a.eps;
      const float denom = invalid ? sqrtf((float)a.Dm) * a.eps : nrm;
      y = invalid ? 0.f : y / denom;
    }
    if (j < a.Dm) a.matching[m * a.Dm + j] = any ? y : 0.f;
  }
}

}  // namespace

extern "C" int snap_vertical_pool_f32(const float* vol, const uint8_t* vvalid, float* plane,
                                      uint8_t* pvalid, int64_t M, int32_t Z, int32_t D,
                                      int32_t pooling, void* stream) {
  if (!vol || !vvalid || !plane || !pvalid) return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || D <= 0 || D % 4 != 0 || D > MAXQ * 128) return SNAP_ERR_BAD_SHAPE;
  if (pooling < SNAP_POOL_MAX || pooling > SNAP_POOL_MEAN) return SNAP_ERR_UNSUPPORTED;
  if (Z <= 64) {
    hipLaunchKernelGGL(vertical_pool_wave_kernel, dim3((unsigned)snap_cdiv(M, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), vol, vvalid, plane, pvalid, M, Z, D,
                       pooling);
  } else {
    hipLaunchKernelGGL(vertical_pool_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), vol, vvalid, plane, pvalid, M, Z, D,
                       pooling);
  }
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_vertical_pool_max_arg_f32(const float* vol, const uint8_t* vvalid, float* plane,
                                              uint8_t* pvalid, uint8_t* argz, uint8_t* ties, int64_t M,
                                              int32_t Z, int32_t D, void* stream) {
  if (!vol || !vvalid || !plane || !pvalid || !argz || !ties) return SNAP_ERR_NULL;
  if (M <= 0 || Z <= 0 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (Z > 64 || D > 128) return SNAP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(vertical_pool_max_arg_kernel, dim3((unsigned)snap_cdiv(M, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), vol, vvalid, plane, pvalid, argz, ties, M, Z, D);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_plane_fuse_match_f32(const float* const* planes, const uint8_t* const* valids,
                                         int32_t num_planes, int64_t M, int32_t D, int32_t pooling,
                                         float* fused, uint8_t* fvalid, const float* Wm,
                                         const float* bm, int32_t Dm, int32_t normalize, float eps,
                                         float* matching, void* stream) {
  if (!planes) return SNAP_ERR_NULL;
  if (num_planes < 1 || num_planes > 4) return SNAP_ERR_UNSUPPORTED;
  if (M <= 0 || D <= 0 || D % 4 != 0 || D > MAXQ * 128) return SNAP_ERR_BAD_SHAPE;
  if (pooling < SNAP_POOL_MAX || pooling > SNAP_POOL_MEAN) return SNAP_ERR_UNSUPPORTED;
  if (matching && (!Wm || !bm)) return SNAP_ERR_NULL;
  if (matching && (Dm < 1 || Dm > 32)) return SNAP_ERR_UNSUPPORTED;
  FuseArgs a;
  for (int i = 0; i < 4; ++i) {
    a.planes[i] = i < num_planes ? planes[i] : nullptr;
    a.valids[i] = (i < num_planes && valids) ? valids[i] : nullptr;
    if (i < num_planes && !a.planes[i]) return SNAP_ERR_NULL;
  }
  a.num_planes = num_planes;
  a.M = M; a.D = D; a.pooling = pooling;
  a.fused = fused; a.fvalid = fvalid;
  a.Wm = Wm; a.bm = bm; a.Dm = Dm; a.normalize = normalize; a.eps = eps;
  a.matching = matching;
  if (matching && D == 128 && M >= 8192)
    hipLaunchKernelGGL(plane_fuse_match_d128_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(plane_fuse_match_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}


// ---------------------------------------------------------------------------
// Confidence head of the BEV plane (snap/models/bev_mapper.py:154-157,292-295):
//   bev_confidence = where(valid, log_sigmoid(Dense(1)(features)), 0)
// One half-wave per cell (float4 per lane, D <= 128 ... any multiple of 4 by striding),
// log_sigmoid(s) = min(s, 0) - log1p(exp(-|s|))  (== -softplus(-s), stable).
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void confidence_head_kernel(
    const float* __restrict__ f, const uint8_t* __restrict__ valid, const float* __restrict__ w,
    const float* __restrict__ bias, int64_t M, int D, float* __restrict__ out) {
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M) return;
  float acc = 0.f;
  for (int c = 4 * hl; c < D; c += 128) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(f + m * D + c);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
    acc += ((x[0] * ww[0] + x[1] * ww[1]) + x[2] * ww[2]) + x[3] * ww[3];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 32);
  if (hl == 0) {
    const float s = acc + bias[0];
    const float ls = fminf(s, 0.f) - log1pf(expf(-fabsf(s)));
    out[m] = (valid == nullptr || valid[m]) ? ls : 0.f;
  }
}
}  // namespace

extern "C" int snap_confidence_head_f32(const float* features, const uint8_t* valid, const float* w,
                                        const float* bias, int64_t M, int32_t D, float* out, void* stream) {
  if (!features || !w || !bias || !out) return SNAP_ERR_NULL;
  if (M <= 0 || D <= 0 || D % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(features) | reinterpret_cast<uintptr_t>(w)) & 15) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(confidence_head_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), features, valid, w, bias, M, D, out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

// Voxel-centre query points (bev_mapper.py:162-196): xyz[b, x, y, k] = (xy[b or 0, x, y, :], z[b, k]).
// The small factors (the BEV cell centres, the per-scene level heights) are computed by the host
// module exactly as the reference does; this is the broadcast into [B, X, Y, Z, 3] in one pass.
namespace {
__global__ __launch_bounds__(256) void voxel_points_kernel(const float* __restrict__ xy, int64_t xy_bstride,
                                                           const float* __restrict__ z, int64_t total,
                                                           int XY, int Z, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // voxel (b, xy, k), level fastest
  if (i >= total) return;
  const int k = (int)(i % Z);
  const int64_t t = i / Z;
  const int c = (int)(t % XY);
  const int64_t b = t / XY;
  const float* p = xy + b * xy_bstride + (int64_t)c * 2;
  float* o = out + i * 3;
  o[0] = p[0];
  o[1] = p[1];
  o[2] = z[b * Z + k];
}
}  // namespace

extern "C" int snap_voxel_points_f32(const float* xy, int32_t xy_batched, const float* z, int32_t B,
                                     int32_t XY, int32_t Z, float* out, void* stream) {
  if (!xy || !z || !out) return SNAP_ERR_NULL;
  if (B <= 0 || XY <= 0 || Z <= 0) return SNAP_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)B * XY * Z;
  hipLaunchKernelGGL(voxel_points_kernel, dim3((unsigned)snap_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), xy, xy_batched ? (int64_t)XY * 2 : 0, z, total, XY, Z,
                     out);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

