// Frequency-domain exhaustive voting: the gfx950 kernels around voting_fft_body.h and the C ABI.
//
// Replaces snap/models/pose_exhaustive_voting.py:72-104 (template_matching, padded mode) as ONE
// entry point: templates [R, H, W, D] + validity, map [Hm, Wm, D] + validity -> scores
// [R, 3 Hm - 1 - H, 3 Wm - 1 - W], finalised (-inf where the overlap count fails, / tcount).
// See voting_fft_body.h for the formulation; DESIGN.md section 5m for traffic and timings.
#include "common.h"
#include "voting_fft_body.h"

namespace {

constexpr int kNT = 64 * vfft::kCols;     // 512 threads at 8 columns: two workgroups per CU

__global__ __launch_bounds__(256) void vf_twiddle_kernel(float2* tw, int N) {
  vfft::twiddle_body(tw, N, (int)(blockIdx.x * 256 + threadIdx.x));
}

__global__ __launch_bounds__(kNT) void vf_slow_kernel(vfft::SlowArgs a) {
  extern __shared__ __align__(16) unsigned char vf_smem[];
  float2* buf = reinterpret_cast<float2*>(vf_smem);
  vfft::slow_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, (int)blockDim.x, buf,
                  buf + a.pl.N * vfft::kCols);
}

__global__ __launch_bounds__(kNT) void vf_fast_kernel(vfft::FastArgs a) {
  extern __shared__ __align__(16) unsigned char vf_smem[];
  float2* buf = reinterpret_cast<float2*>(vf_smem);
  float2* twl = buf + a.pl.N * vfft::kCols;
  vfft::fast_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, (int)blockDim.x, buf, twl,
                  twl + a.pl.N);
}

__global__ __launch_bounds__(kNT) void vf_inv_kernel(vfft::InvArgs a) {
  extern __shared__ __align__(16) unsigned char vf_smem[];
  float2* buf = reinterpret_cast<float2*>(vf_smem);
  vfft::inv_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, (int)blockDim.x, buf,
                 buf + a.pl.N * vfft::kCols);
}

// template masks + valid-cell counts of a rotated call (one atomic per wave and quadrant: the counts
// are integers, the sum is order independent)
__global__ __launch_bounds__(256) void vf_rot_mask_kernel(vfft::RotSource rs, int H, int W, int R,
                                                          uint8_t* tvalid, float* tcount) {
  const int64_t total = (int64_t)(R >> 2) * H * W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int r0 = 0;
  bool ok = false;
  if (idx < total) ok = vfft::rot_mask_body(rs, H, W, R, idx, tvalid, &r0);
  else r0 = (R >> 2) - 1;
  const int RQ = R >> 2;
  const int r0_first = __builtin_amdgcn_readfirstlane(r0);
  if (__all(r0 == r0_first)) {
    const int n = __popcll(__ballot(ok));
    if ((threadIdx.x & 63) == 0 && n > 0)
      for (int k = 0; k < 4; ++k) atomicAdd(tcount + k * RQ + r0_first, (float)n);
  } else if (ok) {
    for (int k = 0; k < 4; ++k) atomicAdd(tcount + k * RQ + r0, 1.f);
  }
}

// threads per workgroup: one fused radix-16 item (16 elements of one column) per thread
inline int threads_for(int N) {
  const int want = N * vfft::kCols / 16;
  int nt = 64;
  while (nt < want && nt < kNT) nt *= 2;
  return nt;
}

struct HipLaunch {
  hipStream_t s;
  bool set_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
  }
  bool twiddle(float2* tw, int N) {
    hipLaunchKernelGGL(vf_twiddle_kernel, dim3((unsigned)snap_cdiv(N, 256)), dim3(256), 0, s, tw, N);
    return hipGetLastError() == hipSuccess;
  }
  bool slow(const vfft::SlowArgs& a, int gx, int gy) {
    const size_t lds = sizeof(float2) * (size_t)a.pl.N * (vfft::kCols + 1);
    if (!set_lds((const void*)&vf_slow_kernel, lds)) return false;
    hipLaunchKernelGGL(vf_slow_kernel, dim3(gx, gy), dim3(threads_for(a.pl.N)), lds, s, a);
    return hipGetLastError() == hipSuccess;
  }
  bool fast(const vfft::FastArgs& a, int gx, int gy) {
    const size_t lds = sizeof(float2) * (size_t)a.pl.N * (vfft::kCols + 2);
    if (!set_lds((const void*)&vf_fast_kernel, lds)) return false;
    hipLaunchKernelGGL(vf_fast_kernel, dim3(gx, gy), dim3(threads_for(a.pl.N)), lds, s, a);
    return hipGetLastError() == hipSuccess;
  }
  bool inv(const vfft::InvArgs& a, int gx, int gy) {
    const size_t lds = sizeof(float2) * (size_t)a.pl.N * (vfft::kCols + 1);
    if (!set_lds((const void*)&vf_inv_kernel, lds)) return false;
    hipLaunchKernelGGL(vf_inv_kernel, dim3(gx, gy), dim3(threads_for(a.pl.N)), lds, s, a);
    return hipGetLastError() == hipSuccess;
  }
};

}  // namespace

extern "C" size_t snap_voting_fft_workspace_bytes(int32_t R, int32_t H, int32_t W, int32_t D,
                                                  int32_t Hm, int32_t Wm) {
  vfft::Geometry g;
  if (!vfft::make_geometry(R, H, W, D, Hm, Wm, &g)) return 0;
  if ((int64_t)R * g.G > 65535 || g.N1 > 65535) return 0;
  return g.total;
}

extern "C" int snap_voting_fft_f32(const float* templates, const uint8_t* tvalid, const float* map,
                                   const uint8_t* mvalid, const float* tcount, int32_t R, int32_t H,
                                   int32_t W, int32_t D, int32_t Hm, int32_t Wm,
                                   float overlap_threshold, int32_t use_overlap, void* workspace,
                                   size_t workspace_bytes, float* scores, void* stream) {
  if (!templates || !map || !tcount || !workspace || !scores) return SNAP_ERR_NULL;
  if (use_overlap && (!tvalid || !mvalid)) return SNAP_ERR_NULL;
  vfft::Geometry g;
  if (R <= 0 || H <= 0 || W <= 0 || D <= 0 || Hm <= 0 || Wm <= 0) return SNAP_ERR_BAD_SHAPE;
  if (!vfft::make_geometry(R, H, W, D, Hm, Wm, &g)) return SNAP_ERR_UNSUPPORTED;
  if ((int64_t)R * g.G > 65535 || g.N1 > 65535) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < g.total) return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) ||
      ((reinterpret_cast<uintptr_t>(templates) | reinterpret_cast<uintptr_t>(map)) & 7))
    return SNAP_ERR_BAD_SHAPE;
  HipLaunch L{static_cast<hipStream_t>(stream)};
  if (!vfft::run_voting(g, templates, tvalid, map, mvalid, tcount, overlap_threshold, use_overlap ? 1 : 0,
                        static_cast<char*>(workspace), scores, L))
    return SNAP_ERR_LAUNCH;
  return SNAP_OK;
}

// The same with the templates sampled on the fly (snap/models/pose_exhaustive_voting.py:107-124:
// exhaustive_pose_voting = sample_query_templates + template_matching): feat [H, H, D] (the query
// plane, already multiplied by its confidence), valid [H, H], tfm [R / 4, 4] = (cos, sin, tx, ty) of
// templates_t_grid for the first quadrant of rotations, cell_size.  The [R, H, H, D] template tensor
// is never written: the first transform interpolates its rows (rotate_sample.h: the arithmetic of
// snap_rotate_templates_f32, bit for bit), the other three quadrants are rot90 index maps.
extern "C" int snap_voting_fft_rotated_f32(const float* feat, const uint8_t* valid, const float* tfm,
                                           float cell_size, const float* map, const uint8_t* mvalid,
                                           int32_t R, int32_t H, int32_t D, int32_t Hm, int32_t Wm,
                                           float min_overlap, void* workspace, size_t workspace_bytes,
                                           float* scores, void* stream) {
  if (!feat || !valid || !tfm || !map || !mvalid || !workspace || !scores) return SNAP_ERR_NULL;
  if (R <= 0 || R % 4 != 0 || H <= 0 || D <= 0 || Hm <= 0 || Wm <= 0) return SNAP_ERR_BAD_SHAPE;
  vfft::Geometry g;
  if (!vfft::make_geometry(R, H, H, D, Hm, Wm, &g)) return SNAP_ERR_UNSUPPORTED;
  if ((int64_t)R * g.G > 65535 || g.N1 > 65535) return SNAP_ERR_UNSUPPORTED;
  if (workspace_bytes < g.total) return SNAP_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) ||
      ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(map)) & 7))
    return SNAP_ERR_BAD_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  uint8_t* tvalid = reinterpret_cast<uint8_t*>(ws + g.o_tvalid);
  float* tcount = reinterpret_cast<float*>(ws + g.o_tcount);
  if (hipMemsetAsync(tcount, 0, sizeof(float) * R, s) != hipSuccess) return SNAP_ERR_LAUNCH;
  vfft::RotSource rs{feat, valid, tfm, cell_size};
  const int64_t cells = (int64_t)(R / 4) * H * H;
  hipLaunchKernelGGL(vf_rot_mask_kernel, dim3((unsigned)snap_cdiv(cells, 256)), dim3(256), 0, s, rs, H, H, R,
                     tvalid, tcount);
  SNAP_CHECK_LAUNCH();
  HipLaunch L{s};
  if (!vfft::run_voting(g, nullptr, tvalid, map, mvalid, tcount, min_overlap * (float)H * (float)H, 1, ws, scores,
                        L, &rs))
    return SNAP_ERR_LAUNCH;
  return SNAP_OK;
}
