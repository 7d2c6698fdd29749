// Frequency-domain exhaustive voting: the gfx950 kernels around voting_fft_body.h and the C ABI.
//
// Replaces snap/models/pose_exhaustive_voting.py:72-104 (template_matching, padded mode) as ONE
// entry point: templates [R, H, W, D] + validity, map [Hm, Wm, D] + validity -> scores
// [R, 3 Hm - 1 - H, 3 Wm - 1 - W], finalised (-inf where the overlap count fails, / tcount).
// See voting_fft_body.h for the formulation; DESIGN.md section 5m for traffic and timings.
#include "common.h"
#include "voting_fft_body.h"

namespace {

constexpr int kNT = 1024;

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

// threads per workgroup: every thread should own at least one radix-4 butterfly of the 16 columns
inline int threads_for(int N) {
  const int want = N * vfft::kCols / 4;
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
