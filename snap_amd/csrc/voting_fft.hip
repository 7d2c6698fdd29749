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

// Template masks + valid-cell counts of a rotated call: the arithmetic of vfft::rot_mask_body (the CPU emulation's
// form: one cell per call), a 64 x 64 tile of first-quadrant cells per workgroup.  The verdicts of the tile go through
// LDS so that all four rot90 copies leave as 64-byte runs -- the copies k = 1, 3 run along si: written by the thread
// that computed the cell they were one byte per 256-byte row -- and the tile's count is ONE atomic per quadrant (the
// per-wave form sent 37 k atomics to 36 addresses).  Counts are integers: the sums are order independent.
constexpr int kMaskTile = 64;
__global__ __launch_bounds__(256) void vf_rot_mask_kernel(vfft::RotSource rs, int H, int R, uint8_t* tvalid,
                                                          float* tcount) {
  __shared__ uint8_t okb[kMaskTile][kMaskTile + 4];
  __shared__ int wcount[4];
  const int RQ = R >> 2;
  const int r0 = blockIdx.z, si0 = blockIdx.y * kMaskTile, sj0 = blockIdx.x * kMaskTile;
  const int t = threadIdx.x;
  int n = 0;
#pragma unroll 4
  for (int c = 0; c < kMaskTile * kMaskTile / 256; ++c) {
    const int li = (c * 256 + t) / kMaskTile, lj = (c * 256 + t) % kMaskTile;
    const int si = si0 + li, sj = sj0 + lj;
    bool ok = false;
    if (si < H && sj < H) {
      const SnapRotSample g = snap_rot_geom(rs.tfm + r0 * 4, si, sj, H, H, rs.cell);
      // (the four taps are clamped into the plane: fetched side by side; every tap must be valid, snap_rot_sample)
      const uint8_t v00 = rs.valid[g.i0 * H + g.j0], v01 = rs.valid[g.i0 * H + g.j1];
      const uint8_t v10 = rs.valid[g.i1 * H + g.j0], v11 = rs.valid[g.i1 * H + g.j1];
      ok = g.ok && v00 && v01 && v10 && v11;
    }
    okb[li][lj] = ok ? 1 : 0;
    n += ok ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((t & 63) == 0) wcount[t >> 6] = n;
  __syncthreads();
  if (t == 0) {
    const int total = ((wcount[0] + wcount[1]) + wcount[2]) + wcount[3];
    if (total > 0)
      for (int k = 0; k < 4; ++k) atomicAdd(tcount + k * RQ + r0, (float)total);
  }
  const int64_t plane = (int64_t)H * H;
#pragma unroll 4
  for (int c = 0; c < kMaskTile * kMaskTile / 256; ++c) {
    const int a = (c * 256 + t) / kMaskTile, b = (c * 256 + t) % kMaskTile;    // b = the fast index of the run
    // k = 0, 2: runs along sj (li = a, lj = b)
    {
      const int si = si0 + a, sj = sj0 + b;
      if (si < H && sj < H) {
        const uint8_t v = okb[a][b];
        tvalid[(int64_t)r0 * plane + (int64_t)si * H + sj] = v;                                   // (di, dj) = (si, sj)
        tvalid[(int64_t)(2 * RQ + r0) * plane + (int64_t)(H - 1 - si) * H + (H - 1 - sj)] = v;   // (H-1-si, W-1-sj)
      }
    }
    // k = 1, 3: runs along si (lj = a, li = b)
    {
      const int si = si0 + b, sj = sj0 + a;
      if (si < H && sj < H) {
        const uint8_t v = okb[b][a];
        tvalid[(int64_t)(RQ + r0) * plane + (int64_t)sj * H + (H - 1 - si)] = v;                  // (sj, H-1-si)
        tvalid[(int64_t)(3 * RQ + r0) * plane + (int64_t)(H - 1 - sj) * H + si] = v;              // (H-1-sj, si)
      }
    }
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
  const unsigned mtiles = (unsigned)snap_cdiv(H, kMaskTile);
  hipLaunchKernelGGL(vf_rot_mask_kernel, dim3(mtiles, mtiles, (unsigned)(R / 4)), dim3(256), 0, s, rs, H, R, tvalid,
                     tcount);
  SNAP_CHECK_LAUNCH();
  HipLaunch L{s};
  if (!vfft::run_voting(g, nullptr, tvalid, map, mvalid, tcount, min_overlap * (float)H * (float)H, 1, ws, scores,
                        L, &rs))
    return SNAP_ERR_LAUNCH;
  return SNAP_OK;
}
