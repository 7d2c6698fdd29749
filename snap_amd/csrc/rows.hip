// Row compaction for masked voxel lists.
//
// The lift marks every voxel as observed / unobserved (streetview_encoder.py:146-178);
// the fusion MLP of the reference then runs over ALL voxels and the unobserved ones are
// zeroed (streetview_encoder.py:281-283).  Here the observed rows are listed once
// (ascending, deterministic) and the MLP GEMMs walk that list through the row-indexed
// conv entry point; the masked rows of the dense volume are zero-filled.  Nothing is
// synchronised with the host: the row count stays in device memory.
#include "common.h"

namespace {

constexpr int CR_THREADS = 256;
constexpr int CR_PER_THREAD = 16;  // mask bytes per thread (one 16-byte load)
constexpr int CR_BLOCK_ROWS = CR_THREADS * CR_PER_THREAD;

__device__ __forceinline__ int load_flags(const uint8_t* __restrict__ mask, int64_t base, int64_t M,
                                          uint32_t lo, uint32_t span, uint32_t* bits) {
  // returns the number of selected bytes among mask[base .. base+16) (bounds-checked);
  // *bits has bit e set when lo <= mask[base+e] <= lo + span  (1, 254: mask != 0).
  uint32_t b = 0;
  if (base + CR_PER_THREAD <= M && ((reinterpret_cast<uintptr_t>(mask + base) & 15) == 0)) {
    const uint4 v = *reinterpret_cast<const uint4*>(mask + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((((w[i] >> (8 * j)) & 0xffu) - lo) <= span) b |= 1u << (4 * i + j);
  } else {
#pragma unroll
    for (int e = 0; e < CR_PER_THREAD; ++e)
      if (base + e < M && ((uint32_t)mask[base + e] - lo) <= span) b |= 1u << e;
  }
  *bits = b;
  return __popc(b);
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  // 256 threads = 4 waves: wave scan by shuffles, wave totals through LDS.
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < wid) off += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(CR_THREADS) void count_rows_kernel(const uint8_t* __restrict__ mask,
                                                                int64_t M, uint32_t lo, uint32_t span,
                                                                int32_t* __restrict__ block_count) {
  const int64_t base = ((int64_t)blockIdx.x * CR_THREADS + threadIdx.x) * CR_PER_THREAD;
  uint32_t bits;
  const int c = load_flags(mask, base, M, lo, span, &bits);
  int total;
  block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the block counts (in place) + the grand total.
__global__ __launch_bounds__(CR_THREADS) void scan_blocks_kernel(int32_t* __restrict__ block_count,
                                                                 int nblocks,
                                                                 int32_t* __restrict__ count) {
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += CR_THREADS) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, &total);
    const int c = carry;
    if (i < nblocks) block_count[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(CR_THREADS) void write_rows_kernel(const uint8_t* __restrict__ mask,
                                                                int64_t M, uint32_t lo, uint32_t span,
                                                                const int32_t* __restrict__ block_off,
                                                                int32_t* __restrict__ index) {
  const int64_t base = ((int64_t)blockIdx.x * CR_THREADS + threadIdx.x) * CR_PER_THREAD;
  uint32_t bits;
  const int c = load_flags(mask, base, M, lo, span, &bits);
  int total;
  int pos = block_off[blockIdx.x] + block_exclusive_scan(c, &total);
#pragma unroll
  for (int e = 0; e < CR_PER_THREAD; ++e)
    if (bits & (1u << e)) index[pos++] = (int32_t)(base + e);
}

__global__ __launch_bounds__(256) void fill_masked_rows_kernel(float* __restrict__ y,
                                                               const uint8_t* __restrict__ mask,
                                                               int64_t M, int C, float value) {
  // a half-wave per row; only rows with mask == 0 are touched.
  const int hl = threadIdx.x & 31;
  const int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (m >= M || mask[m]) return;
  const f32x4 v = {value, value, value, value};
  f32x4* row = reinterpret_cast<f32x4*>(y + m * C);
  for (int q = hl; q < (C >> 2); q += 32) row[q] = v;
}

}  // namespace

extern "C" size_t snap_compact_rows_workspace_bytes(int64_t M) {
  return (size_t)(snap_cdiv(M, CR_BLOCK_ROWS) + 1) * sizeof(int32_t);
}

extern "C" int snap_compact_rows_range_u8(const uint8_t* mask, int64_t M, int32_t lo, int32_t hi,
                                          int32_t* index, int32_t* count, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (!mask || !index || !count || !workspace) return SNAP_ERR_NULL;
  if (M <= 0 || M > 0x7fffffffLL || lo < 0 || hi > 255 || lo > hi) return SNAP_ERR_BAD_SHAPE;
  if (workspace_bytes < snap_compact_rows_workspace_bytes(M)) return SNAP_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nblocks = (int)snap_cdiv(M, CR_BLOCK_ROWS);
  int32_t* bc = static_cast<int32_t*>(workspace);
  const uint32_t ulo = (uint32_t)lo, span = (uint32_t)(hi - lo);
  hipLaunchKernelGGL(count_rows_kernel, dim3(nblocks), dim3(CR_THREADS), 0, s, mask, M, ulo, span, bc);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(CR_THREADS), 0, s, bc, nblocks, count);
  SNAP_CHECK_LAUNCH();
  hipLaunchKernelGGL(write_rows_kernel, dim3(nblocks), dim3(CR_THREADS), 0, s, mask, M, ulo, span,
                     (const int32_t*)bc, index);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

extern "C" int snap_compact_rows_u8(const uint8_t* mask, int64_t M, int32_t* index, int32_t* count,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return snap_compact_rows_range_u8(mask, M, 1, 255, index, count, workspace, workspace_bytes, stream);
}

extern "C" int snap_fill_masked_rows_f32(float* y, const uint8_t* mask, int64_t M, int32_t C,
                                         float value, void* stream) {
  if (!y || !mask) return SNAP_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(y) & 15) return SNAP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(fill_masked_rows_kernel, dim3((unsigned)snap_cdiv(M, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), y, mask, M, C, value);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}
