// Calibration probe (not part of the product): what rate can the CUs pull operand tiles global -> LDS
// (LDS-DMA, the conv / GEMM engines' operand path) as a function of the bytes a workgroup keeps in
// flight, the workgroups per CU and where the data sits (one 1 MB window re-read by everybody = L2
// hits; a private 16 MB window per workgroup = MALL / HBM)?  Every engine of this library measured
// 7.6-8.2 TB/s of (A + B) tile traffic whatever its tile shape; this prints the
// ceiling of the path itself, with no MFMA, VALU or barrier in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o tools/lds_dma_probe && ./tools/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void cglobal_void_t;

// STAGE_KB per ring slot, DEPTH slots in flight; 256 threads x 16 B = 4 KB per DMA instruction
template <int STAGE_KB, int DEPTH, bool BARRIER>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, size_t window, size_t per_wg_stride,
                                             int iters, float* sink) {
  extern __shared__ char ring[];
  constexpr int PIECES = STAGE_KB / 4;                 // DMA instructions per thread and stage
  const char* base = src + (size_t)blockIdx.x * per_wg_stride;
  const int tid = threadIdx.x;
  size_t off = (size_t)tid * 16;
  auto issue = [&](int slot) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      __builtin_amdgcn_global_load_lds((cglobal_void_t*)(base + off), (lds_void_t*)(ring + slot * (STAGE_KB * 1024) + p * 4096 + tid * 16),
                                       16, 0, 0);
      off += 4096;
      if (off >= window) off -= window;
    }
  };
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue(s);
  for (int it = 0; it < iters; ++it) {
    issue((it + DEPTH - 1) % DEPTH);
    // wait for the OLDEST stage: DEPTH - 1 stages may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
    if (BARRIER) __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = reinterpret_cast<float*>(ring)[blockIdx.x & 255];
}

template <int STAGE_KB, int DEPTH, bool BARRIER>
void run(const char* name, const char* src, size_t window, size_t stride, int wgs, float* sink) {
  const size_t lds = (size_t)STAGE_KB * 1024 * DEPTH;
  hipFuncSetAttribute((const void*)probe<STAGE_KB, DEPTH, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 4000 / (STAGE_KB / 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<STAGE_KB, DEPTH, BARRIER>), dim3(wgs), dim3(256), lds, 0, src, window, stride, 50, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<STAGE_KB, DEPTH, BARRIER>), dim3(wgs), dim3(256), lds, 0, src, window, stride, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * iters * STAGE_KB * 1024.0;
  printf("%-6s stage %2d KB depth %d barrier %d wgs %4d (%4.1f / CU, %3d KB in flight / CU): %7.2f TB/s  (%.3f ms)\n", name,
         STAGE_KB, DEPTH, (int)BARRIER, wgs, wgs / 256.0, (int)(STAGE_KB * (DEPTH - 1) * (wgs / 256.0 > 1 ? wgs / 256 : 1)),
         bytes / ms / 1e9, ms);
}

int main() {
  const size_t total = (size_t)4 << 30;                 // 4 GB source
  char* src;
  float* sink;
  hipMalloc(&src, total);
  hipMalloc(&sink, 65536 * 4);
  hipMemset(src, 1, total);
  for (int where = 0; where < 2; ++where) {
    const char* name = where == 0 ? "L2" : "HBM";
    const size_t window = where == 0 ? ((size_t)1 << 20) : ((size_t)4 << 20);
    for (int wgs : {256, 512, 768, 1024}) {
      const size_t stride = where == 0 ? 0 : (total - window) / wgs / 4096 * 4096;   // private windows spread over 4 GB
      run<16, 2, true>(name, src, window, stride, wgs, sink);
      run<16, 3, true>(name, src, window, stride, wgs, sink);
      run<16, 4, true>(name, src, window, stride, wgs, sink);
      run<16, 4, false>(name, src, window, stride, wgs, sink);
      run<32, 3, true>(name, src, window, stride, wgs, sink);
      if (wgs <= 512) run<32, 4, false>(name, src, window, stride, wgs, sink);
    }
  }
  return 0;
}
