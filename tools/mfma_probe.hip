// Calibration probe (not part of the product): sustained f32-MFMA rate on this chip.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe && ./tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = (float)threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<blocks, 256>>>(out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 8 * NACC * 32.0 * 32 * 2 * 2;
  printf("nacc=%d blocks=%d iters=%d ms=%.3f TF=%.1f\n", NACC, blocks, iters, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(256, 20000);       // 1 wave/SIMD
    run<4>(512, 20000);       // 2 waves/SIMD
    run<4>(768, 20000);       // 3 waves/SIMD
    run<1>(768, 40000);       // dependent chain
    run<4>(2048, 20000);      // long run (~clock settles)
  }
  return 0;
}
