// Prints what v_permlane32_swap_b32 does on this GPU (mlp_pool.hip relies on: lanes 32..63 of the
// first operand <-> lanes 0..31 of the second).   hipcc --offload-arch=gfx950 tools/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
  const unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("first : lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("second: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  const bool ok = h[0] == 100 && h[32] == 200 && h[64] == 132 && h[96] == 232;
  printf("%s\n", ok ? "first[32:63] <-> second[0:31]: as assumed" : "DIFFERENT SEMANTICS");
  return ok ? 0 : 1;
}
