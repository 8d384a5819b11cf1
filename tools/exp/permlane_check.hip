// Semantics check of v_permlane16_swap / v_permlane32_swap on gfx950 (used by the weight-gradient flush).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;      // a: blocks a0..a3 of 16 lanes, b: b0..b3
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto q = __builtin_amdgcn_permlane32_swap(r[0], r[1], false, false);
  p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1]; p[128 + threadIdx.x] = q[0]; p[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 first ", "swap16 second", "then32 first ", "then32 second"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int blk = 0; blk < 4; ++blk) printf(" [%u..%u]", h[v * 64 + blk * 16], h[v * 64 + blk * 16 + 15]); printf("\n"); }
  return 0;
}
