// What v_permlane16_swap_b32 moves (gfx950): prints, per lane, the two results for inputs a = lane, b = 100 + lane.
// build: hipcc --offload-arch=gfx950 -O2 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = s[0];
  out[64 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) { printf("result %d:", r); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * r + i]); printf("\n"); }
  return 0;
}
