// Does hipExtAnyOrderLaunch drop the barrier bit on gfx950?  A (few workgroups, ~200 us of spinning) then B (one
// workgroup) in the SAME stream: with the barrier bit B starts after A ends; without it B's start precedes A's end.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder tools/microbench/anyorder.hip && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long* t, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
__global__ void stamp(unsigned long long* t) {
  if (threadIdx.x == 0) { t[2] = wall_clock64(); }
}
int main() {
  unsigned long long* t; hipMalloc(&t, 64); hipMemset(t, 0, 64);
  hipStream_t s; hipStreamCreate(&s);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, t, 20000ull);   // 100 MHz wall clock: 200 us
      if (mode == 0) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, t);
      else hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t);
      hipStreamSynchronize(s);
      unsigned long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("%s: A ran %.1f us; B started %.1f us after A's START (%.1f us %s A's end)\n", mode ? "any-order" : "in-order ",
             (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0,
             (h[2] > h[1] ? (h[2] - h[1]) : (h[1] - h[2])) / 100.0, h[2] > h[1] ? "AFTER" : "BEFORE");
    }
  }
  return 0;
}
