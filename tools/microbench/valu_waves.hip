// VALU issue rate against the number of waves per SIMD: v_fma_f32 v,v,v with independent accumulators,
// 256 / 512 / 768 / 1024 threads per workgroup (1 / 2 / 3 / 4 waves per SIMD), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o valu_waves valu_waves.hip && ./valu_waves
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT, int SREG>
__global__ void __launch_bounds__(NT) k(float* out, int iters, float wv) {
  float acc[24];
  for (int i = 0; i < 24; ++i) acc[i] = threadIdx.x * 1e-6f * i;
  float a1 = threadIdx.x * 1e-3f, w[8];
  for (int i = 0; i < 8; ++i) w[i] = wv + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      if (SREG) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(wv), "v"(a1));
      else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a1), "v"(w[i & 7]));
    }
  }
  float s = 0;
  for (int i = 0; i < 24; ++i) s += acc[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int NT, int SREG>
void run(float* d, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, SREG>), dim3(256), dim3(NT), 0, 0, d, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = ms * 1e6 / (iters * 24.0 * (NT / 256));    // ns per wave-instruction issued on one SIMD
    printf("%-22s %d waves/SIMD  %8.3f ms  -> %.2f ns per instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, NT / 256, ms,
           per_simd, per_simd * 2.4);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<256, 0>(d, "v_fma_f32 v,v,v"); run<512, 0>(d, "v_fma_f32 v,v,v"); run<768, 0>(d, "v_fma_f32 v,v,v"); run<1024, 0>(d, "v_fma_f32 v,v,v");
  run<256, 1>(d, "v_fmac_f32 v,s,v"); run<512, 1>(d, "v_fmac_f32 v,s,v"); run<768, 1>(d, "v_fmac_f32 v,s,v");
  return 0;
}
