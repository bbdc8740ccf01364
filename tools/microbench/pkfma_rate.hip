// Issue rate of v_pk_fma_f32 by operand kind (one wave per SIMD, independent accumulators).
//   hipcc --offload-arch=gfx950 -O3 -o pkfma_rate pkfma_rate.hip && ./pkfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float wv) {
  f32x2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
  f32x2 a = {threadIdx.x * 1e-3f, threadIdx.x * 2e-3f}, w = {wv, wv * 0.5f};
  float a1 = threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(w));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(w));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(w));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "s"(w));
      if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a1), "v"(wv));
      if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a1), "s"(wv));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[6] = {"v_pk_fma_f32 v,v,v", "v_pk_fma_f32 v,s,v", "v_pk_fma_f32 v(bcast),v,v", "v_pk_fma_f32 v(bcast),s,v",
                          "v_fma_f32 v,v,v", "v_fma_f32 v,s,v"};
  for (int m = 0; m < 6; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // cycles per instruction at a nominal 2.4 GHz
      printf("%-28s %8.3f ms  -> %.2f ns per instruction per wave (%.1f cycles @2.4 GHz)\n", names[m], ms,
             ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.4);
    }
  }
  return 0;
}
