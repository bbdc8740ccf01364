// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 at EQUAL work, whole chip, one wave per SIMD:
// registers only; fed 16 KiB per 48 (32x32-equivalent) MFMAs from L2 (lstm_h2o_kernel's operand ratio at
// R = 2); and with that plus 2 VALU instructions per 32x32-equivalent MFMA.  Random operands.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool SMALL, int LOADS, int VALU>
__global__ void __launch_bounds__(256) k(float* out, const f32x4* __restrict__ src, int iters, unsigned seed) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = (threadIdx.x * 2654435761u + i * 40503u + j * 69069u + seed) >> 7;
      a[i][j] = (_Float16)(((h & 1023) - 512) * (1.0f / 512));
      b[i][j] = (_Float16)((((h >> 10) & 1023) - 512) * (1.0f / 512));
    }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x4 ld[LOADS > 0 ? LOADS : 1];
  for (int i = 0; i < (LOADS > 0 ? LOADS : 1); ++i) ld[i] = f32x4{0, 0, 0, 0};
  float vv[4] = {1.0f, 2.0f, 3.0f, 4.0f}, v1 = 0.25f;
  const f32x4* p = src + (blockIdx.x & 63) * 65536 + threadIdx.x;      // 64 streams of 1 MiB: L2 resident
  float sink = 0;
  for (int it = 0; it < iters; ++it) {
    if (LOADS > 0) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) { sink += ld[i][0]; ld[i] = p[((it * LOADS + i) & 255) * 256]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (SMALL) {
          // one 32x32x16 = four 16x16x32 at half the k each... equal MACs: 32*32*16 = 2 x (16*16*32)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 c = {acc[i][8 * h], acc[i][8 * h + 1], acc[i][8 * h + 2], acc[i][8 * h + 3]};
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + i + h) & 3], b[(u * 3 + i) & 3], c, 0, 0, 0);
            acc[i][8 * h] = c[0]; acc[i][8 * h + 1] = c[1]; acc[i][8 * h + 2] = c[2]; acc[i][8 * h + 3] = c[3];
          }
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int w = 0; w < VALU; ++w) {                 // independent chains (w & 3): issue cost, not latency
          vv[w & 3] = vv[w & 3] * v1 + vv[w & 3];
          asm volatile("" : "+v"(vv[w & 3]));
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = sink + vv[0] + vv[1] + vv[2] + vv[3];
  for (int i = 0; i < 8; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 256 * 4 * 4);
  f32x4* src;
  (void)hipMalloc(&src, 64u * 65536 * 16 + 65536 * 16);
  (void)hipMemset(src, 0x3c, 64u * 65536 * 16 + 65536 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int iters : {20000, 20000}) {
      (void)hipEventRecord(e0);
      launch(iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double tf = 48.0 * 32768 * iters * 1024 / (ms * 1e-3) / 1e12;
      printf("%-44s iters %6d  %8.3f ms  %8.1f TFLOP/s\n", name, iters, ms, tf);
    }
  };
#define RUN(NAME, S, L, V) run(NAME, [&](int it) { hipLaunchKernelGGL((k<S, L, V>), dim3(256), dim3(256), 0, 0, d, src, it, 1u); })
  RUN("32x32x16 f16, registers only", false, 0, 0);
  RUN("16x16x32 f16, registers only", true, 0, 0);
  RUN("32x32x16 f16 + 16 KiB / 48 MFMA", false, 16, 0);
  RUN("16x16x32 f16 + 16 KiB / 48 MFMA", true, 16, 0);
  RUN("32x32x16 f16 + 16 KiB + 1 VALU / MFMA", false, 16, 1);
  RUN("16x16x32 f16 + 16 KiB + 1 VALU / MFMA", true, 16, 1);
  RUN("32x32x16 f16 + 16 KiB + 2 VALU / MFMA", false, 16, 2);
  RUN("16x16x32 f16 + 16 KiB + 2 VALU / MFMA", true, 16, 2);
  RUN("32x32x16 f16 + 16 KiB + 3 VALU / MFMA", false, 16, 3);
  RUN("32x32x16 f16 + 16 KiB + 4 VALU / MFMA", false, 16, 4);
  RUN("32x32x16 f16 + 2 VALU / MFMA, no loads", false, 0, 2);
  RUN("32x32x16 f16 + 4 VALU / MFMA, no loads", false, 0, 4);
  return 0;
}
