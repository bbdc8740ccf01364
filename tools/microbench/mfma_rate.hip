// Sustained rate of v_mfma_f32_32x32x16_bf16 (and the f32 32x32x2) on the whole chip, registers only:
// what the matrix pipe delivers under its own power draw, with no memory traffic at all.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool ZERO>
__global__ void __launch_bounds__(256) k_bf16(float* out, int iters, unsigned seed) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = (threadIdx.x * 2654435761u + i * 40503u + j * 69069u + seed) >> 7;
      a[i][j] = ZERO ? (__bf16)0.f : (__bf16)(((h & 1023) - 512) * (1.0f / 512));
      b[i][j] = ZERO ? (__bf16)0.f : (__bf16)((((h >> 10) & 1023) - 512) * (1.0f / 512));
    }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// MFMA stream + the side work of lstm_split_kernel at its real ratios: per 48 MFMAs, LOADS x 16-byte
// per-lane loads (1 KiB per wave each) from an L2-resident buffer and VALU x 2 packed-f32 ops per MFMA.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int LOADS, int VALU, int SPAN>
__global__ void __launch_bounds__(256) k_mix(float* out, const f32x4* __restrict__ src, int iters, unsigned seed) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = (threadIdx.x * 2654435761u + i * 40503u + j * 69069u + seed) >> 7;
      a[i][j] = (__bf16)(((h & 1023) - 512) * (1.0f / 512));
      b[i][j] = (__bf16)((((h >> 10) & 1023) - 512) * (1.0f / 512));
    }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x4 ld[LOADS > 0 ? LOADS : 1];
  for (int i = 0; i < (LOADS > 0 ? LOADS : 1); ++i) ld[i] = f32x4{0, 0, 0, 0};
  f32x2 v0 = {1.0f, 0.5f}, v1 = {0.25f, 0.125f};
  const f32x4* p = src + (blockIdx.x & 63) * 65536 + threadIdx.x;      // 64 streams of 1 MiB: L2 resident
  float sink = 0;
  for (int it = 0; it < iters; ++it) {
    if (LOADS > 0) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) { sink += ld[i][0]; ld[i] = p[((it * LOADS + i) & (SPAN - 1)) * 256]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int w = 0; w < VALU; ++w) {
          v0 = v0 * v1 + v0;
          asm volatile("" : "+v"(v0));
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = sink + v0[0] + v0[1];
  for (int i = 0; i < 8; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_f32(float* out, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = (threadIdx.x % 17 - 8) * 0.1f + i; b[i] = (threadIdx.x % 13 - 6) * 0.1f - i; }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double flop_per_iter_per_wave) {
    for (int iters : {2000, 20000, 20000, 20000}) {
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double tf = flop_per_iter_per_wave * iters * 1024 / (ms * 1e-3) / 1e12;
      printf("%-34s iters %6d  %8.3f ms  %8.1f TFLOP/s\n", name, iters, ms, tf);
    }
  };
  // 256 blocks x 4 waves = one wave per SIMD
  run("bf16 32x32x16, random operands", [&](int it) { hipLaunchKernelGGL((k_bf16<8, false>), dim3(256), dim3(256), 0, 0, d, it, 12345u); }, 32.0 * 32768);
  run("bf16 32x32x16, zero operands", [&](int it) { hipLaunchKernelGGL((k_bf16<8, true>), dim3(256), dim3(256), 0, 0, d, it, 12345u); }, 32.0 * 32768);
  f32x4* src;
  hipMalloc(&src, 64u * 65536 * 16 + 65536 * 16);
  hipMemset(src, 0x3c, 64u * 65536 * 16 + 65536 * 16);
  const double fl = 48.0 * 32768;
  run("bf16 + 16 KiB loads / 48 MFMA", [&](int it) { hipLaunchKernelGGL((k_mix<16, 0, 256>), dim3(256), dim3(256), 0, 0, d, src, it * 2 / 3, 1u); }, fl * 2 / 3);
  run("bf16 + 16 KiB loads, 16 KiB span (L1)", [&](int it) { hipLaunchKernelGGL((k_mix<16, 0, 4>), dim3(256), dim3(256), 0, 0, d, src, it * 2 / 3, 1u); }, fl * 2 / 3);
  run("bf16 + 8 KiB loads / 48 MFMA", [&](int it) { hipLaunchKernelGGL((k_mix<8, 0, 256>), dim3(256), dim3(256), 0, 0, d, src, it * 2 / 3, 1u); }, fl * 2 / 3);
  run("bf16 + 2 VALU / MFMA", [&](int it) { hipLaunchKernelGGL((k_mix<0, 2, 256>), dim3(256), dim3(256), 0, 0, d, src, it * 2 / 3, 1u); }, fl * 2 / 3);
  run("bf16 + loads + 2 VALU / MFMA", [&](int it) { hipLaunchKernelGGL((k_mix<16, 2, 256>), dim3(256), dim3(256), 0, 0, d, src, it * 2 / 3, 1u); }, fl * 2 / 3);
  run("f32 32x32x2", [&](int it) { hipLaunchKernelGGL(k_f32, dim3(256), dim3(256), 0, 0, d, it); }, 32.0 * 4096);
  return 0;
}
