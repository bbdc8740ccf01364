// Why does an MFMA stream fed from L2 lose 25 % as soon as ANY VALU work rides along?  Variants of one loop
// (one wave per SIMD, whole chip; per iteration 48 v_mfma_f32_32x32x16_f16, random operands):
//   L     16 x 16-byte per-lane loads (1 KiB per wave each) from an L2-resident buffer
//   D     the same bytes by ds_read_b128 from LDS instead
//   V     one v_fma_f32 behind every MFMA (four independent chains)     Vtop: the same 48 at the loop top
//   S     one s_add behind every MFMA
// Prints time, TFLOP/s and - from clock64 / wall_clock64 of workgroup 0 - cycles per iteration and the clock.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SRC_NONE = 0, SRC_L2 = 1, SRC_LDS = 2, SRC_L2_SPREAD = 3 };
enum { SIDE_NONE = 0, SIDE_VALU = 1, SIDE_VALU_TOP = 2, SIDE_SALU = 3, SIDE_VALU2 = 4 };

template <int SRC, int SIDE, bool SMALL = false>
__global__ void __launch_bounds__(256) k(float* out, const f32x4* __restrict__ src, int iters, unsigned seed,
                                         unsigned long long* clk) {
  __shared__ f32x4 lds[16 * 256];
  for (int i = threadIdx.x; i < 16 * 256; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
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
  f32x4 ld[16];
  for (int i = 0; i < 16; ++i) ld[i] = f32x4{0, 0, 0, 0};
  float vv[4] = {1.0f, 2.0f, 3.0f, 4.0f}, v1 = 0.25f;
  int sc = 0;
  const f32x4* p = src + (blockIdx.x & 63) * 65536 + threadIdx.x;
  float sink = 0;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (SRC != SRC_NONE && SRC != SRC_L2_SPREAD) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sink += ld[i][0];
        if (SRC == SRC_L2) ld[i] = p[((it * 16 + i) & 255) * 256];
        else ld[i] = lds[i * 256 + threadIdx.x];
      }
    }
    if (SIDE == SIDE_VALU_TOP) {
#pragma unroll
      for (int w = 0; w < 48; ++w) vv[w & 3] = __builtin_fmaf(vv[w & 3], v1, vv[(w + 1) & 3]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (SRC == SRC_L2_SPREAD && (u * 8 + i) % 3 == 0) {          // one load per 3 MFMAs, consumed 48 MFMAs later
          const int j = (u * 8 + i) / 3;
          sink += ld[j][0];
          ld[j] = p[((it * 16 + j) & 255) * 256];
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SMALL) {                         // the same MACs as two v_mfma_f32_16x16x32_f16
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 c = {acc[i][8 * h], acc[i][8 * h + 1], acc[i][8 * h + 2], acc[i][8 * h + 3]};
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + i + h) & 3], b[(u * 3 + i) & 3], c, 0, 0, 0);
            acc[i][8 * h] = c[0]; acc[i][8 * h + 1] = c[1]; acc[i][8 * h + 2] = c[2]; acc[i][8 * h + 3] = c[3];
          }
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
        }
        // no inline asm here: hipcc puts an s_waitcnt vmcnt(0) in front of an asm statement while loads are in
        // flight (the first version of this loop lost 25 % to exactly that); fences pin the order instead
        if (SIDE == SIDE_VALU) { vv[i & 3] = __builtin_fmaf(vv[i & 3], v1, vv[(i + 1) & 3]); __builtin_amdgcn_sched_barrier(0); }
        if (SIDE == SIDE_VALU2) {
          vv[i & 3] = __builtin_fmaf(vv[i & 3], v1, vv[(i + 1) & 3]);
          vv[(i + 2) & 3] = __builtin_fmaf(vv[(i + 2) & 3], v1, vv[(i + 3) & 3]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (SIDE == SIDE_SALU) { sc = sc * 3 + it; __builtin_amdgcn_sched_barrier(0); }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  float s = sink + vv[0] + vv[1] + vv[2] + vv[3] + sc;
  for (int i = 0; i < 8; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 256 * 4 * 4);
  f32x4* src;
  (void)hipMalloc(&src, 64u * 65536 * 16 + 65536 * 16);
  (void)hipMemset(src, 0x3c, 64u * 65536 * 16 + 65536 * 16);
  unsigned long long* clk;
  (void)hipHostMalloc((void**)&clk, 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      launch(20000);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 48.0 * 32768 * 20000 * 1024 / (ms * 1e-3) / 1e12;
    printf("%-28s %8.3f ms %8.1f TFLOP/s  %7.1f cycles/iter (48 MFMA: 1536)  %.3f GHz\n", name, ms, tf,
           (double)clk[0] / 20000, (double)clk[0] / clk[1] / 10);
  };
#define RUN(NAME, S, V) run(NAME, [&](int it) { hipLaunchKernelGGL((k<S, V>), dim3(256), dim3(256), 0, 0, d, src, it, 1u, clk); })
  RUN("MFMA only", SRC_NONE, SIDE_NONE);
  RUN("MFMA + V", SRC_NONE, SIDE_VALU);
  RUN("MFMA + 2V", SRC_NONE, SIDE_VALU2);
  RUN("MFMA + L", SRC_L2, SIDE_NONE);
  RUN("MFMA + L + V", SRC_L2, SIDE_VALU);
  RUN("MFMA + L + Vtop", SRC_L2, SIDE_VALU_TOP);
  RUN("MFMA + Lspread", SRC_L2_SPREAD, SIDE_NONE);
  RUN("MFMA + Lspread + V", SRC_L2_SPREAD, SIDE_VALU);
  RUN("MFMA + Lspread + 2V", SRC_L2_SPREAD, SIDE_VALU2);
  RUN("MFMA + Lspread + S", SRC_L2_SPREAD, SIDE_SALU);
  RUN("MFMA + D", SRC_LDS, SIDE_NONE);
  RUN("MFMA + D + V", SRC_LDS, SIDE_VALU);
#define RUNS(NAME, S, V) run(NAME, [&](int it) { hipLaunchKernelGGL((k<S, V, true>), dim3(256), dim3(256), 0, 0, d, src, it, 1u, clk); })
  RUNS("16x16x32: MFMA only", SRC_NONE, SIDE_NONE);
  RUNS("16x16x32: MFMA + Lspread", SRC_L2_SPREAD, SIDE_NONE);
  RUNS("16x16x32: MFMA + Lspread + V", SRC_L2_SPREAD, SIDE_VALU);
  RUNS("16x16x32: MFMA + Lspread + 2V", SRC_L2_SPREAD, SIDE_VALU2);
  return 0;
}
