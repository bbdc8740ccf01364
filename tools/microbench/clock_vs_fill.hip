// Does the clock give back what a fuller matrix pipe gains?  
// The inner loop of the big Bi-LSTM kernels without their gates and barriers: per ENTRY two 1 KB weight fragments from an
// L2-resident 2.6 MB buffer (ring of 4 entries) feed 12 v_mfma_f32_16x16x32_f16 in chains of three (hi*lo, lo*hi, hi*hi
// pattern) whose A operands come from LDS (8 ds_read_b128 per 4 entries); 256 workgroups = one per CU.  Varied:
//   waves per SIMD (1 or 2; a workgroup is 4 or 8 waves), the DATA (random f16 bits, or all zeros: the same instructions, no toggling in the multipliers),
//   the duty (s_sleep between entries).  Printed per variant: cycles per product and SIMD, the clock the launch held
//   (s_memtime / s_memrealtime of every workgroup, median) and dense-f16 TFLOP/s = products x 16384 / wall.
//   hipcc --offload-arch=gfx950 -O3 -o clock_vs_fill clock_vs_fill.hip && ./clock_vs_fill
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kEntries = 40, kWBytes = kEntries * 2 * 1024;   // one wave's weight slice: 80 KB, as in the layers

// STREAM: on top, every 6th entry one 1 KB request to a 1 GB buffer (every address once per launch: HBM) and every 8th entry
// one 1 KB store: ~1.3 + 1.0 TB/s over the chip at full rate, the traffic of the 192->128 layer's launches
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MF32: the same operand traffic and the same FLOPs per entry through SIX v_mfma_f32_32x32x16_f16 (two 32-row tiles x three
// products; half the A / B register reads per MAC) instead of twelve 16x16x32
// ORD = 1: the three products of a chain as (hi*lo, hi*hi, lo*hi) - each operand changes ONCE per chain - instead of
// (hi*lo, lo*hi, hi*hi)
template <int WAVES, int SLEEP, int STREAM = 0, int MF32 = 0, int ORD = 0>
__global__ void __launch_bounds__(256 * WAVES) k(const char* __restrict__ w, const float* __restrict__ xsrc, float* out, int steps,
                                                unsigned long long* clk, const char* big = nullptr, char* bigw = nullptr) {
  __shared__ __attribute__((aligned(16))) float xl[6 * 4 * 2 * 256];             // 48 KB of "activations"
  for (int i = threadIdx.x; i < 6 * 4 * 2 * 256; i += 256 * WAVES) xl[i] = xsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(w) + (size_t)((blockIdx.x * 4 * WAVES + wave) & 31) * kWBytes, 0, kWBytes, 0x00020000);
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 acci[16];
  for (int i = 0; i < 16; ++i) acci[i] = i32x4{0, 0, 0, 0};
  f32x16 acc32[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  f32x4 b[4][2], a[4][2];
  for (int e = 0; e < 3; ++e)
    for (int t = 0; t < 2; ++t) b[e][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (e * 2 + t) * 1024, 0));
  f32x4 hb = f32x4{0.f, 0.f, 0.f, 0.f};
  size_t boff = ((size_t)(blockIdx.x * 4 * WAVES + wave) * 1024) * 1024 + lane * 16;   // 1 MB per wave and launch
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int e = 0; e < kEntries; ++e) {
      if (e % 4 == 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) a[rt][t] = *(const f32x4*)(xl + ((((e / 4) % 6) * 4 + rt) * 2 + t) * 256 + lane * 4);
      }
      const int en = (e + 3) % kEntries;
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int rt = m / 3, pr = m % 3;
        const f16x8 af = __builtin_bit_cast(f16x8, a[rt][ORD ? (pr == 2 ? 1 : 0) : (pr == 1 ? 1 : 0)]);
        const f16x8 bf = __builtin_bit_cast(f16x8, b[e % 4][pr == 0 ? 1 : 0]);
        if constexpr (MF32 == 2) {                       // v_mfma_i32_16x16x64_i8: twice the MACs per instruction, same operand bytes
          acci[(e % 4) * 4 + rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af), __builtin_bit_cast(i32x4, bf),
                                                                          acci[(e % 4) * 4 + rt], 0, 0, 0);
        } else if constexpr (MF32 == 1) {
          if (rt < 2) acc32[(e % 2) * 2 + rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc32[(e % 2) * 2 + rt], 0, 0, 0);
        } else
        acc[(e % 4) * 4 + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[(e % 4) * 4 + rt], 0, 0, 0);
        if (m < 2) {
          __builtin_amdgcn_sched_barrier(0);
          b[(e + 3) % 4][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (en * 2 + m) * 1024, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (STREAM && e % 6 == 5) {
        acc[0][0] += hb[0];
        hb = *(const f32x4*)(big + boff);
        boff += 1024;
      }
      if (STREAM && e % 8 == 7) *(f32x4*)(bigw + boff) = acc[1];
      if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sink = 0.f;
  for (int i = 0; i < 16; ++i) sink += acc[i][0] + acc[i][3];
  for (int i = 0; i < 4; ++i) sink += acc32[i][0] + acc32[i][15];
  for (int i = 0; i < 16; ++i) sink += (float)(acci[i][0] + acci[i][3]);
  if (sink == 12345.678f) out[threadIdx.x] = sink;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static int g_sustain = 20;
static char *g_big = nullptr, *g_bigw = nullptr;
template <int WAVES, int SLEEP, int STREAM = 0, int MF32 = 0, int ORD = 0>
static void run(const char* name, const char* w, const float* x, float* out, unsigned long long* clk, int steps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // SUSTAIN launches back to back first (argv[1], default 20 = a 3.5 ms burst; 12000 = two seconds of continuous load, after
  // which the clock is the one the chip HOLDS under this load), then the timed 20
  for (int i = 0; i < g_sustain; ++i) hipLaunchKernelGGL((k<WAVES, SLEEP, STREAM, MF32, ORD>), dim3(256), dim3(256 * WAVES), 0, 0, w, x, out, steps, clk, g_big, g_bigw);
  // socket power and shader clock while the queue is still full of these launches
  double watts = 0, mhz = 0;
  {
    for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL((k<WAVES, SLEEP, STREAM, MF32, ORD>), dim3(256), dim3(256 * WAVES), 0, 0, w, x, out, steps, clk, g_big, g_bigw);
    for (int rep = 0; rep < 2; ++rep) {                  // the second reading: the load has lasted a while
      FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null", "r");
      char line[512];
      while (f && fgets(line, sizeof line, f)) {
        const char* p = strstr(line, "Package Power (W):");
        if (p) watts = atof(p + 18);
        p = strstr(line, "sclk clock level:");
        if (p && (p = strchr(p, '('))) mhz = atof(p + 1);
      }
      if (f) pclose(f);
    }
  }
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<WAVES, SLEEP, STREAM, MF32, ORD>), dim3(256), dim3(256 * WAVES), 0, 0, w, x, out, steps, clk, g_big, g_bigw);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz, cyc;
  const double prod_per_wave = (double)steps * kEntries * 12;
  for (int i = 0; i < 256; ++i) {
    ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0));          // s_memrealtime: 100 MHz
    cyc.push_back((double)h[2 * i] / (prod_per_wave * WAVES));                 // cycles per product of the SIMD
  }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double tf = 256.0 * 4 * WAVES * prod_per_wave * 16384.0 / (ms / reps * 1e-3) / 1e12;
  printf("%-58s %6.2f cycles/product  %5.3f GHz  %7.1f TFLOP/s dense f16 (%5.1f f32-grade)  %7.1f us  %6.0f W  sclk %4.0f MHz\n", name,
         cyc[128], ghz[128], tf, tf / 3, ms / reps * 1e3, watts, mhz);
  fflush(stdout);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1) g_sustain = atoi(argv[1]);
  printf("sustain: %d launches in front of the timed 20\n", g_sustain);
  const size_t wbytes = 32 * (size_t)kWBytes;
  std::vector<unsigned short> hw(wbytes / 2), hz(wbytes / 2, 0);
  std::vector<float> hx(6 * 4 * 2 * 256), hxz(6 * 4 * 2 * 256, 0.f);
  srand(7);
  for (auto& v : hw) v = (unsigned short)((rand() & 0x3fff) | ((rand() & 1) << 15) | 0x2000);      // f16 in [2^-7, 2): finite
  for (size_t i = 0; i < hx.size(); ++i) {
    const unsigned short lo = (unsigned short)((rand() & 0x3fff) | 0x2000), hi = (unsigned short)((rand() & 0x3fff) | 0x2000 | ((rand() & 1) << 15));
    const unsigned u = ((unsigned)hi << 16) | lo;
    hx[i] = *(const float*)&u;
  }
  char *w, *wz; float *x, *xz, *out; unsigned long long* clk;
  hipMalloc(&w, wbytes); hipMalloc(&wz, wbytes); hipMalloc(&x, hx.size() * 4); hipMalloc(&xz, hx.size() * 4);
  hipMalloc(&out, 4096); hipMalloc(&clk, 512 * 8);
  hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(wz, hz.data(), wbytes, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(xz, hxz.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const int steps = 26;                                  // ~ two launches' worth of products per wave
  run<1, 0>("1 wave/SIMD, random operands", w, x, out, clk, steps);
  run<1, 0>("1 wave/SIMD, ZERO operands (same instructions)", wz, xz, out, clk, steps);
  run<2, 0>("2 waves/SIMD, random operands", w, x, out, clk, steps);
  run<2, 0>("2 waves/SIMD, ZERO operands", wz, xz, out, clk, steps);
  run<2, 1>("2 waves/SIMD, random, s_sleep 1 per entry", w, x, out, clk, steps);
  run<2, 2>("2 waves/SIMD, random, s_sleep 2 per entry", w, x, out, clk, steps);
  run<2, 4>("2 waves/SIMD, random, s_sleep 4 per entry", w, x, out, clk, steps);
  hipMalloc(&g_big, (size_t)2048 * 1024 * 1024 + 4096); hipMalloc(&g_bigw, (size_t)2048 * 1024 * 1024 + (1 << 20));
  hipMemset(g_big, 1, (size_t)2048 * 1024 * 1024);
  run<2, 0, 1>("2 waves/SIMD, random + HBM stream (1 KB / 6 entries, store / 8)", w, x, out, clk, steps);
  run<2, 2, 1>("2 waves/SIMD, random + HBM stream, s_sleep 2", w, x, out, clk, steps);
  run<1, 0, 1>("1 wave/SIMD, random + HBM stream", w, x, out, clk, steps);
  run<2, 0, 0, 0, 1>("2 waves/SIMD, random, products ordered hi*lo, hi*hi, lo*hi", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0>("2 waves/SIMD, random operands (again)", w, x, out, clk, steps);
  run<2, 0, 0, 0, 1>("2 waves/SIMD, random, products ordered hi*lo, hi*hi, lo*hi (again)", w, x, out, clk, steps);
  run<2, 0, 0, 2>("2 waves/SIMD, random BYTES, v_mfma_i32_16x16x64_i8 (x2 for ops)", w, x, out, clk, steps);
  run<2, 0, 1, 2>("2 waves/SIMD, random bytes + HBM stream, i8 (x2 for ops)", w, x, out, clk, steps);
  run<2, 0, 0, 1>("2 waves/SIMD, random, 32x32x16 tiles (same FLOPs, half the operand reads)", w, x, out, clk, steps);
  run<2, 0, 1, 1>("2 waves/SIMD, random + HBM stream, 32x32x16 tiles", w, x, out, clk, steps);
  run<1, 0, 0, 1>("1 wave/SIMD, random, 32x32x16 tiles", w, x, out, clk, steps);
  run<1, 2>("1 wave/SIMD, random, s_sleep 2 per entry", w, x, out, clk, steps);
  run<1, 4>("1 wave/SIMD, random, s_sleep 4 per entry", w, x, out, clk, steps);
  return 0;
}
