// Does the clock give back what a fuller matrix pipe gains?  
// The inner loop of the big Bi-LSTM kernels without their gates and barriers: per ENTRY two 1 KB weight fragments from an
// L2-resident 2.6 MB buffer (ring of 4 entries) feed 12 v_mfma_f32_16x16x32_f16 in chains of three (hi*lo, lo*hi, hi*hi
// pattern) whose A operands come from LDS (8 ds_read_b128 per 4 entries); 256 workgroups = one per CU.  Varied:
//   waves per SIMD (1 or 2; a workgroup is 4 or 8 waves), the DATA (random f16 bits, or all zeros: the same instructions, no toggling in the multipliers),
//   the duty (s_sleep between entries).  Printed per variant: cycles per product and SIMD, the clock the launch held
//   (s_memtime / s_memrealtime of every workgroup, median) and dense-f16 TFLOP/s = products x 16384 / wall.
// Round 6: EVERY wave stores s_memtime / s_memrealtime at entry and exit plus HW_ID / XCC_ID, and the
// host prints the launch's timeline next to the event-timed wall: first start .. last end over all waves, the spread of the
// starts, the elapsed time of the older (0-3) and the younger (4-7) waves of a workgroup, how many workgroups shared a CU - so
// that "in-kernel" and "wall" TFLOP/s are computed over the same interval.  PRIO: 1 = the younger waves at s_setprio 1;
// 2 = both groups swap priority every entry (A high on even entries, B on odd).  ORD = 2: the twelve products of an entry
// ordered term-outer (consecutive products on different accumulators), so that ONE wave can fill the pipe.  WEVERY: the weight
// stream thinned (2: a fetch for every second entry; 0: none) - is the fill bounded by the L2 -> CU stream or by the pipe?
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
constexpr int kRec = 6;                                       // u64 per wave: c0, r0, c1, r1, HW_ID, XCC_ID
template <int WAVES, int SLEEP, int STREAM = 0, int MF32 = 0, int ORD = 0, int PRIO = 0, int WEVERY = 1, int PPE = 12, int DEDUP = 0, int NB = 4, int LOADER = 0>
__global__ void __launch_bounds__(256 * WAVES) k(const char* __restrict__ w, const float* __restrict__ xsrc, float* out, int steps,
                                                unsigned long long* clk, const char* big = nullptr, char* bigw = nullptr) {
  __shared__ __attribute__((aligned(16))) float xl[6 * 4 * 2 * 256];             // 48 KB of "activations"
  unsigned long long ce, re;                             // kernel entry, in front of the staging
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ce), "=s"(re)::"memory");
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
  static_assert(kEntries % NB == 0, "ring");
  f32x4 b[NB][2], a[4][2];                               // NB: weight ring entries (NB - 1 in flight per wave: 2 KB each)
  for (int e = 0; e < (WEVERY == 1 ? NB - 1 : NB); ++e)
    for (int t = 0; t < 2; ++t) b[e][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (e * 2 + t) * 1024, 0));
  f32x4 hb = f32x4{0.f, 0.f, 0.f, 0.f};
  size_t boff = ((size_t)(blockIdx.x * 4 * WAVES + wave) * 1024) * 1024 + lane * 16;   // 1 MB per wave and launch
  unsigned long long c0, r0, c1, r1;
  float sink0 = 0.f;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // LOADER (two waves per SIMD): waves 4-7 do NOTHING but issue the weight stream of one consumer wave each (2 KB per entry, at
  // most 8 KB in flight; 1: into registers, 2: by LDS-DMA into a 4 KB ring of their own), waves 0-3 multiply without any
  // vector-memory instruction (run with WEVERY = 0).  Does a 1 KB request cost the ISSUING wave its time, or the SIMD?
  if (LOADER && wave >= 4) {
    f32x4 sinkv = f32x4{0.f, 0.f, 0.f, 0.f};
    __shared__ __attribute__((aligned(16))) float ring[4 * 4 * 256];
    for (int s = 0; s < steps; ++s) {
#pragma unroll 8
      for (int e = 0; e < kEntries; ++e) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (LOADER == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + ((wave - 4) * 4 + (e & 1) * 2 + t) * 256), 16, lane * 16,
                                                 (e * 2 + t) * 1024, 0, 0);
          } else {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (e * 2 + t) * 1024, 0));
            asm volatile("" ::"v"(v));
          }
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sink0 = sinkv[0];
  } else
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int e = 0; e < kEntries; ++e) {
      if (e % 4 == 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) a[rt][t] = *(const f32x4*)(xl + ((((e / 4) % 6) * 4 + rt) * 2 + t) * 256 + lane * 4);
      }
      const int en = (e + NB - 1) % kEntries;
      if (PRIO == 2) {
        if (((e & 1) != 0) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      // DEDUP (SURVEY 7's second dedup, modelled): every fifth entry (2 of a step's 10 k-blocks: the signal branch's share of the
      // 192 -> 128 input projection) is neither fetched nor multiplied; the wave loads 2 KB of precomputed accumulator-start values
      // instead - the same bytes into the CU, a fifth fewer products.  1: from a 32 MB table (what one launch group's Z_sig
      // would be: re-read, Infinity-Cache resident); 2: every address once (HBM)
      if (DEDUP && e % 5 == 4) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[(e % 4) * 4 + t] += *(const f32x4*)(big + (DEDUP == 1 ? boff % ((size_t)32 << 20) : boff));   // (+=: an overwrite would make the products in front of it dead code)
          boff += 1024;
        }
      } else
#pragma unroll
      for (int m = 0; m < PPE; ++m) {                    // PPE = 6: a weight entry feeds two row tiles only (a 32-row wave)
        const int rt = ORD == 2 ? m % 4 : m / 3, pr = ORD == 2 ? m / 4 : m % 3;
        const f16x8 af = __builtin_bit_cast(f16x8, a[rt][ORD == 1 ? (pr == 2 ? 1 : 0) : (pr == 1 ? 1 : 0)]);
        const f16x8 bf = __builtin_bit_cast(f16x8, b[e % NB][pr == 0 ? 1 : 0]);
        if constexpr (MF32 == 2) {                       // v_mfma_i32_16x16x64_i8: twice the MACs per instruction, same operand bytes
          acci[(e % 4) * 4 + rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af), __builtin_bit_cast(i32x4, bf),
                                                                          acci[(e % 4) * 4 + rt], 0, 0, 0);
        } else if constexpr (MF32 == 1) {
          if (rt < 2) acc32[(e % 2) * 2 + rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc32[(e % 2) * 2 + rt], 0, 0, 0);
        } else
        acc[(e % 4) * 4 + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[(e % 4) * 4 + rt], 0, 0, 0);
        if (m < 2 && WEVERY > 0 && e % WEVERY == 0 && !(DEDUP && en % 5 == 4)) {    // WEVERY = 2: a weight entry is fetched for every second entry only (the ring
          __builtin_amdgcn_sched_barrier(0);             // slot keeps its old bits otherwise); 0: no weight stream at all
          b[(e + NB - 1) % NB][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (en * 2 + m) * 1024, 0));
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
  float sink = sink0;
  for (int i = 0; i < 16; ++i) sink += acc[i][0] + acc[i][3];
  for (int i = 0; i < 4; ++i) sink += acc32[i][0] + acc32[i][15];
  for (int i = 0; i < 16; ++i) sink += (float)(acci[i][0] + acci[i][3]);
  if (sink == 12345.678f) out[threadIdx.x] = sink;
  if (lane == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    unsigned long long* r = clk + ((size_t)blockIdx.x * 4 * WAVES + wave) * kRec;
    r[0] = c0; r[1] = r0; r[2] = c1; r[3] = r1; r[4] = hwid; r[5] = ((unsigned long long)(re & 0xffffffffull) << 32) | xcc;
    (void)ce;
  }
}

static int g_sustain = 20;
static char *g_big = nullptr, *g_bigw = nullptr;
static FILE* g_tsv = nullptr;                             // one row per variant, machine-readable (argv[2])
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }
static double vmin(const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); }
static double vmax(const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); }

template <int WAVES, int SLEEP, int STREAM = 0, int MF32 = 0, int ORD = 0, int PRIO = 0, int WEVERY = 1, int PPE = 12, int DEDUP = 0, int NB = 4, int LOADER = 0>
static void run(const char* name, const char* w, const float* x, float* out, unsigned long long* clk, int steps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    hipLaunchKernelGGL((k<WAVES, SLEEP, STREAM, MF32, ORD, PRIO, WEVERY, PPE, DEDUP, NB, LOADER>), dim3(256), dim3(256 * WAVES), 0, 0, w, x, out, steps, clk, g_big, g_bigw);
  };
  // SUSTAIN launches back to back first (argv[1], default 20 = a 3.5 ms burst; 12000 = two seconds of continuous load, after
  // which the clock is the one the chip HOLDS under this load), then the timed 20
  for (int i = 0; i < g_sustain; ++i) launch();
  // socket power and shader clock while the queue is still full of these launches
  double watts = 0, mhz = 0;
  {
    for (int i = 0; i < 4000; ++i) launch();
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
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // the LAST of those launches left its stamps: one record per wave
  const int nw = 256 * 4 * WAVES;
  std::vector<unsigned long long> h((size_t)nw * kRec);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  const double prod_per_wave = (double)steps * kEntries * PPE * (DEDUP ? 0.8 : 1.0);   // products executed
  const double flop = 256.0 * 4 * WAVES * prod_per_wave * 16384.0 * (MF32 == 2 ? 1.0 : 1.0);
  unsigned long long first = ~0ull, last = 0;
  for (int i = 0; i < nw; ++i) { first = std::min(first, h[(size_t)i * kRec + 1]); last = std::max(last, h[(size_t)i * kRec + 3]); }
  std::vector<double> ghz, el_old, el_young, start_us, end_us, cyc_old, cyc_young;
  std::vector<int> cu_count(8 * 8 * 2 * 16, 0);
  for (int wg = 0; wg < 256; ++wg) {
    for (int wv = 0; wv < 4 * WAVES; ++wv) {
      const unsigned long long* r = h.data() + ((size_t)wg * 4 * WAVES + wv) * kRec;
      const double el = (double)(r[3] - r[1]) / 100.0;                          // s_memrealtime: 100 MHz -> us
      ghz.push_back((double)(r[2] - r[0]) / ((double)(r[3] - r[1]) * 10.0));
      (wv < 4 ? el_old : el_young).push_back(el);
      (wv < 4 ? cyc_old : cyc_young).push_back((double)(r[2] - r[0]) / prod_per_wave);
      start_us.push_back((double)(r[1] - first) / 100.0);
      end_us.push_back((double)(r[3] - first) / 100.0);
    }
    const unsigned long long* r = h.data() + ((size_t)wg * 4 * WAVES) * kRec;
    const unsigned hw = (unsigned)r[4], xcc = (unsigned)(r[5] & 0xf);
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cu_count[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
  }
  int cus_used = 0, cus_shared = 0, max_on_cu = 0;
  for (int c : cu_count) { cus_used += c > 0; cus_shared += c > 1; max_on_cu = std::max(max_on_cu, c); }
  const double wall_us = ms / reps * 1e3, span_us = (double)(last - first) / 100.0;
  const double tf_wall = flop / (wall_us * 1e-6) / 1e12, tf_span = flop / (span_us * 1e-6) / 1e12;
  // products of the SIMD per cycle over the SPAN (first start .. last end) at the median clock: the pipe's real fill
  const double g = med(ghz), cyc_span = span_us * 1e-6 * g * 1e9 / (prod_per_wave * WAVES);
  printf("%-58s wall %7.1f us  span %7.1f us (gap %5.1f)  %5.3f GHz  %6.0f W  sclk %4.0f\n", name, wall_us, span_us, wall_us - span_us, g, watts, mhz);
  printf("    TFLOP/s dense f16: wall %7.1f  span %7.1f  (f32-grade %5.1f / %5.1f)   cycles/product over the span %5.2f (pipe fill %4.2f)\n",
         tf_wall, tf_span, tf_wall / 3, tf_span / 3, cyc_span, 16.0 / cyc_span);
  printf("    starts: median +%5.1f us, last +%5.1f us;  ends: first +%6.1f, median +%6.1f, last +%6.1f us\n", med(start_us), vmax(start_us),
         vmin(end_us), med(end_us), vmax(end_us));
  printf("    wave elapsed us  older (0-3): %6.1f / %6.1f / %6.1f   cycles per OWN product %5.2f", vmin(el_old), med(el_old), vmax(el_old), med(cyc_old));
  if (WAVES == 2)
    printf("\n                     younger(4-7): %6.1f / %6.1f / %6.1f   cycles per OWN product %5.2f", vmin(el_young), med(el_young), vmax(el_young), med(cyc_young));
  printf("\n    CUs used %d, CUs holding more than one workgroup %d (max %d on one)\n", cus_used, cus_shared, max_on_cu);
  if (g_tsv)
    fprintf(g_tsv, "%s\t%d\t%.1f\t%.1f\t%.3f\t%.0f\t%.1f\t%.1f\t%.2f\t%.1f\t%.1f\t%d\t%d\n", name, WAVES, wall_us, span_us, g, watts, tf_wall, tf_span, cyc_span,
            med(el_old), WAVES == 2 ? med(el_young) : 0.0, cus_used, cus_shared);
  fflush(stdout);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1) g_sustain = atoi(argv[1]);
  if (argc > 2) {
    g_tsv = fopen(argv[2], "w");
    if (g_tsv) fprintf(g_tsv, "variant\twaves_per_simd\twall_us\tspan_us\tghz\twatts\ttf_wall\ttf_span\tcycles_per_product_span\tolder_us\tyounger_us\tcus_used\tcus_shared\n");
  }
  const bool full = argc > 3 && atoi(argv[3]) != 0;        // the round-4 list on top of the round-6 one
  printf("sustain: %d launches in front of the timed 20; stamps are those of the LAST timed launch\n", g_sustain);
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
  hipMalloc(&out, 4096); hipMalloc(&clk, (size_t)256 * 8 * kRec * 8);
  hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(wz, hz.data(), wbytes, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(xz, hxz.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const int steps = 26;                                  // ~ two launches' worth of products per wave
  hipMalloc(&g_big, (size_t)2048 * 1024 * 1024 + 4096); hipMalloc(&g_bigw, (size_t)2048 * 1024 * 1024 + (1 << 20));
  hipMemset(g_big, 0, (size_t)2048 * 1024 * 1024);
  run<1, 0>("1 wave/SIMD, random operands", w, x, out, clk, steps);
  run<1, 0>("1 wave/SIMD, ZERO operands (same instructions)", wz, xz, out, clk, steps);
  run<1, 0, 0, 0, 2>("1 wave/SIMD, random, term-outer (independent neighbours)", w, x, out, clk, steps);
  run<1, 0, 0, 0, 2>("1 wave/SIMD, ZERO, term-outer", wz, xz, out, clk, steps);
  run<2, 0>("2 waves/SIMD, random operands", w, x, out, clk, steps);
  run<2, 0>("2 waves/SIMD, ZERO operands", wz, xz, out, clk, steps);
  run<2, 0, 0, 0, 0, 1>("2 waves/SIMD, random, younger waves at s_setprio 1", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 2>("2 waves/SIMD, random, priority swapped every entry", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 2>("2 waves/SIMD, ZERO, priority swapped every entry", wz, xz, out, clk, steps);
  run<2, 0, 0, 0, 2>("2 waves/SIMD, random, term-outer", w, x, out, clk, steps);
  run<2, 0, 0, 0, 2>("2 waves/SIMD, ZERO, term-outer", wz, xz, out, clk, steps);
  // what bounds the fill: the weight stream (2 KB from L2 per wave and 12 products) halved / removed, the rest unchanged
  run<1, 0, 0, 0, 0, 0, 2>("1 wave/SIMD, random, weight entry fetched every 2nd entry", w, x, out, clk, steps);
  run<1, 0, 0, 0, 0, 0, 0>("1 wave/SIMD, random, NO weight stream", w, x, out, clk, steps);
  run<1, 0, 0, 0, 2, 0, 0>("1 wave/SIMD, random, NO weight stream, term-outer", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 0, 2>("2 waves/SIMD, random, weight entry fetched every 2nd entry", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 2, 2>("2 waves/SIMD, random, every 2nd entry, priority swapped", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 0, 0>("2 waves/SIMD, random, NO weight stream", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 0, 0>("2 waves/SIMD, ZERO, NO weight stream", wz, xz, out, clk, steps);
  run<2, 0, 0, 0, 0, 2, 0>("2 waves/SIMD, random, NO weight stream, priority swapped", w, x, out, clk, steps);
  run<2, 0, 0, 0, 2, 0, 0>("2 waves/SIMD, random, NO weight stream, term-outer", w, x, out, clk, steps);
  // a weight entry feeding 6 products instead of 12: what an eight-wave form of the 256->64 layer with 32-row waves would stream
  run<2, 0, 0, 0, 0, 0, 1, 6>("2 waves/SIMD, random, 6 products per weight entry", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 2, 1, 6>("2 waves/SIMD, random, 6 products per entry, priority swapped", w, x, out, clk, steps);
  run<1, 0, 0, 0, 0, 0, 1, 6>("1 wave/SIMD, random, 6 products per weight entry", w, x, out, clk, steps);
  // memory-level parallelism: the weight ring deeper (8 entries: 7 x 2 KB in flight per wave instead of 3)
  run<2, 0, 0, 0, 0, 0, 1, 12, 0, 8>("2 waves/SIMD, random, weight ring of 8 entries", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 2, 1, 12, 0, 8>("2 waves/SIMD, random, ring of 8, priority swapped", w, x, out, clk, steps);
  run<1, 0, 0, 0, 0, 0, 1, 12, 0, 8>("1 wave/SIMD, random, weight ring of 8 entries", w, x, out, clk, steps);
  run<1, 0, 0, 0, 0, 0, 1, 12, 0, 20>("1 wave/SIMD, random, weight ring of 20 entries", w, x, out, clk, steps);
  // who pays for a vector-memory instruction: loader waves beside consumer waves
  run<2, 0, 0, 0, 0, 0, 0, 12, 0, 4, 1>("2 waves/SIMD: waves 0-3 multiply (no weight stream), waves 4-7 only LOAD it (registers)", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 0, 0, 12, 0, 4, 2>("2 waves/SIMD: waves 0-3 multiply (no weight stream), waves 4-7 only LOAD it (LDS-DMA)", w, x, out, clk, steps);
  // the second dedup, modelled on this loop: the wall time of a launch is the figure (the useful work is the same)
  run<2, 0, 0, 0, 0, 0, 1, 12, 1>("2 waves/SIMD, random, DEDUP: 2 of 10 k-blocks as 2 KB loads from a 32 MB table", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0, 0, 1, 12, 2>("2 waves/SIMD, random, DEDUP: ... every address once (HBM)", w, x, out, clk, steps);
  run<2, 1>("2 waves/SIMD, random, s_sleep 1 per entry", w, x, out, clk, steps);
  run<2, 2>("2 waves/SIMD, random, s_sleep 2 per entry", w, x, out, clk, steps);
  run<2, 4>("2 waves/SIMD, random, s_sleep 4 per entry", w, x, out, clk, steps);
  run<2, 2, 0, 0, 0, 2>("2 waves/SIMD, random, priority swapped, s_sleep 2", w, x, out, clk, steps);
  run<2, 4, 0, 0, 0, 2>("2 waves/SIMD, random, priority swapped, s_sleep 4", w, x, out, clk, steps);
  run<1, 2>("1 wave/SIMD, random, s_sleep 2 per entry", w, x, out, clk, steps);
  run<1, 4>("1 wave/SIMD, random, s_sleep 4 per entry", w, x, out, clk, steps);
  run<2, 0, 1>("2 waves/SIMD, random + HBM stream (1 KB / 6 entries, store / 8)", w, x, out, clk, steps);
  run<2, 0, 1, 0, 0, 2>("2 waves/SIMD, random + HBM stream, priority swapped", w, x, out, clk, steps);
  run<1, 0, 1>("1 wave/SIMD, random + HBM stream", w, x, out, clk, steps);
  run<2, 0, 0, 0, 0>("2 waves/SIMD, random operands (again)", w, x, out, clk, steps);
  if (full) {
    run<2, 2, 1>("2 waves/SIMD, random + HBM stream, s_sleep 2", w, x, out, clk, steps);
    run<2, 0, 0, 0, 1>("2 waves/SIMD, random, products ordered hi*lo, hi*hi, lo*hi", w, x, out, clk, steps);
    run<2, 0, 0, 2>("2 waves/SIMD, random BYTES, v_mfma_i32_16x16x64_i8 (x2 for ops)", w, x, out, clk, steps);
    run<2, 0, 1, 2>("2 waves/SIMD, random bytes + HBM stream, i8 (x2 for ops)", w, x, out, clk, steps);
    run<2, 0, 0, 1>("2 waves/SIMD, random, 32x32x16 tiles (same FLOPs, half the operand reads)", w, x, out, clk, steps);
    run<2, 0, 1, 1>("2 waves/SIMD, random + HBM stream, 32x32x16 tiles", w, x, out, clk, steps);
    run<1, 0, 0, 1>("1 wave/SIMD, random, 32x32x16 tiles", w, x, out, clk, steps);
  }
  if (g_tsv) fclose(g_tsv);
  return 0;
}
