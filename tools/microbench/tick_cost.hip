// What does an instruction cost when it rides between v_mfma_f32_16x16x32_f16 of ONE wave per SIMD?
// The 192->128 Bi-LSTM kernel (nrv_lstm_f16x2s.h) issues one MFMA per 16-cycle "tick" with gate arithmetic, loads and
// scalar moves between them; in-kernel stamps (scripts/gpu_stamps.py, r04a) show ~5 cycles per non-MFMA instruction
// ON TOP of the 16 per MFMA.  This loop isolates each kind: per iteration 96 MFMAs in chains of CH products per
// accumulator, and per MFMA (fenced with sched_barrier) one of the side instructions below.
//   hipcc --offload-arch=gfx950 -O3 -o tick_cost tick_cost.hip && ./tick_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { S_NONE, S_FMA1, S_FMA2, S_FMA3, S_EXP1, S_EXP_FMA, S_SMOV1, S_SMOV2, S_DSW, S_DSR, S_LOAD12, S_LOAD12_SMOV, S_ACCRD, S_NOP1, S_CVT, S_EXP_RCP, S_FMA4, S_FMA_FIRST, S_ALOAD, S_ALOAD_FMA2, S_ADSR, S_ADSW, S_AACC, S_ALOAD4, S_ASTORE, S_ABUF, S_ADSR64, S_ADSW32, S_ADSW64, S_ADSW128, S_ADSR2, S_ABUF1, S_ADSW16_4, S_WAITONLY, S_ABUF_NOWAIT, S_ADSR_NOWAIT, S_ABUF_SPREAD, S_ABUF_MID, S_ADSW64_NOW, S_ABUF8, S_P_WAIT_IN, S_P_2LD_SAME, S_P_DSR_IN, S_P_DSW_IN, S_P_DSW_EDGE, S_P_FMA_IN, S_P_ACC_IN, S_P_ST_IN, S_P_ENTRY_OLD, S_P_ENTRY_NEW, S_P_ENTRY_NEW2, S_P_EXP_IN, S_P_EXP_EDGE, S_P_LD_LAST, S_P_ENTRY_NEW3 };

template <int SIDE, int CH, int NACC = 8>
__global__ void __launch_bounds__(256) k(float* out, const f32x4* __restrict__ src, int iters, unsigned seed,
                                         unsigned long long* clk) {
  __shared__ f32x4 lds[4 * 256];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = (threadIdx.x * 2654435761u + i * 40503u + j * 69069u + seed) >> 7;
      a[i][j] = (_Float16)(((h & 1023) - 512) * (1.0f / 512));
      b[i][j] = (_Float16)((((h >> 10) & 1023) - 512) * (1.0f / 512));
    }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float vv[4] = {1.0f, 2.0f, 3.0f, 4.0f}, v1 = 0.25f;
  f32x4 ld[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  int sc = seed, sc2 = seed * 3;
  const f32x4* p = src + (blockIdx.x & 63) * 65536 + threadIdx.x;
  float sink = 0;
  _Float16 hh = (_Float16)0.f;
  const char* gbase = (const char*)src + (size_t)(blockIdx.x & 3) * 640 * 1024 + (threadIdx.x >> 6) * 160 * 1024 + (threadIdx.x & 63) * 16;
  const char* gp = gbase;
  unsigned goff = 0;
  const unsigned ldsa = threadIdx.x * 16;
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i rsrc;
  {
    const unsigned long long ba = (unsigned long long)((const char*)src + (size_t)(blockIdx.x & 3) * 640 * 1024 + (threadIdx.x >> 6) * 160 * 1024);
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)(ba >> 32));
    rsrc[2] = 0x7fffffff;
    rsrc[3] = 0x00020000;
  }
  const unsigned boff = (threadIdx.x & 63) * 16;
  const float hv = 1.5f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 ld2 = {0.f, 0.f};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 96; ++m) {
      const int ai = (m / CH) % NACC;
      if ((SIDE == S_LOAD12 || SIDE == S_LOAD12_SMOV) && m % 12 == 0) {          // two 1 KiB loads per 12 MFMAs, consumed 84 MFMAs later
        sink += ld[0][0] + ld[1][1];
        ld[0] = p[((it * 16 + m / 12) & 255) * 256];
        ld[1] = p[((it * 16 + m / 12 + 8) & 255) * 256];
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SIDE == S_LOAD12 || SIDE == S_LOAD12_SMOV || SIDE == S_DSR || SIDE == S_ACCRD || SIDE == S_DSW) {
        // compiler-placed forms (an asm statement beside loads in flight would make hipcc wait vmcnt(0) in front of it)
        acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 3], b[(m * 3) & 3], acc[ai], 0, 0, 0);
        if (SIDE == S_LOAD12_SMOV) sc = sc * 3 + it;
        if (SIDE == S_DSW) ((_Float16*)lds)[threadIdx.x * 8 + (m & 7)] = (_Float16)vv[m & 3];
        if (SIDE == S_DSR && (m & 1)) sink += lds[(m & 3) * 256 + threadIdx.x][m & 3];
        if (SIDE == S_ACCRD) sink += acc[(ai + NACC / 2) % NACC][m & 3];
      } else {
        // ONE asm statement per 12 ticks (hipcc puts an s_nop behind every asm statement): the MFMAs and exactly the
        // side instructions named, in this order.  %0..%3 accumulators, %4..%7 vv, %8 sc, %9 a, %10 b, %11 v1
        if (m % 12 == 0) {
#define MFA(A) "v_mfma_f32_16x16x32_f16 " A ", %9, %10, " A "\n\t"
#define FMA0 "v_fma_f32 %4, %4, %11, %5\n\t"
#define FMA1 "v_fma_f32 %6, %6, %11, %7\n\t"
#define FMA2 "v_fma_f32 %5, %5, %11, %6\n\t"
#define FMA3 "v_fma_f32 %7, %7, %11, %4\n\t"
#define EXP "v_exp_f32 %7, %7\n\t"
#define RCPX "v_rcp_f32 %4, %5\n\t"
#define SM0 "s_mul_i32 %8, %8, 3\n\t"
#define SM1 "s_add_u32 %8, %8, 5\n\t"
#define CVTX "v_cvt_f16_f32 %4, %5\n\tv_fma_mix_f32 %6, %4, %11, %5 op_sel_hi:[1,0,0]\n\t"
#define T12(S, A0, A1, A2, A3, A4, A5, A6, A7, A8, A9, A10, A11)                                                     \
  asm volatile(MFA(A0) S MFA(A1) S MFA(A2) S MFA(A3) S MFA(A4) S MFA(A5) S MFA(A6) S MFA(A7) S MFA(A8) S MFA(A9) S MFA(A10) S MFA(A11) S \
               : "+a"(acc[(m / 12 * 4) % NACC]), "+a"(acc[(m / 12 * 4 + 1) % NACC]), "+a"(acc[(m / 12 * 4 + 2) % NACC]),          \
                 "+a"(acc[(m / 12 * 4 + 3) % NACC]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+s"(sc)                 \
               : "v"(a[(m / 12) & 3]), "v"(b[(m / 4) & 3]), "v"(v1))
#define TCH(S)                                                                                                       \
  do {                                                                                                               \
    if constexpr (CH == 1) T12(S, "%0", "%1", "%2", "%3", "%0", "%1", "%2", "%3", "%0", "%1", "%2", "%3");                 \
    else if constexpr (CH == 3) T12(S, "%0", "%0", "%0", "%1", "%1", "%1", "%2", "%2", "%2", "%3", "%3", "%3");            \
    else T12(S, "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0");                                   \
  } while (0)
          if constexpr (SIDE == S_NONE) TCH("");
          if constexpr (SIDE == S_FMA1) TCH(FMA0);
          if constexpr (SIDE == S_FMA2) TCH(FMA0 FMA1);
          if constexpr (SIDE == S_FMA3) TCH(FMA0 FMA1 FMA2);
          if constexpr (SIDE == S_FMA4) TCH(FMA0 FMA1 FMA2 FMA3);
          if constexpr (SIDE == S_EXP1) TCH(EXP);
          if constexpr (SIDE == S_EXP_FMA) TCH(EXP FMA0);
          if constexpr (SIDE == S_EXP_RCP) TCH(EXP RCPX);
          if constexpr (SIDE == S_SMOV1) TCH(SM0);
          if constexpr (SIDE == S_SMOV2) TCH(SM0 SM1);
          if constexpr (SIDE == S_NOP1) TCH("s_nop 0\n\t");
          if constexpr (SIDE == S_CVT) TCH(CVTX);
          if constexpr (SIDE == S_FMA_FIRST) TCH(FMA0 FMA1 "s_nop 7\n\t");
          // memory forms, asm too (no compiler waits): per 12 MFMAs two 1 KiB loads out of a 160 KB L2-resident stream per
          // wave (the kernel's weight entries), at most 16 in flight; %12 address pair, %13/%14 destinations, %15 LDS address
#define T12M(PRE, S)                                                                                                 \
  asm volatile(PRE MFB("%0") S MFB("%0") S MFB("%0") S MFB("%1") S MFB("%1") S MFB("%1") S MFB("%2") S MFB("%2") S MFB("%2") S MFB("%3") S MFB("%3") S MFB("%3") S \
               : "+a"(acc[(m / 12 * 4) % NACC]), "+a"(acc[(m / 12 * 4 + 1) % NACC]), "+a"(acc[(m / 12 * 4 + 2) % NACC]),          \
                 "+a"(acc[(m / 12 * 4 + 3) % NACC]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+s"(sc),                \
                 "+v"(ld[0]), "+v"(ld[1]), "+v"(ld2)                                                                              \
               : "v"(a[(m / 12) & 3]), "v"(b[(m / 4) & 3]), "v"(v1), "v"(gp), "v"(ldsa), "v"(hv), "v"(boff), "s"(rsrc), "s"(goff) \
               : "memory")
// operands of the memory forms: %9 %10 16-byte registers, %11 8-byte register, %12 a, %13 b, %14 v1, %15 global address,
// %16 LDS address, %17 a float, %18 buffer offset, %19 buffer resource, %20 scalar offset
#define MFB(A) "v_mfma_f32_16x16x32_f16 " A ", %12, %13, " A "\n\t"
#define FMB0 "v_fma_f32 %4, %4, %14, %5\n\t"
#define FMB1 "v_fma_f32 %6, %6, %14, %7\n\t"
#define LD2 "global_load_dwordx4 %9, %15, off\n\tglobal_load_dwordx4 %10, %15, off offset:1024\n\ts_waitcnt vmcnt(14)\n\t"
#define LD4 LD2 "global_load_dwordx4 %9, %15, off offset:2048\n\tglobal_load_dwordx4 %10, %15, off offset:3072\n\ts_waitcnt vmcnt(28)\n\t"
          if constexpr (SIDE == S_ALOAD) T12M(LD2, "");
          if constexpr (SIDE == S_ALOAD4) T12M(LD4, "");
          if constexpr (SIDE == S_ALOAD_FMA2) T12M(LD2, FMB0 FMB1);
          if constexpr (SIDE == S_ADSR) T12M("ds_read_b128 %9, %16\n\ts_waitcnt lgkmcnt(4)\n\t", "");
          if constexpr (SIDE == S_ADSW) T12M("", "ds_write_b16 %16, %17\n\t");
          if constexpr (SIDE == S_AACC) T12M("", "v_accvgpr_read_b32 %4, a100\n\t");
          if constexpr (SIDE == S_ASTORE) T12M("global_store_dwordx4 %15, %9, off offset:2048\n\ts_waitcnt vmcnt(14)\n\t", "");
          if constexpr (SIDE == S_ABUF) T12M("buffer_load_dwordx4 %9, %18, %19, %20 offen\n\tbuffer_load_dwordx4 %10, %18, %19, %20 offen offset:1024\n\ts_waitcnt vmcnt(14)\n\t", "");
          if constexpr (SIDE == S_ABUF1) T12M("buffer_load_dwordx4 %9, %18, %19, %20 offen\n\ts_waitcnt vmcnt(14)\n\t", "");
          if constexpr (SIDE == S_ADSR64) T12M("ds_read_b64 %11, %16\n\ts_waitcnt lgkmcnt(4)\n\t", "");
          if constexpr (SIDE == S_ADSR2) T12M("ds_read_b128 %9, %16\n\tds_read_b128 %10, %16 offset:4096\n\ts_waitcnt lgkmcnt(4)\n\t", "");
          if constexpr (SIDE == S_ADSW32) T12M("", "ds_write_b32 %16, %17\n\t");
          if constexpr (SIDE == S_ADSW64) T12M("ds_write_b64 %16, %11\n\t", "");
          if constexpr (SIDE == S_ADSW128) T12M("ds_write_b128 %16, %9\n\t", "");
          if constexpr (SIDE == S_ADSW16_4) T12M("ds_write_b16 %16, %17\n\tds_write_b16 %16, %17 offset:2\n\tds_write_b16 %16, %17 offset:4\n\tds_write_b16 %16, %17 offset:6\n\t", "");
          if constexpr (SIDE == S_WAITONLY) T12M("s_waitcnt vmcnt(14)\n\t", "");
          if constexpr (SIDE == S_ABUF_NOWAIT) T12M("buffer_load_dwordx4 %9, %18, %19, %20 offen\n\tbuffer_load_dwordx4 %10, %18, %19, %20 offen offset:1024\n\t", "");
          if constexpr (SIDE == S_ADSR_NOWAIT) T12M("ds_read_b128 %9, %16\n\t", "");
          if constexpr (SIDE == S_ABUF8) T12M("buffer_load_dwordx4 %9, %18, %19, %20 offen\n\tbuffer_load_dwordx4 %10, %18, %19, %20 offen offset:1024\n\tbuffer_load_dwordx4 %9, %18, %19, %20 offen offset:2048\n\tbuffer_load_dwordx4 %10, %18, %19, %20 offen offset:3072\n\t", "");
          if constexpr (SIDE == S_ABUF_SPREAD || SIDE == S_ABUF_MID) {
            // the two loads apart: behind MFMA 0 and MFMA 6 (SPREAD: chain boundaries), or behind MFMA 1 and MFMA 7 (MID: inside a chain)
#define BL0 "buffer_load_dwordx4 %9, %18, %19, %20 offen\n\t"
#define BL1 "buffer_load_dwordx4 %10, %18, %19, %20 offen offset:1024\n\t"
#define T12X(P0, P1, P2, P6, P7)                                                                                     \
  asm volatile(P0 MFB("%0") P1 MFB("%0") P2 MFB("%0") MFB("%1") MFB("%1") MFB("%1") P6 MFB("%2") P7 MFB("%2") MFB("%2") MFB("%3") MFB("%3") MFB("%3") \
               : "+a"(acc[(m / 12 * 4) % NACC]), "+a"(acc[(m / 12 * 4 + 1) % NACC]), "+a"(acc[(m / 12 * 4 + 2) % NACC]),          \
                 "+a"(acc[(m / 12 * 4 + 3) % NACC]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+s"(sc),                \
                 "+v"(ld[0]), "+v"(ld[1]), "+v"(ld2)                                                                              \
               : "v"(a[(m / 12) & 3]), "v"(b[(m / 4) & 3]), "v"(v1), "v"(gp), "v"(ldsa), "v"(hv), "v"(boff), "s"(rsrc), "s"(goff) \
               : "memory")
            if constexpr (SIDE == S_ABUF_SPREAD) T12X(BL0, "", "", BL1, "");
            else T12X("", BL0, "", "", BL1);
          }
          // placement forms: P(i) is what stands BEHIND MFMA i of the 12 (chains of three: 0-2, 3-5, 6-8, 9-11; so
          // i % 3 == 0, 1 are INSIDE a chain - the next MFMA depends on this one -, i % 3 == 2 is a chain edge)
#define T12P(P0, P1, P2, P3, P4, P5, P6, P7, P8, P9, P10, P11)                                                      \
  asm volatile(MFB("%0") P0 MFB("%0") P1 MFB("%0") P2 MFB("%1") P3 MFB("%1") P4 MFB("%1") P5 MFB("%2") P6 MFB("%2") P7 MFB("%2") P8 MFB("%3") P9 MFB("%3") P10 MFB("%3") P11 \
               : "+a"(acc[(m / 12 * 4) % NACC]), "+a"(acc[(m / 12 * 4 + 1) % NACC]), "+a"(acc[(m / 12 * 4 + 2) % NACC]),          \
                 "+a"(acc[(m / 12 * 4 + 3) % NACC]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+s"(sc),                \
                 "+v"(ld[0]), "+v"(ld[1]), "+v"(ld2)                                                                              \
               : "v"(a[(m / 12) & 3]), "v"(b[(m / 4) & 3]), "v"(v1), "v"(gp), "v"(ldsa), "v"(hv), "v"(boff), "s"(rsrc), "s"(goff) \
               : "memory")
#define WT "s_waitcnt vmcnt(14)\n\t"
#define DR "ds_read_b128 %9, %16\n\t"
#define DW "ds_write_b16 %16, %17\n\t"
#define AR "v_accvgpr_read_b32 %4, a100\n\t"
#define ST "global_store_dwordx4 %15, %9, off offset:2048\n\t"
#define EX "v_exp_f32 %7, %7\n\t"
#define F2 FMB0 FMB1
          if constexpr (SIDE == S_P_WAIT_IN) T12P("", WT, "", "", "", "", "", "", "", "", "", "");
          if constexpr (SIDE == S_P_2LD_SAME) T12P(BL0, BL1, "", "", "", "", "", "", "", "", "", "");
          if constexpr (SIDE == S_P_LD_LAST) T12P("", "", BL0, "", "", "", "", "", BL1, "", "", "");
          if constexpr (SIDE == S_P_DSR_IN) T12P("", DR, "", "", "", "", "", "", "", "", "", "");
          if constexpr (SIDE == S_P_DSW_IN) T12P(DW, "", "", DW, "", "", DW, "", "", DW, "", "");
          if constexpr (SIDE == S_P_DSW_EDGE) T12P("", "", DW, "", "", DW, "", "", DW, "", "", DW);
          if constexpr (SIDE == S_P_FMA_IN) T12P(F2 FMB0, F2 FMB0, "", F2 FMB0, F2 FMB0, "", F2 FMB0, F2 FMB0, "", F2 FMB0, F2 FMB0, "");
          if constexpr (SIDE == S_P_ACC_IN) T12P(AR, AR, "", AR, AR, "", AR, AR, "", AR, AR, "");
          if constexpr (SIDE == S_P_ST_IN) T12P("", ST, "", "", "", "", "", "", "", "", "", "");
          if constexpr (SIDE == S_P_EXP_IN) T12P(EX FMB0, "", "", EX FMB0, "", "", EX FMB0, "", "", EX FMB0, "", "");
          if constexpr (SIDE == S_P_EXP_EDGE) T12P("", "", EX FMB0, "", "", EX FMB0, "", "", EX FMB0, "", "", EX FMB0);
          // a whole weight entry of the Bi-LSTM kernel: two loads, one counted wait, twelve products, a gate piece per tick
          if constexpr (SIDE == S_P_ENTRY_OLD)
            asm volatile(BL0 BL1 WT MFB("%0") F2 MFB("%0") F2 MFB("%0") F2 MFB("%1") F2 MFB("%1") F2 MFB("%1") F2 MFB("%2") F2 MFB("%2") F2 MFB("%2") F2 MFB("%3") F2 MFB("%3") F2 MFB("%3") F2
               : "+a"(acc[(m / 12 * 4) % NACC]), "+a"(acc[(m / 12 * 4 + 1) % NACC]), "+a"(acc[(m / 12 * 4 + 2) % NACC]),
                 "+a"(acc[(m / 12 * 4 + 3) % NACC]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+s"(sc),
                 "+v"(ld[0]), "+v"(ld[1]), "+v"(ld2)
               : "v"(a[(m / 12) & 3]), "v"(b[(m / 4) & 3]), "v"(v1), "v"(gp), "v"(ldsa), "v"(hv), "v"(boff), "s"(rsrc), "s"(goff)
               : "memory");
          if constexpr (SIDE == S_P_ENTRY_NEW) T12P(BL0 F2, F2, F2, BL1 F2, F2, F2, F2, F2, F2, F2, WT F2, F2);
          if constexpr (SIDE == S_P_ENTRY_NEW2) T12P(BL0, F2 F2, F2, BL1, F2 F2, F2, F2, F2, F2, WT, F2 F2, F2);
          if constexpr (SIDE == S_P_ENTRY_NEW3) T12P(BL0, F2, F2 F2, BL1, F2, F2 F2, F2, F2, F2, WT, F2, F2 F2);
          if constexpr (SIDE == S_ADSW64_NOW) T12M("", "ds_write_b64 %16, %11\n\t");
          if constexpr (SIDE >= S_ALOAD) {
            goff = goff + 2048 < 160 * 1024 ? goff + 2048 : 0;
            gp = gbase + goff;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  float s = sink + vv[0] + vv[1] + vv[2] + vv[3] + sc + sc2 + (float)hh;
  for (int i = 0; i < NACC; ++i) s += acc[i][threadIdx.x & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  float* d;
  (void)hipMalloc(&d, 256 * 256 * 4 * 4);
  f32x4* src;
  (void)hipMalloc(&src, 64u * 65536 * 16 + 65536 * 16);
  (void)hipMemset(src, 0x3c, 64u * 65536 * 16 + 65536 * 16);
  unsigned long long* clk;
  (void)hipHostMalloc((void**)&clk, 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  auto run = [&](const char* name, auto launch) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      launch(iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s %8.3f ms  %6.2f cycles/MFMA  %.3f GHz\n", name, ms, (double)clk[0] / iters / 96, (double)clk[0] / clk[1] / 10);
  };
#define RUN(NAME, S, CH) run(NAME, [&](int it) { hipLaunchKernelGGL((k<S, CH>), dim3(256), dim3(256), 0, 0, d, src, it, 1u, clk); })
  RUN("MFMA only, chains of 1 (8 acc)", S_NONE, 1);
  RUN("MFMA only, chains of 3", S_NONE, 3);
  RUN("MFMA only, chains of 12", S_NONE, 12);
  RUN("chains of 3 + 1 v_fma", S_FMA1, 3);
  RUN("chains of 3 + 2 v_fma", S_FMA2, 3);
  RUN("chains of 3 + 3 v_fma", S_FMA3, 3);
  RUN("chains of 1 + 2 v_fma", S_FMA2, 1);
  RUN("chains of 3 + 1 v_exp", S_EXP1, 3);
  RUN("chains of 3 + v_exp + v_fma", S_EXP_FMA, 3);
  RUN("chains of 3 + 1 s_mul/add", S_SMOV1, 3);
  RUN("chains of 3 + 2 s_mul/add", S_SMOV2, 3);
  RUN("chains of 3 + s_nop 0", S_NOP1, 3);
  RUN("chains of 3 + 4 v_fma", S_FMA4, 3);
  RUN("chains of 3 + v_exp + v_rcp", S_EXP_RCP, 3);
  RUN("chains of 3 + 2 v_fma + s_nop 7", S_FMA_FIRST, 3);
  RUN("chains of 12 + 2 v_fma", S_FMA2, 12);
  RUN("asm: 2 x 1 KiB L2 loads per 12 MFMAs", S_ALOAD, 3);
  RUN("asm: 4 x 1 KiB L2 loads per 12 MFMAs", S_ALOAD4, 3);
  RUN("asm: 2 loads per 12 + 2 v_fma per MFMA", S_ALOAD_FMA2, 3);
  RUN("asm: 1 ds_read_b128 per 12 MFMAs", S_ADSR, 3);
  RUN("asm: 1 ds_write_b16 per MFMA", S_ADSW, 3);
  RUN("asm: 1 v_accvgpr_read per MFMA", S_AACC, 3);
  RUN("asm: 1 x 1 KiB store per 12 MFMAs", S_ASTORE, 3);
  RUN("asm: 2 x buffer_load_dwordx4 (32-bit off) per 12", S_ABUF, 3);
  RUN("asm: 1 x buffer_load_dwordx4 per 12", S_ABUF1, 3);
  RUN("asm: 1 ds_read_b64 per 12 MFMAs", S_ADSR64, 3);
  RUN("asm: 2 ds_read_b128 per 12 MFMAs", S_ADSR2, 3);
  RUN("asm: 1 ds_write_b32 per MFMA", S_ADSW32, 3);
  RUN("asm: 1 ds_write_b64 per 12 MFMAs", S_ADSW64, 3);
  RUN("asm: 1 ds_write_b128 per 12 MFMAs", S_ADSW128, 3);
  RUN("asm: 4 ds_write_b16 per 12 MFMAs", S_ADSW16_4, 3);
  RUN("asm: s_waitcnt vmcnt(14) alone per 12", S_WAITONLY, 3);
  RUN("asm: 2 buffer_load per 12, NO waitcnt", S_ABUF_NOWAIT, 3);
  RUN("asm: 1 ds_read_b128 per 12, NO waitcnt", S_ADSR_NOWAIT, 3);
  RUN("asm: 2 buffer_load per 12 at chain edges 0 / 6", S_ABUF_SPREAD, 3);
  RUN("asm: 2 buffer_load per 12 inside chains 1 / 7", S_ABUF_MID, 3);
  RUN("asm: 1 ds_write_b64 per MFMA", S_ADSW64_NOW, 3);
  RUN("place: s_waitcnt inside a chain", S_P_WAIT_IN, 3);
  RUN("place: 2 loads inside the SAME chain (0, 1)", S_P_2LD_SAME, 3);
  RUN("place: 2 loads behind chain ends (2, 8)", S_P_LD_LAST, 3);
  RUN("place: 1 ds_read_b128 inside a chain", S_P_DSR_IN, 3);
  RUN("place: 4 ds_write_b16 inside chains (0,3,6,9)", S_P_DSW_IN, 3);
  RUN("place: 4 ds_write_b16 at chain ends (2,5,8,11)", S_P_DSW_EDGE, 3);
  RUN("place: 3 v_fma inside chains only (24 per 12)", S_P_FMA_IN, 3);
  RUN("place: 8 v_accvgpr_read inside chains", S_P_ACC_IN, 3);
  RUN("place: 1 store inside a chain", S_P_ST_IN, 3);
  RUN("place: 4 (v_exp + v_fma) inside chains (0,3,6,9)", S_P_EXP_IN, 3);
  RUN("place: 4 (v_exp + v_fma) at chain ends", S_P_EXP_EDGE, 3);
  RUN("entry, r03 form: 2 loads + wait in front, 2 v_fma per tick", S_P_ENTRY_OLD, 3);
  RUN("entry, loads at 0 / 3, wait at 10, 2 v_fma per tick", S_P_ENTRY_NEW, 3);
  RUN("entry, loads / wait alone in their tick, v_fma moved on", S_P_ENTRY_NEW2, 3);
  RUN("entry, the same with the v_fma moved to the chain end", S_P_ENTRY_NEW3, 3);
  RUN("chains of 3 + cvt f16 + ds_write_b16", S_DSW, 3);
  RUN("chains of 3 + v_cvt_f16 + v_fma_mix", S_CVT, 3);
  RUN("chains of 3 + ds_read_b128 every 2nd", S_DSR, 3);
  RUN("chains of 3 + accvgpr read + add", S_ACCRD, 3);
  RUN("chains of 3 + 2 x 1 KiB L2 loads per 12", S_LOAD12, 3);
  RUN("chains of 3 + loads + 1 salu per MFMA", S_LOAD12_SMOV, 3);
  return 0;
}
