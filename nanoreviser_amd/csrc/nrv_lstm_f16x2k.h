// f16x2 Bi-LSTM layer with H = 64, EIGHT waves per workgroup: the reduction dimension split between the two waves of a SIMD.
#pragma once
#include "nrv_lstm_f16x2s.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// lstm_h2k_kernel: the 256 -> 64 layer with two waves per SIMD.
//
// Why (round 4): lstm_h2s_kernel runs this layer with ONE wave per SIMD (four unit groups of 16 = four waves
// for 64 rows) and hides the gate arithmetic between the products of the next step's input projection, which makes that
// phase issue-bound: 11.9 k cycles per step against 7.7 k of matrix pipe (in-kernel stamps, profiles/r04al_stamps_lstm_h2s_lstm4).
// lstm_h2w_kernel's remedy - two waves per SIMD that own different units - needs eight unit groups; H = 64 has four.
//
// Here the two waves of a SIMD own the SAME 16 units x 4 gates x 64 rows and split the REDUCTION: wave w (group A, w < 4)
// multiplies the first KA input k-blocks and the recurrent ones and does the gate arithmetic, wave w + 4 (group B) multiplies
// the other input k-blocks - which depend on no h - and hands its 16 partial accumulator tiles over through LDS:
//     group A:  rec(s): Z += h_{s-1} U;  [requests for x_{s+1}, A's k-blocks]  gates(s) from Z + B's tiles -> h_s;  A's part
//               of x_{s+1} -> LDS            | barrier M |  in_A(s+1): Z = x_{s+1}[0..KA) W;  copy-out of h_s      | barrier E
//     group B:  in_B(s+1): Zb = x_{s+1}[KA..) W;  [requests for x_{s+2}, B's k-blocks]
//                                            | barrier M |  Zb -> LDS;  B's part of x_{s+2} -> LDS;  copy-out      | barrier E
// Every buffer is single (x: 64 KB, B's tiles: 64 KB) and changes hands at a barrier: A's part of x is written in the first
// half and read in the second, B's part and B's tiles the other way round; only h has two images (gates(s) write h_s while
// other waves still read h_{s-1}): 32 + 64 + 64 = 160 KB of LDS, all of it.  While group A does gate arithmetic group B feeds
// the matrix pipe; group B has no vector work at all.  Requests ride behind products inside chains of three, staging requests
// go out where a wave waits for no weight entry (group A: at the start of its gates; group B: behind its last product), as
// in lstm_h2w_kernel.  The sum of a tile is (x[0..KA) W + h U) + x[KA..) W: another order of f32 additions than
// lstm_h2s_kernel's (not bit-identical to it; same products).
//
// Operand layouts, scales, packed weights (pack_lstm_h2s, one unit half per wave) and the raw copy-out of h x 2^13 are
// lstm_h2s_kernel's.
// ---------------------------------------------------------------------------------------
#ifndef NRV_L4_CPRIDE
#define NRV_L4_CPRIDE 0
#endif
#ifndef NRV_L4_BPRIO
#define NRV_L4_BPRIO 2
#endif
#if NRV_STAMP
// diagnostic build: s_memtime stamps [wave 8][step 15][slot 16] per workgroup in the 256->64 layer's region of the stamp buffer,
// by scalar stores (the kernel's LDS is full); scripts/gpu_stamps_k.py
#define NRV_STAMP_K(slot)                                                                                    \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(t_), "s"(stamp_base), "s"((unsigned)((stamp_step * 16 + (slot)) * 8)) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#else
#define NRV_STAMP_K(slot) do { } while (0)
#endif

template <int KQ0, int H, int ACT, int NBG, int KA>
__global__ void __launch_bounds__(512)
lstm_h2k_kernel(const LstmH2Args args) {
  constexpr int NG = H / 16;                                   // unit groups of 16
  static_assert(NG == 4, "four unit groups x two reduction halves = eight waves: H = 64");
  constexpr int KK_IN = KQ0 / 8, KK_REC = H / 32, KK = KK_IN + KK_REC, KB = KK_IN - KA;
  constexpr int R = 2, RT = 2 * R, EPK = 4, ROWS = 32 * R;
  constexpr int GS = ROWS * 8 + 8, TERM = (H / 8) * GS, HBUF = TERM, NTHREADS = 512;
  constexpr int NAS = 6, LBG = NBG - 1;
  constexpr int NE = RT * 4, XF = 256;
  constexpr int SEQ_A = EPK * (KK_REC + KA), SEQ_B = EPK * KB;  // weight entries per step, in the order a group uses them
  static_assert(KA >= 1 && KB >= 1 && SEQ_A % NBG == 0 && SEQ_B % NBG == 0 && LBG < SEQ_B && LBG < EPK * KK_REC, "weight ring");
  static_assert(RT == NG, "staging: wave w of a group fetches row tile w");
  static_assert((EPK * KK_REC) % NBG == 0, "A's input blocks start on ring slot 0");
  __shared__ __attribute__((aligned(16))) float hbuf[HBUF];                      // the split image of h (hi | lo)
  __shared__ __attribute__((aligned(16))) float zst[4 * 16 * 64 * 4];           // group B's 16 partial tiles per wave
  __shared__ __attribute__((aligned(16))) float xst[KK_IN * RT * 2 * XF];        // x: fragment (kk, rt, term) lane-linear

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if NRV_STAMP
  int stamp_step = 0;
  unsigned long long* const stamp_base =
      &nrv_stamp_buf[1][blockIdx.x < kStampBlocks ? blockIdx.x : kStampBlocks - 1][0][0][0] + wave * (kStampSteps * 16);
  unsigned long long stamp_rt0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt0)::"memory");
#endif
  const bool grp_a = wave < 4;
  const int ug = wave & 3;
  const int l15 = lane & 15, kq = lane >> 4;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmH2ModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS;

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + ug) * KK) * (EPK * 2 * 1024), KK * EPK * 2 * 1024);
  const unsigned wlane = lane * 16;
  const int u0 = ug * 16 + l15;                                // this lane's unit
  float* const zw = zst + (size_t)((ug * 16) * 64 + lane) * 4;                      // tile (g, rt) at + (g RT + rt) 256
  float* const xw = xst + (size_t)((ug * 2) * 64 + lane) * 4;                       // row tile ug: + (kk RT 2 + term) XF
  const float* const xr = xst + (size_t)lane * 4;                                    // fragment (kk, rt, term): + ((kk RT + rt) 2 + term) XF

  for (int i = threadIdx.x; i < HBUF; i += NTHREADS) hbuf[i] = 0.f;              // h_{-1} = 0 (rec(0) multiplies it: exact)

  // ---- staging of x_s: wave w of a group fetches row tile w, both terms, of its group's k-blocks
  struct SBase {
    __amdgpu_buffer_rsrc_t r0;
    unsigned v0;
  };
  auto mk_stage = [&](int s) __attribute__((always_inline)) {
    const int sc = s < T ? s : T - 1;                    // past the end: staged again, read by nobody
    const int t = dir ? (T - 1 - sc) : sc;
    SBase sb;
    sb.r0 = make_rsrc(P.in0.ubase(row0 + (ug >> 1) * 32, t), 0xffffffffu);
    sb.v0 = (P.in0.voff(row0 + (ug >> 1) * 32, t, l15 + 16 * (ug & 1), kq & 1) + (kq >> 1) * 512) * 4;
    return sb;
  };
  constexpr int NXS = 2 * (KA > KB ? KA : KB);
  f32x4 xs[NXS];                                               // a group's staged fragments: (block j of its part, term)
  auto stage_load = [&](const SBase& sb, int k0, int nk) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < nk; ++j)
#pragma unroll
      for (int term = 0; term < 2; ++term) xs[2 * j + term] = buf_load16(sb.r0, sb.v0, (k0 + j) * 4096 + term * 1024);
  };
  auto stage_store1 = [&](int k0, int i) __attribute__((always_inline)) {          // i = 2 j + term
    *(f32x4*)(xw + ((k0 + i / 2) * RT * 2 + (i & 1)) * XF) = xs[i];
  };

  struct BReg { f16x8 t[2]; };
  struct AReg { f32x4 v[2]; };
  BReg b[NBG];
  AReg a[NAS];                                                 // fragment (block kl of the phase, row tile rt) in slot (4 kl + rt) % 6
  auto loadB1 = [&](int e, int term, BReg& bb) __attribute__((always_inline)) {
    bb.t[term] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2 + term) * 1024));
  };
  // weight entry of position p of a group's step: A: the recurrent blocks, then its input blocks; B: its input blocks
  auto seq_a = [&](int p) __attribute__((always_inline)) { return p < EPK * KK_REC ? EPK * KK_IN + p : p - EPK * KK_REC; };
  auto seq_b = [&](int p) __attribute__((always_inline)) { return EPK * KA + p; };
  auto loadA_in1 = [&](int kk, int rt, int term, AReg& d) __attribute__((always_inline)) {
    d.v[term] = *(const f32x4*)(xr + ((kk * RT + rt) * 2 + term) * XF);
  };
  auto loadA_rec1 = [&](const _Float16* hp, int kkr, int rt, int term, AReg& d) __attribute__((always_inline)) {
    d.v[term] = *(const f32x4*)(hp + kkr * 4 * GS + rt * 128 + term * TERM);
  };
  // next block's fragment pieces by entry and product (as in lstm_h2w_kernel): rt' = 0, 1 behind products 4, 5 of entries 0, 1
  // (the ring's two free slots), rt' = 2, 3 behind products 4, 5 / 7, 8 of entry 3 (the slots this block's rt 0, 1 just left)
  auto frag_piece = [&](int g, int m, int& rt_next, int& term) __attribute__((always_inline)) {
    rt_next = -1;
    term = 0;
    if ((g == 0 || g == 1) && (m == 4 || m == 5)) { rt_next = g; term = m - 4; }
    if (g == EPK - 1 && (m == 4 || m == 5)) { rt_next = 2; term = m - 4; }
    if (g == EPK - 1 && (m == 7 || m == 8)) { rt_next = 3; term = m - 7; }
  };
  constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // hi*lo, lo*hi, hi*hi

  // ---- copy-out of h_s x 2^13 as it lies in the image: one item per thread (a read of the two term planes, later their
  // stores through a descriptor on this workgroup's first output tile)
  constexpr int KBH = H / 16;
  static_assert((H / 8) * ROWS == NTHREADS, "copy-out: one item per thread");
  const int c_kbh = threadIdx.x / ROWS, c_rr = threadIdx.x % ROWS;
  const __amdgpu_buffer_rsrc_t ors = make_rsrc(P.out + (size_t)(blk.rowblk * R) * T * (2 * H / 4) * 128, 0xffffffffu);
  const unsigned cw_off =
      (unsigned)((((c_rr / 32) * T * (2 * H / 4) + (dir * KBH + (c_kbh >> 1)) * 4 + (c_kbh & 1)) * 128 + (c_rr & 31) * 4) * 4);
  f16x8 chi, clo;
  auto copy_read = [&](const _Float16* himg) __attribute__((always_inline)) {
    chi = *(const f16x8*)(himg + c_kbh * GS + c_rr * 8);
    clo = *(const f16x8*)(himg + c_kbh * GS + c_rr * 8 + TERM);
  };
  auto copy_write1 = [&](int term, int t) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)t * ((2 * H / 4) * 128 * 4) + term * (2 * 128 * 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, term == 0 ? chi : clo), ors, cw_off, soff, 0);
  };
  auto t_of = [&](int s) __attribute__((always_inline)) { return dir ? (T - 1 - s) : s; };

  // ======================================= group A: all recurrent products + KA input blocks, no vector work ===============
  auto run_a = [&]() __attribute__((always_inline)) {
    const int hp_off = kq * GS + l15 * 8;                                         // rec: + kkr 4 GS + rt 128; lo: + TERM
    f32x4 Z[4][RT];
    // rec(s): positions 0 .. EPK KK_REC - 1 of A's sequence; in the last block the tiles of an entry are final when it is
    // through and are parked in LDS (for group B's gates) behind products of the next one; the staged fragments of A's part of
    // the next x (requested behind the last product of in_A) go to LDS behind products of the first entries
    auto rec_phase = [&](const _Float16* hp, int t_out) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        loadA_rec1(hp, 0, rt, 0, a[rt]);
        loadA_rec1(hp, 0, rt, 1, a[rt]);
      }
#pragma unroll
      for (int kr = 0; kr < KK_REC; ++kr) {
        if (kr == 1) NRV_STAMP_K(10);
#pragma unroll
        for (int g = 0; g < EPK; ++g) {
          const int p = EPK * kr + g;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 3 * RT; ++m) {
            const int rt = m / 3, pr = m % 3;
            Z[g][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[(4 * kr + rt) % NAS].v[PA[pr]]), b[p % NBG].t[PB[pr]], Z[g][rt]);
            int rn, tn;
            frag_piece(g, m, rn, tn);
            const bool frag = rn >= 0 && kr + 1 < KK_REC;
            const bool park = kr == KK_REC - 1 && g > 0 && (m == 4 || m == 5 || m == 7 || m == 8);
            const bool cp = NRV_L4_CPRIDE && ((p == 1 && m == 10) || (p == 3 && m >= 10));   // copy-out of h_{s-1} (this thread's item)
            if (m < 2 || frag || park || cp) {
              __builtin_amdgcn_sched_barrier(0);
              if (m < 2) loadB1(seq_a((p + LBG) % SEQ_A), m, b[(p + LBG) % NBG]);
              if (frag) loadA_rec1(hp, kr + 1, rn, tn, a[(4 * (kr + 1) + rn) % NAS]);
              if (park) {
                const int rs = m == 4 ? 0 : m == 5 ? 1 : m == 7 ? 2 : 3;
                *(f32x4*)(zw + ((g - 1) * RT + rs) * 256) = Z[g - 1][rs];
              }
              if (cp && p == 1) copy_read((const _Float16*)hbuf);
              if (cp && p == 3) copy_write1(m - 10, t_out);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *(f32x4*)(zw + ((EPK - 1) * RT + rt) * 256) = Z[EPK - 1][rt];
      if (!NRV_L4_CPRIDE) {                              // (stores in front of weight requests hold the products behind them back)
        copy_read((const _Float16*)hbuf);
        copy_write1(0, t_out);
        copy_write1(1, t_out);
      }
    };
    // in_A(s+1): positions EPK KK_REC .. SEQ_A - 1
    auto in_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        loadA_in1(0, rt, 0, a[rt]);
        loadA_in1(0, rt, 1, a[rt]);
      }
#pragma unroll
      for (int kk = 0; kk < KA; ++kk) {
        if (kk > 0 && kk < 4) NRV_STAMP_K(10 + kk);
#pragma unroll
        for (int g = 0; g < EPK; ++g) {
          const int p = EPK * KK_REC + EPK * kk + g;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 3 * RT; ++m) {
            const int rt = m / 3, pr = m % 3;
            Z[g][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[(4 * kk + rt) % NAS].v[PA[pr]]), b[p % NBG].t[PB[pr]],
                                  (kk == 0 && pr == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : Z[g][rt]);
            int rn, tn;
            frag_piece(g, m, rn, tn);
            const bool frag = rn >= 0 && kk + 1 < KA;
            if (m < 2 || frag) {
              __builtin_amdgcn_sched_barrier(0);
              if (m < 2) loadB1(seq_a((p + LBG) % SEQ_A), m, b[(p + LBG) % NBG]);
              if (frag) loadA_in1(kk + 1, rn, tn, a[(4 * (kk + 1) + rn) % NAS]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    auto t_prev = [&](int s) __attribute__((always_inline)) { return t_of(s > 0 ? s - 1 : 0); };   // (s = 0: zeros to slot 0, overwritten later)
    {
      stage_load(mk_stage(0), 0, KA);
#pragma unroll
      for (int p = 0; p < LBG; ++p) loadB1(seq_a(EPK * KK_REC + p), 0, b[p]), loadB1(seq_a(EPK * KK_REC + p), 1, b[p]);
#pragma unroll
      for (int i = 0; i < 2 * KA; ++i) stage_store1(0, i);
      __syncthreads();                                   // P1: x_0 staged, the image of h zeroed
      stage_load(mk_stage(1), 0, KA);                    // (x_1's requests fly during in_A(0))
      in_phase();                                        // Z = x_0[0..KA) W
      __syncthreads();                                   // P2: nobody reads A's part of x_0 any more
    }
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
#if NRV_STAMP
      stamp_step = s < kStampSteps - 1 ? s : kStampSteps - 2;
#endif
      NRV_STAMP_K(0);
      rec_phase((const _Float16*)hbuf + hp_off, t_prev(s));              // Z += h_{s-1} U, parked; copy-out of h_{s-1}
      NRV_STAMP_K(1);
#pragma unroll
      for (int i = 0; i < 2 * KA; ++i) stage_store1(0, i);              // A's part of x_{s+1} (requested a step ago)
      stage_load(mk_stage(s + 2), 0, KA);                // behind the last product: the next weight wait is a barrier away
      NRV_STAMP_K(2);
      __syncthreads();                                   // Y: A's tiles of step s parked, x_{s+1}[0..KA) staged, h_{s-1} no longer read
      NRV_STAMP_K(3);
      if (s + 1 < T) in_phase();                         // Z = x_{s+1}[0..KA) W, while group B does the gates of step s
      NRV_STAMP_K(4);
      __syncthreads();                                   // X: h_s complete, x_{s+1}[KA..) staged
      NRV_STAMP_K(5);
    }
    copy_read((const _Float16*)hbuf);
    copy_write1(0, t_of(T - 1));
    copy_write1(1, t_of(T - 1));
  };

  // ======================================= group B: the other input blocks + the gate arithmetic ==============================
  auto run_b = [&]() __attribute__((always_inline)) {
    const float* bp = P.bias + (size_t)(dir * NG + ug) * 4 * 16 + l15;
    const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
    const float bi = bp[0] * dsc, bf = bp[16] * dsc, bg = bp[32] * dsc, bo = bp[48] * dsc;
    const float kI = __builtin_fmaf(bi, 0.2f, 0.5f), kF = __builtin_fmaf(bf, 0.2f, 0.5f), kO = __builtin_fmaf(bo, 0.2f, 0.5f),
                kG = bg * 2.885390081777927f;
    _Float16* const hw = (_Float16*)hbuf + (u0 >> 3) * GS + (4 * kq) * 8 + (u0 & 7);   // gates: + (16 rt + reg) 8; lo: + TERM
    float c[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) c[i] = 0.f;
    f32x4 Z[4][RT];
    // in_B(s): positions 0 .. SEQ_B - 1
    auto in_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        loadA_in1(KA, rt, 0, a[rt]);
        loadA_in1(KA, rt, 1, a[rt]);
      }
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        if (kk > 0 && kk < 4) NRV_STAMP_K(5 + kk);
#pragma unroll
        for (int g = 0; g < EPK; ++g) {
          const int p = EPK * kk + g;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 3 * RT; ++m) {
            const int rt = m / 3, pr = m % 3;
            Z[g][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[(4 * kk + rt) % NAS].v[PA[pr]]), b[p % NBG].t[PB[pr]],
                                  (kk == 0 && pr == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : Z[g][rt]);
            int rn, tn;
            frag_piece(g, m, rn, tn);
            const bool frag = rn >= 0 && kk + 1 < KB;
            if (m < 2 || frag) {
              __builtin_amdgcn_sched_barrier(0);
              if (m < 2) loadB1(seq_b((p + LBG) % SEQ_B), m, b[(p + LBG) % NBG]);
              if (frag) loadA_in1(KA + kk + 1, rn, tn, a[(4 * (kk + 1) + rn) % NAS]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // gates(s) from Z (x_s[KA..) W) + group A's tiles (x_s[0..KA) W + h_{s-1} U): c, h -> the split image of h_s
    auto gates = [&]() __attribute__((always_inline)) {
      f32x4 zt[2][4];                                    // A's four gate tiles of a row tile, one row tile ahead
#pragma unroll
      for (int g = 0; g < 4; ++g) zt[0][g] = *(const f32x4*)(zw + (g * RT) * 256);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int rt = e / 4, reg = e % 4;
        if (reg == 0 && rt + 1 < RT) {
#pragma unroll
          for (int g = 0; g < 4; ++g) zt[(rt + 1) & 1][g] = *(const f32x4*)(zw + (g * RT + rt + 1) * 256);
        }
        float zi = zt[rt & 1][0][reg] + Z[0][rt][reg], zf = zt[rt & 1][1][reg] + Z[1][rt][reg];
        float zg = zt[rt & 1][2][reg] + Z[2][rt][reg], zo = zt[rt & 1][3][reg] + Z[3][rt][reg];
        if constexpr (ACT == 0) {
          zi = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(zi, dsc02, kI), 0.0f), 1.0f);
          zf = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(zf, dsc02, kF), 0.0f), 1.0f);
          zo = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(zo, dsc02, kO), 0.0f), 1.0f);
        } else {
          zi = sigmoid_exact(__builtin_fmaf(zi, dsc, bi));
          zf = sigmoid_exact(__builtin_fmaf(zf, dsc, bf));
          zo = sigmoid_exact(__builtin_fmaf(zo, dsc, bo));
        }
        zg = __builtin_amdgcn_exp2f(__builtin_fmaf(zg, dsc2, kG));
        float t = __builtin_amdgcn_rcpf(zg + 1.0f);
        const float p = zi * __builtin_fmaf(t, -2.0f, 1.0f);
        const float cn = __builtin_fmaf(zf, c[e], p);
        c[e] = cn;
        zg = __builtin_amdgcn_exp2f(cn * 2.885390081777927f);
        t = __builtin_amdgcn_rcpf(zg + 1.0f);
        const float hv = zo * __builtin_fmaf(t, -2.0f * kHScale, kHScale);      // og * tanh(c) * 2^13
        const _Float16 hh = (_Float16)hv;
        const _Float16 hl = (_Float16)(hv - (float)hh);
        hw[(rt * 16 + reg) * 8] = hh;
        hw[(rt * 16 + reg) * 8 + TERM] = hl;
        if (reg == 3) __builtin_amdgcn_sched_barrier(0); // four elements in flight are enough: the other wave fills the gaps
      }
    };
    {
      stage_load(mk_stage(0), KA, KB);
#pragma unroll
      for (int p = 0; p < LBG; ++p) loadB1(seq_b(p), 0, b[p]), loadB1(seq_b(p), 1, b[p]);
#pragma unroll
      for (int i = 0; i < 2 * KB; ++i) stage_store1(KA, i);
      __syncthreads();                                   // P1
      __syncthreads();                                   // P2
    }
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
#if NRV_STAMP
      stamp_step = s < kStampSteps - 1 ? s : kStampSteps - 2;
#endif
      NRV_STAMP_K(0);
      in_phase();                                        // Z = x_s[KA..) W
      NRV_STAMP_K(1);
      copy_read((const _Float16*)hbuf);                  // copy-out of h_{s-1}: its image is overwritten behind barrier Y
      copy_write1(0, t_of(s > 0 ? s - 1 : 0));
      copy_write1(1, t_of(s > 0 ? s - 1 : 0));
      NRV_STAMP_K(2);
      __syncthreads();                                   // Y
      NRV_STAMP_K(3);
      stage_load(mk_stage(s + 1), KA, KB);               // at the start of the gate arithmetic: no weight wait until the next in_B
      gates();
#pragma unroll
      for (int i = 0; i < 2 * KB; ++i) stage_store1(KA, i);
      NRV_STAMP_K(4);
      __syncthreads();                                   // X
      NRV_STAMP_K(5);
    }
    copy_read((const _Float16*)hbuf);
    copy_write1(0, t_of(T - 1));
    copy_write1(1, t_of(T - 1));
  };
  if (grp_a) run_a();
  else run_b();
#if NRV_STAMP
  {
    unsigned long long stamp_rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt1)::"memory");
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(stamp_rt0), "s"(stamp_base), "s"((unsigned)(((kStampSteps - 1) * 16 + 0) * 8)) : "memory");
    asm volatile("s_store_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(stamp_rt1), "s"(stamp_base),
                 "s"((unsigned)(((kStampSteps - 1) * 16 + 1) * 8)) : "memory");
  }
#endif
}

}  // namespace nrv
