// f16x2 Bi-LSTM layer, EIGHT waves per workgroup in two groups that run a step in opposite order.
#pragma once
#include "nrv_lstm_f16x2s.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// lstm_h2w_kernel: the 192 -> 128 layer with two waves per SIMD whose matrix and vector phases interleave.
//
// Why (round 4): in lstm_h2s_kernel - ONE wave per SIMD, 32 units x 4 gates x 64 rows = 256 accumulator
// registers - the gate arithmetic of a step (150 cycles of vector issue per element, 32 elements) can only hide behind the
// products of the next step's input projection, which offer 8.6 free issue cycles each: that phase is issue-bound (in-kernel
// stamps: 2 260-2 510 cycles per 96 products against 1 596 of matrix pipe), while the recurrent phase, on which nothing can ride,
// leaves 3.3 k cycles of issue idle.  A single in-order wave cannot move the one into the other.
//
// Here a wave owns 16 units x 4 gates x the same 64 rows (64 accumulators; a weight entry still feeds 4 row tiles x 3 products),
// eight waves make the workgroup, and the two waves that share a SIMD (w and w + 4) run a step in DIFFERENT order:
//     waves 0-3 (group A):  rec(s)  [requests for x_{s+3}] gates(s) in registers | barrier 1 | h_s -> image, in(s+1) -> Z        |
//     waves 4-7 (group B):  rec(s), Z -> LDS                                     | barrier 1 | in(s+1) -> Z, [requests for      | barrier 2
//                                                                                |           | x_{s+2}] gates(s) from LDS -> h_s |
//     barrier 1: nobody reads h_{s-1} any more, x_{s+1} is staged;  barrier 2: h_s complete, nobody reads x_{s+1} any more
// so that while one wave of a SIMD turns its accumulators into (c, h) on the vector pipe the other one feeds the matrix pipe.
// Group A - the older waves, which the SIMD's arbiter prefers when both have a product ready - is through rec(s) first and
// does its gate arithmetic while group B still multiplies (only the 32 two-byte stores of h_s wait for barrier 1, and ride on
// the first entries of its in(s+1): as a burst at the head of the second half they held group B's first fragment reads back); group B's
// input projection runs at raised priority, in front of group A's, and its gates hide behind the rest of that.  No gate piece
// rides between products any more (no hand-cut stages, no ticks): each wave's stream is plain, the arbiter interleaves.
// Group B's in(s+1) needs Z's accumulators while its gates(s) still need Z's values: the 16 tiles are parked in LDS (one
// 16-byte store per tile, behind products of the last recurrent block as the tiles become final; read back four tiles per
// row tile by the gate code), so both groups live on 64 accumulators and fit the 256 registers of a two-wave SIMD.
//
// The activation fragments of x_{s+1} are the same for all eight waves, and with 16 units per wave each would feed only 12
// products: fetched per wave they double the fragment-shaped requests at the CU's vector-memory front end, which the weight
// stream (80 KB per wave and step) already loads to two thirds (measured: 186 us against lstm_h2s_kernel's 163 on one box with
// per-wave x loads).  So x goes through LDS once per workgroup: wave w fetches the fragments (row tile w / 2, term w % 2)
// of all six k blocks - requested at the start of its gate arithmetic, the one stretch in which it waits for no weight entry
// (the memory counter is in order) - and writes them lane-linear behind products of the next rec(); in() reads them with
// ds_read_b128.  Barrier 1 in the middle of the step makes that legal and lets h live in ONE image (gates(s) overwrite
// h_{s-1} behind it): 33 KB (h) + 64 KB (group B's tiles) + 48 KB (x) = 145 KB of LDS.
//
// Every request is issued BEHIND a product inside a chain of three (tools/microbench/tick_cost.hip: two 1 KB requests in front
// of an entry's 12 products cost 2.3 cycles per product, inside a chain 0.6).
//
// Registers: 256 per wave (two waves per SIMD), none to spare: hipcc parks five address values of the epilogue's copy-out in
// scratch (24 B per lane) in front of the loop and fetches them back behind it; there is no scratch instruction inside the
// loop (tools/isa_of.sh; a weight ring of 8 entries, or precomputing those addresses, puts some there).
//
// In-kernel stamps (scripts/gpu_stamps_w.py; profiles/r04*_stamps_lstm_h2w.json): 18.8 k cycles per step against
// lstm_h2s_kernel's 22.7-23.6 k and 15.5 k of matrix pipe; the second half runs pipe-bound, the first at 80 % (group B's rec()
// gets 23 % of the pipe while group A's runs, then shares the SIMD with group A's gate arithmetic).  The time gained is a
// fraction of the cycles gained: the launch runs at the socket's power cap and the fuller pipe holds a lower clock
// (1.81 GHz against lstm_h2s_kernel's 2.03 on one box: 148.5 us against 152.9).
//
// Everything else is lstm_h2s_kernel's: operand layouts, the split h image, the scales, the packed weights
// (pack_lstm_h2s with one unit half per wave), the raw copy-out of h x 2^13.
// ---------------------------------------------------------------------------------------
#ifndef NRV_STAMP_REC_ENTRIES
#define NRV_STAMP_REC_ENTRIES 0
#endif
#if NRV_STAMP
// diagnostic build: [wave 8][step 15][slot 16] s_memtime stamps per workgroup, in the stamp buffer's region of this layer
#define NRV_STAMP_W(slot)                                                                                    \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
    stamp_lds[(wave * kStampSteps + stamp_step) * 16 + (slot)] = t_;                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
// blocks inside a phase: s_memtime into an SGPR pair of its own, NOT waited for (a wait would drain the LDS reads in flight);
// NRV_STAMP_FLUSH at the end of the iteration waits once and stores them (slots 7..15)
#define NRV_STAMP_D(i)                                                                                       \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_memtime %0" : "=s"(stamp_d[i])::"memory");                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#define NRV_STAMP_FLUSH()                                                                                    \
  do {                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    for (int i_ = 0; i_ < 9; ++i_) stamp_lds[(wave * kStampSteps + stamp_step) * 16 + 7 + i_] = stamp_d[i_];  \
  } while (0)
#else
#define NRV_STAMP_W(slot) do { } while (0)
#define NRV_STAMP_D(i) do { } while (0)
#define NRV_STAMP_FLUSH() do { } while (0)
#endif

template <int KQ0, int KQ1, int H, int ACT, int NBG>
__global__ void __launch_bounds__(512)
lstm_h2w_kernel(const LstmH2Args args) {
  constexpr int NG = H / 16;                                   // waves = unit groups of 16
  static_assert(NG == 8, "eight waves: H = 128");
  constexpr int KK0 = KQ0 / 8, KK1 = KQ1 / 8, KK_IN = KK0 + KK1, KK_REC = H / 32, KK = KK_IN + KK_REC;
  constexpr int R = 2, RT = 2 * R, EPK = 4, ROWS = 32 * R;
  constexpr int GS = ROWS * 8 + 8, TERM = (H / 8) * GS, HBUF = TERM, NTHREADS = 64 * NG;
  constexpr int NAS = 6, LBG = NBG - 1;                        // activation ring: 6 fragment slots = 1.5 k blocks
  static_assert(RT * 2 == NG && KK_IN % 2 == 0 && RT == 4, "staging: one (row tile, term) per wave, in two halves");
  static_assert((EPK * KK) % NBG == 0 && (EPK * KK_IN) % NBG == 0 && LBG <= EPK * KK_REC && LBG <= EPK * KK_IN, "weight ring");
  constexpr int NE = RT * 4;                                   // gate elements per lane
  constexpr int XF = 256;                                      // floats per staged fragment (64 lanes x 16 B)
  __shared__ __attribute__((aligned(16))) float hbuf[HBUF];                       // the split image of h (hi | lo), ONE buffer
  __shared__ __attribute__((aligned(16))) float zst[4 * 16 * 64 * 4];           // group B: a wave's 16 accumulator tiles (64 KB)
  __shared__ __attribute__((aligned(16))) float xst[KK_IN * RT * 2 * XF];        // x_{s+1}: fragment (kk, rt, term) lane-linear (48 KB)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if NRV_STAMP
  __shared__ unsigned long long stamp_lds[8 * kStampSteps * 16];
  int stamp_step = 0;
  unsigned long long stamp_d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8 * kStampSteps * 16; i += 512) stamp_lds[i] = 0;
  unsigned long long stamp_rt0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt0)::"memory");
#endif
  const bool grp_a = wave < 4;
  const int hg = wave;
  const int l15 = lane & 15, kq = lane >> 4;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmH2ModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS;

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + hg) * KK) * (EPK * 2 * 1024), KK * EPK * 2 * 1024);
  const unsigned wlane = lane * 16;
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * 16 + l15;
  const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
  const float bi = bp[0] * dsc, bf = bp[16] * dsc, bg = bp[32] * dsc, bo = bp[48] * dsc;
  const float kI = __builtin_fmaf(bi, 0.2f, 0.5f), kF = __builtin_fmaf(bf, 0.2f, 0.5f), kO = __builtin_fmaf(bo, 0.2f, 0.5f),
              kG = bg * 2.885390081777927f;
  const int u0 = hg * 16 + l15;                                // this lane's unit
  _Float16* const hw = (_Float16*)hbuf + (u0 >> 3) * GS + (4 * kq) * 8 + (u0 & 7);   // gates: + (16 rt + reg) 8; lo: + TERM
  const _Float16* const hp = (const _Float16*)hbuf + kq * GS + l15 * 8;              // rec: + kkr 4 GS + rt 128; lo: + TERM
  float* const zw = zst + (size_t)(((wave & 3) * 16) * 64 + lane) * 4;               // tile (g, rt) at + (g RT + rt) 256
  float* const xw = xst + (size_t)(wave * 64 + lane) * 4;                            // this wave's fragments: + kk RT 2 XF
  const float* const xr = xst + (size_t)lane * 4;                                    // fragment (kk, rt, term): + ((kk RT + rt) 2 + term) XF

  for (int i = threadIdx.x; i < HBUF; i += NTHREADS) hbuf[i] = 0.f;              // image of h_{-1}
  float c[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) c[i] = 0.f;

  // ---- staging of x_s: this wave's share is (row tile wave / 2, term wave % 2) of every k block
  struct SBase {
    __amdgpu_buffer_rsrc_t r0, r1;
    unsigned v0, v1;
  };
  auto mk_stage = [&](int s) __attribute__((always_inline)) {
    const int t = dir ? (T - 1 - s) : s;
    const int r = wave >> 2, sub = (wave >> 1) & 1, term = wave & 1;
    SBase sb;
    sb.r0 = make_rsrc(P.in0.ubase(row0 + r * 32, t), 0xffffffffu);
    sb.v0 = (P.in0.voff(row0 + r * 32, t, l15 + 16 * sub, kq & 1) + (kq >> 1) * 512) * 4 + term * 1024;
    if constexpr (KQ1 > 0) {
      sb.r1 = make_rsrc(P.in1.ubase(row0 + r * 32, t), 0xffffffffu);
      sb.v1 = (P.in1.voff(row0 + r * 32, t, l15 + 16 * sub, kq & 1) + (kq >> 1) * 512) * 4 + term * 1024;
    } else {
      sb.r1 = sb.r0;
      sb.v1 = 0;
    }
    return sb;
  };
  f32x4 xs[KK_IN];
  auto stage_load1 = [&](const SBase& sb, int kk) __attribute__((always_inline)) {
    xs[kk] = (KQ1 == 0 || kk < KK0) ? buf_load16(sb.r0, sb.v0, kk * 4096) : buf_load16(sb.r1, sb.v1, (kk - KK0) * 4096);
  };
  // The six requests of a wave's share go out TOGETHER at the start of its gate arithmetic - the one stretch of a step in
  // which the wave waits for no weight entry: the memory counter is in order, so a request to HBM in front of a weight
  // request holds the products behind that entry back until it has landed (with the requests riding on rec() entries, rec()
  // ran at 21 cycles per product with both waves of a SIMD in it - stamps, r04aa).
  auto stage_load = [&](const SBase& sb) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KK_IN; ++kk) stage_load1(sb, kk);
  };
  auto stage_store1 = [&](int kk) __attribute__((always_inline)) {
    *(f32x4*)(xw + kk * RT * 2 * XF) = xs[kk];
  };
  auto stage_all = [&](const SBase& sb) __attribute__((always_inline)) {      // prologue only: nothing to hide behind
    stage_load(sb);
#pragma unroll
    for (int kk = 0; kk < KK_IN; ++kk) stage_store1(kk);
  };

  struct BReg { f16x8 t[2]; };
  struct AReg { f32x4 v[2]; };
  BReg b[NBG];
  AReg a[NAS];                                                 // fragment (block kl of the phase, row tile rt) in slot (4 kl + rt) % 6
  auto loadB1 = [&](int e, int term, BReg& bb) __attribute__((always_inline)) {
    bb.t[term] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2 + term) * 1024));
  };
  auto loadB = [&](int e, BReg& bb) __attribute__((always_inline)) {
    loadB1(e, 0, bb);
    loadB1(e, 1, bb);
  };
  auto loadA_in1 = [&](int kk, int rt, int term, AReg& d) __attribute__((always_inline)) {
    d.v[term] = *(const f32x4*)(xr + ((kk * RT + rt) * 2 + term) * XF);
  };
  auto loadA_rec1 = [&](int kkr, int rt, int term, AReg& d) __attribute__((always_inline)) {
    d.v[term] = *(const f32x4*)(hp + kkr * 4 * GS + rt * 128 + term * TERM);
  };
  auto loadA_in = [&](int kk, int rt, AReg& d) __attribute__((always_inline)) {
    loadA_in1(kk, rt, 0, d);
    loadA_in1(kk, rt, 1, d);
  };
  auto loadA_rec = [&](int kkr, int rt, AReg& d) __attribute__((always_inline)) {
    loadA_rec1(kkr, rt, 0, d);
    loadA_rec1(kkr, rt, 1, d);
  };
  // Where a request rides (tools/microbench/tick_cost.hip: two 1 KB requests in front of an entry's 12 products cost 2.3 cycles
  // per product, inside a chain of three 0.6): the pieces of an entry's side work are issued BEHIND its product number m,
  // fenced so that they stay there.  Next block's fragment (rt' = 0, 1: entries 0, 1, into the ring's two free slots;
  // rt' = 2, 3: entry 3, behind the last products of this block's fragments 0, 1, whose slots they take).
  auto frag_piece = [&](int g, int m, int& rt_next, int& term) __attribute__((always_inline)) {
    rt_next = -1;
    term = 0;
    if ((g == 0 || g == 1) && (m == 4 || m == 5)) { rt_next = g; term = m - 4; }
    if (g == EPK - 1 && (m == 4 || m == 5)) { rt_next = 2; term = m - 4; }
    if (g == EPK - 1 && (m == 7 || m == 8)) { rt_next = 3; term = m - 7; }
  };
  constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // hi*lo, lo*hi, hi*hi

  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  f16x2 hpk[NE];                                         // DEFER: (hi, lo) of the element's h: stored by in() / gates_write()
  // ---- in(): D = x W over the input blocks from the staged x (first product of a tile: C = 0).  FIRST: the prologue's
  // in(0), whose weight requests wrap around to in(1)'s entries (step 0 has no rec()).  STASH: group B, whose rec() parked
  // the tiles of entries 0..2 already; entry 3's follow behind products of entries 0 and 1.
  auto in_phase = [&](auto first_tag, auto stash_tag, auto hw_tag, f32x4 (&D)[4][RT]) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value, STASH = decltype(stash_tag)::value, HWRITE = decltype(hw_tag)::value;
    // Group B's projection goes first on the SIMD (its gates still have to follow and want group A's projection to hide
    // behind); either one goes in front of the other wave's gate arithmetic, which fills the gaps.
    __builtin_amdgcn_s_setprio(STASH ? 3 : 2);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) loadA_in(0, rt, a[rt]);
#pragma unroll
    for (int kk = 0; kk < KK_IN; ++kk) {
#if !NRV_STAMP_REC_ENTRIES
      if (!FIRST && kk > 0) NRV_STAMP_D(kk - 1);
#endif
#pragma unroll
      for (int g = 0; g < EPK; ++g) {
        const int e = EPK * kk + g;
        const int en = FIRST ? (e + LBG) % (EPK * KK_IN) : (e + LBG) % (EPK * KK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 3 * RT; ++m) {
          const int rt = m / 3, pr = m % 3;
          D[g][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[(4 * kk + rt) % NAS].v[PA[pr]]), b[e % NBG].t[PB[pr]],
                                (kk == 0 && pr == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : D[g][rt]);
          int rn, tn;
          frag_piece(g, m, rn, tn);
          const bool stash = STASH && kk == 0 && g < 2 && (m == 7 || m == 10);
          const bool hwr = HWRITE && e < NE && (m == 3 || m == 10);  // group A: element e of h_s, hi behind product 3, lo behind 10
          if (m < 2 || (rn >= 0 && kk + 1 < KK_IN) || stash || hwr) {
            __builtin_amdgcn_sched_barrier(0);
            if (hwr) hw[((e / 4) * 16 + e % 4) * 8 + (m == 10 ? TERM : 0)] = hpk[e][m == 10 ? 1 : 0];   // (an int index: a bool one reads element -1)
            if (m < 2) loadB1(en, m, b[(e + LBG) % NBG]);
            if (rn >= 0 && kk + 1 < KK_IN) loadA_in1(kk + 1, rn, tn, a[(4 * (kk + 1) + rn) % NAS]);
            if (stash) {
              const int rs = g * 2 + (m == 10);
              *(f32x4*)(zw + ((EPK - 1) * RT + rs) * 256) = D[EPK - 1][rs];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- copy-out of h_s x 2^13 as it lies in the image (the BatchNorm behind the layer lives in the next layer's weights):
  // two items per thread, each a read of the two term planes and, some products later, their stores
  constexpr int ITEMS = (H / 16) * 2 * ROWS, NIT = ITEMS / NTHREADS;
  static_assert(ITEMS % NTHREADS == 0, "copy-out items must divide evenly");
  f16x8 chi[NIT], clo[NIT];
  auto copy_read = [&](int i) __attribute__((always_inline)) {
    const _Float16* himg = (const _Float16*)hbuf;
    const int it = threadIdx.x + i * NTHREADS;
    const int kbh = it / ROWS, rr = it % ROWS;
    chi[i] = *(const f16x8*)(himg + kbh * GS + rr * 8);
    clo[i] = *(const f16x8*)(himg + kbh * GS + rr * 8 + TERM);
  };
  // (stores through a descriptor on this workgroup's first output tile: the step enters as a scalar offset, the item's
  // place is a per-lane constant - no 64-bit address arithmetic between the products)
  const __amdgpu_buffer_rsrc_t ors = make_rsrc(P.out + (size_t)(blk.rowblk * R) * T * (2 * H / 4) * 128, 0xffffffffu);
  unsigned cw_off[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    constexpr int KBH = H / 16;
    const int it = threadIdx.x + i * NTHREADS;
    const int kbh = it / ROWS, rr = it % ROWS;
    cw_off[i] = (unsigned)((((rr / 32) * T * (2 * H / 4) + (dir * KBH + (kbh >> 1)) * 4 + (kbh & 1)) * 128 + (rr & 31) * 4) * 4);
  }
  auto copy_write1 = [&](int i, int term, int t) __attribute__((always_inline)) {
    if (term < 0 || term > 1) return;
    const unsigned soff = (unsigned)t * ((2 * H / 4) * 128 * 4) + term * (2 * 128 * 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, term == 0 ? chi[i] : clo[i]), ors, cw_off[i], soff, 0);
  };
  auto copy_write = [&](int i, int t) __attribute__((always_inline)) {
    copy_write1(i, 0, t);
    copy_write1(i, 1, t);
  };
  auto copy_out = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) copy_read(i);
#pragma unroll
    for (int i = 0; i < NIT; ++i) copy_write(i, t);
  };
  // ---- rec(): Z += h U, h from the image behind barrier 2.  The step's other memory work rides on its entries (all eight
  // waves enter together: issued in one burst these pieces held the phase's first products back by 1.3-3.7 k cycles -
  // in-kernel stamps, r04q): the staged fragments' way into LDS and the copy-out of h_{s-1}.  STASH (group B): in the last block the tiles of an entry are final when it is through, and are parked in
  // LDS behind products of the next one.
  auto rec_phase = [&](auto stash_tag, f32x4 (&Z)[4][RT], int t_out) __attribute__((always_inline)) {
    constexpr bool STASH = decltype(stash_tag)::value;
    // side work by entry, behind product 10 (and 11): 0..5 the staged fragments (requested during the gates before) -> LDS |
    // 6, 8 copy-out reads | 7, 9 copy-out stores
    static_assert(KK_IN == 6 && NIT == 2 && EPK * KK_REC == 16, "the side-work table below");
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) loadA_rec(0, rt, a[rt]);
#pragma unroll
    for (int kr = 0; kr < KK_REC; ++kr) {
      const int kk = KK_IN + kr;
#if !NRV_STAMP_REC_ENTRIES
      if (kr > 0) NRV_STAMP_D(4 + kr);
#endif
#pragma unroll
      for (int g = 0; g < EPK; ++g) {
        const int e = EPK * kk + g, er = EPK * kr + g;
        const int en = (e + LBG) % (EPK * KK);
#if NRV_STAMP_REC_ENTRIES
        if (er >= 1 && er <= 8) NRV_STAMP_D(er - 1);           // one-off: the first eight entries of rec(), slots 7..14
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 3 * RT; ++m) {
          const int rt = m / 3, pr = m % 3;
          Z[g][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[(4 * kr + rt) % NAS].v[PA[pr]]), b[e % NBG].t[PB[pr]], Z[g][rt]);
          int rn, tn;
          frag_piece(g, m, rn, tn);
          const bool frag = rn >= 0 && kr + 1 < KK_REC;
          const bool stash = STASH && kr == KK_REC - 1 && g > 0 && (m == 4 || m == 5 || m == 7 || m == 8);
          const bool side = m >= 10 && er < 10;
          if (m < 2 || frag || stash || side) {
            __builtin_amdgcn_sched_barrier(0);
            if (m < 2) loadB1(en, m, b[(e + LBG) % NBG]);
            if (frag) loadA_rec1(kr + 1, rn, tn, a[(4 * (kr + 1) + rn) % NAS]);
            if (stash) {
              const int rs = m == 4 ? 0 : m == 5 ? 1 : m == 7 ? 2 : 3;
              *(f32x4*)(zw + ((g - 1) * RT + rs) * 256) = Z[g - 1][rs];
            }
            if (m == 10) {
              if (er < KK_IN) stage_store1(er);
              if (er == 6) copy_read(0);
              if (er == 8) copy_read(1);
            }
            if (er == 7) copy_write1(0, m - 10, t_out);       // m = 10, 11: the two term planes
            if (er == 9) copy_write1(1, m - 10, t_out);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- gates of step s from Z: c, h -> the split image of h_s (plain: the SIMD's other wave owns the matrix pipe meanwhile)
  auto gates = [&](auto lds_tag, auto defer_tag, const f32x4 (&Z)[4][RT]) __attribute__((always_inline)) {
    constexpr bool FROM_LDS = decltype(lds_tag)::value, DEFER = decltype(defer_tag)::value;
    f32x4 zt[2][4];                                      // FROM_LDS: the four gate tiles of a row tile, one row tile ahead
    if constexpr (FROM_LDS) {
#pragma unroll
      for (int g = 0; g < 4; ++g) zt[0][g] = *(const f32x4*)(zw + (g * RT) * 256);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int rt = e / 4, reg = e % 4;
      if constexpr (FROM_LDS) {
        if (reg == 0 && rt + 1 < RT) {
#pragma unroll
          for (int g = 0; g < 4; ++g) zt[(rt + 1) & 1][g] = *(const f32x4*)(zw + (g * RT + rt + 1) * 256);
        }
      }
      float zi, zf, zg, zo;
      if constexpr (FROM_LDS) { zi = zt[rt & 1][0][reg]; zf = zt[rt & 1][1][reg]; zg = zt[rt & 1][2][reg]; zo = zt[rt & 1][3][reg]; }
      else { zi = Z[0][rt][reg]; zf = Z[1][rt][reg]; zg = Z[2][rt][reg]; zo = Z[3][rt][reg]; }
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
      const float hv = zo * __builtin_fmaf(t, -2.0f * kHScale, kHScale);        // og * tanh(c) * 2^13
      const _Float16 hh = (_Float16)hv;
      const _Float16 hl = (_Float16)(hv - (float)hh);
      if constexpr (DEFER) {
        hpk[e] = f16x2{hh, hl};
      } else {
        hw[(rt * 16 + reg) * 8] = hh;
        hw[(rt * 16 + reg) * 8 + TERM] = hl;
      }
      if (reg == 3) __builtin_amdgcn_sched_barrier(0);   // four elements in flight are enough: the other wave fills the gaps
    }
  };
  auto gates_write = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int rt = e / 4, reg = e % 4;
      hw[(rt * 16 + reg) * 8] = hpk[e][0];
      hw[(rt * 16 + reg) * 8 + TERM] = hpk[e][1];
    }
  };
  f32x4 Z[4][RT];
  auto t_of = [&](int s) __attribute__((always_inline)) { return dir ? (T - 1 - s) : s; };

  // The two groups run separate instruction streams from here on (a wave-uniform branch): each is a plain loop whose
  // registers are assigned on its own.
  auto run = [&](auto a_tag) __attribute__((always_inline)) {
    constexpr bool GRP_A = decltype(a_tag)::value;
    auto s_clamp = [&](int s) __attribute__((always_inline)) { return s < T ? s : T - 1; };   // past the end: staged again, read by nobody
    // Group A's gate arithmetic of a step follows its rec() at once, in FRONT of barrier 1 - group A (the older waves, which
    // the SIMD's arbiter prefers) is through rec() 2.8 k cycles before group B (stamps, r04y) and would wait there; only the
    // 32 two-byte stores of h_s have to stay behind the barrier (other waves still read h_{s-1}).
    constexpr bool DEFER = GRP_A;
    typedef std::integral_constant<bool, DEFER> defer_t;
    {
      stage_all(mk_stage(0));
#pragma unroll
      for (int e = 0; e < LBG; ++e) loadB(e % (EPK * KK_IN), b[e]);
      __syncthreads();                                   // x_0 staged, the image of h_{-1} zeroed
      stage_load(mk_stage(s_clamp(1)));                  // (x_1's requests fly during in(0))
      in_phase(std::true_type{}, std::false_type{}, std::false_type{}, Z);
      __syncthreads();                                   // nobody reads x_0 any more
#pragma unroll
      for (int kk = 0; kk < KK_IN; ++kk) stage_store1(kk);
      if constexpr (!GRP_A) {                            // step 0 has no rec() to park group B's tiles
#pragma unroll
        for (int g = 0; g + 1 < EPK; ++g)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) *(f32x4*)(zw + (g * RT + rt) * 256) = Z[g][rt];
      }
      if constexpr (DEFER) {
        stage_load(mk_stage(s_clamp(2)));
        gates(std::false_type{}, defer_t{}, Z);
      }
      __syncthreads();                                   // barrier 1 of step 0 (which has no rec)
    }
    // The loop is rotated: the second half of step s, then the first half of step s + 1.
#pragma unroll 1
    for (int s = 0; s + 1 < T; ++s) {
#if NRV_STAMP
      stamp_step = s;
#endif
      NRV_STAMP_W(0);
      if constexpr (GRP_A) {
        if constexpr (!DEFER) {
          stage_load(mk_stage(s_clamp(s + 2)));
          gates(std::false_type{}, std::false_type{}, Z);
        }
        NRV_STAMP_W(1);
        in_phase(std::false_type{}, std::false_type{}, defer_t{}, Z);   // (DEFER: the stores of h_s ride on its first entries)
      } else {
        in_phase(std::false_type{}, std::true_type{}, std::false_type{}, Z);
        NRV_STAMP_W(1);
        stage_load(mk_stage(s_clamp(s + 2)));
        gates(std::true_type{}, std::false_type{}, Z);
      }
      NRV_STAMP_W(2);
      __syncthreads();                                   // barrier 2: h_s complete, Z = x_{s+1} W
      NRV_STAMP_W(3);
      NRV_STAMP_W(4);
      rec_phase(std::integral_constant<bool, !GRP_A>{}, Z, t_of(s));     // (the fragments of x_{s+2} go to LDS in here)
      NRV_STAMP_W(5);
      if constexpr (DEFER) {
        stage_load(mk_stage(s_clamp(s + 3)));
        gates(std::false_type{}, defer_t{}, Z);
      }
      NRV_STAMP_D(8);
      __syncthreads();                                   // barrier 1 of step s + 1
      NRV_STAMP_W(6);
      NRV_STAMP_FLUSH();
    }
    if constexpr (DEFER) gates_write();                  // the last step: nothing follows its gates
    else gates(std::false_type{}, std::false_type{}, Z);
    __syncthreads();
    copy_out(t_of(T - 1));
#if NRV_STAMP
    {
      unsigned long long stamp_rt1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt1)::"memory");
      stamp_lds[(wave * kStampSteps + kStampSteps - 1) * 16 + 0] = stamp_rt0;
      stamp_lds[(wave * kStampSteps + kStampSteps - 1) * 16 + 1] = stamp_rt1;
      __syncthreads();
      if (blockIdx.x < kStampBlocks)
        for (int i = threadIdx.x; i < 8 * kStampSteps * 16; i += 512) (&nrv_stamp_buf[0][blockIdx.x][0][0][0])[i] = stamp_lds[i];
    }
#endif
  };
  if (grp_a) run(std::true_type{});
  else run(std::false_type{});
}

}  // namespace nrv
