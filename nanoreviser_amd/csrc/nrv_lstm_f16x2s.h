// f16x2 Bi-LSTM layer on v_mfma_f32_16x16x32_f16 tiles.
#pragma once
#include "nrv_lstm_f16x2.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// lstm_h2s_kernel: lstm_h2o_kernel's schedule (rec(s) -> in(s+1) with the gate / copy-out pieces of step s
// between its MFMAs -> Z <- N) on the SMALL matrix tile.
//
// Why: the chip throttles its clock on the 32x32x16 f16 MFMA and not on the 16x16x32 one
// (tools/microbench/mfma_mix.hip, registers only, one wave per SIMD: 1540 cycles per 48 MFMAs at 1.93 GHz
// against 1556 cycles for the same MACs at 2.39 GHz), and lstm_h2o_kernel runs its 60 % pipe-busy stream at
// 1.71-1.77 GHz.  The small tile costs twice the MFMA instructions (16 cycles each, 8 of them holding the issue
// port) for the same operand bytes.
//
// Ownership: a wave owns UH groups of 16 hidden units x 4 gates x 2R row tiles of 16 rows: accumulator tile
// (gate g, unit half uh, row tile rt) is a float4 per lane - lane l holds column (unit) l & 15, rows
// 4 (l >> 4) .. + 3 - so i, f, g, o of one (row, unit) still meet in one lane.  H = 128: UH = 2, four waves;
// H = 64: UH = 1, four waves (what lstm_h2o_kernel needs GPT = 2 and a lane exchange for).
// A fragment of (row tile, 32-k block kk, term): lane l = (row l & 15, k-group l >> 4) reads 16 bytes of the
// split planes at chunk 4 (2 kk + (kq >> 1)) + 2 term + (kq & 1) - the same buffers, four 256-byte runs per
// wave instead of one KiB.  B fragments (host: pack_lstm_h2s): entry (kk, gate, uh) = [term][64 lanes][8 f16],
// lane l: k = 32 kk + 8 (l >> 4) + j, unit 16 uh + (l & 15); an entry feeds 2R row tiles x 3 products = 6R MFMAs.
// The recurrent operand is kept in LDS ALREADY SPLIT: the gate code writes h_s x 2^13 as two f16 terms into
// the image [term][unit group of 8][row][8 f16] (a group is ROWS x 16 B + 16 B of padding), so its A fragments
// are two 16-byte LDS reads like those of x and the recurrent phase has no VALU work at all - on this tile a
// VALU instruction is not free (two per 16-cycle tick), and the split of 8 values per (row tile, k-block) and
// lane was 40 % of the kernel's VALU instructions.  The copy-out adds hi + lo back before the BatchNorm.
// ---------------------------------------------------------------------------------------
// Diagnostic build only (-DNRV_STAMP=1, tools/lstm_exp.sh stamp; no stamp executes in the product): s_memtime at the
// phase edges of every step and wave, kept in LDS during the launch and copied at its end to a buffer of their own
// that nothing else reads (scripts/gpu_stamps.py reads it through nrv_exp_stamps).  The data, and with it the clock,
// are the product's; the stamps' fences forbid some overlap, so read SHARES of a step, not its length.
#ifndef NRV_STAMP
#define NRV_STAMP 0
#endif
#if NRV_STAMP
constexpr int kStampSlots = 32, kStampSteps = 15, kStampWaves = 4, kStampBlocks = 256;
__device__ unsigned long long nrv_stamp_buf[2][kStampBlocks][kStampWaves][kStampSteps][kStampSlots];
#define NRV_STAMP_AT(slot)                                                                                   \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
    stamp_lds[(wave * kStampSteps + stamp_step) * kStampSlots + (slot)] = t_;                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#else
#define NRV_STAMP_AT(slot) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma16_f16(const f16x8& a, const f16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// KBL: the weight fragments of the first KBL input blocks stay in LDS for the whole launch (each wave its own
// 2 EPK KBL KB, copied once in the prologue) and enter the ring by ds_read instead of from L2.
// RAW: the layer's output is h x 2^13 itself (no BatchNorm behind it, or one that the host folded into the next
// layer's weights): the copy-out moves the two term planes of the image to memory as they are, no arithmetic.
template <int KQ0, int KQ1, int H, int R, int WR, int UH, int ACT, int NBG, int NA, int KBL = 0, bool RAW = false>
__global__ void __launch_bounds__(64 * (H / (16 * UH)) * WR)
lstm_h2s_kernel(const LstmH2Args args) {
  constexpr int NG = H / (16 * UH);                            // unit groups = waves per wave-row
  constexpr int KK0 = KQ0 / 8, KK1 = KQ1 / 8, KK_IN = KK0 + KK1, KK_REC = H / 32, KK = KK_IN + KK_REC;
  constexpr int RT = 2 * R;                                    // 16-row tiles of a wave
  constexpr int EPK = 4 * UH, TPE = 3 * RT;                    // weight entries per k-block / MFMA ticks per entry
  constexpr int ROWS = 32 * R * WR;
  constexpr int GS = ROWS * 8 + 8;                             // f16 per unit group of the h image (8 of padding)
  constexpr int TERM = (H / 8) * GS;                           // f16 per term plane
  constexpr int HBUF = TERM;                                   // floats per image = 2 terms x TERM f16
  constexpr int NTHREADS = 64 * NG * WR;
  constexpr int LBG = NBG - 1, LA = NA - 1;
  static_assert(KQ0 % 8 == 0 && KQ1 % 8 == 0 && H % 32 == 0 && H % (16 * UH) == 0, "K must come in blocks of 32");
  static_assert((EPK * KK) % NBG == 0 && (EPK * KK_IN) % NBG == 0 && KK % NA == 0 && KK_IN % NA == 0,
                "ring sizes must divide the block counts");
  static_assert(LA >= 1 && LA <= KK_IN && LA <= KK_REC && LBG <= EPK * KK_REC, "leads must stay inside a phase");

  constexpr int NE = UH * RT * 4;                              // gate elements per lane
  constexpr bool CLDS = NE > 32;                               // cell state in LDS when the registers are needed elsewhere (not since h is kept split: no split buffers)
  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];
  __shared__ __attribute__((aligned(16))) float bnl[2 * H];
  __shared__ float cl[CLDS ? NE * NTHREADS : 1];
  __shared__ __attribute__((aligned(16))) float wl[KBL > 0 ? NG * WR * KBL * EPK * 512 : 4];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if NRV_STAMP
  __shared__ unsigned long long stamp_lds[kStampWaves * kStampSteps * kStampSlots];
  int stamp_step = 0;
  for (int i = threadIdx.x; i < kStampWaves * kStampSteps * kStampSlots; i += 64 * (H / (16 * UH)) * WR) stamp_lds[i] = 0;
  unsigned long long stamp_rt0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt0)::"memory");
#endif
  const int hg = wave % NG, wr = wave / NG;
  const int l15 = lane & 15, kq = lane >> 4;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmH2ModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS + wr * (32 * R);
  const int lrow0 = wr * (32 * R);

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + hg) * KK) * (EPK * 2 * 1024), KK * EPK * 2 * 1024);
  const unsigned wlane = lane * 16;
  // the bias (x 2^E) rides on the gate constants, per unit half (see lstm_h2o_kernel)
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * (16 * UH) + l15;
  const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
  float kI[UH], kF[UH], kO[UH], kG[UH], bzI[UH], bzF[UH], bzO[UH];
#pragma unroll
  for (int uh = 0; uh < UH; ++uh) {
    const float bi = bp[0 * 16 * UH + 16 * uh] * dsc, bf = bp[1 * 16 * UH + 16 * uh] * dsc,
                bg = bp[2 * 16 * UH + 16 * uh] * dsc, bo = bp[3 * 16 * UH + 16 * uh] * dsc;
    kI[uh] = __builtin_fmaf(bi, 0.2f, 0.5f); kF[uh] = __builtin_fmaf(bf, 0.2f, 0.5f);
    kO[uh] = __builtin_fmaf(bo, 0.2f, 0.5f); kG[uh] = bg * 2.885390081777927f;
    bzI[uh] = bi; bzF[uh] = bf; bzO[uh] = bo;
  }
  const int u0 = hg * 16 * UH + l15;                           // this lane's unit of half 0 (half uh: + 16 uh)
  const int hw_off = (u0 >> 3) * GS + (lrow0 + 4 * kq) * 8 + (u0 & 7);          // f16; + uh 2 GS + (16 rt + reg) 8; lo: + TERM
  const int hp_off = kq * GS + (lrow0 + l15) * 8;                               // f16; + kkr 4 GS + rt 128; lo: + TERM

  float* const wlw = wl + (wave * KBL * EPK) * 512 + lane * 4;     // this wave's resident weight fragments
  if constexpr (KBL > 0) {
#pragma unroll 1
    for (int e0 = 0; e0 < KBL * EPK; e0 += 4) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j / 2 < KBL * EPK) v[j] = buf_load16(wrs, wlane, ((e0 + j / 2) * 2 + (j & 1)) * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j / 2 < KBL * EPK) *(f32x4*)(wlw + ((e0 + j / 2) * 2 + (j & 1)) * 256) = v[j];
    }
  }
  for (int i = threadIdx.x; i < 2 * H; i += NTHREADS)
    bnl[i] = i < H ? P.out_scale[dir * H + i] : P.out_shift[dir * H + i - H];
  for (int i = threadIdx.x; i < HBUF; i += NTHREADS) hbuf[i] = 0.f;          // image of h_{-1} (buffer 0)
  float c[CLDS ? 1 : NE];
  if constexpr (CLDS) {
#pragma unroll
    for (int i = 0; i < NE; ++i) cl[i * NTHREADS + threadIdx.x] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) c[i] = 0.f;
  }
  __syncthreads();

  struct ABase {
    __amdgpu_buffer_rsrc_t r0[R], r1[R];
    unsigned v0[R][2], v1[R][2];
  };
  auto mk_base = [&](int s) __attribute__((always_inline)) {
    const int sc = s < T ? s : T - 1;                    // a step past the end aliases the last one (requests nobody consumes)
    const int t = dir ? (T - 1 - sc) : sc;
    ABase ab;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ab.r0[r] = make_rsrc(P.in0.ubase(row0 + r * 32, t), 0xffffffffu);
      if constexpr (KQ1 > 0) ab.r1[r] = make_rsrc(P.in1.ubase(row0 + r * 32, t), 0xffffffffu);
      else ab.r1[r] = ab.r0[r];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        ab.v0[r][sub] = (P.in0.voff(row0 + r * 32, t, l15 + 16 * sub, kq & 1) + (kq >> 1) * 512) * 4;
        if constexpr (KQ1 > 0) ab.v1[r][sub] = (P.in1.voff(row0 + r * 32, t, l15 + 16 * sub, kq & 1) + (kq >> 1) * 512) * 4;
        else ab.v1[r][sub] = 0;
      }
    }
    return ab;
  };
  struct BReg { f16x8 t[2]; };
  struct AReg { f32x4 v[2]; };
  BReg b[NBG];
  AReg a[NA][RT];
  // weight entry e = (EPK kk + 4 uh... ) in the order (kk, gate, uh) over the step's block sequence
  auto loadB = [&](int e, BReg& bb) __attribute__((always_inline)) {
    if (KBL > 0 && e < KBL * EPK) {
      bb.t[0] = __builtin_bit_cast(f16x8, *(const f32x4*)(wlw + (e * 2) * 256));
      bb.t[1] = __builtin_bit_cast(f16x8, *(const f32x4*)(wlw + (e * 2 + 1) * 256));
    } else {
      bb.t[0] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2) * 1024));
      bb.t[1] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2 + 1) * 1024));
    }
  };
  auto loadA_in = [&](const ABase& ab, int kk, int rt, AReg& d) __attribute__((always_inline)) {
    const int r = rt >> 1, sub = rt & 1;
    // the lo term first: the block's first product takes the hi term, so ONE counted wait covers both
    if (KQ1 == 0 || kk < KK0) {
      d.v[1] = buf_load16(ab.r0[r], ab.v0[r][sub], kk * 4096 + 1024);
      d.v[0] = buf_load16(ab.r0[r], ab.v0[r][sub], kk * 4096);
    } else {
      d.v[1] = buf_load16(ab.r1[r], ab.v1[r][sub], (kk - KK0) * 4096 + 1024);
      d.v[0] = buf_load16(ab.r1[r], ab.v1[r][sub], (kk - KK0) * 4096);
    }
  };
  auto loadA_rec = [&](const _Float16* hp, int kkr, int rt, AReg& d) __attribute__((always_inline)) {
    const _Float16* qh = hp + kkr * 4 * GS + rt * 128;
    d.v[1] = *(const f32x4*)(qh + TERM);                 // lo first, as for x
    d.v[0] = *(const f32x4*)(qh);
  };
  // One request at a time, issued BEHIND a product inside a chain of three instead of in front of the
  // entry / block (tools/microbench/tick_cost.hip: two 1 KB requests in front of an entry's 12 products cost 2.3 cycles
  // per product, inside a chain 0.6; lstm_h2w_kernel is built that way)
  auto loadB1 = [&](int e, int term, BReg& bb) __attribute__((always_inline)) {
    if (KBL > 0 && e < KBL * EPK) bb.t[term] = __builtin_bit_cast(f16x8, *(const f32x4*)(wlw + (e * 2 + term) * 256));
    else bb.t[term] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2 + term) * 1024));
  };
  auto loadA_in1 = [&](const ABase& ab, int kk, int rt, int term, AReg& d) __attribute__((always_inline)) {
    const int r = rt >> 1, sub = rt & 1;
    if (KQ1 == 0 || kk < KK0) d.v[term] = buf_load16(ab.r0[r], ab.v0[r][sub], kk * 4096 + term * 1024);
    else d.v[term] = buf_load16(ab.r1[r], ab.v1[r][sub], (kk - KK0) * 4096 + term * 1024);
  };
  auto loadA_rec1 = [&](const _Float16* hp, int kkr, int rt, int term, AReg& d) __attribute__((always_inline)) {
    d.v[term] = *(const f32x4*)(hp + kkr * 4 * GS + rt * 128 + term * TERM);
  };
  constexpr int APPE = (2 * RT + EPK - 1) / EPK;             // fragment pieces (row tile, term) per entry, behind products 4..
  static_assert(4 + APPE <= 3 * RT, "fragment pieces must fit behind an entry's products");
  // hi*lo, lo*hi, hi*hi: the first product of an entry takes the LAST-requested fragment of both operands
  constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};

  // ---- the VALU work of a step, cut into pieces of at most ~5 instructions (see lstm_h2o_kernel) ----
  // a tick is 16 cycles of which the MFMA holds the issue port for 8: at most two VALU instructions ride along
  // for free, so a gate element is cut into TWELVE stages here (a heavier piece delays its MFMA and the pipe
  // does not get the time back on the next, lighter tick)
  constexpr int GST = 14;
  struct GateSt { float zi, zf, zg, zo, cp, p, hv, t; _Float16 hh, hl; };
  auto gate_stage = [&](GateSt& g, const f32x4 (&Z)[4][UH][RT], _Float16* hw, int e, int st) __attribute__((always_inline)) {
    const int uh = e / (RT * 4), rt = (e / 4) % RT, reg = e % 4;
    if (st == 0) {
      g.zi = Z[0][uh][rt][reg]; g.zf = Z[1][uh][rt][reg];
      if constexpr (CLDS) g.cp = cl[e * NTHREADS + threadIdx.x];
      else g.cp = c[e];
    } else if (st == 1) {
      g.zg = Z[2][uh][rt][reg]; g.zo = Z[3][uh][rt][reg];
    } else if (st == 2) {
      if constexpr (ACT == 0) {
        g.zi = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zi, dsc02, kI[uh]), 0.0f), 1.0f);
        g.zf = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zf, dsc02, kF[uh]), 0.0f), 1.0f);
      } else {
        g.zi = sigmoid_exact(__builtin_fmaf(g.zi, dsc, bzI[uh]));
        g.zf = sigmoid_exact(__builtin_fmaf(g.zf, dsc, bzF[uh]));
      }
    } else if (st == 3) {
      if constexpr (ACT == 0) g.zo = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zo, dsc02, kO[uh]), 0.0f), 1.0f);
      else g.zo = sigmoid_exact(__builtin_fmaf(g.zo, dsc, bzO[uh]));
      g.zg = __builtin_fmaf(g.zg, dsc2, kG[uh]);
    } else if (st == 4) {
      g.zg = __builtin_amdgcn_exp2f(g.zg);
    } else if (st == 5) {
      g.t = __builtin_amdgcn_rcpf(g.zg + 1.0f);
    } else if (st == 6) {
      g.p = g.zi * __builtin_fmaf(g.t, -2.0f, 1.0f);
    } else if (st == 7) {
      const float cn = __builtin_fmaf(g.zf, g.cp, g.p);
      if constexpr (CLDS) cl[e * NTHREADS + threadIdx.x] = cn;
      else c[e] = cn;
      g.zg = cn * 2.885390081777927f;
    } else if (st == 8) {
      g.zg = __builtin_amdgcn_exp2f(g.zg);
    } else if (st == 9) {
      g.t = __builtin_amdgcn_rcpf(g.zg + 1.0f);
    } else if (st == 10) {
      // og * tanh(c) * 2^13, the scale riding on tanh's last fma
      g.hv = g.zo * __builtin_fmaf(g.t, -2.0f * kHScale, kHScale);
    } else if (st == 11) {
      g.hh = (_Float16)g.hv;
    } else if (st == 12) {
      g.hl = (_Float16)(g.hv - (float)g.hh);
    } else {
      hw[uh * 2 * GS + (rt * 16 + reg) * 8] = g.hh;
      hw[uh * 2 * GS + (rt * 16 + reg) * 8 + TERM] = g.hl;
    }
  };
  constexpr int ITEMS = (H / 16) * 2 * ROWS;
  constexpr int NIT = ITEMS / NTHREADS;
  static_assert(ITEMS % NTHREADS == 0, "copy-out items must divide evenly");
  constexpr int CST = RAW ? 4 : 12;                            // stages of one copy-out item
  struct CopySt { f16x8 hi, lo; f32x4 x0, x1; Split2 o; float* dst; };
  auto copy_stage = [&](CopySt& k, const _Float16* himg, int t, int i, int st) __attribute__((always_inline)) {
    const int it = threadIdx.x + i * NTHREADS;
    constexpr int KBH = H / 16;
    const int kbh = it / ROWS, rr = it % ROWS;          // kbh = 2*kbo + hf: features 8*kbh .. 8*kbh + 7 = unit group kbh
    if (st == 0) {
      k.hi = *(const f16x8*)(himg + kbh * GS + rr * 8);
      k.lo = *(const f16x8*)(himg + kbh * GS + rr * 8 + TERM);
      const int tile = blk.rowblk * (R * WR) + rr / 32;
      k.dst = P.out + ((size_t)(tile * T + t) * (2 * H / 4) + (dir * KBH + (kbh >> 1)) * 4 + (kbh & 1)) * 128 + (rr & 31) * 4;
    } else if (RAW) {
      if (st == 1) *(f16x8*)k.dst = k.hi;
      else if (st == 2) *(f16x8*)(k.dst + 2 * 128) = k.lo;
    } else if (st < 5) {                                // BatchNorm of hi + lo, two features per stage
      const int j0 = 2 * (st - 1);
#pragma unroll
      for (int j = j0; j < j0 + 2; ++j) {
        const float sc = bnl[kbh * 8 + j], sh = bnl[H + kbh * 8 + j];
        const float v = __builtin_fmaf((float)k.hi[j], sc, __builtin_fmaf((float)k.lo[j], sc, sh));
        if (j < 4) k.x0[j] = v; else k.x1[j - 4] = v;
      }
    } else if (st < 9) {                                // the split, two elements per stage
      const int j0 = 2 * (st - 5);
#pragma unroll
      for (int j = j0; j < j0 + 2; ++j) {
        const float x = j < 4 ? k.x0[j] : k.x1[j - 4];
        const _Float16 hh = (_Float16)x;
        k.o.t[0][j] = hh;
        k.o.t[1][j] = (_Float16)(x - (float)hh);
      }
    } else if (st == 9) {
      *(f16x8*)k.dst = k.o.t[0];
    } else if (st == 10) {
      *(f16x8*)(k.dst + 2 * 128) = k.o.t[1];
    }
  };

  // ---- in(): N = x W over the input blocks, one MFMA per TICK of 16 cycles; the gate stages of Z take the
  // ticks [0, TG), the barrier follows tick TG - 1, the copy-out stages take the ticks behind it.
  constexpr int NTICK = KK_IN * EPK * TPE;
  constexpr int NGP = NE * GST, NCP = NIT * CST;               // pieces
  constexpr int TG_WANT = NGP < (2 * NTICK) / 3 ? NGP : (2 * NTICK) / 3;
  constexpr int TG_MAX = (KK_IN - LA) * EPK * TPE;             // the rec() operands are requested from block KK_IN - LA on
  constexpr int TG = TG_WANT < TG_MAX ? TG_WANT : TG_MAX;
  constexpr int TC = NTICK - TG;
  static_assert(TG >= 1 && TC >= 1, "no room for the gates / copy-out in the input phase");
  auto in_phase = [&](auto work_tag, f32x4 (&N)[4][UH][RT], const f32x4 (&Z)[4][UH][RT], const ABase& xb,
                      const _Float16* hp_next, _Float16* himg_w, int t_out, const ABase& xb_wrap) __attribute__((always_inline)) {
    constexpr bool WORK = decltype(work_tag)::value;
    GateSt gs;
    CopySt cs;
    // (N is not zeroed: the first product of every tile takes a constant 0 as its C operand)
#pragma unroll
    for (int kk = 0; kk < KK_IN; ++kk) {
      if constexpr (WORK) NRV_STAMP_AT(8 + kk);
      const int ka = kk + LA;                            // activations LA blocks ahead: input, then recurrent
      {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (ka < KK_IN) loadA_in(xb, ka, rt, a[ka % NA][rt]);
          else if (WORK) loadA_rec(hp_next, ka - KK_IN, rt, a[ka % NA][rt]);
          else loadA_in(xb_wrap, ka - KK_IN, rt, a[ka % NA][rt]);
        }
      }
#pragma unroll
      for (int ge = 0; ge < EPK; ++ge) {
        const int e = EPK * kk + ge, g = ge / UH, uh = ge % UH;
        const int en = WORK ? (e + LBG) % (EPK * KK) : (e + LBG) % (EPK * KK_IN);
        loadB(en, b[(e + LBG) % NBG]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr) {
            const int tk = (e * RT + rt) * 3 + pr;
            N[g][uh][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[kk % NA][rt].v[PA[pr]]), b[e % NBG].t[PB[pr]],
                                      (kk == 0 && pr == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : N[g][uh][rt]);
            if constexpr (WORK) {
              if (tk < TG) {
#pragma unroll
                for (int pc = (tk * NGP) / TG; pc < ((tk + 1) * NGP) / TG; ++pc) {
                  gate_stage(gs, Z, himg_w + hw_off, pc / GST, pc % GST);
                }
              } else {
#pragma unroll
                for (int pc = ((tk - TG) * NCP) / TC; pc < ((tk - TG + 1) * NCP) / TC; ++pc)
                  copy_stage(cs, himg_w, t_out, pc / CST, pc % CST);
              }
              __builtin_amdgcn_sched_barrier(0);
              if (tk == TG - 1) {
                NRV_STAMP_AT(24);
                __syncthreads();                         // h_s complete: rec(s+1) operands and the copy-out may read it
                NRV_STAMP_AT(25);
              }
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- rec(): Z += h U over the recurrent blocks: MFMAs and requests only (the operand arrives split).
  auto rec_phase = [&](f32x4 (&Z)[4][UH][RT], const _Float16* hp, const ABase& xb_next) __attribute__((always_inline)) {
#pragma unroll
    for (int kr = 0; kr < KK_REC; ++kr) {
      const int kk = KK_IN + kr;
      NRV_STAMP_AT(1 + kr);
      const int ka = kk + LA;
#pragma unroll
      for (int ge = 0; ge < EPK; ++ge) {
        const int e = EPK * kk + ge, g = ge / UH, uh = ge % UH;
        const int en = (e + LBG) % (EPK * KK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr) {
            Z[g][uh][rt] = mfma16_f16(__builtin_bit_cast(f16x8, a[kk % NA][rt].v[PA[pr]]), b[e % NBG].t[PB[pr]], Z[g][uh][rt]);
            const int m = rt * 3 + pr, pa = ge * APPE + (m - 4);
            const bool fa = m >= 4 && m < 4 + APPE && pa < 2 * RT;
            if (m < 2 || fa) {
              __builtin_amdgcn_sched_barrier(0);
              if (m < 2) loadB1(en, m, b[(e + LBG) % NBG]);
              if (fa) {
                const int rn = pa / 2, tn = pa % 2;
                if (ka < KK) loadA_rec1(hp, ka - KK_IN, rn, tn, a[ka % NA][rn]);
                else loadA_in1(xb_next, ka - KK, rn, tn, a[ka % NA][rn]);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  f32x4 Z[4][UH][RT], N[4][UH][RT];
  auto himg = [&](int s) __attribute__((always_inline)) { return (_Float16*)(hbuf + ((s + 1) & 1) * HBUF); };   // image of h_s
  auto t_of = [&](int s) __attribute__((always_inline)) { return dir ? (T - 1 - s) : s; };

  // prologue: the rings' first entries, then in(0) straight into Z
  {
    const ABase x0 = mk_base(0), x1 = mk_base(1);
#pragma unroll
    for (int e = 0; e < LBG; ++e) loadB(e % (EPK * KK_IN), b[e]);
#pragma unroll
    for (int i = 0; i < LA; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) loadA_in(x0, i, rt, a[i][rt]);
    in_phase(std::false_type{}, Z, Z, x0, nullptr, nullptr, 0, x1);
  }
  // The steps 0 .. T - 2 (rec, then in() of the next step with this step's gates between its products) are the loop;
  // the LAST step (rec, plain gates, copy-out) stands behind it.  As two arms of one `if` inside the loop (r02-r03)
  // LLVM hoisted what both arms begin with - the read-out of Z and the hard_sigmoids of ALL 32 gate elements, 230
  // vector instructions - in front of the branch, i.e. behind rec()'s last product with nothing to hide behind, and
  // in()'s 128 accumulator-zeroing moves came on top: in-kernel stamps (scripts/gpu_stamps.py, r04a) put 1500 of a
  // step's 23.6 k cycles there.
  auto last_step = [&](int s) __attribute__((always_inline)) {
    NRV_STAMP_AT(26);
    // no input projection to hide behind: plain gates, barrier, copy-out
    GateSt gs;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int st = 0; st < GST; ++st) gate_stage(gs, Z, himg(s) + hw_off, e, st);
      if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    CopySt cs;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
      for (int st = 0; st < CST; ++st) copy_stage(cs, himg(s), t_of(s), i, st);
    NRV_STAMP_AT(27);
  };
#pragma unroll 1
  for (int s = 0; s + 1 < T; ++s) {
    // step s: Z holds x_s W on entry; on exit it holds x_{s+1} W and h_s has been written out
#if NRV_STAMP
    stamp_step = s;
#endif
    NRV_STAMP_AT(0);
    const ABase xn = mk_base(s + 1);
    if (s > 0) rec_phase(Z, himg(s - 1) + hp_off, xn);
    in_phase(std::true_type{}, N, Z, xn, himg(s) + hp_off, himg(s), t_of(s), xn);
    NRV_STAMP_AT(26);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int uh = 0; uh < UH; ++uh)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) Z[g][uh][rt] = N[g][uh][rt];
    NRV_STAMP_AT(27);
  }
  {
    const int s = T - 1;
#if NRV_STAMP
    stamp_step = s;
#endif
    NRV_STAMP_AT(0);
    const ABase xn = mk_base(s + 1);
    if (s > 0) rec_phase(Z, himg(s - 1) + hp_off, xn);
    last_step(s);
  }
#if NRV_STAMP
  {
    unsigned long long stamp_rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt1)::"memory");
    stamp_step = kStampSteps - 1;
    stamp_lds[(wave * kStampSteps + stamp_step) * kStampSlots + 0] = stamp_rt0;
    stamp_lds[(wave * kStampSteps + stamp_step) * kStampSlots + 1] = stamp_rt1;
    __syncthreads();
    if (blockIdx.x < kStampBlocks)
      for (int i = threadIdx.x; i < kStampWaves * kStampSteps * kStampSlots; i += NTHREADS)
        (&nrv_stamp_buf[H == 128 ? 0 : 1][blockIdx.x][0][0][0])[i] = stamp_lds[i];
  }
#endif
}

}  // namespace nrv
