// Bi-LSTM layer kernel on the f16 matrix pipe with the scaled two-term split (NRV_PREC_F16X2).
#pragma once
#include "nrv_lstm_f32.h"

// Experiment builds (tools/lstm_exp.sh, -DNRV_EXP=<bits>; 0 in the product): parts of lstm_h2o_kernel compiled
// out - results WRONG by construction, only cycles / clock / time mean something, and even those only with
// care: the chip's clock follows the kernel's power, which follows the DATA.
//   1 no gate / copy-out pieces   2 no weight loads behind the prologue   4 no activation loads
//   8 no split of the recurrent operand   64 device printf of clock64 / wall_clock64 deltas of workgroup 3
//   (of the 192->128 layer; +128: the 256->64 layer; +256: the 32->64 layer) - scripts/gpu_clk.sh
#ifndef NRV_EXP
#define NRV_EXP 0
#endif
namespace nrv {

// ---------------------------------------------------------------------------------------
// f16x2: f32-grade products from THREE matrix products instead of bf16x3's six.
//
// A value x, first multiplied by a power of two 2^s chosen so that the tensor's largest magnitude
// sits near 2^14 (exact; the f16 range is what made the unscaled form lose bits), is written as
// hi + lo with two f16 terms, both rounded to nearest even: |x 2^s - hi - lo| <= 2^-22 |x 2^s| for
// values whose lo term is a normal f16, and <= 2^-25 absolute (the f16 subnormal spacing) for the
// small ones - i.e. <= 2^-39 of the tensor's maximum, far below the rounding of an f32 dot product.
// a*b is formed as  lo*hi + hi*lo + hi*hi  on v_mfma_f32_32x32x16_f16 with f32 accumulation; the
// dropped lo*lo is below 2^-22 of the product.  tools/bf16_split_study.py (modes h3s / h3sz): max
// |dp| vs fp64 equal to the f32 pipe's on the fixture windows of both species, no argmax flips, with
// or without subnormal flushing.
//
// What makes it cheap on the device, beyond the halved MFMA count:
//   * activations travel BETWEEN kernels already split: every producer writes, instead of the f32
//     tile [kq][32 rows][4 f32], the same 512-byte chunks as [k-block][term hi|lo][half][32 rows][8 f16]
//     (same bytes, same chunk addressing: chunk index 4*kb + 2*term + half takes the place of kq), so
//     one wave-wide 16-byte load IS an A fragment of one term and the consumer splits nothing.  The
//     split happens once per element in the producer's epilogue instead of once per (element, hidden
//     group) in every consumer;
//   * the power-of-two scales are folded into constants that exist anyway: the producer's BatchNorm
//     scale/shift (x 2^s_out), the packed weights (x 2^(E - s_in)), the bias (x 2^E) and the gate
//     constants (x 2^-E), E being the layer's accumulator exponent (nrv_api.hip: plan_f16x2);
//   * only the recurrent operand h_{t-1} (|h| < 1, kept in LDS as f32 x 2^13) is split in the loop:
//     24 VALU ops per 12 MFMAs, in the shadow of the previous unit.
//
// Structure: as lstm_split_kernel (a wave owns 32 hidden units x 4 gates x R row tiles, c in
// registers, h through a double-buffered LDS image, one barrier per step), but the k-blocks of a
// step are FULLY unrolled: ring slots, operand types (pre-split input block / raw recurrent block)
// and the counted waits are static, weights and activations are requested a fixed number of entries
// ahead, and the requests run across the step boundary (the first blocks of step s+1 do not depend on h_s).
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Split2 { f16x8 t[2]; };

// x[0..7] (already scaled) -> hi + lo
__device__ __forceinline__ Split2 split2(const f32x4& lo4, const f32x4& hi4) {
  Split2 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? lo4[j] : hi4[j - 4];
    const _Float16 h = (_Float16)x;
    o.t[0][j] = h;
    o.t[1][j] = (_Float16)(x - (float)h);
  }
  return o;
}

// The same split with the low terms as v_fma_mix_f32(hi, -1, x): four instructions per pair instead of six.  (With the
// low terms converted by the same instruction, v_fma_mixlo / mixhi_f16 - three per pair - the signal branch and the
// 32->64 layer were measured SLOWER, r03.)  m1 is
// -1.0f in a register the compiler cannot see through (neg_one_opaque(), once per kernel): written as x - (float)h
// the subtraction needs v_cvt_f32_f16 first.
__device__ __forceinline__ float neg_one_opaque() {
  float m1;
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  return m1;
}
__device__ __forceinline__ Split2 split2(const f32x4& lo4, const f32x4& hi4, const float m1) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  Split2 o;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const float x0 = j < 4 ? lo4[j] : hi4[j - 4], x1 = j < 4 ? lo4[j + 1] : hi4[j - 3];
    const h2_t h = __builtin_convertvector(f2_t{x0, x1}, h2_t);
    const h2_t l = __builtin_convertvector(f2_t{__builtin_fmaf((float)h[0], m1, x0), __builtin_fmaf((float)h[1], m1, x1)}, h2_t);
    o.t[0][j] = h[0]; o.t[0][j + 1] = h[1];
    o.t[1][j] = l[0]; o.t[1][j + 1] = l[1];
  }
  return o;
}

__device__ __forceinline__ f32x16 mfma_f16(const f16x8& a, const f16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// gate nonlinearities of an accumulator that holds z * 2^E: the descale rides on constants
template <int ACT>
__device__ __forceinline__ float gate_act_scaled(float acc, float d, float d02) {
  if constexpr (ACT == 0) return __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(acc, d02, 0.5f), 0.0f), 1.0f);
  else return sigmoid_exact(acc * d);
}
__device__ __forceinline__ float tanh_fast_scaled(float acc, float d2) {    // d2 = 2 log2(e) * 2^-E
  const float e = __builtin_amdgcn_exp2f(acc * d2);
  return __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
}

constexpr float kHScale = 8192.0f;      // h_{t-1} in LDS is h * 2^13 (|h| < 1)

struct LstmH2ModelParams {
  const void* wsplit;     // [dir][hg][kb][gate][term 2][64 lanes][8 f16], already x 2^(E - s_in); kb: input then recurrent
  const float* bias;      // [dir][hg][gate][32], x 2^E
  const float* out_scale; // [2H]  BatchNorm scale x 2^(s_out - 13)   (the image holds h * 2^13)
  const float* out_shift; // [2H]  BatchNorm shift x 2^s_out
  ActView in0, in1;       // split-plane buffers (chunk index = 4*kb + 2*term + half)
  float* out;             // split planes [tiles][T][2H/16][term][half][32][8 f16], or f32 tiles when OUT_F32
  float descale;          // 2^-E
};
struct LstmH2Args {
  LstmH2ModelParams m[2];
  int T;
  int n_rows;
  int n_blk;
};

// ---------------------------------------------------------------------------------------
// lstm_h2o_kernel: grid = lstm_grid(ceil(tiles/(R*WR))), block = 64*NG*WR.  K0 = 4*KQ0, K1 = 4*KQ1, H multiples
// of 16.  The VALU work of a step runs in the matrix shadow of the next step's input blocks.
//
// PMC of its predecessor with the gates AFTER the matrix phase (lstm_h2_kernel, round 2, in the history;
// 192->128, profiles/r02b_*): per step a wave spends 15.4 k cycles issuing its
// 480 MFMAs and another ~8.6 k on ~1700 VALU instructions (gates, the split of h, BatchNorm + split of
// the output) that run AFTER the matrix phase - the kernel's time is the SUM (matrix pipe 54 % busy).
// Here the input blocks of step s+1, which do not depend on h_s, are issued while the VALU turns z_s
// into (c_s, h_s), passes the barrier and writes h_s out:
//     in(0)                                                  prologue
//     rec(s):   Z += h_{s-1} U          recurrent blocks, the operand split in their own shadow
//     in(s+1):  N  = x_{s+1} W          input blocks (the bias rides on the gate constants); between their MFMAs: the gate elements of
//                                       Z, then the barrier, then the copy-out items of h_s
//     Z <- N                            two accumulator sets; N is moved into Z at the end of the step
// so that the serial section of a step is rec(s) alone.  The weight ring is kept per (k-block, gate)
// entry (8 registers, NBG - 1 entries of lead) instead of per k-block: at R = 2 the two accumulator
// sets take all 256 AGPRs and everything else must fit the 256 VGPRs.  Step 0 has no rec() (h_{-1} = 0:
// the prologue's requests wrap around to step 1's input blocks instead) and the last step no in()
// (its gates run plainly): no MFMA is issued whose product is not used.
//
// GPT: gates per 32-column accumulator tile.  1: a wave owns 32 hidden units, one tile per gate.  2: a
// wave owns 16 units, tile 0 = [i | f], tile 1 = [g | o] (16 columns each): H = 64 then spreads over four
// waves instead of two pairs of waves that fetch the same weights, each weight fragment feeds R = 2 row
// tiles, and the CU's vector-memory path carries the layer's weights once per step instead of twice
// (that path, 64 B/clk, was what paced the 256->64 layer: 768 KB per step against 7.7 k MFMA cycles).
// The halves of a tile are brought together for the gate math by one v_permlane16_swap per register
// pair: lanes 0-15 end up with i, f, g, o of rows 0-3 / 8-11 of their unit, lanes 16-31 with rows 16-19 /
// 24-27 (+4 in the upper half of the wave).
// ---------------------------------------------------------------------------------------
// KBL: the weight fragments of the first KBL input blocks stay in LDS for the whole launch (each wave its own
// 2 NGT KBL KB, copied once in the prologue) and enter the ring by ds_read instead of from L2.
template <int KQ0, int KQ1, int H, int R, int WR, int ACT, bool OUT_F32, int NBG, int NA, int GPT = 1, int KBL = 0>
__global__ void __launch_bounds__(64 * ((H * GPT + 31) / 32) * WR)
lstm_h2o_kernel(const LstmH2Args args) {
  static_assert(GPT == 1 || GPT == 2, "gates per tile");
  constexpr int UPW = 32 / GPT, NGT = 4 / GPT;                 // hidden units per wave / accumulator tiles per row tile
  constexpr int NG = (H + UPW - 1) / UPW;
  constexpr int KB0 = KQ0 / 4, KB1 = KQ1 / 4, KB_IN = KB0 + KB1, KB_REC = H / 16, KB = KB_IN + KB_REC;
  constexpr int ROWS = 32 * R * WR;
  constexpr int PLANE = ROWS * 4 + 4;
  constexpr int HBUF = (NG * UPW / 4) * PLANE;
  constexpr int NTHREADS = 64 * NG * WR;
  constexpr int LBG = NBG - 1, LA = NA - 1;
  static_assert(KQ0 % 4 == 0 && KQ1 % 4 == 0 && H % 16 == 0, "K must come in blocks of 16");
  static_assert((NGT * KB) % NBG == 0 && (NGT * KB_IN) % NBG == 0 && KB % NA == 0 && KB_IN % NA == 0,
                "ring sizes must divide the block counts");
  static_assert(LA >= 1 && LA <= KB_IN && LA <= KB_REC && LBG <= NGT * KB_REC, "leads must stay inside a phase");
  static_assert(H % UPW == 0, "hidden units per wave");

  constexpr bool CLDS = R >= 2 && GPT == 1;
  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];
  __shared__ __attribute__((aligned(16))) float bnl[2 * H];
  __shared__ float cl[CLDS ? 16 * R * NTHREADS : 1];
  constexpr int NWV = NG * WR;
  __shared__ __attribute__((aligned(16))) float wl[KBL > 0 ? NWV * KBL * NGT * 512 : 4];

#if NRV_EXP & 64
  const unsigned long long exp_c0 = clock64(), exp_w0 = wall_clock64();    // shader clock / 100 MHz
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hg = wave % NG, wr = wave / NG;
  const int half = lane >> 5, l31 = lane & 31;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmH2ModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS + wr * (32 * R);
  const int lrow0 = wr * (32 * R);

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + hg) * KB) * (NGT * 2 * 1024), KB * NGT * 2 * 1024);
  const unsigned wlane = lane * 16;
  const int ul = l31 % UPW;                                   // this lane's unit inside the wave's group
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * UPW + ul;
  // The bias (x 2^E in memory) is not added to the accumulators - a pre-splatted 16-register tile per
  // gate would sit in the register file for the whole launch - it rides on the gate constants:
  //   hard_sigmoid(z) = clamp(acc * 0.2d + (0.5 + 0.2 b)),  tanh(z) from exp2(acc * 2log2e d + 2log2e b)
  const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
  const float bz[4] = {bp[0] * dsc, bp[UPW] * dsc, bp[2 * UPW] * dsc, bp[3 * UPW] * dsc};      // exact: d is a power of two
  const float kI = __builtin_fmaf(bz[0], 0.2f, 0.5f), kF = __builtin_fmaf(bz[1], 0.2f, 0.5f),
              kO = __builtin_fmaf(bz[3], 0.2f, 0.5f), kG = bz[2] * 2.885390081777927f;
  const int u = hg * UPW + ul;
  // rows of this lane's elements: accumulator register j -> (j & 3) + 8 (j >> 2) + 4 half; with two gates
  // per tile lanes 16-31 take the registers 8-15 of their tile (16 rows further down) after the exchange
  const int hw_off = (u >> 2) * PLANE + (u & 3) + (lrow0 + 4 * half + (GPT == 2 ? 16 * (l31 >> 4) : 0)) * 4;
  const int hp_off = (2 * half) * PLANE + (lrow0 + l31) * 4;

  float* const wlw = wl + (wave * KBL * NGT) * 512 + lane * 4;     // this wave's resident weight fragments
  if constexpr (KBL > 0) {
#pragma unroll 1
    for (int e0 = 0; e0 < KBL * NGT; e0 += 4) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j / 2 < KBL * NGT) v[j] = buf_load16(wrs, wlane, ((e0 + j / 2) * 2 + (j & 1)) * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j / 2 < KBL * NGT) *(f32x4*)(wlw + ((e0 + j / 2) * 2 + (j & 1)) * 256) = v[j];
    }
  }
  for (int i = threadIdx.x; i < 2 * H; i += NTHREADS)
    bnl[i] = i < H ? P.out_scale[dir * H + i] : P.out_shift[dir * H + i - H];
  for (int i = threadIdx.x; i < HBUF; i += NTHREADS) hbuf[i] = 0.f;          // image of h_{-1} (buffer 0)
  constexpr int EPR = 16 / GPT, NE = EPR * R;                  // gate elements per lane: per row tile / in all
  float c[CLDS ? 1 : NE];
  if constexpr (CLDS) {
#pragma unroll
    for (int i = 0; i < 16 * R; ++i) cl[i * NTHREADS + threadIdx.x] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) c[i] = 0.f;
  }
  __syncthreads();

  struct ABase {
    __amdgpu_buffer_rsrc_t r0[R], r1[R];
    unsigned v0[R], v1[R];
  };
  auto mk_base = [&](int s) __attribute__((always_inline)) {
    const int sc = s < T ? s : T - 1;                    // a step past the end aliases the last one (requests nobody consumes)
    const int t = dir ? (T - 1 - sc) : sc;
    ABase ab;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ab.r0[r] = make_rsrc(P.in0.ubase(row0 + r * 32, t), 0xffffffffu);
      ab.v0[r] = P.in0.voff(row0 + r * 32, t, l31, half) * 4;
      if constexpr (KQ1 > 0) {
        ab.r1[r] = make_rsrc(P.in1.ubase(row0 + r * 32, t), 0xffffffffu);
        ab.v1[r] = P.in1.voff(row0 + r * 32, t, l31, half) * 4;
      } else {
        ab.r1[r] = ab.r0[r];
        ab.v1[r] = 0;
      }
    }
    return ab;
  };
  struct BReg { f16x8 t[2]; };
  struct AReg { f32x4 v[2]; };
  BReg b[NBG];
  AReg a[NA][R];
#if NRV_EXP
  bool exp_steady = false;      // experiments (tools/lstm_exp.sh): requests compiled out behind the prologue
#endif
  // weight entry e = NGT*kb + g over the step's block sequence (input blocks, then recurrent blocks)
  auto loadB = [&](int e, BReg& bb) __attribute__((always_inline)) {
#if NRV_EXP & 2
    if (exp_steady) return;
#endif
    if (KBL > 0 && e < KBL * NGT) {
      bb.t[0] = __builtin_bit_cast(f16x8, *(const f32x4*)(wlw + (e * 2) * 256));
      bb.t[1] = __builtin_bit_cast(f16x8, *(const f32x4*)(wlw + (e * 2 + 1) * 256));
    } else {
      bb.t[0] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2) * 1024));
      bb.t[1] = __builtin_bit_cast(f16x8, buf_load16(wrs, wlane, (e * 2 + 1) * 1024));
    }
  };
  auto loadA_in = [&](const ABase& ab, int kb, int r, AReg& d) __attribute__((always_inline)) {
#if NRV_EXP & 4
    if (exp_steady) return;
#endif
    // the lo term first: the block's first product takes the hi term, so ONE counted wait covers both
    if (KQ1 == 0 || kb < KB0) {
      d.v[1] = buf_load16(ab.r0[r], ab.v0[r], kb * 2048 + 1024);
      d.v[0] = buf_load16(ab.r0[r], ab.v0[r], kb * 2048);
    } else {
      d.v[1] = buf_load16(ab.r1[r], ab.v1[r], (kb - KB0) * 2048 + 1024);
      d.v[0] = buf_load16(ab.r1[r], ab.v1[r], (kb - KB0) * 2048);
    }
  };
  auto loadA_rec = [&](const float* hp, int kbr, int r, AReg& d) __attribute__((always_inline)) {
#if NRV_EXP & 4
    if (exp_steady) return;
#endif
    const float* qh = hp + kbr * 4 * PLANE + r * 128;
    d.v[0] = *(const f32x4*)(qh);
    d.v[1] = *(const f32x4*)(qh + PLANE);
  };
  // hi*lo, lo*hi, hi*hi: the first product of an entry takes the LAST-requested fragment of both operands
  constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};

  // ---- the VALU work of a step, cut into PIECES of at most ~5 instructions ------------------------
  // An MFMA holds the SIMD's issue port for 8 of its 32 cycles; what a wave issues in the other 24 is
  // free, what exceeds them delays the next MFMA (the wave issues in order).  hipcc's scheduler does
  // not spread a dependent chain (one gate element is ~25 dependent instructions) between MFMAs by
  // itself, so the chains are cut by hand into stages, one stage per MFMA "tick", each tick fenced.
  constexpr int GST = 6;                                       // stages of one gate element
  struct GateSt { float zi, zf, zg, zo, cp, p, hv; };
  auto gate_stage = [&](GateSt& g, const f32x16 (&Z)[NGT][R], float* hw, int e, int st) __attribute__((always_inline)) {
    const int r = e / EPR, reg = e % EPR;
    if (st == 0) {
      if constexpr (GPT == 1) {
        g.zi = Z[0][r][reg]; g.zf = Z[1][r][reg]; g.zg = Z[2][r][reg]; g.zo = Z[3][r][reg];
      } else {
        // register j of a tile holds [gate a | gate b] of rows(j), register 8+j the same of rows(j) + 16:
        // swapping lanes 16-31 of the first with lanes 0-15 of the second leaves (a, b) of one row set per lane
        // (scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this hipcc)
        const float a0 = Z[0][r][reg], b0 = Z[0][r][8 + reg], a1 = Z[1][r][reg], b1 = Z[1][r][8 + reg];
        const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
        const unsigned ui = s0[0], uf = s0[1], ug = s1[0], uo = s1[1];
        g.zi = __builtin_bit_cast(float, ui); g.zf = __builtin_bit_cast(float, uf);
        g.zg = __builtin_bit_cast(float, ug); g.zo = __builtin_bit_cast(float, uo);
      }
      if constexpr (CLDS) g.cp = cl[e * NTHREADS + threadIdx.x];
      else g.cp = c[e];
    } else if (st == 1) {
      if constexpr (ACT == 0) {
        g.zi = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zi, dsc02, kI), 0.0f), 1.0f);
        g.zf = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zf, dsc02, kF), 0.0f), 1.0f);
        g.zo = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(g.zo, dsc02, kO), 0.0f), 1.0f);
      } else {
        g.zi = sigmoid_exact(__builtin_fmaf(g.zi, dsc, bz[0]));
        g.zf = sigmoid_exact(__builtin_fmaf(g.zf, dsc, bz[1]));
        g.zo = sigmoid_exact(__builtin_fmaf(g.zo, dsc, bz[3]));
      }
      g.zg = __builtin_amdgcn_exp2f(__builtin_fmaf(g.zg, dsc2, kG));
    } else if (st == 2) {
      const float gg = __builtin_fmaf(__builtin_amdgcn_rcpf(g.zg + 1.0f), -2.0f, 1.0f);
      g.p = g.zi * gg;
    } else if (st == 3) {
      const float cn = __builtin_fmaf(g.zf, g.cp, g.p);
      if constexpr (CLDS) cl[e * NTHREADS + threadIdx.x] = cn;
      else c[e] = cn;
      g.zg = __builtin_amdgcn_exp2f(cn * 2.885390081777927f);
    } else if (st == 4) {
      // og * tanh(c) * 2^13, the scale riding on tanh's last fma
      const float th = __builtin_fmaf(__builtin_amdgcn_rcpf(g.zg + 1.0f), -2.0f * kHScale, kHScale);
      g.hv = g.zo * th;
    } else {
      hw[(r * 32 + (reg & 3) + 8 * (reg >> 2)) * 4] = g.hv;
    }
  };
  constexpr int ITEMS = OUT_F32 ? (H / 4) * ROWS : (H / 16) * 2 * ROWS;
  constexpr int NIT = (ITEMS + NTHREADS - 1) / NTHREADS;
  static_assert(ITEMS % NTHREADS == 0, "copy-out items must divide evenly");
  constexpr int CST = OUT_F32 ? 3 : 8;                         // stages of one copy-out item
  struct CopySt { f32x4 x0, x1, s0, s1, h0, h1; Split2 o; float* dst; };
  auto copy_stage = [&](CopySt& k, const float* himg, int t, int i, int st) __attribute__((always_inline)) {
    const int it = threadIdx.x + i * NTHREADS;
    if constexpr (OUT_F32) {
      constexpr int KQH = H / 4;
      const int kq = it / ROWS, rr = it % ROWS;
      if (st == 0) {
        k.x0 = *(const f32x4*)(himg + kq * PLANE + rr * 4);
        k.s0 = *(const f32x4*)(bnl + kq * 4);
        k.h0 = *(const f32x4*)(bnl + H + kq * 4);
        const int tile = blk.rowblk * (R * WR) + rr / 32;
        k.dst = P.out + ((size_t)(tile * T + t) * (2 * KQH) + dir * KQH + kq) * 128 + (rr & 31) * 4;
      } else if (st == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) k.x0[q] = k.x0[q] * k.s0[q] + k.h0[q];
      } else {
        *(f32x4*)k.dst = k.x0;
      }
    } else {
      constexpr int KBH = H / 16;
      const int kbh = it / ROWS, rr = it % ROWS;        // kbh = 2*kbo + hf: features 8*kbh .. 8*kbh + 7
      const int kq = 2 * kbh;
      if (st == 0) {
        k.x0 = *(const f32x4*)(himg + kq * PLANE + rr * 4);
        k.x1 = *(const f32x4*)(himg + (kq + 1) * PLANE + rr * 4);
        k.s0 = *(const f32x4*)(bnl + kq * 4); k.s1 = *(const f32x4*)(bnl + kq * 4 + 4);
        k.h0 = *(const f32x4*)(bnl + H + kq * 4); k.h1 = *(const f32x4*)(bnl + H + kq * 4 + 4);
        const int tile = blk.rowblk * (R * WR) + rr / 32;
        k.dst = P.out + ((size_t)(tile * T + t) * (2 * H / 4) + (dir * KBH + (kbh >> 1)) * 4 + (kbh & 1)) * 128 + (rr & 31) * 4;
      } else if (st == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) k.x0[q] = k.x0[q] * k.s0[q] + k.h0[q];
      } else if (st == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) k.x1[q] = k.x1[q] * k.s1[q] + k.h1[q];
      } else if (st < 7) {                              // the split, two elements per stage
        const int j0 = 2 * (st - 3);
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
          const float x = j < 4 ? k.x0[j] : k.x1[j - 4];
          const _Float16 hh = (_Float16)x;
          k.o.t[0][j] = hh;
          k.o.t[1][j] = (_Float16)(x - (float)hh);
        }
      } else {
        *(f16x8*)k.dst = k.o.t[0];
        *(f16x8*)(k.dst + 2 * 128) = k.o.t[1];
      }
    }
  };
  // the split stage above covers elements 0..7 in stages 3..6
  static_assert(OUT_F32 || CST == 8, "copy-out stage table");

  // ---- in(): N = b + x W over the input blocks of step sx (bases xb), one MFMA per TICK.  The gate
  // stages of Z occupy the ticks [0, TG), the barrier follows tick TG - 1, the copy-out stages of the
  // finished image take the ticks behind it.  WORK = false: the prologue (no Z yet).
  // hp_next: image the following rec() reads; its first LA blocks are requested here, behind the barrier.
  constexpr int NTICK = KB_IN * NGT * 3 * R;
  constexpr int NGP = NE * GST, NCP = NIT * CST;               // pieces
  constexpr int TG_WANT = NGP < (2 * NTICK) / 3 ? NGP : (2 * NTICK) / 3;
  constexpr int TG_MAX = (KB_IN - LA) * NGT * 3 * R;             // the rec() operands are requested from block KB_IN - LA on
  constexpr int TG = TG_WANT < TG_MAX ? TG_WANT : TG_MAX;
  constexpr int TC = NTICK - TG;
  static_assert(TG >= 1 && TC >= 1, "no room for the gates / copy-out in the input phase");
  auto in_phase = [&](auto work_tag, f32x16 (&N)[NGT][R], const f32x16 (&Z)[NGT][R], const ABase& xb,
                      const float* hp_next, float* himg_w, int t_out, const ABase& xb_wrap) __attribute__((always_inline)) {
    // WORK = false is the prologue, in(0): step 0 has no recurrent blocks (h_{-1} = 0), so what follows
    // it is in(1), and its tail requests wrap around to the input blocks / weights of step 1 (xb_wrap)
    constexpr bool WORK = decltype(work_tag)::value;
    GateSt gs;
    CopySt cs;
#pragma unroll
    for (int g = 0; g < NGT; ++g)
#pragma unroll
      for (int r = 0; r < R; ++r) N[g][r] = splat16(0.0f);
#pragma unroll
    for (int kb = 0; kb < KB_IN; ++kb) {
      {
        const int ka = kb + LA;                          // activations LA blocks ahead: input, then recurrent
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (ka < KB_IN) loadA_in(xb, ka, r, a[ka % NA][r]);
          else if (WORK) loadA_rec(hp_next, ka - KB_IN, r, a[ka % NA][r]);
          else loadA_in(xb_wrap, ka - KB_IN, r, a[ka % NA][r]);
        }
      }
#pragma unroll
      for (int g = 0; g < NGT; ++g) {
        const int e = NGT * kb + g;
        loadB(WORK ? (e + LBG) % (NGT * KB) : (e + LBG) % (NGT * KB_IN), b[(e + LBG) % NBG]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr) {
            const int tk = (e * R + r) * 3 + pr;
            N[g][r] = mfma_f16(__builtin_bit_cast(f16x8, a[kb % NA][r].v[PA[pr]]), b[e % NBG].t[PB[pr]], N[g][r]);
            if constexpr (WORK) {
#if NRV_EXP & 1
              if (false) {
#else
              if (tk < TG) {
#endif
#pragma unroll
                for (int pc = (tk * NGP) / TG; pc < ((tk + 1) * NGP) / TG; ++pc)
                  gate_stage(gs, Z, himg_w + hw_off, pc / GST, pc % GST);
              } else if (!(NRV_EXP & 1)) {
#pragma unroll
                for (int pc = ((tk - TG) * NCP) / TC; pc < ((tk - TG + 1) * NCP) / TC; ++pc)
                  copy_stage(cs, himg_w, t_out, pc / CST, pc % CST);
              }
              __builtin_amdgcn_sched_barrier(0);
              if (tk == TG - 1) __syncthreads();         // h_s complete: rec(s+1) operands and the copy-out may read it
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- rec(): Z += h U over the recurrent blocks of the image at hp; the split of the next block's
  // units runs in the shadow of this block's MFMAs.  The first LA input blocks of the in() phase that
  // follows (bases xb_next) are requested here.
  auto rec_phase = [&](f32x16 (&Z)[NGT][R], const float* hp, const ABase& xb_next) __attribute__((always_inline)) {
    Split2 sp[2][R];
    // the split of one block, cut like the gates: two values of one row tile go through three stages
    // (hi terms; residuals; lo terms), one stage per MFMA tick: 4R pairs over the block's 12R ticks
    constexpr int NSP = 3 * 4 * R, NTK = NGT * 3 * R;          // stage pieces / MFMA ticks per k-block
    struct SplitSt { float d0, d1; };
    SplitSt ss;
    auto split_piece = [&](Split2& o, const AReg& src, int j0, int st) __attribute__((always_inline)) {
#if NRV_EXP & 8
      if (j0 == 0 && st == 0) { o.t[0] = __builtin_bit_cast(f16x8, src.v[0]); o.t[1] = __builtin_bit_cast(f16x8, src.v[1]); }
#else
      const float x0 = j0 < 4 ? src.v[0][j0] : src.v[1][j0 - 4], x1 = j0 < 4 ? src.v[0][j0 + 1] : src.v[1][j0 - 3];
      if (st == 0) {
        o.t[0][j0] = (_Float16)x0;
        o.t[0][j0 + 1] = (_Float16)x1;
      } else if (st == 1) {
        ss.d0 = x0 - (float)o.t[0][j0];
        ss.d1 = x1 - (float)o.t[0][j0 + 1];
      } else {
        o.t[1][j0] = (_Float16)ss.d0;
        o.t[1][j0 + 1] = (_Float16)ss.d1;
      }
#endif
    };
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j0 = 0; j0 < 8; j0 += 2)
#pragma unroll
        for (int st = 0; st < 3; ++st) split_piece(sp[0][r], a[KB_IN % NA][r], j0, st);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kr = 0; kr < KB_REC; ++kr) {
      const int kb = KB_IN + kr;
      {
        const int ka = kb + LA;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (ka < KB) loadA_rec(hp, ka - KB_IN, r, a[ka % NA][r]);
          else loadA_in(xb_next, ka - KB, r, a[ka % NA][r]);
        }
      }
#pragma unroll
      for (int g = 0; g < NGT; ++g) {
        const int e = NGT * kb + g;
        loadB((e + LBG) % (NGT * KB), b[(e + LBG) % NBG]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr) {
            const int tk = (g * R + r) * 3 + pr;
            Z[g][r] = mfma_f16(sp[kr & 1][r].t[PA[pr]], b[e % NBG].t[PB[pr]], Z[g][r]);
            if (kr + 1 < KB_REC) {
#pragma unroll
              for (int pc = (tk * NSP) / NTK; pc < ((tk + 1) * NSP) / NTK; ++pc)
                split_piece(sp[(kr + 1) & 1][pc / 12], a[(kb + 1) % NA][pc / 12], 2 * ((pc / 3) % 4), pc % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
  };

  // Z: x_s W (+ h_{s-1} U after rec()); N: the next step's input projection.  At the end of a step N is
  // moved into Z (128 v_accvgpr_mov at R = 2, ~3 % of a step).  Alternating the roles of two sets in a
  // body unrolled x2 avoids the moves on paper, but hipcc then cannot keep either set in place across
  // the loop edge (it inserted more moves than this, plus 64 spilled registers).
  f32x16 Z[NGT][R], N[NGT][R];
  auto himg = [&](int s) __attribute__((always_inline)) { return hbuf + ((s + 1) & 1) * HBUF; };   // image of h_s
  auto t_of = [&](int s) __attribute__((always_inline)) { return dir ? (T - 1 - s) : s; };

  // prologue: the rings' first entries, then in(0) straight into Z
  {
    const ABase x0 = mk_base(0), x1 = mk_base(1);
#pragma unroll
    for (int e = 0; e < LBG; ++e) loadB(e % (NGT * KB_IN), b[e]);
#pragma unroll
    for (int i = 0; i < LA; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r) loadA_in(x0, i, r, a[i][r]);
    in_phase(std::false_type{}, Z, Z, x0, nullptr, nullptr, 0, x1);
  }
#if NRV_EXP
  exp_steady = true;
#endif
#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    // step s: Z holds x_s W on entry; on exit it holds x_{s+1} W and h_s has been written out
    const ABase xn = mk_base(s + 1);
    if (s > 0) rec_phase(Z, himg(s - 1) + hp_off, xn);
    if (s + 1 < T) {
      in_phase(std::true_type{}, N, Z, xn, himg(s) + hp_off, himg(s), t_of(s), xn);
#pragma unroll
      for (int g = 0; g < NGT; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) Z[g][r] = N[g][r];
    } else {
      // the last step has no input projection to hide behind: plain gates, barrier, copy-out
      GateSt gs;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
#pragma unroll
        for (int st = 0; st < GST; ++st) gate_stage(gs, Z, himg(s) + hw_off, e, st);
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the accumulator read-out local
      }
      __syncthreads();
      CopySt cs;
#pragma unroll
      for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int st = 0; st < CST; ++st) copy_stage(cs, himg(s), t_of(s), i, st);
    }
  }
#if NRV_EXP & 64
  if (blockIdx.x == 3 && threadIdx.x == 0 && ((NRV_EXP & 256) ? (H == 64 && KQ0 == 8) : (NRV_EXP & 128) ? (H == 64 && KQ0 == 64) : H == 128))
    printf("CLK %llu %llu\n", (unsigned long long)(clock64() - exp_c0), (unsigned long long)(wall_clock64() - exp_w0));
#endif
}

}  // namespace nrv
