// Second read-branch layer, Bi-LSTM(32 -> 64), f16x2 mode: TRANSPOSED products, wave-private recurrence.
#pragma once
#include "nrv_lstm_f16x2s.h"   // mfma16_f16

namespace nrv {

// ---------------------------------------------------------------------------------------
// lstm2_t_kernel.  output_handeler.py:220 (Bidirectional(LSTM(64)) on the 32 features of read_rnn1).
//
// The layer is small (10.5 GFLOP per 4096-window group, 98 KB of f16x2 weights per direction) and was
// LATENCY-bound in the generic layout (lstm_h2o_kernel<8,0,64>, round 2: 44-48 us, matrix pipe 27 % busy):
// with hidden units on the lanes, h_t has to cross the workgroup through an LDS image every step - split,
// 2-byte stores, a barrier, fragment reads - before the next recurrent product can start, and at 4096
// windows there is exactly one such dependent chain per SIMD.
//
// Here the products are transposed, z^T = [W | U]^T [x ; h]: the WEIGHTS are the A operand (M = the 256
// gate-units), the activations the B operand (N = 16 data rows), so a result tile has the data row on the
// lane and the gate-units in the registers:  tile (gate g, unit tile ut) of v_mfma_f32_16x16x32_f16 gives
// lane l = (row n = l & 15, q = l >> 4) the units 16 ut + 4 q + r, r = 0..3.  One wave owns ALL 256 gate-units
// of its 16 rows (16 tiles = 64 accumulator registers), so i, f, g, o of a (row, unit) meet in one lane, c and h
// never leave the lane, and the 16 values of h_t a lane computes ARE - as two f16x8 terms, elements 8 kb + j =
// unit 16 (2 kb + (j >> 2)) + 4 q + (j & 3) - its B fragments of the next step's two recurrent k-blocks: the host
// packs the rows of U in that k order.  No LDS image of h, no barrier, no inter-wave traffic at all; the
// four waves of a workgroup only share the weight image.
//   * weights: the whole f16x2 set of one (direction, model), [kb 3][tile 16][term 2] fragments of 1 KiB
//     (96 KiB), is copied to LDS once per workgroup and every wave streams it from there each step
//     (96 ds_read_b128 per wave and step, ~140 of the LDS's 256 B/clk); the bias enters as the initial
//     value of the accumulators, read from a [tile][lane] image (16 KiB) - no constant registers;
//   * schedule of step s, ONE accumulator set:  rec(s) by unit tile - the gate math of unit tile ut - 1
//     runs between the MFMAs of unit tile ut, cut into stages of <= 2 instructions per 16-cycle MFMA tick
//     as in lstm_h2s_kernel - then the input projection of step s+1 into the tiles whose gates are done,
//     with the gates of the last unit tile and the output stores between its MFMAs;
//   * output: h x 2^13 as f16 split planes, RAW - the BatchNorm(128) behind this layer lives in the weights
//     of the 192->128 layer's first 128 input rows (nrv_api.hip upload_model) - and each lane's 4 units of
//     a unit tile are 8 contiguous bytes of a plane row.
// grid = (ceil(rows / 64), 2 directions, 2 models), block = 256: four independent waves of 16 rows.
// ---------------------------------------------------------------------------------------
struct Lstm2TModelParams {
  const void* wfrag;      // [dir][kb 3 (input, rec 0, rec 1)][tile 16][term 2][64 lanes][8 f16], x 2^(E - s)
  const float* bias;      // [dir][tile 16][64 lanes][4], x 2^E, in accumulator layout
  const float* in;        // X1 split planes [tile32][T][kb16 2][term][half][32][8 f16]
  float* out;             // X2 split planes [tile32][T][kb16 8][term][half][32][8 f16], h x 2^13
  float descale;          // 2^-E
};
struct Lstm2TArgs {
  Lstm2TModelParams m[2];
  int T;
  int n_rows;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kL2tThreads = 256;
constexpr int kL2tWFrags = 3 * 16 * 2;                 // 1 KiB fragments of one direction's weights

#ifndef NRV_L2T_NR
#define NRV_L2T_NR 4
#define NRV_L2T_LEAD 3
#endif

// Staging of one (direction, model)'s weights and bias image into LDS, by all `nthreads` threads of the workgroup
// (the caller puts a barrier behind it).
__device__ __forceinline__ void lstm2_t_stage(const Lstm2TModelParams& P, const int dir, float* wl, float* bl,
                                              const int tid, const int nthreads) {
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc((const char*)P.wfrag + (size_t)dir * kL2tWFrags * 1024, kL2tWFrags * 1024);
  for (int base = 0; base < kL2tWFrags * 64; base += 8 * nthreads) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = buf_load16(wrs, (unsigned)(base + j * nthreads + tid) * 16, 0);   // past the end: 0
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + j * nthreads + tid < kL2tWFrags * 64) ((f32x4*)wl)[base + j * nthreads + tid] = v[j];
  }
  const f32x4* bsrc = (const f32x4*)(P.bias + (size_t)dir * 16 * 256);
  for (int i = tid; i < 16 * 64; i += nthreads) ((f32x4*)bl)[i] = bsrc[i];
}

// One wave's unit: the 16 rows of block rb, one direction, all T steps; wl / bl: the staged weights and bias image.
template <int ACT, bool GLC>
__device__ __forceinline__ void lstm2_t_unit(const Lstm2TModelParams& P, const int T, const int dir, const int rb,
                                             const int lane, const float* wl, const float* bl) {
  constexpr int NR = NRV_L2T_NR, LEAD = NRV_L2T_LEAD;  // weight-pair ring: slots / pairs of lead
  static_assert(32 % NR == 0 && 16 % NR == 0 && LEAD < NR, "ring");
  const int n = lane & 15, q = lane >> 4;
  const int tile = rb >> 1, trow = 16 * (rb & 1) + n;  // 32-row tile and this lane's row inside it
  const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
  // input fragment of timestep t: features 8 q .. + 7 of row trow (chunk 4 (q >> 1) + 2 term + (q & 1))
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(P.in + (size_t)tile * T * 8 * 128, (unsigned)T * 8 * 512);
  const unsigned xv = (unsigned)((4 * (q >> 1) + (q & 1)) * 512 + trow * 16);
  auto t_of = [&](int s) __attribute__((always_inline)) { const int sc = s < T ? s : T - 1; return dir ? (T - 1 - sc) : sc; };
  struct XFrag { f32x4 hi, lo; };
  auto load_x = [&](int s) __attribute__((always_inline)) {
    XFrag x;
    const unsigned so = (unsigned)t_of(s) * 8 * 512;
    // GLC: when this wave produced X1 itself a moment ago (the fused launch), its loads must come from L2, not from a line
    // the vector cache may hold
    x.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xv + 1024, so, GLC ? 1 : 0));
    x.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xv, so, GLC ? 1 : 0));
    return x;
  };
  // output: unit tile ut -> feature block dir * 4 + ut, this lane's units 4 q .. + 3 = 8 bytes at element 4 (q & 1) of
  // half q >> 1
  float* const obase = P.out + ((size_t)tile * T * 32 + (dir * 4) * 4 + (q >> 1)) * 128 + trow * 4 + (q & 1) * 2;

  // ---- weight pairs.  Canonical order of a step: 32 recurrent pairs (ut, g, kb = 1, 2), then 16 input pairs
  // (ut, g, kb = 0); ring slot = canonical index % NR (both phase lengths are multiples of NR).
  struct WPair { f32x4 hi, lo; };
  WPair wr[NR];
  auto frag_of = [&](int ci) __attribute__((always_inline)) {              // canonical index -> first fragment (hi) of the pair
    if (ci < 32) { const int ut = ci >> 3, g = (ci >> 1) & 3, kb = 1 + (ci & 1); return ((kb * 16) + g * 4 + ut) * 2; }
    const int ut = (ci - 32) >> 2, g = (ci - 32) & 3;
    return (g * 4 + ut) * 2;
  };
  auto load_w = [&](int ci) __attribute__((always_inline)) {
    const int f = frag_of(ci);
    wr[ci % NR].lo = *(const f32x4*)(wl + (f + 1) * 256 + lane * 4);
    wr[ci % NR].hi = *(const f32x4*)(wl + f * 256 + lane * 4);
  };

  f32x4 Z[16];                                         // tile g * 4 + ut
  float c[16];                                         // cell state of unit 16 ut + 4 q + r at index 4 ut + r
  f16x8 hB[2][2];                                      // h_{s-1} x 2^13 as B fragments [kb][term]: what rec(s) multiplies
  f16x8 hN[2][2];                                      // h_s, filled by the gates of step s while rec(s) still reads hB
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0.f;

  // ---- gate math of one element (ut, r), cut into stages of <= 2 VALU instructions (lstm_h2s_kernel's cut)
  constexpr int GST = 13;
  struct GateSt { float zi, zf, zg, zo, cp, p, t, hv[4]; };
  GateSt gs;
  auto gate_stage = [&](int ut, int r, int st, int t_out) __attribute__((always_inline)) {
    const int e = 4 * ut + r;
    if (st == 0) { gs.zi = Z[0 + ut][r]; gs.zf = Z[4 + ut][r]; gs.cp = c[e]; }
    else if (st == 1) { gs.zg = Z[8 + ut][r]; gs.zo = Z[12 + ut][r]; }
    else if (st == 2) {
      if constexpr (ACT == 0) {
        gs.zi = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(gs.zi, dsc02, 0.5f), 0.0f), 1.0f);
        gs.zf = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(gs.zf, dsc02, 0.5f), 0.0f), 1.0f);
      } else {
        gs.zi = sigmoid_exact(gs.zi * dsc);
        gs.zf = sigmoid_exact(gs.zf * dsc);
      }
    } else if (st == 3) {
      if constexpr (ACT == 0) gs.zo = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(gs.zo, dsc02, 0.5f), 0.0f), 1.0f);
      else gs.zo = sigmoid_exact(gs.zo * dsc);
      gs.zg = gs.zg * dsc2;
    } else if (st == 4) gs.zg = __builtin_amdgcn_exp2f(gs.zg);
    else if (st == 5) gs.t = __builtin_amdgcn_rcpf(gs.zg + 1.0f);
    else if (st == 6) gs.p = gs.zi * __builtin_fmaf(gs.t, -2.0f, 1.0f);
    else if (st == 7) {
      const float cn = __builtin_fmaf(gs.zf, gs.cp, gs.p);
      c[e] = cn;
      gs.zg = cn * 2.885390081777927f;
    } else if (st == 8) gs.zg = __builtin_amdgcn_exp2f(gs.zg);
    else if (st == 9) gs.t = __builtin_amdgcn_rcpf(gs.zg + 1.0f);
    else if (st == 10) gs.hv[r] = gs.zo * __builtin_fmaf(gs.t, -2.0f * kHScale, kHScale);   // o tanh(c) 2^13
    else if (st == 11) {
      if (r & 1) {                                     // pairs: two values per conversion instruction
        const f16x2 hp = __builtin_convertvector(f32x2{gs.hv[r - 1], gs.hv[r]}, f16x2);
        hN[ut >> 1][0][4 * (ut & 1) + r - 1] = hp[0];
        hN[ut >> 1][0][4 * (ut & 1) + r] = hp[1];
      }
    } else {
      if (r & 1) {
        const int j = 4 * (ut & 1) + r;
        const f16x2 lp = __builtin_convertvector(
            f32x2{gs.hv[r - 1] - (float)hN[ut >> 1][0][j - 1], gs.hv[r] - (float)hN[ut >> 1][0][j]}, f16x2);
        hN[ut >> 1][1][j - 1] = lp[0];
        hN[ut >> 1][1][j] = lp[1];
      }
      if (r == 3) {                                    // the unit tile is complete: its 4 units are 8 contiguous bytes per term
        typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
        float* d = obase + ((size_t)t_out * 32 + ut * 4) * 128;
        const f16x8 a = hN[ut >> 1][0], b = hN[ut >> 1][1];
        const int o = 4 * (ut & 1);
        *(f16x4v*)d = f16x4v{a[o], a[o + 1], a[o + 2], a[o + 3]};
        *(f16x4v*)(d + 2 * 128) = f16x4v{b[o], b[o + 1], b[o + 2], b[o + 3]};
      }
    }
  };
  auto gates_plain = [&](int ut, int t_out) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int st = 0; st < GST; ++st) gate_stage(ut, r, st, t_out);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int NGP = 4 * GST;                         // stage pieces of one unit tile
  constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};  // (weight term, activation term): hi*hi, hi*lo, lo*hi
  // the first product of a pair takes the LAST-requested fragment of the weights (hi), so one counted wait covers it

  // ---- rec(s): Z += U^T h_{s-1}, unit tile by unit tile; the gates of unit tile ut - 1 between the MFMAs of ut
  auto rec_phase = [&](int t_out) __attribute__((always_inline)) {
#pragma unroll
    for (int ci = 0; ci < 32; ++ci) {
      const int ut = ci >> 3, g = (ci >> 1) & 3, kb = ci & 1;
      load_w(ci + LEAD < 32 ? ci + LEAD : 32 + (ci + LEAD - 32));          // the tail requests the input pairs that follow
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const f16x8 a = __builtin_bit_cast(f16x8, PA[pr] ? wr[ci % NR].lo : wr[ci % NR].hi);
        Z[g * 4 + ut] = mfma16_f16(a, hB[kb][PB[pr]], Z[g * 4 + ut]);
        if (ut > 0) {
          const int tk = (ci & 7) * 3 + pr;                                  // tick inside this unit tile: 0..23
#pragma unroll
          for (int pc = (tk * NGP) / 24; pc < ((tk + 1) * NGP) / 24; ++pc) gate_stage(ut - 1, pc / GST, pc % GST, t_out);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- in(s+1): Z = b + W^T x_{s+1} into the tiles whose gates are done: unit tiles 0..2 first, with the gates of unit
  // tile 3 between their MFMAs (GATES3: those are the gates of step s; false in the prologue), then unit tile 3.
  // NEXT_REC: the tail requests the first pairs of the recurrent phase that follows.
  auto in_phase = [&](auto gates3_tag, const XFrag& x, int t_out) __attribute__((always_inline)) {
    constexpr bool GATES3 = decltype(gates3_tag)::value;
    const f16x8 xb[2] = {__builtin_bit_cast(f16x8, x.hi), __builtin_bit_cast(f16x8, x.lo)};
#pragma unroll
    for (int i = 0; i < 12; ++i) Z[(i & 3) * 4 + (i >> 2)] = *(const f32x4*)(bl + ((i & 3) * 4 + (i >> 2)) * 256 + lane * 4);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ci = 32 + k, ut = k >> 2, g = k & 3;
      load_w(ci + LEAD < 48 ? ci + LEAD : ci + LEAD - 48);                 // the tail requests the recurrent pairs of the next step
      if (k == 9) {                                    // unit tile 3's accumulators: its gates have read them by now
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) Z[gg * 4 + 3] = *(const f32x4*)(bl + (gg * 4 + 3) * 256 + lane * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const f16x8 a = __builtin_bit_cast(f16x8, PA[pr] ? wr[ci % NR].lo : wr[ci % NR].hi);
        Z[g * 4 + ut] = mfma16_f16(a, xb[PB[pr]], Z[g * 4 + ut]);
        if constexpr (GATES3) {
          const int tk = k * 3 + pr;                                         // 0..47; the gates of unit tile 3 take ticks 0..26
          if (tk < 27) {
#pragma unroll
            for (int pc = (tk * NGP) / 27; pc < ((tk + 1) * NGP) / 27; ++pc) gate_stage(3, pc / GST, pc % GST, t_out);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // prologue: in(0) straight into Z (no gates yet), the ring primed with its first pairs
  XFrag xn = load_x(1);
  {
    const XFrag x0 = load_x(0);
#pragma unroll
    for (int i = 0; i < LEAD; ++i) load_w(32 + i);
    in_phase(std::false_type{}, x0, 0);
    // in_phase's tail requested recurrent pairs; step 0 has no recurrent phase and starts with the input pairs again
#pragma unroll
    for (int i = 0; i < LEAD; ++i) load_w(32 + i);
  }
  // step 0: h_{-1} = 0, no recurrent product: the gates of unit tiles 0..2 run plainly, unit tile 3 under in(1)
  {
    const int t0 = t_of(0);
#pragma unroll
    for (int ut = 0; ut < 3; ++ut) gates_plain(ut, t0);
    if (T > 1) {
      const XFrag x = xn;
      xn = load_x(2);
      in_phase(std::true_type{}, x, t0);
    } else {
      gates_plain(3, t0);
    }
  }
#pragma unroll 1
  for (int s = 1; s < T; ++s) {
    const int t = t_of(s);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) hB[kb][tm] = hN[kb][tm];             // h_{s-1} is complete
    rec_phase(t);                                      // ... the gates of unit tiles 0..2 of step s inside
    if (s + 1 < T) {
      const XFrag x = xn;
      xn = load_x(s + 2);
      in_phase(std::true_type{}, x, t);                // ... and those of unit tile 3
    } else {
      gates_plain(3, t);                               // the last step has no input projection to hide behind
    }
  }
}

template <int ACT>
__global__ void __launch_bounds__(kL2tThreads) lstm2_t_kernel(const Lstm2TArgs args) {
  __shared__ __attribute__((aligned(16))) float wl[kL2tWFrags * 256];       // 96 KiB
  __shared__ __attribute__((aligned(16))) float bl[16 * 256];               // 16 KiB
  const Lstm2TModelParams& P = args.m[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lstm2_t_stage(P, blockIdx.y, wl, bl, threadIdx.x, kL2tThreads);
  __syncthreads();                                     // the only barrier: from here on the waves are independent
  lstm2_t_unit<ACT, false>(P, args.T, blockIdx.y, blockIdx.x * 4 + wave, lane, wl, bl);
}

}  // namespace nrv
