// Second read-branch layer, Bi-LSTM(32 -> 64), f16x2 mode: transposed products, TWO waves per chain (r05).
#pragma once
#include "nrv_lstm_f16x2s.h"   // mfma16_f16

namespace nrv {

// ---------------------------------------------------------------------------------------
// lstm2_u_kernel.  output_handeler.py:220 (Bidirectional(LSTM(64)) on the 32 features of read_rnn1).
//
// The layer is small (10.5 GFLOP per 4096-window group, 98 KB of f16x2 weights per direction) and was LATENCY-bound with
// hidden units on the lanes (round 2: 44-48 us, matrix pipe 27 % busy).  Here the products are TRANSPOSED,
// z^T = [W | U]^T [x ; h]: the weights are the A operand (M = the 256 gate-units), the activations the B operand (N = 16 data
// rows), so a result tile has the data row on the lane and the gate-units in the registers: tile (gate g, unit tile ut) of
// v_mfma_f32_16x16x32_f16 gives lane l = (row n = l & 15, q = l >> 4) the units 16 ut + 4 q + r, r = 0..3; i, f, g, o of a
// (row, unit) meet in one lane, c never leaves the lane, and the values of h_t a lane computes are - as two f16x8 terms,
// elements 8 kb + j = unit 16 (2 kb + (j >> 2)) + 4 q + (j & 3) - B fragments of the next step's recurrent k-blocks: the host
// packs the rows of U in that k order (pack_lstm2_t).
//   * weights: the whole f16x2 set of one (direction, model), [kb 3][tile 16][term 2] fragments of 1 KiB (96 KiB), is copied
//     to LDS once per workgroup and every wave streams its half from there each step; the bias enters as the initial value of
//     the accumulators, read from a [tile][lane] image (16 KiB);
//   * output: h x 2^13 as f16 split planes, RAW - the BatchNorm(128) behind this layer lives in the weights of the 192->128
//     layer's first 128 input rows (nrv_api.hip upload_model) - each lane's 4 units of a unit tile are 8 contiguous bytes.
//
// Round 4's form (lstm2_t_kernel, removed in r06; HISTORY.md) ran one wave per 16-row chain: 4096 windows x 2 directions x 2
// models are exactly 1024 chains, ONE wave per SIMD, 533 vector + 144 matrix instructions + 96 LDS reads per step in one
// instruction stream: 5.5 k cycles for 2.4 k of matrix pipe.
//
// Here a chain is TWO waves on one SIMD (wave w and w + 4 of an 8-wave workgroup): wave `hf` owns unit tiles 2 hf and
// 2 hf + 1 - 32 hidden units x 4 gates = 8 accumulator tiles, 8 cell states per lane - and therefore k-block hf of h_t (the
// host packs U's rows so that k-block kb holds unit tiles 2 kb, 2 kb + 1).  The two
// halves of h_t meet in LDS: each wave stores its two f16x8 terms (32 B per lane), ONE workgroup barrier, each reads both
// halves back as the B fragments of the next step's recurrent product - 2 KiB per chain and step through a double-buffered
// 16 KiB image, on top of the 96 + 16 KiB of weights and bias.  As in the one-wave form: i, f, g, o of a
// (row, unit) in one lane, c in registers, the gates of a unit tile between the products of the next one, the input
// projection of step s+1 behind the recurrent product of step s, products in the same order - results bit-identical to it.
// While one wave of the SIMD does gate arithmetic the other feeds the matrix pipe.
// grid = (ceil(rows / 64), 2 directions, 2 models), block = 512: four chains of two waves.
// ---------------------------------------------------------------------------------------
constexpr int kL2uThreads = 512;

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

constexpr int kL2tWFrags = 3 * 16 * 2;                 // 1 KiB fragments of one direction's weights

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


template <int ACT>
__device__ __forceinline__ void lstm2_u_unit(const Lstm2TModelParams& P, const int T, const int dir, const int rb,
                                             const int lane, const int hf, const float* wl, const float* bl, f32x4* hx) {
  constexpr int NR = 4, LEAD = 3;                      // weight-pair ring: slots / pairs of lead
  const int n = lane & 15, q = lane >> 4;
  const int tile = rb >> 1, trow = 16 * (rb & 1) + n;  // 32-row tile and this lane's row inside it
  const float dsc = P.descale, dsc02 = 0.2f * dsc, dsc2 = 2.885390081777927f * dsc;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(P.in + (size_t)tile * T * 8 * 128, (unsigned)T * 8 * 512);
  const unsigned xv = (unsigned)((4 * (q >> 1) + (q & 1)) * 512 + trow * 16);
  auto t_of = [&](int s) __attribute__((always_inline)) { const int sc = s < T ? s : T - 1; return dir ? (T - 1 - sc) : sc; };
  struct XFrag { f32x4 hi, lo; };
  auto load_x = [&](int s) __attribute__((always_inline)) {
    XFrag x;
    const unsigned so = (unsigned)t_of(s) * 8 * 512;
    x.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xv + 1024, so, 0));
    x.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xv, so, 0));
    return x;
  };
  // this wave's unit tiles are ut = 2 hf + u, u = 0, 1: output block, weight fragments and bias image shifted by 2 hf tiles
  float* const obase = P.out + ((size_t)tile * T * 32 + (dir * 4) * 4 + (q >> 1)) * 128 + trow * 4 + (q & 1) * 2 + (size_t)hf * 2 * 4 * 128;
  const float* const wlh = wl + hf * 2 * 2 * 256 + lane * 4;      // fragment (kb * 16 + g * 4 + ut) * 2 + term, 256 floats each
  const float* const blh = bl + hf * 2 * 256 + lane * 4;          // tile g * 4 + ut

  // ---- weight pairs.  Canonical order of a step: 16 recurrent pairs (u, g, kb = 1, 2), then 8 input pairs (u, g, kb = 0)
  struct WPair { f32x4 hi, lo; };
  WPair wr[NR];
  auto frag_of = [&](int ci) __attribute__((always_inline)) {
    if (ci < 16) { const int u = ci >> 3, g = (ci >> 1) & 3, kb = 1 + (ci & 1); return ((kb * 16) + g * 4 + u) * 2; }
    const int u = (ci - 16) >> 2, g = (ci - 16) & 3;
    return (g * 4 + u) * 2;
  };
  auto load_w = [&](int ci) __attribute__((always_inline)) {
    const int f = frag_of(ci);
    wr[ci % NR].lo = *(const f32x4*)(wlh + (f + 1) * 256);
    wr[ci % NR].hi = *(const f32x4*)(wlh + f * 256);
  };

  f32x4 Z[8];                                          // tile g * 2 + u
  float c[8];                                          // cell state of unit 16 (2 hf + u) + 4 q + r at index 4 u + r
  f16x8 hB[2][2];                                      // h_{s-1} x 2^13 as B fragments [kb][term], both halves, from LDS
  f16x8 hN[2];                                         // this wave's half of h_s [term]
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = 0.f;

  constexpr int GST = 13;
  struct GateSt { float zi, zf, zg, zo, cp, p, t; };
  GateSt gsr;                                          // the element of the unit tile in flight
  float hv[4];
  auto gate_stage = [&](int u, int r, int st, int t_out) __attribute__((always_inline)) {
    const int e = 4 * u + r;
    GateSt& gs = gsr;
    if (st == 0) { gs.zi = Z[0 + u][r]; gs.zf = Z[2 + u][r]; gs.cp = c[e]; }
    else if (st == 1) { gs.zg = Z[4 + u][r]; gs.zo = Z[6 + u][r]; }
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
    else if (st == 10) hv[r] = gs.zo * __builtin_fmaf(gs.t, -2.0f * kHScale, kHScale);   // o tanh(c) 2^13
    else if (st == 11) {
      if (r & 1) {
        const f16x2 hp = __builtin_convertvector(f32x2{hv[r - 1], hv[r]}, f16x2);
        hN[0][4 * u + r - 1] = hp[0];
        hN[0][4 * u + r] = hp[1];
      }
    } else {
      if (r & 1) {
        const int j = 4 * u + r;
        const f16x2 lp = __builtin_convertvector(
            f32x2{hv[r - 1] - (float)hN[0][j - 1], hv[r] - (float)hN[0][j]}, f16x2);
        hN[1][j - 1] = lp[0];
        hN[1][j] = lp[1];
      }
      if (r == 3) {                                    // the unit tile is complete: its 4 units are 8 contiguous bytes per term
        typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
        float* d = obase + ((size_t)t_out * 32 + u * 4) * 128;
        const f16x8 a = hN[0], b = hN[1];
        const int o = 4 * u;
        *(f16x4v*)d = f16x4v{a[o], a[o + 1], a[o + 2], a[o + 3]};
        *(f16x4v*)(d + 2 * 128) = f16x4v{b[o], b[o + 1], b[o + 2], b[o + 3]};
      }
    }
  };
  // piece pc (0 .. 4 GST - 1) of a unit tile, element-major: pc = GST r + st.  (Stage-major - the four elements' stages as
  // independent neighbours - was measured in r05: 35.3-36.0 vs 34.9-35.7 us, nothing: the launch is issue-bound.)
  auto gate_piece = [&](int u, int pc, int t_out) __attribute__((always_inline)) { gate_stage(u, pc / GST, pc % GST, t_out); };
  auto gates_plain = [&](int u, int t_out) __attribute__((always_inline)) {
#pragma unroll
    for (int pc = 0; pc < 4 * GST; ++pc) {
      gate_piece(u, pc, t_out);
      if ((pc & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int NGP = 4 * GST;                         // stage pieces of one unit tile
  constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};  // (weight term, activation term): hi*hi, hi*lo, lo*hi

  // ---- rec(s): Z += U^T h_{s-1}: unit tile 0, then unit tile 1 with the gates of unit tile 0 between its MFMAs
  auto rec_phase = [&](int t_out) __attribute__((always_inline)) {
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
      const int u = ci >> 3, g = (ci >> 1) & 3, kb = ci & 1;
      load_w(ci + LEAD);                                // 16 .. 18: the input pairs that follow
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const f16x8 a = __builtin_bit_cast(f16x8, PA[pr] ? wr[ci % NR].lo : wr[ci % NR].hi);
        Z[g * 2 + u] = mfma16_f16(a, hB[kb][PB[pr]], Z[g * 2 + u]);
        if (u > 0) {
          const int tk = (ci & 7) * 3 + pr;             // tick inside this unit tile: 0..23
#pragma unroll
          for (int pc = (tk * NGP) / 24; pc < ((tk + 1) * NGP) / 24; ++pc) gate_piece(0, pc, t_out);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- in(s+1): Z = b + W^T x_{s+1}: unit tile 0 (its gates are done) with the gates of unit tile 1 between its MFMAs
  // (GATES1: false in the prologue), then unit tile 1.
  auto in_phase = [&](auto gates1_tag, const XFrag& x, int t_out) __attribute__((always_inline)) {
    constexpr bool GATES1 = decltype(gates1_tag)::value;
    const f16x8 xb[2] = {__builtin_bit_cast(f16x8, x.hi), __builtin_bit_cast(f16x8, x.lo)};
#pragma unroll
    for (int g = 0; g < 4; ++g) Z[g * 2] = *(const f32x4*)(blh + (g * 4) * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = 16 + k, u = k >> 2, g = k & 3;
      load_w(ci + LEAD < 24 ? ci + LEAD : ci + LEAD - 24);                 // the tail requests the recurrent pairs of the next step
      if (k == 4) {                                    // unit tile 1's accumulators: its gates have read them by now
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) Z[gg * 2 + 1] = *(const f32x4*)(blh + (gg * 4 + 1) * 256);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const f16x8 a = __builtin_bit_cast(f16x8, PA[pr] ? wr[ci % NR].lo : wr[ci % NR].hi);
        Z[g * 2 + u] = mfma16_f16(a, xb[PB[pr]], Z[g * 2 + u]);
        if constexpr (GATES1) {
          const int tk = k * 3 + pr;                    // 0..23; the gates of unit tile 1 take ticks 0..11
          if (tk < 12) {
#pragma unroll
            for (int pc = (tk * NGP) / 12; pc < ((tk + 1) * NGP) / 12; ++pc) gate_piece(1, pc, t_out);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // h_s: this wave's half to the image of parity p, one barrier, both halves back
  auto exchange = [&](int p) __attribute__((always_inline)) {
    f32x4* const img = hx + p * 4 * 64 + lane;          // [kb 2][term 2][64 lanes]
    img[(hf * 2 + 0) * 64] = __builtin_bit_cast(f32x4, hN[0]);
    img[(hf * 2 + 1) * 64] = __builtin_bit_cast(f32x4, hN[1]);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) hB[kb][tm] = __builtin_bit_cast(f16x8, img[(kb * 2 + tm) * 64]);
  };

  // prologue: in(0) straight into Z (no gates yet), the ring primed with its first pairs
  XFrag xn = load_x(1);
  {
    const XFrag x0 = load_x(0);
#pragma unroll
    for (int i = 0; i < LEAD; ++i) load_w(16 + i);
    in_phase(std::false_type{}, x0, 0);
    // in_phase's tail requested recurrent pairs; step 0 has no recurrent phase and starts with the input pairs again
#pragma unroll
    for (int i = 0; i < LEAD; ++i) load_w(16 + i);
  }
  // step 0: h_{-1} = 0, no recurrent product: the gates of unit tile 0 run plainly, unit tile 1 under in(1)
  {
    const int t0 = t_of(0);
    gates_plain(0, t0);
    if (T > 1) {
      const XFrag x = xn;
      xn = load_x(2);
      in_phase(std::true_type{}, x, t0);
      exchange(0);
    } else {
      gates_plain(1, t0);
    }
  }
#pragma unroll 1
  for (int s = 1; s < T; ++s) {
    const int t = t_of(s);
    rec_phase(t);                                      // ... the gates of unit tile 0 of step s inside
    if (s + 1 < T) {
      const XFrag x = xn;
      xn = load_x(s + 2);
      in_phase(std::true_type{}, x, t);                // ... and those of unit tile 1
      exchange(s & 1);
    } else {
      gates_plain(1, t);                               // the last step has no input projection to hide behind
    }
  }
}

template <int ACT>
__global__ void __launch_bounds__(kL2uThreads) lstm2_u_kernel(const Lstm2TArgs args) {
  __shared__ __attribute__((aligned(16))) float wl[kL2tWFrags * 256];       // 96 KiB
  __shared__ __attribute__((aligned(16))) float bl[16 * 256];               // 16 KiB
  __shared__ __attribute__((aligned(16))) f32x4 hx[4 * 2 * 4 * 64];         // 32 KiB: [chain][parity][kb][term][lane]
  const Lstm2TModelParams& P = args.m[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chain = wave & 3, hf = wave >> 2;          // waves w and w + 4 share a SIMD and a chain
  lstm2_t_stage(P, blockIdx.y, wl, bl, threadIdx.x, kL2uThreads);
  __syncthreads();
  lstm2_u_unit<ACT>(P, args.T, blockIdx.y, blockIdx.x * 4 + chain, lane, hf, wl, bl, hx + chain * 2 * 4 * 64);
}

}  // namespace nrv
