// Signal branch in f16x2 mode, register-resident: conv1 (VALU) -> conv2 (MFMA) -> dense 400->64 (MFMA) per wave,
// no feature image, no roles, no barrier per tile (cnn_r_kernel).
#pragma once
#include "nrv_common.h"
#include "nrv_lstm_f16x2s.h"   // f16x8, mfma16_f16
#include "nrv_lstm1.h"         // lstm1_unit: the 6 -> 16 Bi-LSTM rides along as four more waves

namespace nrv {

// ---------------------------------------------------------------------------------------
// cnn_r_kernel.  nanorevcnn.py:17-38 + output_handeler.py:209-215.
// (cnn_h2_kernel / cnn_m_kernel below are round 3's forms of it, removed since.)
//
// What was measured on those (scripts/gpu_variants.py, parts compiled out, r03): with the 8 -> 8 convolution on the
// matrix pipe the conv role needs 7 us per launch, the 400 -> 64 layer's MFMAs 18 us - and the launch takes 75 us,
// because the 400-feature image that carries one role's output to the other through LDS costs 2.7 M LDS instructions
// per launch (8-byte stores in accumulator layout on one side, four waves re-reading the whole image on the other):
// the LDS is busy half the launch and every read queues behind it.  So the image is gone.  All three layers are
// TRANSPOSED products with the 16 events of a unit on N (the lane), and each layer's result tile is, in registers,
// the next layer's B fragment:
//   conv1   lane (event n, q = l >> 4) computes the 8 channels of position p + q - 1 on the VALU (48 constants in
//           SGPRs) and splits them: that IS the B fragment (k-group q = tap) of
//   conv2   A = [16][32] weights: rows 0-7 = W2[tap = kg][ci][co] (kg < 3), rows 8-15 = W2[tap = kg - 1][ci][co]
//           (kg > 0): one accumulating triple (hi*hi, hi*lo, lo*hi) gives positions p AND p + 1, no padding row:
//           lane (n, q) holds position p + (q >> 1), channels 4 (q & 1) .. + 3.  Bias, ReLU, BatchNorm, the residual
//           sample, x 2^6 and the f16 split happen in the lane; two such triples (p = 4 ks, 4 ks + 2) fill the 8
//           elements of the B fragment of k-step ks of
//   dense   S^T[64][16 events] = Wd^T feat^T, k order (kg, j) = position 4 ks + 2 (j >> 2) + (kg >> 1), channel
//           4 (kg & 1) + (j & 3) - the host packs Wd's rows that way.  Its A fragments (104 KB for one model) are
//           the only big thing in LDS, read once per unit by every wave (8 x 16-byte reads per k-step of 18 MFMAs).
//   out     lane (n, q) holds output features 16 mt + 4 q .. + 3 of event n: 8 contiguous bytes per term of the
//           split-plane row (instead of 16 two-byte stores), behind the range guard (nrv_cnn_f16x2.h).
// conv1 runs once per position: a wave-private LDS ring carries its result from the quarter that computed it to the
// three quarters that need it as their tap (2 stores + 4 reads per k-step).  The 13 k-steps are unrolled, so ring
// slots and sample addresses are immediate offsets; per 16-event unit a wave issues ~1360 vector instructions,
// 246 MFMAs, 104 + 52 LDS reads, 26 LDS writes, ~40 loads.
// A wave owns whole units (dealt workgroup-first: the bench step's 3328 units per model are 26 per workgroup), eight
// such waves per workgroup (two per SIMD), persistent: one workgroup per CU and model.
//
// FOUR MORE WAVES per workgroup run the first read-branch layer, Bi-LSTM(6 -> 16) (lstm1_unit, nrv_lstm1.h: one wave
// = 16 rows of one direction, f32 16x16x4 tiles, weights in registers, wave-private).  That layer is independent of
// the signal branch until the 192 -> 128 layer, is a 13 us launch of 1024 latency-bound waves on its own - one per
// SIMD - and needs 96 registers and 1 KB of LDS per wave: beside two conv waves of <= 168 registers it fits the SIMD,
// its dependent chain runs in their shadow, and the launch is gone.  (As a second STREAM the same overlap cost ~20 us
// of event fork / join in round 2; as a role of this launch it costs nothing.)
// ---------------------------------------------------------------------------------------
struct CnnRConsts {        // per model, BY VALUE in the kernel arguments: scalar loads
  float w1[24];            // first convolution [tap][co]
  float b1[8];
  float s1[8], h1[8];      // BatchNorm 1 scale / shift x 2^6 (c1 enters conv2 x 2^6)
};
struct CnnRModelParams {
  const void* w2frag;      // conv2 A operand x 2^u: [term 2][64 lanes][8 f16] (two-position form, see above)
  const float* ep;         // [3][8]: accumulator init b2 x 2^(6+u), k1 = s2 x 2^-u, k2 = h2 x 2^6 per output channel;
                           // [24]: the sample magnitude below which no conv1 output can leave the f16 range (range guard)
  const void* dfrag;       // dense A operand x 2^10: [ks 13][mt 4][term 2][64 lanes][8 f16]
  const float* dbias;      // [64] x 2^16
  float* out;              // S x 2^6 as f16 split planes: window-major [wtile][t][16 chunks][32][8 f16] or event-major
};
struct CnnRArgs {
  CnnRModelParams m[2];
  CnnRConsts k[2];
  const float* signal;     // [n][T][50] (window mode) or [N][50] (event mode)
  int T, n_rows, n_tiles;  // as CnnH2Args: 32-event tiles
  unsigned* sat;           // range guard counter
  Lstm1ModelParams l1[2];  // the 6 -> 16 Bi-LSTM of the same launch group (rows = windows)
  int l1_T, l1_rows;
};

constexpr int kCnnRWaves = 8;                           // conv / dense waves
constexpr int kCnnRL1Waves = 4;                         // 6 -> 16 Bi-LSTM waves
constexpr int kCnnRThreads = 64 * (kCnnRWaves + kCnnRL1Waves);
constexpr float kCnnRImgScale = 64.0f;                  // 2^6
constexpr float kCnnRDenseDescale = 1.0f / 1024.0f;     // 2^-10

typedef float f32x2r __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32r;

// Wave-private ring of conv1 results: 12 position slots (9 are live at any time; 12 so that a quarter's four
// positions wrap in two of the six slot bases only), each [term 2][event 16][8 f16] = 512 B; behind the rings one
// all-zero slot (the second convolution's zero padding) and one write-only slot for positions past the window (so that
// the conv1 code has no branch and stays in one basic block with the MFMAs it is interleaved with), shared by all waves.
constexpr int kCnnRRing = 12, kCnnRSlot = 2 * 16 * 8, kCnnRC1Wave = kCnnRRing * kCnnRSlot;   // f16

template <int ACT>
__global__ void __launch_bounds__(kCnnRThreads) cnn_r_kernel(const CnnRArgs args) {
  constexpr int NKS = 13;
  __shared__ __attribute__((aligned(16))) float wd_s[NKS * 4 * 2 * 256];          // dense fragments (104 KiB)
  __shared__ __attribute__((aligned(16))) _Float16 c1_s[kCnnRWaves * kCnnRC1Wave + 2 * kCnnRSlot];  // 6 KiB per wave + 1
  __shared__ __attribute__((aligned(16))) float l1h_s[kCnnRL1Waves][16 * 16 + 16];  // lstm1_unit's wave-private images
  const CnnRModelParams& P = args.m[blockIdx.y];
  const CnnRConsts& K = args.k[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(P.dfrag, NKS * 8 * 1024);
  for (int base = 0; base < NKS * 8 * 64; base += 8 * kCnnRThreads) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = buf_load16(wrs, (unsigned)(base + j * kCnnRThreads + threadIdx.x) * 16, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + j * kCnnRThreads + threadIdx.x < NKS * 8 * 64) ((f32x4*)wd_s)[base + j * kCnnRThreads + threadIdx.x] = v[j];
  }
  static_assert((kCnnRWaves * kCnnRC1Wave + 2 * kCnnRSlot) % 8 == 0, "ring image in 16-byte pieces");
  for (int i = threadIdx.x; i < (kCnnRWaves * kCnnRC1Wave + 2 * kCnnRSlot) / 8; i += kCnnRThreads) ((f32x4*)c1_s)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // incl. the zero slot
  __syncthreads();                                   // the only barrier: from here on the waves are independent

  if (wave >= kCnnRWaves) {
    // ================================ 6 -> 16 Bi-LSTM role ====================================
    const int nu = 2 * ((args.l1_rows + 15) / 16);   // units: (16-row block, direction)
    for (int idx = blockIdx.x * kCnnRL1Waves + (wave - kCnnRWaves); idx < nu; idx += gridDim.x * kCnnRL1Waves)
      lstm1_unit<ACT, true>(args.l1[blockIdx.y], args.l1_T, args.l1_rows, idx & 1, idx >> 1, lane, l1h_s[wave - kCnnRWaves]);
    return;
  }

  typedef __attribute__((address_space(3))) _Float16 lds_h;
  typedef __attribute__((address_space(3))) f16x8 lds_h8;
  lds_h* const c1 = (lds_h*)c1_s + wave * kCnnRC1Wave;
  const f16x8 a2_hi = *(const f16x8*)((const char*)P.w2frag + lane * 16);
  const f16x8 a2_lo = *(const f16x8*)((const char*)P.w2frag + 1024 + lane * 16);
  const int c0 = 4 * (q & 1);                        // this lane's conv2 output channels c0 .. c0 + 3
  const f32x4 binit = *(const f32x4*)(P.ep + c0);
  const f32x4 k1 = *(const f32x4*)(P.ep + 8 + c0);
  const f32x4 k2 = *(const f32x4*)(P.ep + 16 + c0);
  f32x4 dbias[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) dbias[mt] = *(const f32x4*)(P.dbias + 16 * mt + 4 * q);
  const float* wd = wd_s + lane * 4;
  bool bad = false;
  // The first convolution's 48 constants are kernel arguments (scalar loads).  They do not stay in SGPRs through the
  // k-step loop (hipcc re-loads them with three s_load_dwordx16 per k-step); holding them in 48 VGPRs instead was
  // built and measured: the same time, and twelve waves per workgroup leave no room for it (<= 168 registers).
#define w1v K.w1
#define b1v K.b1
#define s1v K.s1
#define h1v K.h1
  // ReLU of a matrix-instruction result as v_med3_f32(x, 0, 3e38): written as fmaxf, hipcc puts a canonicalising
  // v_max_f32 x, x in front of every v_max_f32 0, x (104 extra instructions per unit).
  auto relu_acc = [](float x) __attribute__((always_inline)) { return __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f); };
  float m1; asm("s_mov_b32 %0, 0xbf800000" : "=s"(m1));   // -1, opaque: x - (float)hi as v_fma_mix_f32 (no v_cvt_f32_f16)
#define NRV_LO(x, h) __builtin_fmaf((float)(h), m1, (x))
  // Ring slot of position pos: pos % 12.  A quarter's positions C + q never wrap for C % 12 <= 8, so the slot is an
  // immediate offset on ONE per-lane address (rq); the two bases that can wrap (C % 12 = 9, 11) have their own.
  lds_h* const rq = c1 + q * kCnnRSlot + n * 8;
  lds_h* const rq9 = c1 + ((9 + q) % kCnnRRing) * kCnnRSlot + n * 8;
  lds_h* const rq11 = c1 + ((11 + q) % kCnnRRing) * kCnnRSlot + n * 8;
  lds_h* const zslot = (lds_h*)c1_s + kCnnRWaves * kCnnRC1Wave + n * 8;     // all-zero slot, then the write-only one
  auto slot_c = [&](int C) __attribute__((always_inline)) {       // slot of position C + q (C a compile-time constant)
    const int cm = ((C % kCnnRRing) + kCnnRRing) % kCnnRRing;
    return cm == 9 ? rq9 : cm == 11 ? rq11 : rq + cm * kCnnRSlot;
  };

  const int n_units = 2 * args.n_tiles;
  // conv1 + BatchNorm of one position (this lane's event) from its three samples -> ring slot d, both terms.
  // (Written as packed f32 math, v_pk_fma_f32, it was measured SLOWER, 73 vs 67 us.)
  // Range guard of c1: a conv1 output beyond 65504 / 2^6 would become hi = +inf, lo = -inf, their products
  // NaN, and the ReLU behind conv2 (v_med3) turns a NaN into 0 - a silently wrong feature that the check on S cannot
  // see.  So the centre samples (every position of the window is one lane's centre sample exactly once) feed a running
  // maximum, one v_med3_f32 each, which is compared once per unit with the static bound of upload_model (ep[24]).
  // This guard covers FINITE overflow only: v_med3_f32 returns min3 when an operand is NaN, so a NaN sample drops out
  // of xmax (and +Inf is clamped to 3e38, still above every bound).  NaN samples are caught by the check on S, which a
  // NaN always reaches through conv2 and the dense layer (tests/test_gpu_range.py::test_nan_and_inf_samples pins both).
  float xmax = 0.f;
  auto conv1_store = [&](lds_h* d, float xm, float xc, float xp) __attribute__((always_inline)) {
    xmax = __builtin_amdgcn_fmed3f(__builtin_fabsf(xc), xmax, 3.0e38f);
    float c[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float tt = b1v[o];
      tt = __builtin_fmaf(xm, w1v[0 * 8 + o], tt);
      tt = __builtin_fmaf(xc, w1v[1 * 8 + o], tt);
      tt = __builtin_fmaf(xp, w1v[2 * 8 + o], tt);
      tt = __builtin_fmaxf(tt, 0.f);
      c[o] = __builtin_fmaf(tt, s1v[o], h1v[o]);
    }
    f16x8 hi, lo;
#pragma unroll
    for (int o = 0; o < 8; o += 2) {
      const f16x2r hp = __builtin_convertvector(f32x2r{c[o], c[o + 1]}, f16x2r);
      hi[o] = hp[0]; hi[o + 1] = hp[1];
      const f16x2r lp = __builtin_convertvector(f32x2r{NRV_LO(c[o], hp[0]), NRV_LO(c[o + 1], hp[1])}, f16x2r);
      lo[o] = lp[0]; lo[o + 1] = lp[1];
    }
    *(lds_h8*)d = hi;
    *(lds_h8*)(d + 16 * 8) = lo;
  };


  // units of 16 events: unit u = 2 * tile + sub
  // Units are dealt WORKGROUP-first (unit u -> workgroup u % G, then that workgroup's waves in turn): the bench step's
  // 3328 units per model over 128 workgroups are exactly 26 each, 6 or 7 per SIMD.  Dealt wave-first (u -> wave
  // u % 1024 of the launch, r03e-r03k) the first 32 workgroups had 8 per SIMD and the others 6, and the launch waited
  // for those 32: 53.3 -> 47.1 us on the same box.
  for (int u = blockIdx.x + gridDim.x * wave; u < n_units; u += gridDim.x * kCnnRWaves) {
    const int b = u >> 1, sub = u & 1;
    const int wt = b / T, t = b % T;
    const int row = 16 * sub + n;
    const bool rok = wt * 32 + row < args.n_rows;
    // This lane's event: samples straight from memory (200 B per event, L1-resident for the length of the unit).
    // ONE per-lane address (+ one for the residual samples); the position is the instruction's immediate offset, and
    // what lies outside the window or past the rows is addressed past the descriptor's end: the load returns 0.
    const size_t ev0 = (size_t)wt * 32 * T + t, left = ((size_t)args.n_rows * T - ev0) * (kSig * 4);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(args.signal + ev0 * kSig, left < 0x7fffffffu ? (unsigned)left : 0x7fffffffu);
    constexpr unsigned kOut = 0x80000000u;
    const unsigned xq = rok ? (unsigned)(row * T * kSig + q) * 4u : kOut;          // position q
    const unsigned xh = rok ? (unsigned)(row * T * kSig + (q >> 1)) * 4u : kOut;   // position q >> 1
    // x[C + q] / x[C + (q >> 1)], C >= 0 a compile-time constant; the compare only where some quarter is outside
    auto ldq = [&](int C) __attribute__((always_inline)) {
      if (C >= kSig) return 0.f;
      return buf_load4(rs, (C + 3 < kSig || q < kSig - C ? xq : kOut) + 4u * C, 0);
    };
    auto ldh = [&](int C) __attribute__((always_inline)) {
      if (C >= kSig) return 0.f;
      return buf_load4(rs, (C + 1 < kSig || (q >> 1) < kSig - C ? xh : kOut) + 4u * C, 0);
    };
    // prologue: conv1 of positions 0..3 (set 0), the samples of set 1 and the residual samples of k-step 0 in flight.
    // (Fetching the NEXT unit's first samples in this unit's last two k-steps, which load nothing useful, was built
    // and measured: 73 us instead of 65.)
    float xa[3], xres[2];
    {
      const float x0 = buf_load4(rs, rok && q > 0 ? xq - 4u : kOut, 0), x1 = ldq(0), x2 = ldq(1);
#pragma unroll
      for (int k = 0; k < 3; ++k) xa[k] = ldq(3 + k);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) xres[pi] = ldh(2 * pi);
      conv1_store(slot_c(0), x0, x1, x2);
    }

    f32x4 S[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) S[mt] = dbias[mt];

    f16x8 fb_hi, fb_lo;                                // B fragment of a dense k-step (carried into the next iteration)
    f32x4 wf[4][2];                                    // ... and that k-step's weights
#pragma unroll
    for (int j = 0; j < 8; ++j) { fb_hi[j] = (_Float16)0.f; fb_lo[j] = (_Float16)0.f; }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { wf[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; wf[mt][1] = wf[mt][0]; }
    auto dense = [&]() __attribute__((always_inline)) { // 4 output tiles x (hi*hi, hi*lo, lo*hi)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f16x8 w_hi = __builtin_bit_cast(f16x8, wf[mt][0]), w_lo = __builtin_bit_cast(f16x8, wf[mt][1]);
        S[mt] = mfma16_f16(w_hi, fb_hi, S[mt]);
        S[mt] = mfma16_f16(w_hi, fb_lo, S[mt]);
        S[mt] = mfma16_f16(w_lo, fb_hi, S[mt]);
      }
    };
    // The 13 k-steps are UNROLLED: every position is then a compile-time constant + q, so ring slots and sample
    // addresses are immediate offsets, and the window-edge selects exist only in the four k-steps that touch an edge
    // (0, 10, 11, 12).  As a rolled loop the same body spent ~60 of its ~200 vector instructions per k-step on
    // pos % 10, range compares and selects.
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      // The dense products of k-step ks - 1 go FIRST: they depend on nothing computed in this iteration, and the
      // conv1 arithmetic below is interleaved with them (one MFMA, three vector instructions, ...): measured with
      // parts compiled out, the MFMAs at the END of a k-step added their full pipe time to the launch (12.6 us).
      dense();                                         // (k-step "-1": zero fragments)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {                 // this k-step's dense weights (used at the top of the next iteration)
        wf[mt][1] = *(const f32x4*)(wd + ((ks * 4 + mt) * 2 + 1) * 256);
        wf[mt][0] = *(const f32x4*)(wd + ((ks * 4 + mt) * 2) * 256);
      }
      // conv1 of the NEXT set (positions 4 ks + 4 .. + 7; pair B below needs its first position), samples one step ahead
      const float xr0 = xres[0] * kCnnRImgScale, xr1 = xres[1] * kCnnRImgScale;
      {
        const float x0 = xa[0], x1 = xa[1], x2 = xa[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) xa[k] = ldq(4 * (ks + 2) - 1 + k);
        if (ks + 1 < NKS) {
#pragma unroll
          for (int pi = 0; pi < 2; ++pi) xres[pi] = ldh(4 * (ks + 1) + 2 * pi);
        }
        const int C = 4 * (ks + 1);                    // positions C + q; the last k-step's lie outside: nothing to do
        if (C < kSig) {
          lds_h* d = slot_c(C);
          if (C + 3 >= kSig) d = q < kSig - C ? d : zslot + kCnnRSlot;
          conv1_store(d, x0, x1, x2);
        }
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
      }
      wave_lds_fence();                                // other quarters' conv1 results -> this lane's fragment reads
      // both position pairs' conv2 triples FIRST, then both epilogues: the epilogue of pair 0 no longer waits (s_nop 9) for the
      // matrix result it reads - the other pair's three products stand in between
      f32x4 acc2[2];
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int p = 4 * ks + 2 * pi;
        const lds_h* src = slot_c(p - 1);
        if (p - 1 < 0) src = q > 0 ? src : zslot;
        if (p - 1 + 3 >= kSig) src = q < kSig - (p - 1) ? src : zslot;
        const f16x8 b_lo = *(const lds_h8*)(src + 16 * 8);
        const f16x8 b_hi = *(const lds_h8*)src;
        f32x4 acc = binit;
        acc = mfma16_f16(a2_hi, b_hi, acc);
        acc = mfma16_f16(a2_hi, b_lo, acc);
        acc = mfma16_f16(a2_lo, b_hi, acc);
        acc2[pi] = acc;
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const float xs = pi ? xr1 : xr0;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(relu_acc(acc2[pi][r]), k1[r], k2[r] + xs);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f16x2r hp = __builtin_convertvector(f32x2r{v[r], v[r + 1]}, f16x2r);
          fb_hi[4 * pi + r] = hp[0]; fb_hi[4 * pi + r + 1] = hp[1];
          const f16x2r lp = __builtin_convertvector(f32x2r{NRV_LO(v[r], hp[0]), NRV_LO(v[r + 1], hp[1])}, f16x2r);
          fb_lo[4 * pi + r] = lp[0]; fb_lo[4 * pi + r + 1] = lp[1];
        }
      }
    }
    dense();                                           // the last k-step's products
    // ---- S x 2^6 as f16 split planes: output features 16 mt + 4 q .. + 3 of event n = 8 bytes per term
    float* dst = P.out + (size_t)b * 16 * 128 + (q >> 1) * 128 + (16 * sub + n) * 4 + (q & 1) * 2;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = S[mt][r] * kCnnRDenseDescale;
        bad |= !(__builtin_fabsf(v[r]) <= 65504.f);
      }
      const f16x2r h01 = __builtin_convertvector(f32x2r{v[0], v[1]}, f16x2r);
      const f16x2r h23 = __builtin_convertvector(f32x2r{v[2], v[3]}, f16x2r);
      const f16x2r l01 = __builtin_convertvector(f32x2r{v[0] - (float)h01[0], v[1] - (float)h01[1]}, f16x2r);
      const f16x2r l23 = __builtin_convertvector(f32x2r{v[2] - (float)h23[0], v[3] - (float)h23[1]}, f16x2r);
      *(f16x4r*)(dst + (4 * mt) * 128) = f16x4r{h01[0], h01[1], h23[0], h23[1]};
      *(f16x4r*)(dst + (4 * mt + 2) * 128) = f16x4r{l01[0], l01[1], l23[0], l23[1]};
    }
    wave_lds_fence();                                  // the next unit overwrites the ring
  }
  bad |= !(xmax <= P.ep[24]);
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicAdd(args.sat, 1u);
}

#undef NRV_LO
#undef w1v
#undef b1v
#undef s1v
#undef h1v

}  // namespace nrv
