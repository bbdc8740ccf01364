// Signal branch in f16x2 mode, register-resident: conv1 (VALU) -> conv2 (MFMA) -> dense 400->64 (MFMA) per wave,
// no feature image, no roles, no barrier per tile (cnn_r_kernel).
#pragma once
#include "nrv_common.h"
#include "nrv_lstm_f16x2s.h"   // f16x8, mfma16_f16

namespace nrv {

// ---------------------------------------------------------------------------------------
// cnn_r_kernel.  nanorevcnn.py:17-38 + output_handeler.py:209-215, as cnn_h2_kernel / cnn_m_kernel computed them.
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
// Each conv1 position is computed by two quarters (the price of keeping everything in registers: 2 x 40 VALU per
// pair); per 16-event unit a wave issues ~2300 vector instructions, 231 MFMAs, 104 + ~100 LDS reads.
// A wave owns whole units, eight waves per workgroup (two per SIMD), persistent: one workgroup per CU and model.
// ---------------------------------------------------------------------------------------
struct CnnRConsts {        // per model, BY VALUE in the kernel arguments: scalar loads
  float w1[24];            // first convolution [tap][co]
  float b1[8];
  float s1[8], h1[8];      // BatchNorm 1 scale / shift x 2^6 (c1 enters conv2 x 2^6)
};
struct CnnRModelParams {
  const void* w2frag;      // conv2 A operand x 2^u: [term 2][64 lanes][8 f16] (two-position form, see above)
  const float* ep;         // [3][8]: accumulator init b2 x 2^(6+u), k1 = s2 x 2^-u, k2 = h2 x 2^6 per output channel
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
};

#ifndef NRV_CNNR_WAVES
#define NRV_CNNR_WAVES 8
#endif
constexpr int kCnnRWaves = NRV_CNNR_WAVES, kCnnRThreads = 64 * kCnnRWaves;
constexpr int kCnnRXs = 52;                             // samples per event in the wave-private image: index p + 1, zero halo
constexpr float kCnnRImgScale = 64.0f;                  // 2^6
constexpr float kCnnRDenseDescale = 1.0f / 1024.0f;     // 2^-10

typedef float f32x2r __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32r;

__global__ void __launch_bounds__(kCnnRThreads) cnn_r_kernel(const CnnRArgs args) {
  constexpr int NKS = 13;
  __shared__ __attribute__((aligned(16))) float wd_s[NKS * 4 * 2 * 256];          // 104 KiB
  __shared__ __attribute__((aligned(16))) float xr_s[kCnnRWaves * 16 * kCnnRXs];  // 26 KiB
  const CnnRModelParams& P = args.m[blockIdx.y];
  const CnnRConsts& K = args.k[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;

  {
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(P.dfrag, NKS * 8 * 1024);
    for (int base = 0; base < NKS * 8 * 64; base += 8 * kCnnRThreads) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load16(wrs, (unsigned)(base + j * kCnnRThreads + threadIdx.x) * 16, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (base + j * kCnnRThreads + threadIdx.x < NKS * 8 * 64) ((f32x4*)wd_s)[base + j * kCnnRThreads + threadIdx.x] = v[j];
    }
    for (int i = threadIdx.x; i < kCnnRWaves * 16 * kCnnRXs; i += kCnnRThreads) xr_s[i] = 0.f;   // the halos stay zero
  }
  __syncthreads();                                   // the only barrier: from here on the waves are independent

  lds_f32r* const xr = (lds_f32r*)xr_s + wave * 16 * kCnnRXs;
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

  // units of 16 events: unit u = 2 * tile + sub, dealt round-robin over all waves of the launch's workgroups
  const int n_units = 2 * args.n_tiles;
  for (int u = blockIdx.x * kCnnRWaves + wave; u < n_units; u += gridDim.x * kCnnRWaves) {
    const int b = u >> 1, sub = u & 1;
    const int wt = b / T, t = b % T;
    // ---- the unit's samples -> wave-private image xr[event][p + 1]
    {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(args.signal + ((size_t)wt * 32 * T + t) * kSig, 0xffffffffu);
      float v[13];
#pragma unroll
      for (int it = 0; it < 13; ++it) {
        const int idx = lane + 64 * it;                // 0 .. 831; 800 real
        const int ev = idx / kSig, p = idx - ev * kSig;
        const int row = 16 * sub + ev;
        const bool ok = idx < 16 * kSig && wt * 32 + row < args.n_rows;
        const float x = buf_load4(rs, ok ? (unsigned)((row * T * kSig + p) * 4) : 0u, 0);
        v[it] = ok ? x : 0.f;
      }
#pragma unroll
      for (int it = 0; it < 13; ++it) {
        const int idx = lane + 64 * it;
        const int ev = idx / kSig, p = idx - ev * kSig;
        if (idx < 16 * kSig) xr[ev * kCnnRXs + p + 1] = v[it];
      }
    }
    wave_lds_fence();
    const lds_f32r* const xe = xr + n * kCnnRXs;       // this lane's event

    f32x4 S[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) S[mt] = dbias[mt];

#pragma unroll 1
    for (int ks = 0; ks < NKS; ++ks) {
      // dense weights of this k-step: requested now, used after the two conv pairs
      f32x4 wf[4][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        wf[mt][1] = *(const f32x4*)(wd + ((ks * 4 + mt) * 2 + 1) * 256);
        wf[mt][0] = *(const f32x4*)(wd + ((ks * 4 + mt) * 2) * 256);
      }
      f16x8 fb_hi, fb_lo;                              // B fragment of the dense k-step
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int p = 4 * ks + 2 * pi;                 // this triple gives positions p, p + 1
        // ---- conv1 + BatchNorm at position pp = p + q - 1 of this lane's event (zero outside the window)
        const int pp = p + q - 1;
        const bool inside = pp >= 0 && pp < kSig;
        const int rp = pp < 0 ? 0 : (pp > kSig - 1 ? kSig - 1 : pp);
        const float xm = xe[rp], xc = xe[rp + 1], xp = xe[rp + 2];      // x[rp - 1], x[rp], x[rp + 1]
        float c[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          float tt = K.b1[o];
          tt = __builtin_fmaf(xm, K.w1[0 * 8 + o], tt);
          tt = __builtin_fmaf(xc, K.w1[1 * 8 + o], tt);
          tt = __builtin_fmaf(xp, K.w1[2 * 8 + o], tt);
          tt = __builtin_fmaxf(tt, 0.f);
          tt = __builtin_fmaf(tt, K.s1[o], K.h1[o]);
          c[o] = inside ? tt : 0.f;
        }
        f16x8 b_hi, b_lo;
#pragma unroll
        for (int o = 0; o < 8; o += 2) {
          const f16x2r hp = __builtin_convertvector(f32x2r{c[o], c[o + 1]}, f16x2r);
          b_hi[o] = hp[0]; b_hi[o + 1] = hp[1];
          const f16x2r lp = __builtin_convertvector(f32x2r{c[o] - (float)hp[0], c[o + 1] - (float)hp[1]}, f16x2r);
          b_lo[o] = lp[0]; b_lo[o + 1] = lp[1];
        }
        // ---- conv2: positions p (rows 0-7) and p + 1 (rows 8-15) of 16 events
        f32x4 acc = binit;
        acc = mfma16_f16(a2_hi, b_hi, acc);
        acc = mfma16_f16(a2_hi, b_lo, acc);
        acc = mfma16_f16(a2_lo, b_hi, acc);
        // ---- bias is in, ReLU, BatchNorm, + sample (x 2^6), split: elements 4 pi .. 4 pi + 3 of the dense B fragment
        const int P2 = p + (q >> 1);                   // this lane's position (<= 51; 50, 51 meet zero weights)
        const float xs = xe[(P2 < kSig ? P2 : kSig - 1) + 1] * kCnnRImgScale;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(__builtin_fmaxf(acc[r], 0.f), k1[r], k2[r] + xs);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f16x2r hp = __builtin_convertvector(f32x2r{v[r], v[r + 1]}, f16x2r);
          fb_hi[4 * pi + r] = hp[0]; fb_hi[4 * pi + r + 1] = hp[1];
          const f16x2r lp = __builtin_convertvector(f32x2r{v[r] - (float)hp[0], v[r + 1] - (float)hp[1]}, f16x2r);
          fb_lo[4 * pi + r] = lp[0]; fb_lo[4 * pi + r + 1] = lp[1];
        }
      }
      // ---- dense k-step: 4 output tiles x (hi*hi, hi*lo, lo*hi)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f16x8 w_hi = __builtin_bit_cast(f16x8, wf[mt][0]), w_lo = __builtin_bit_cast(f16x8, wf[mt][1]);
        S[mt] = mfma16_f16(w_hi, fb_hi, S[mt]);
        S[mt] = mfma16_f16(w_hi, fb_lo, S[mt]);
        S[mt] = mfma16_f16(w_lo, fb_hi, S[mt]);
      }
    }
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
    wave_lds_fence();                                  // the next unit overwrites the sample image
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicAdd(args.sat, 1u);
}

}  // namespace nrv
