// Signal branch in f16x2 mode with the 8 -> 8 convolution on the matrix pipe (cnn_m_kernel).
#pragma once
#include "nrv_cnn_f16x2.h"     // image layout constants, CnnH2ModelParams, the dense (matrix) role
#include "nrv_lstm_f16x2s.h"   // mfma16_f16

#ifdef NRV_EXPERIMENTS      // superseded by cnn_r_kernel (nrv_cnn_r.h); kept for A/B timing (NRV_CNN=m)
namespace nrv {

// ---------------------------------------------------------------------------------------
// cnn_m_kernel.  Same computation as cnn_h2_kernel (nanorevcnn.py:17-38, output_handeler.py:209-215) and the same
// workgroup shape - persistent, twelve waves, four of them running the 400 -> 64 layer of tile i - 1 while
// eight build tile i's feature image, one barrier per tile - but the conv waves no longer spend 4800 of their
// ~7500 vector instructions per tile on the 8 -> 8 convolution's FMAs: that convolution is three accumulating
// GEMM taps, and with the output channels on M, the POSITIONS on N and (tap, input channel) on K it needs no
// im2col at all:
//     out^T[co][p] = sum_{tap, ci} W2[tap][ci][co] * c1[p + tap - 1][ci]
//   A (weights, resident: 8 VGPRs)  lane (m = l & 15, kg = l >> 4):  W2[tap = kg][ci = j][co = m]   (co >= 8, tap 3: zero)
//   B (activations)                 lane (n = l & 15, kg = l >> 4):  c1[position p(n) + kg - 1][ci = j]
// i.e. the B fragment of a tile of 16 positions is ONE 16-byte LDS read per lane and term from a
// [position][8 channels] image of the first convolution's output - the natural layout its producer writes
// with one 16-byte store per position and term; the three taps are the three k-groups reading the same
// image one position apart.  Result tile: lane (position n, q = l >> 4) holds output channels 4 q .. 4 q + 3
// (q < 2): bias, ReLU, BatchNorm, the residual sample, the x 2^6 scale and the f16 split happen in the lane,
// and the 4 channels leave as 8 contiguous bytes per term of the dense layer's A-fragment image.
// f32-grade as everywhere in this mode: both operands are scaled two-term f16 splits, three products per tile
// (hi*hi, hi*lo, lo*hi), f32 accumulation.
//
// A conv wave owns 8 of the tile's 32 events, two at a time, start to finish - samples -> first convolution
// (VALU, one position per lane, its 48 constants in SGPRs: they are kernel arguments) -> c1 image (wave-private
// LDS, 3.5 KB) -> 7 matrix tiles of 16 positions -> feature image - so nothing but the tile barrier that
// exists anyway synchronises it with anybody.  Per wave and tile ~930 instructions instead of ~1900, 42 MFMAs.
// Out-of-range input: nothing is clamped; an overflowing c1 or feature becomes an f16 infinity, the event's
// outputs become non-finite and the dense role's range guard (nrv_cnn_f16x2.h) reports the launch group.
// ---------------------------------------------------------------------------------------
struct CnnMConsts {        // per model, BY VALUE in the kernel arguments: scalar loads, no vector-memory traffic
  float w1[24];            // first convolution [tap][co]
  float b1[8];
  float s1[8], h1[8];      // BatchNorm 1 scale / shift x 2^6 (c1 is kept x 2^6)
};
struct CnnMModelParams {
  const void* w2frag;      // second convolution x 2^u as the A operand: [term 2][64 lanes][8 f16]
  const float* ep;         // [3][16]: accumulator init b2 x 2^(6+u), k1 = s2 x 2^-u, k2 = h2 x 2^6 per output channel (8..15: 0)
};
struct CnnMArgs {
  CnnH2ModelParams m[2];   // dense layer + output (conv pointer unused)
  CnnMModelParams c[2];
  CnnMConsts k[2];
  const float* signal;
  int T, n_rows, n_tiles;
  unsigned* sat;
};

constexpr int kCnnMC1Ev = 56;                            // positions per event in the c1 image: 52 (one halo each side) + pad
constexpr int kCnnMC1Wave = 2 * 2 * kCnnMC1Ev * 8;       // f16 per conv wave: [term][event 2][position][8]
constexpr int kCnnMXr = 2 * 60;                          // f32 per conv wave: samples of two events, index p + 2, zero halo

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Role split of cnn_m_kernel's twelve waves.  Measured with parts compiled out (r03, scripts/gpu_variants.py): with the
// second convolution on the matrix pipe the conv role of EIGHT waves needs 7 us per launch, while the dense role of
// four waves (cnn_dense_role: both row halves per wave, one k-step of read-ahead) needs 35 us alone and 78 us beside the
// conv role - it waits for its LDS reads, 200 cycles per k-step of 96 MFMA cycles.  So the waves are dealt the other
// way round: EIGHT dense waves - wave (ct, rh) owns 16 output columns and ONE row half, two of them per SIMD so that one's
// read latency is the other's matrix time, and a ring of four k-steps - and FOUR conv waves of 8 events each.
#ifndef NRV_CNNM_DENSE
#define NRV_CNNM_DENSE 8
#endif
constexpr int kCnnMDenseWaves = NRV_CNNM_DENSE, kCnnMConvWaves = 12 - NRV_CNNM_DENSE;
static_assert(64 * (kCnnMDenseWaves + kCnnMConvWaves) == kCnnH2Threads, "twelve waves");

__device__ __forceinline__ void cnn_m_dense_role(const CnnH2ModelParams& P, lds_f16* const img, const int nloc, const int G,
                                                 const int wave, const int lane, unsigned* sat) {
  constexpr int CH = kCnnH2CH, IMG = kCnnH2IMG;
  constexpr int NRH = 8 / kCnnMDenseWaves;        // row halves per wave: 1 (eight dense waves) or 2 (four)
  const int ct = wave & 3, rh0 = NRH == 1 ? (wave >> 2) : 0;
  const int n16 = lane & 15, kg = lane >> 4;
  constexpr int NKS = 13, NRING = NRH == 1 ? 4 : 2, LEAD = NRING - 1;
  f16x8 bw[NKS][2];
#pragma unroll
  for (int k = 0; k < NKS; ++k)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
      bw[k][tm] = *(const f16x8*)((const char*)P.dsplit + ((size_t)((k * 4 + ct) * 2 + tm) * 64 + lane) * 16);
  const float bias = P.dbias[ct * 16 + n16];
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};           // lo*hi, hi*lo, hi*hi
  bool bad = false;                              // range guard: an output that is not |v| <= f16 max (NaN included)
  __syncthreads();                               // tile 0 is being convolved
  for (int i = 1; i <= nloc; ++i) {
#if !(NRV_EXP & 2048)                               // timing experiment: no dense layer at all
    {
      const lds_f16* im = img + ((i - 1) & 1) * IMG;
      // A fragment of k-step ks: lane (row n16, k-group kg) reads k-block 2 ks + (kg >> 1), half kg & 1, event 16 rh + n16
      const lds_f16* ap = im + ((kg >> 1) * 4 + (kg & 1)) * CH + (16 * rh0 + n16) * 8;
      f32x4 acc[NRH];
#pragma unroll
      for (int r = 0; r < NRH; ++r) acc[r] = f32x4{bias, bias, bias, bias};
      f16x8 at[NRING][NRH][2];
      auto rd = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NRH; ++r) {
#if NRV_EXP & 131072                                // timing experiment: no LDS reads in the dense role
          at[k % NRING][r][0] = bw[k][0]; at[k % NRING][r][1] = bw[k][1];
#else
          at[k % NRING][r][0] = *(const lds_f16x8*)(ap + k * 8 * CH + r * 16 * 8);
          at[k % NRING][r][1] = *(const lds_f16x8*)(ap + k * 8 * CH + r * 16 * 8 + 2 * CH);
#endif
        }
      };
#pragma unroll
      for (int k = 0; k < LEAD; ++k) rd(k);
#pragma unroll
      for (int k = 0; k < NKS; ++k) {
        if (k + LEAD < NKS) rd(k + LEAD);
        __builtin_amdgcn_sched_barrier(0);
#if !(NRV_EXP & 65536)                              // timing experiment: no MFMAs in the dense role
#pragma unroll
        for (int r = 0; r < NRH; ++r)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at[k % NRING][r][PA[pr]], bw[k][PB[pr]], acc[r], 0, 0, 0);
#else
#pragma unroll
        for (int r = 0; r < NRH; ++r) acc[r][0] += (float)at[k % NRING][r][0][0] + (float)at[k % NRING][r][1][0];
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: S x 2^6 as f16 split planes
      float* dst = P.out + (size_t)(blockIdx.x + (i - 1) * G) * 16 * 128;
      const int u = ct * 16 + n16;                // output feature of this lane
#pragma unroll
      for (int r = 0; r < NRH; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (rh0 + r) * 16 + 4 * kg + q;
          const float v = acc[r][q] * kDenseDescale;
          bad |= !(__builtin_fabsf(v) <= 65504.f);
          const _Float16 hi = (_Float16)v;
          const _Float16 lo = (_Float16)(v - (float)hi);
          _Float16* d16 = (_Float16*)(dst + ((u >> 4) * 4 + ((u >> 3) & 1)) * 128 + row * 4) + (u & 7);
          d16[0] = hi;
          d16[2 * 128 * 2] = lo;
        }
    }
#endif
    __syncthreads();
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicAdd(sat, 1u);
}

__global__ void __launch_bounds__(kCnnH2Threads) cnn_m_kernel(const CnnMArgs args) {
  constexpr int CH = kCnnH2CH, IMG = kCnnH2IMG;
  __shared__ __attribute__((aligned(16))) _Float16 img_s[2 * IMG];
  __shared__ __attribute__((aligned(16))) _Float16 c1_s[kCnnMConvWaves * kCnnMC1Wave];
  __shared__ __attribute__((aligned(16))) float xr_s[kCnnMConvWaves * kCnnMXr];
  lds_f16* const img = (lds_f16*)img_s;

  const CnnH2ModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ntile = args.n_tiles, G = gridDim.x;
  const int nloc = (ntile - (int)blockIdx.x + G - 1) / G;     // tiles of this workgroup: b = blockIdx.x + i*G

  // the padding k-block (features 400..415) of both images is zero for the whole launch; so are the halos of the
  // sample images (index 0, 1 and 52..59 of each event)
  for (int i = threadIdx.x; i < 2 * 4 * CH; i += kCnnH2Threads)
    img_s[(i / (4 * CH)) * IMG + 25 * 4 * CH + i % (4 * CH)] = (_Float16)0.f;
  for (int i = threadIdx.x; i < kCnnMConvWaves * kCnnMXr; i += kCnnH2Threads) xr_s[i] = 0.f;
  for (int i = threadIdx.x; i < kCnnMConvWaves * kCnnMC1Wave; i += kCnnH2Threads) c1_s[i] = (_Float16)0.f;
  __syncthreads();

  if (wave >= kCnnMDenseWaves) {
    // ================================ CONV role ==============================================
    const int cwv = wave - kCnnMDenseWaves;
    const CnnMConsts& K = args.k[blockIdx.y];
    const CnnMModelParams& C = args.c[blockIdx.y];
    const int n16 = lane & 15, q = lane >> 4;
    lds_f16* const c1 = (lds_f16*)c1_s + cwv * kCnnMC1Wave;
    lds_f32* const xr = (lds_f32*)xr_s + cwv * kCnnMXr;
    // second convolution's weights (A operand) and this lane's epilogue constants (channels 4 q .. 4 q + 3)
    const f16x8 a_hi = *(const f16x8*)((const char*)C.w2frag + lane * 16);
    const f16x8 a_lo = *(const f16x8*)((const char*)C.w2frag + 1024 + lane * 16);
    const f32x4 binit = *(const f32x4*)(C.ep + 4 * q);
    const f32x4 k1 = *(const f32x4*)(C.ep + 16 + 4 * q);
    const f32x4 k2 = *(const f32x4*)(C.ep + 32 + 4 * q);

    // samples of this wave's eight events of tile b: x.v[pass][e] of lane p < 50 is sample p of event 8 cwv + 2 pass + e
    constexpr int NPS = 32 / kCnnMConvWaves / 2;           // passes of two events
    struct XRegs { float v[NPS][2]; };
    auto load_x = [&](int b) __attribute__((always_inline)) {
      XRegs x;
      const int bt = b < ntile ? b : 0;
      const int wt = bt / T, t = bt % T;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(args.signal + ((size_t)wt * 32 * T + t) * kSig, 0xffffffffu);
      const int p = lane < kSig ? lane : kSig - 1;
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ev = 2 * NPS * cwv + 2 * ps + e;
          const bool ok = b < ntile && wt * 32 + ev < args.n_rows && lane < kSig;
          const float v = buf_load4(rs, ok ? (unsigned)((ev * T * kSig + p) * 4) : 0u, 0);
          x.v[ps][e] = ok ? v : 0.f;
        }
      return x;
    };

    XRegs x = load_x(blockIdx.x);
    for (int i = 0; i < nloc; ++i) {
      const XRegs xn = load_x(blockIdx.x + (i + 1) * G);     // next tile's samples: a whole iteration of lead
      lds_f16* const im = img + (i & 1) * IMG;
#pragma unroll
      for (int ps = 0; ps < ((NRV_EXP & 4096) ? 0 : NPS); ++ps) {   // (4096: timing experiment, no conv work)
        // ---- samples -> wave-private image (index p + 2, zeros around)
        if (lane < kSig) {
          xr[lane + 2] = x.v[ps][0];
          xr[60 + lane + 2] = x.v[ps][1];
        }
        wave_lds_fence();
        // ---- first convolution + BatchNorm, one position per lane: j = lane + 64 round over (event 2, pp 52), the
        // image position pp = p + 1 (pp = 0 and 51 are the zero padding the second convolution sees)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int j = lane + 64 * rd;
          const int el = j >= 52 ? 1 : 0, pp = j - 52 * el;
          if (j < 104) {
            const lds_f32* xs = xr + el * 60 + pp;            // x[p - 1], x[p], x[p + 1] with p = pp - 1 -> index p + 1 ..
            const float xm = xs[0], xc = xs[1], xp = xs[2];
            const bool inside = pp >= 1 && pp <= kSig;
            f16x8 hi, lo;
            float v[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
              float t = K.b1[o];
              t = __builtin_fmaf(xm, K.w1[0 * 8 + o], t);
              t = __builtin_fmaf(xc, K.w1[1 * 8 + o], t);
              t = __builtin_fmaf(xp, K.w1[2 * 8 + o], t);
              t = __builtin_fmaxf(t, 0.f);
              t = __builtin_fmaf(t, K.s1[o], K.h1[o]);
              v[o] = inside ? t : 0.f;
            }
#pragma unroll
            for (int o = 0; o < 8; o += 2) {
              const f16x2 hp = __builtin_convertvector(f32x2{v[o], v[o + 1]}, f16x2);
              hi[o] = hp[0]; hi[o + 1] = hp[1];
              const f16x2 lp = __builtin_convertvector(f32x2{v[o] - (float)hp[0], v[o + 1] - (float)hp[1]}, f16x2);
              lo[o] = lp[0]; lo[o + 1] = lp[1];
            }
            lds_f16* d = c1 + (el * kCnnMC1Ev + pp) * 8;
            *(lds_f16x8*)d = hi;
            *(lds_f16x8*)(d + 2 * kCnnMC1Ev * 8) = lo;
          }
        }
        wave_lds_fence();
        // ---- second convolution on the matrix pipe: 7 tiles of 16 positions over the linear index i2 = 50 event + p
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const int i2 = 16 * k + n16;
          const bool valid = i2 < 100;
          const int ic = valid ? i2 : 99;
          const int el = ic >= kSig ? 1 : 0, p = ic - kSig * el;
          const lds_f16* bsrc = c1 + (el * kCnnMC1Ev + p + q) * 8;       // image position (p + tap - 1) + 1, tap = q
          const f16x8 b_hi = *(const lds_f16x8*)bsrc;
          const f16x8 b_lo = *(const lds_f16x8*)(bsrc + 2 * kCnnMC1Ev * 8);
          const float xres = xr[el * 60 + p + 2];
          f32x4 acc = binit;
#if !(NRV_EXP & 16384)                              // timing experiment: no MFMAs in the conv role
          acc = mfma16_f16(a_hi, b_hi, acc);
          acc = mfma16_f16(a_hi, b_lo, acc);
          acc = mfma16_f16(a_lo, b_hi, acc);
#else
          acc[0] += (float)b_hi[0] + (float)b_lo[1] + (float)a_hi[0] + (float)a_lo[0];
#endif
          if (valid && q < 2) {
            const float xs = xres * kImgScale;              // Add(): the raw sample, broadcast over the channels
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(__builtin_fmaxf(acc[r], 0.f), k1[r], k2[r] + xs);
            const f16x2 h01 = __builtin_convertvector(f32x2{v[0], v[1]}, f16x2);
            const f16x2 h23 = __builtin_convertvector(f32x2{v[2], v[3]}, f16x2);
            const f16x2 l01 = __builtin_convertvector(f32x2{v[0] - (float)h01[0], v[1] - (float)h01[1]}, f16x2);
            const f16x2 l23 = __builtin_convertvector(f32x2{v[2] - (float)h23[0], v[3] - (float)h23[1]}, f16x2);
            // flat index p*8 + co: k-block p >> 1, half p & 1; chunk (kb, term, half) = 32 events x 16 B (+ pad)
            const int ev = 2 * NPS * cwv + 2 * ps + el;
            lds_f16* d = im + ((p >> 1) * 4 + (p & 1)) * CH + ev * 8 + 4 * q;
            typedef __attribute__((address_space(3))) f16x4 lds_f16x4;
#if NRV_EXP & 32768                                 // timing experiment: no feature-image writes
            if (v[0] == 1.2345f)
#endif
            {
              *(lds_f16x4*)d = f16x4{h01[0], h01[1], h23[0], h23[1]};
              *(lds_f16x4*)(d + 2 * CH) = f16x4{l01[0], l01[1], l23[0], l23[1]};
            }
          }
        }
        wave_lds_fence();                                  // the next pass overwrites the sample and c1 images
      }
      x = xn;
      __syncthreads();
    }
    __syncthreads();                               // the matrix role's last tile
  } else {
    cnn_m_dense_role(P, img, nloc, G, wave, lane, args.sat);
  }
}

}  // namespace nrv
#endif
