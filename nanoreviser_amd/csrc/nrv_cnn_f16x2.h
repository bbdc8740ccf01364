// Signal branch in f16x2 mode: conv1d x2 + BN + residual on ALL eight waves, dense(400 -> 64) on the f16
// matrix pipe with the scaled two-term split (nrv_lstm_f16x2.h).
#pragma once
#include "nrv_cnn.h"
#include "nrv_lstm_f16x2.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// cnn_h2_kernel.  Same computation as cnn_kernel (nanorevcnn.py:17-38, output_handeler.py:209-215),
// re-balanced around two measurements:
//   * one wave alone on a SIMD issues a VALU instruction every 4 cycles, two waves together one every
//     2 (MI355X_MICROARCH.md, constants table).  cnn_kernel's four conv waves each sit alone on the
//     VALU of their SIMD (their partner is a matrix wave) and are the critical path of a tile: ~1900
//     instructions per thread and tile at 4 cycles.  Here the workgroup has TWELVE waves, eight of
//     them convolving (3-4 positions per thread instead of 6-7): every SIMD holds two conv waves (and
//     one matrix wave) and the same instructions issue at twice the rate;
//   * splitting an activation once in its PRODUCER is 3x cheaper than splitting it in every consumer:
//     the conv threads store the 400 features of an event already scaled (x 2^6) and split into two
//     f16 terms, as the A fragments the dense layer reads -
//     [k-block 26][term][half][32 events][8 f16], one 16-byte LDS store per (event, position, term),
//     exactly the two stores the f32 image needed.  The four matrix waves then run the dense layer
//     with NO VALU work in the loop: wave ct owns 16 output columns as v_mfma_f32_16x16x32_f16 tiles
//     (two row halves x 13 k-steps x 3 products = 78 MFMAs of 16 cycles per tile), A fragments straight
//     from LDS, B fragments resident in registers (104 VGPRs), no partial sums to exchange.
// Roles are separate code paths (waves 0-3 matrix, 4-11 conv), so each has the whole 168-register
// budget of a 12-wave workgroup to itself; the conv path reads its 264 constants as LDS broadcasts
// (hand-issued scalar loads, s_load_dwordx8 through inline asm with SGPR operands for the FMAs, run
// at the same speed; from global memory hipcc makes them per-lane vector loads, because the kernel
// also stores to global memory).
// Measured (same box, parts compiled out): whole kernel 80 us; the 8->8 convolution's FMAs 43 us -
// tools/microbench/valu_waves.hip puts the practical v_fma_f32 rate at 1.4-1.7 ns per wave-instruction
// and SIMD with 2-4 waves per SIMD (4.4 ns with one), i.e. a floor of ~22 us for them; the dense layer's
// MFMAs 6 us, its split-plane epilogue stores 2.5 us, everything else (conv1, loads, barriers) 20 us.
// cnn_kernel<true,true> (4 conv waves, bf16x3 dense): 90 us.
// Persistent: one workgroup per CU and model, tile i is convolved while tile i-1 is multiplied; one
// barrier per tile.  Output: f16 split planes of S x 2^6 (the layout lstm_h2o_kernel reads).
//
// RANGE GUARD.  The conv features and S are the one activation of the f16x2 mode without a static bound
// (the reference normalises samples as (raw - median) / MAD without any clipping, preprocessing.py:120-131,
// so a spike sample or a tiny MAD gives arbitrarily large inputs, and f32 arithmetic gives them a
// well-defined answer).  Nothing is clamped here: a feature beyond the f16 range becomes an f16 infinity, a
// NaN sample a NaN, either makes every one of the event's 64 outputs non-finite, and the matrix role's
// epilogue - which sees every output anyway - counts outputs that are not |S x 2^6| <= 65504 into *sat.
// The host re-runs a launch group with a non-zero count on the f32 kernels (nrv_api.hip), so the f16x2
// mode never returns a result that differs from the f32 mode's for out-of-range input.
// ---------------------------------------------------------------------------------------
struct CnnH2ModelParams {
  const float* conv;      // 24 w1[k][o], 8 b1, 8 s1, 8 h1, 192 w2[k][c][o], 8 b2, 8 s2 x 2^6, 8 h2 x 2^6  (=264)
  const void* dsplit;     // dense 400->64 x 2^10 as f16x2 B fragments of the 16x16x32 tile:
                          // [ks 13][ct 4][term 2][64 lanes][8 f16], lane l: k = 32 ks + 8 (l>>4) + j, n = 16 ct + (l&15)
  const float* dbias;     // [64] x 2^16 (image scale 2^6 x weight scale 2^10)
  float* out;             // split planes, KQ=16 chunks per tile: window-major [wtile][t][..] or event-major [etile][..]
};
struct CnnH2Args {
  CnnH2ModelParams m[2];
  const float* signal;    // [n][T][50] (window mode) or [N][50] (event mode)
  int T;                  // window mode: T; event mode: 1
  int n_rows;             // windows (window mode) or events (event mode)
  int n_tiles;            // 32-event tiles to process (per model)
  unsigned* sat;          // range guard: += waves that produced an output outside the f16 range (or NaN)
};

constexpr float kImgScale = 64.0f;          // 2^6: conv features in the LDS image and S in HBM
constexpr float kDenseDescale = 1.0f / 1024.0f;   // 2^-10: undo the dense weights' own scale
typedef float f32x8 __attribute__((ext_vector_type(8)));
// LDS pointers keep their address space across the not-inlined role functions (as generic pointers
// every access would become a flat instruction)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) _Float16 lds_f16;
typedef __attribute__((address_space(3))) f16x8 lds_f16x8;

#ifdef NRV_EXPERIMENTS   // the VALU form of the convolutions: experiments build only (NRV_CNNM=0)
// conv1+BN -> conv2+BN -> +signal for NP consecutive positions of one event; the 8 channels of a position
// leave as one scaled hi chunk and one lo chunk of the A-fragment image.  x[] holds samples p0-2 .. p0+NP+1.
template <int NP>
__device__ __forceinline__ void conv_positions_h2(const lds_f32* cw, const float (&x)[8], int p0,
                                                  int r, lds_f16* img) {
  float b1v[NP + 2][8];                            // bn1 at positions p0-1 .. p0+NP
  {
    float w1[48];                                  // w1[3][8], b1[8], bn1 scale[8], shift[8]
#pragma unroll
    for (int k = 0; k < 48; ++k) w1[k] = cw[k];
#pragma unroll
    for (int q = 0; q < NP + 2; ++q) {
      const int p = p0 - 1 + q;
      const bool inside = (p >= 0) && (p < kSig);
      const float xm = x[q], xc = x[q + 1], xp = x[q + 2];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float v = w1[24 + o];
        v = __builtin_fmaf(xm, w1[0 * 8 + o], v);
        v = __builtin_fmaf(xc, w1[1 * 8 + o], v);
        v = __builtin_fmaf(xp, w1[2 * 8 + o], v);
        v = __builtin_fmaxf(v, 0.f);
        v = __builtin_fmaf(v, w1[32 + o], w1[40 + o]);
        b1v[q][o] = inside ? v : 0.f;
      }
    }
  }
  // conv2, weight-stationary: each (tap k, in-channel ci) row of 8 weights is read once (two 16-byte LDS
  // broadcast reads, one row ahead of its use) and applied to all NP positions of this thread.  The
  // fences keep the reads next to their rows: left alone they are hoisted to the top and spilled.
  const lds_f32* w2 = cw + 48;
  float acc[NP][8];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[q][o] = w2[192 + o];
  float wnext[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) wnext[o] = w2[o];
#pragma unroll
  for (int kc = 0; kc < 24; ++kc) {
    float wrow[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) wrow[o] = wnext[o];
    if (kc + 1 < 24) {
      // the next row is requested HERE, in front of this row's FMAs (one row of lead).  Its address hangs
      // on an opaque zero that the asm produces from the previous row's last result, which pins the read
      // between the two rows (fences alone did not stop the reads of all 24 rows from being clustered at
      // the top and spilled)
      int z;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(acc[NP - 1][7]));
#pragma unroll
      for (int o = 0; o < 8; ++o) wnext[o] = w2[(kc + 1) * 8 + o + z];
    }
    const int k = kc >> 3, ci = kc & 7;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const float av = b1v[q + k][ci];
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[q][o] = __builtin_fmaf(av, wrow[o], acc[q][o]);
    }
  }
  float s2[8], h2[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) { s2[o] = w2[200 + o]; h2[o] = w2[208 + o]; }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = p0 + q;
    const float xs = x[q + 2] * kImgScale;         // Add(): the raw sample, broadcast over the channels
    f16x8 hi, lo;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float v = __builtin_fmaxf(acc[q][o], 0.f);
      v = __builtin_fmaf(v, s2[o], h2[o] + xs);
      const _Float16 hh = (_Float16)v;             // beyond the f16 range: +-inf, caught by the matrix role's guard
      hi[o] = hh;
      lo[o] = (_Float16)(v - (float)hh);
    }
    // flat index p*8+o: k-block p>>1, half p&1; image chunk (kb, term, half) = 32 events x 16 B (+16 B pad)
    lds_f16* d = img + ((p >> 1) * 4 + (p & 1)) * (33 * 8) + r * 8;
    *(lds_f16x8*)d = hi;
    *(lds_f16x8*)(d + 2 * 33 * 8) = lo;
  }
}

#endif

constexpr int kCnnH2MatWaves = 4, kCnnH2ConvWaves = 8;
constexpr int kCnnH2Threads = 64 * (kCnnH2MatWaves + kCnnH2ConvWaves);
constexpr int kCnnH2CH = 33 * 8;                   // f16 per (kb, term, half) chunk: 32 events x 8 + one event of padding
constexpr int kCnnH2NKB = 26;                      // k-blocks of 16 in the image: 25 real + one of zeros (K = 400 -> 416)
constexpr int kCnnH2IMG = kCnnH2NKB * 4 * kCnnH2CH;  // f16 per image

// The dense role of the signal branch (waves 0-3 of cnn_h2_kernel / cnn_m_kernel): wave ct owns output columns
// 16 ct .. + 15 of the 400 -> 64 layer, all 13 k-steps of 32, both row halves; its B fragments (26 KB) stay in
// registers for the whole launch; tile i - 1's image is multiplied while tile i is being built (one barrier per
// tile, matched by the conv role).  Range guard: see the header comment above.
__device__ __forceinline__ void cnn_dense_role(const CnnH2ModelParams& P, lds_f16* const img, const int nloc, const int G,
                                               const int wave, const int lane, unsigned* sat) {
  constexpr int CH = kCnnH2CH, IMG = kCnnH2IMG;
  // wave ct: output columns 16 ct .. +15, all 13 k-steps of 32, both row halves; its B fragments (26 KB)
  // stay in registers for the whole launch
  const int ct = wave;
  const int n16 = lane & 15, kg = lane >> 4;
  constexpr int NKS = 13;
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
      // A fragment of k-step ks, row half rh: lane (row n16, k-group kg) reads k-block 2 ks + (kg >> 1),
      // half kg & 1, event 16 rh + n16
      const lds_f16* ap = im + ((kg >> 1) * 4 + (kg & 1)) * CH + n16 * 8;
      f32x4 acc[2] = {{bias, bias, bias, bias}, {bias, bias, bias, bias}};
      f16x8 at[2][2][2];                                             // A: ring of two k-steps [slot][row half][term]
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        at[0][rh][0] = *(const lds_f16x8*)(ap + rh * 16 * 8);
        at[0][rh][1] = *(const lds_f16x8*)(ap + rh * 16 * 8 + 2 * CH);
      }
#pragma unroll
      for (int k = 0; k < NKS; ++k) {
        if (k + 1 < NKS) {
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) {
            at[(k + 1) & 1][rh][0] = *(const lds_f16x8*)(ap + (k + 1) * 8 * CH + rh * 16 * 8);
            at[(k + 1) & 1][rh][1] = *(const lds_f16x8*)(ap + (k + 1) * 8 * CH + rh * 16 * 8 + 2 * CH);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
            acc[rh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at[k & 1][rh][PA[pr]], bw[k][PB[pr]], acc[rh], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: S x 2^6 as f16 split planes
      float* dst = P.out + (size_t)(blockIdx.x + (i - 1) * G) * 16 * 128;
      const int u = ct * 16 + n16;                // output feature of this lane
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rh * 16 + 4 * kg + q;
          const float v = acc[rh][q] * kDenseDescale;
          bad |= !(__builtin_fabsf(v) <= 65504.f);
          const _Float16 hi = (_Float16)v;
          const _Float16 lo = (_Float16)(v - (float)hi);
          _Float16* d16 = (_Float16*)(dst + ((u >> 4) * 4 + ((u >> 3) & 1)) * 128 + row * 4) + (u & 7);
#if !(NRV_EXP & 8192)                               // timing experiment: no epilogue stores
          d16[0] = hi;
          d16[2 * 128 * 2] = lo;
#else
          if (v == 1.2345f) { d16[0] = hi; d16[2 * 128 * 2] = lo; }
#endif
        }
    }
#endif
    __syncthreads();
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicAdd(sat, 1u);
}

#ifdef NRV_EXPERIMENTS
__global__ void __launch_bounds__(kCnnH2Threads) cnn_h2_kernel(const CnnH2Args args) {
  constexpr int CH = kCnnH2CH, IMG = kCnnH2IMG;
  __shared__ __attribute__((aligned(16))) _Float16 img_s[2 * IMG];
  // The 264 conv constants, read as LDS broadcasts (from global memory they are 147 wave-wide loads of one
  // address each per thread and tile, every one of them a full 1 KB return through the vector-memory path).
  __shared__ __attribute__((aligned(16))) float cwl_s[264];
  lds_f16* const img = (lds_f16*)img_s;
  const lds_f32* const cwl = (const lds_f32*)cwl_s;

  const CnnH2ModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ntile = args.n_tiles, G = gridDim.x;
  const int nloc = (ntile - (int)blockIdx.x + G - 1) / G;     // tiles of this workgroup: b = blockIdx.x + i*G

  // the padding k-block (features 400..415) of both images is zero for the whole launch
  for (int i = threadIdx.x; i < 2 * 4 * CH; i += kCnnH2Threads)
    img_s[(i / (4 * CH)) * IMG + 25 * 4 * CH + i % (4 * CH)] = (_Float16)0.f;
  for (int i = threadIdx.x; i < 264; i += kCnnH2Threads) cwl_s[i] = P.conv[i];
  __syncthreads();

  if (wave >= kCnnH2MatWaves) {
    // ================================ CONV role ==============================================
    const int cwv = wave - kCnnH2MatWaves;
    const int r = lane & 31, hsel = lane >> 5;
    // chunks: waves 0-3 -> 3-wide 0..7 (positions 0..23); wave 4 -> 4-wide (24..31); waves 5-7 -> 3-wide (32..49)
    const int p0 = cwv < 4 ? 3 * (2 * cwv + hsel) : (cwv == 4 ? 24 + 4 * hsel : 32 + 3 * (2 * (cwv - 5) + hsel));
    float x[8];                                    // samples p0-2 .. p0+5 of the tile being convolved
    auto load_x = [&](int b, float (&xo)[8]) __attribute__((always_inline)) {
      const int bt = b < ntile ? b : 0;
      const int wt = bt / T, t = bt % T;
      const bool ok = b < ntile && wt * 32 + r < args.n_rows;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(args.signal + ((size_t)wt * 32 * T + t) * kSig, 0xffffffffu);
      const unsigned rowoff = ok ? (unsigned)(r * T * kSig) : 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = p0 - 2 + i;
        const int pc = p < 0 ? 0 : (p > kSig - 1 ? kSig - 1 : p);
        const float v = buf_load4(rs, (rowoff + (unsigned)pc) * 4, 0);
        xo[i] = (ok && p >= 0 && p < kSig) ? v : 0.f;
      }
    };
    load_x(blockIdx.x, x);
    // Two things hipcc's register allocation is sensitive to here (each alone turns 120-133 VGPRs into
    // > 168 + spills): the loop must not be of the form `for (i = 0; i <= nloc; ++i) { if (i < nloc) {...}
    // barrier }`, and the two chunk widths must not meet inside one loop body (`cwv == 4 ? <4> : <3>` in the
    // loop: both variants' registers are summed) - hence one whole tile loop per width.
    auto tile_loop = [&](auto np_tag) __attribute__((always_inline)) {
      constexpr int NP = decltype(np_tag)::value;
      for (int i = 0; i < nloc; ++i) {
        float xn[8];
        load_x(blockIdx.x + (i + 1) * G, xn);      // next tile's samples: a whole iteration of lead
        lds_f16* im = img + (i & 1) * IMG;
        int zoff = 0;                              // opaque zero: keeps LICM from hoisting the 264 reads out of the tile loop
        asm volatile("" : "+v"(zoff));
        const lds_f32* cw = cwl + zoff;
        conv_positions_h2<NP>(cw, x, p0, r, im);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = xn[k];
        __syncthreads();
      }
    };
    if (cwv == 4) tile_loop(std::integral_constant<int, 4>{});
    else tile_loop(std::integral_constant<int, 3>{});
    __syncthreads();                               // the matrix role's last tile
  } else {
    cnn_dense_role(P, img, nloc, G, wave, lane, args.sat);
  }
}

#endif

}  // namespace nrv
