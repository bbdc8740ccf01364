// Signal branch: conv1d x2 + BN + residual + flatten + dense(400 -> 64).
#pragma once
#include "nrv_lstm_bf16x3.h"   // Split3, split3, mfma_bf16

namespace nrv {

// ---------------------------------------------------------------------------------------
// Signal branch: conv1d(1->8,k3)+ReLU+BN, conv1d(8->8,k3)+ReLU+BN, + signal, flatten(400),
// dense(400->64).  nanorevcnn.py:17-38, output_handeler.py:209-215.
//
// Persistent, wave-specialised workgroups (one per CU): four CONV waves turn the next 32-event tile
// into its 400-feature A-fragment image in LDS on the VALU (f32 VALU rate == f32 MFMA rate on
// gfx950, and N=8 would waste 3/4 of a matrix tile) while four MATRIX waves run the 400->64 dense
// of the previous tile out of the other image buffer, their share of the dense kernel resident in
// registers; one barrier per tile, no other synchronisation.
// Measured: conv alone 59 us, dense alone 49 us, together 90 us per 4096-window group - f32 MFMA
// and f32 VALU work of co-resident waves add up rather than overlap (they price against the same
// 64 FLOP/clk/SIMD), so what this structure buys is the removed barriers/staging (113 -> 90 us).
// ---------------------------------------------------------------------------------------
struct CnnModelParams {
  const float* conv;      // 24 w1[k][o], 8 b1, 8 s1, 8 h1, 192 w2[k][c][o], 8 b2, 8 s2, 8 h2  (=264)
  const float* dpack;     // dense 400->64 packed for 16x16x4 MFMA: [ct 4][kg 25][64][4]
  const float* dbias;     // [64]
  const void* dsplit;     // dense 400->64 as split-bf16 B fragments: [kb 25][nh 2][term 3][64 lanes][8 bf16]
  float* out;             // tiled, KQ=16: window-major [wtile][t][16][32][4] or event-major [etile][16][32][4]
};
struct CnnArgs {
  CnnModelParams m[2];
  const float* signal;    // [n][T][50] (window mode) or [N][50] (event mode)
  int T;                  // window mode: T; event mode: 1
  int n_rows;             // windows (window mode) or events (event mode)
  int n_tiles;            // 32-event tiles to process (per model)
};

constexpr int kCnnMatWaves = 4;
constexpr int kCnnConvWaves = 4;   // 32 events x 8 position chunks (7,7,6,6,6,6,6,6) = 256 threads
constexpr int kCnnThreads = 64 * (kCnnMatWaves + kCnnConvWaves);

typedef float f32x2 __attribute__((ext_vector_type(2)));

// conv1+BN -> conv2+BN -> +signal for NP consecutive positions of one event, written into the
// A-fragment image.  x[] holds samples p0-2 .. p0+NP+1.
template <int NP>
__device__ __forceinline__ void conv_positions(const float* __restrict__ cw, const float (&x)[11], int p0,
                                               int r, float* flat, int plane) {
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
        v = v * w1[32 + o] + w1[40 + o];
        b1v[q][o] = inside ? v : 0.f;
      }
    }
  }
  // conv2, weight-stationary: each (tap k, in-channel ci) row of 8 weights is fetched once and
  // applied to all NP positions of this thread as 4 packed FMAs each.
  const float* w2 = cw + 48;
  f32x2 acc[NP][4];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[q][o] = f32x2{w2[192 + 2 * o], w2[192 + 2 * o + 1]};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
      f32x2 wrow[4];
#pragma unroll
      for (int o = 0; o < 4; ++o)
        wrow[o] = f32x2{w2[(k * 8 + ci) * 8 + 2 * o], w2[(k * 8 + ci) * 8 + 2 * o + 1]};
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const float av = b1v[q + k][ci];
        const f32x2 a2 = f32x2{av, av};
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[q][o] = __builtin_elementwise_fma(a2, wrow[o], acc[q][o]);
      }
    }
  float s2[8], h2[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) { s2[o] = w2[200 + o]; h2[o] = w2[208 + o]; }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float xc = x[q + 2];
    float o8[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float v = __builtin_fmaxf(acc[q][o >> 1][o & 1], 0.f);
      v = v * s2[o] + h2[o];
      o8[o] = v + xc;                              // Add(): broadcast the raw signal over channels
    }
    const int p = p0 + q;                          // flat index p*8+o -> kq = 2p, 2p+1
    *(f32x4*)(flat + (2 * p) * plane + r * 4) = f32x4{o8[0], o8[1], o8[2], o8[3]};
    *(f32x4*)(flat + (2 * p + 1) * plane + r * 4) = f32x4{o8[4], o8[5], o8[6], o8[7]};
  }
}

// SPLIT (bf16x3 mode): the dense layer runs on v_mfma_f32_32x32x16_bf16 with the exact three-term
// split.  The bf16 pipe overlaps with the conv waves' VALU work where the f32 one adds to it, and
// the four matrix waves are arranged 2 column halves x 2 K halves so that no activation is split
// more than twice: wave (nh, kh) holds its 13 (12) k-blocks of split weights in registers, the kh = 1
// wave hands its partial tile to its partner through a double-buffered 4 KB LDS slab, and the
// partner adds it one iteration later (behind the tile barrier that exists anyway).
// The f16x2 mode has its own signal-branch kernel (cnn_r_kernel, nrv_cnn_r.h).
template <bool SPLIT>
__global__ void __launch_bounds__(kCnnThreads) cnn_kernel(const CnnArgs args) {
  constexpr int PLANE = 32 * 4 + 4;         // floats per kq plane of the image (+4: conflict-free)
  constexpr int IMG = 100 * PLANE;
  __shared__ __attribute__((aligned(16))) float img[2 * IMG];
  __shared__ __attribute__((aligned(16))) float part[SPLIT ? 2 * 2 * 16 * 64 : 4];   // [buf][nh][reg][lane]
  // The 264 conv constants, read as LDS broadcasts.  From global memory every conv thread needs all of
  // them for every tile: 172 wave-wide loads of ONE address each, and each of them occupies the
  // vector-memory return path like any other 1 KB load (round 2: that path, not the FMAs, bounded the
  // convolution; a 12-wave variant with twice the conv threads ran 127 us instead of 91 for that reason).
  __shared__ __attribute__((aligned(16))) float cwl[264];

  const CnnModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ntile = args.n_tiles, G = gridDim.x;
  const int nloc = (ntile - (int)blockIdx.x + G - 1) / G;     // tiles of this workgroup: b = blockIdx.x + i*G

  for (int i = threadIdx.x; i < 264; i += kCnnThreads) cwl[i] = P.conv[i];
  __syncthreads();

  if (wave >= kCnnMatWaves) {
    // ================================ CONV role ==============================================
    // The conv waves are the critical path of a tile (the matrix waves finish early and wait at the
    // barrier): they win the VALU arbitration against the matrix wave of their SIMD.
    __builtin_amdgcn_s_setprio(3);
    // wave cw0 holds the two 7-position chunks, the others 6-position chunks (wave-uniform count)
    const int cwv = wave - kCnnMatWaves;
    const int chunk = cwv * 2 + (lane >> 5);
    const int r = lane & 31;
    const int p0 = chunk < 2 ? 7 * chunk : 14 + 6 * (chunk - 2);
    float x[11];                                   // samples p0-2 .. p0+8 of the tile being convolved
    // One buffer resource per tile (64-bit base wave-uniform) + one 32-bit lane offset + immediates:
    // eleven independent loads, no per-sample 64-bit address and no branch.  Out-of-window samples
    // and rows past the end are clamped to a valid address and masked to zero.  (Per-sample
    // predicated loads spilled an address pair, and the reload's vmcnt(0) serialised the loads.)
    auto load_x = [&](int b, float (&xo)[11]) {
      const int bt = b < ntile ? b : 0;
      const int wt = bt / T, t = bt % T;
      const bool ok = b < ntile && wt * 32 + r < args.n_rows;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(args.signal + ((size_t)wt * 32 * T + t) * kSig, 0xffffffffu);
      const unsigned rowoff = ok ? (unsigned)(r * T * kSig) : 0u;
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        const int p = p0 - 2 + i;
        const int pc = p < 0 ? 0 : (p > kSig - 1 ? kSig - 1 : p);
        const float v = buf_load4(rs, (rowoff + (unsigned)pc) * 4, 0);
        xo[i] = (ok && p >= 0 && p < kSig) ? v : 0.f;
      }
    };
    load_x(blockIdx.x, x);
    for (int i = 0; i <= nloc; ++i) {
      // iteration i builds the image of local tile i (the matrix waves consume tile i-1)
      if (i < nloc) {
        float xn[11];
        load_x(blockIdx.x + (i + 1) * G, xn);        // next tile's samples: a whole iteration of lead
        float* flat = img + (i & 1) * IMG;
        int zoff = 0;                                // opaque zero: keeps LICM from hoisting the 264 reads out of the tile loop
        asm volatile("" : "+v"(zoff));
        const float* cw = cwl + zoff;
        if (cwv == 0) conv_positions<7>(cw, x, p0, r, flat, PLANE);
        else conv_positions<6>(cw, x, p0, r, flat, PLANE);
#pragma unroll
        for (int k = 0; k < 11; ++k) x[k] = xn[k];
      }
      __syncthreads();
    }
  } else {
    // ================================ MATRIX role ============================================
    if constexpr (SPLIT) {
      const int nh = wave & 1, kh = wave >> 1;
      const int half = lane >> 5, l31 = lane & 31;
      constexpr int NKB = 13;                                   // k-blocks of 16: kh = 0 takes 0..12, kh = 1 13..24
      const int kb0 = kh * NKB, nkb = kh ? 12 : 13;
      bf16x8 bw[NKB][3];
#pragma unroll
      for (int k = 0; k < NKB; ++k)
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
          const int kb = kb0 + (k < nkb ? k : 0);                 // (the unused 13th block of kh = 1 is never multiplied)
          bw[k][tm] = *(const bf16x8*)((const char*)P.dsplit + ((size_t)((kb * 2 + nh) * 3 + tm) * 64 + lane) * 16);
        }
      const float bias = kh == 0 ? P.dbias[nh * 32 + l31] : 0.f;
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      f32x16 prev = splat16(0.f);                                // kh = 0: finished K-half of the previous tile
      auto finish = [&](int i_tile) __attribute__((always_inline)) {   // kh = 0: tile i_tile = prev + partner's half
        const float* ps = part + ((i_tile & 1) * 2 + nh) * 1024 + lane;
        float* dst = P.out + (size_t)(blockIdx.x + i_tile * G) * 16 * 128;
        const int u = nh * 32 + l31;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
          dst[(u >> 2) * 128 + row * 4 + (u & 3)] = prev[reg] + ps[reg * 64];
        }
      };
      for (int i = 0; i <= nloc; ++i) {
        if (kh == 0 && i >= 2) finish(i - 2);
        if (i > 0) {
          const float* flat = img + ((i - 1) & 1) * IMG;
          const float* ap = flat + (4 * kb0 + 2 * half) * PLANE + l31 * 4;
          f32x16 acc = splat16(bias);
          f32x4 a[2][2];
          a[0][0] = *(const f32x4*)(ap);
          a[0][1] = *(const f32x4*)(ap + PLANE);
#pragma unroll
          for (int k = 0; k < NKB; ++k) {
            if (k < nkb) {                                         // wave-uniform
              if (k + 1 < NKB) {
                const int kn = (k + 1 < nkb) ? k + 1 : k;
                a[(k + 1) & 1][0] = *(const f32x4*)(ap + kn * 4 * PLANE);
                a[(k + 1) & 1][1] = *(const f32x4*)(ap + kn * 4 * PLANE + PLANE);
              }
              const Split3 sa = split3(a[k & 1][0], a[k & 1][1]);
#pragma unroll
              for (int pr = 0; pr < 6; ++pr) acc = mfma_bf16(sa.t[PA[pr]], bw[k][PB[pr]], acc);
            }
          }
          if (kh == 0) {
            prev = acc;
          } else {
            float* ps = part + (((i - 1) & 1) * 2 + nh) * 1024 + lane;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) ps[reg * 64] = acc[reg];
          }
        }
        __syncthreads();
      }
      if (kh == 0 && nloc >= 1) finish(nloc - 1);
    } else {
    // wave w owns 16 output features (column tile w) for all 32 rows (two 16-row accumulators).
    // Its share of the 400x64 kernel (25 k-groups x 4 VGPRs) stays in registers for the whole
    // launch, so a tile costs 50 LDS reads + 200 v_mfma_f32_16x16x4_f32 and no weight traffic.
    const int ct = wave;
    const int q4 = lane >> 4, r16 = lane & 15;
    f32x4 bw[25];
#pragma unroll
    for (int kg = 0; kg < 25; ++kg) bw[kg] = *(const f32x4*)(P.dpack + ((size_t)ct * 25 + kg) * 256 + lane * 4);
    const float bias = P.dbias[ct * 16 + r16];
    for (int i = 0; i <= nloc; ++i) {
      if (i > 0) {
        const float* flat = img + ((i - 1) & 1) * IMG;
        const int b = blockIdx.x + (i - 1) * G;
        f32x4 acc0 = {bias, bias, bias, bias}, acc1 = acc0;
        const float* ap = flat + q4 * PLANE + r16 * 4;
        f32x4 a0[2], a1[2];
        a0[0] = *(const f32x4*)(ap);
        a1[0] = *(const f32x4*)(ap + 64);
#pragma unroll
        for (int kg = 0; kg < 25; ++kg) {
          const int cur = kg & 1;
          if (kg + 1 < 25) {
            a0[cur ^ 1] = *(const f32x4*)(ap + (kg + 1) * 4 * PLANE);
            a1[cur ^ 1] = *(const f32x4*)(ap + (kg + 1) * 4 * PLANE + 64);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cur][j], bw[kg][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cur][j], bw[kg][j], acc1, 0, 0, 0);
          }
        }
        float* dst = P.out + (size_t)b * 16 * 128;
        const int u = ct * 16 + r16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = 4 * q4 + reg;
          dst[(u >> 2) * 128 + row * 4 + (u & 3)] = acc0[reg];
          dst[(u >> 2) * 128 + (row + 16) * 4 + (u & 3)] = acc1[reg];
        }
      }
      __syncthreads();
    }
    }
  }
}


}  // namespace nrv
