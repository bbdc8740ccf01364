// First read-branch layer, Bi-LSTM(6 -> 16).
#pragma once
#include "nrv_common.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// First read-branch layer, Bi-LSTM(6 -> 16), on 16x16x4 MFMA tiles.
// With H = 16 the generic kernel above pads to 32 hidden units and is latency-bound (17 % matrix
// pipe busy).  Here one WAVE owns 16 rows x 16 units x 4 gates: the whole weight set of a
// direction (6+16 rows x 64 columns) lives in 24 VGPRs, h_t goes through a 1 KB wave-private LDS
// image (no barrier), and the float4 a lane reads back from it is at once the A fragment of the
// next recurrent product and its share of the BatchNorm'd output row.
// grid = (ceil(rows/64), 2 directions, 2 models), block = 256 (4 independent waves).
// ---------------------------------------------------------------------------------------
struct Lstm1ModelParams {
  const float* wpack;     // [dir][ (kstep 2 + s 4) ][gate 4][64 lanes]   (see pack_lstm1_16)
  const float* bias;      // [dir][gate][16]
  const float* bn_scale;  // [32]
  const float* bn_shift;  // [32]
  const float* plain_in;  // [n][T][6] (ev_stride 0) or [N][6] (ev_stride 1)
  int plain_ev_stride;
  float* out;             // tiled window-major, KQ = 8: [tile32][T][8][32][4]; OUT_SPLIT: the same chunks as
                          // f16 split planes [kb 2][term][half][32][8 f16] (nrv_lstm_f16x2.h), scale in bn_*
};
struct Lstm1Args {
  Lstm1ModelParams m[2];
  int T;
  int n_rows;
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// One wave's unit: 16 rows (block rb) of one direction.  hb: 272 floats of wave-private LDS.
template <int ACT, bool OUT_SPLIT>
__device__ __forceinline__ void lstm1_unit(const Lstm1ModelParams& P, const int T, const int n_rows, const int dir,
                                           const int rb, const int lane, float* hb) {
  const int q = lane >> 4, c = lane & 15;
  const int row = rb * 16 + c;                           // the row this lane feeds as A operand
  // weights of this direction, register-resident
  float win[2][4], wrec[4][4], bias[4];
  {
    const float* wp = P.wpack + (size_t)dir * 6 * 4 * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 4; ++g) win[s][g] = wp[(s * 4 + g) * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int g = 0; g < 4; ++g) wrec[s][g] = wp[((2 + s) * 4 + g) * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = P.bias[(dir * 4 + g) * 16 + c];
  }
  const f32x4 bsc = *(const f32x4*)(P.bn_scale + dir * 16 + 4 * q);
  const f32x4 bsh = *(const f32x4*)(P.bn_shift + dir * 16 + 4 * q);

  auto load_x = [&](int t, float& x0, float& x1) {
    x0 = 0.f; x1 = 0.f;
    if (row < n_rows) {
      const float* src = P.plain_in +
          (P.plain_ev_stride ? (size_t)(row + t) * kFeat : ((size_t)row * T + t) * kFeat);
      x0 = src[q];                                       // k = q        (k-step 0)
      if (q < 2) x1 = src[4 + q];                        // k = 4 + q    (k-step 1; k = 6,7 are padding)
    }
  };
  // f32 tiles: chunk kq = dir*4 + q holds units 4q..4q+3 of this direction.  Split planes: the 16
  // units of a direction are exactly k-block `dir`; units 4q..4q+3 are elements 4(q&1).. of half q>>1.
  float* out_base = OUT_SPLIT
      ? P.out + ((size_t)(rb >> 1) * T * 8 + dir * 4 + (q >> 1)) * 128 + (16 * (rb & 1) + c) * 4 + (q & 1) * 2
      : P.out + ((size_t)(rb >> 1) * T * 8 + dir * 4 + q) * 128 + (16 * (rb & 1) + c) * 4;

  f32x4 cc = {0.f, 0.f, 0.f, 0.f};                       // cell state of (rows 4q+reg, unit c)
  f32x4 hprev = {0.f, 0.f, 0.f, 0.f};                    // h_{t-1}[row c][units 4q..4q+3]

  float x0, x1;
  load_x(dir ? T - 1 : 0, x0, x1);
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    float nx0 = 0.f, nx1 = 0.f;
    if (s + 1 < T) load_x(dir ? t - 1 : t + 1, nx0, nx1);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{bias[g], bias[g], bias[g], bias[g]};
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, win[0][g], acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, win[1][g], acc[g], 0, 0, 0);
    if (s > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hprev[k], wrec[k][g], acc[g], 0, 0, 0);
    }
    // gates: lane holds unit c for rows 4q + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float ig = gate_act<ACT>(acc[0][reg]);
      const float fg = gate_act<ACT>(acc[1][reg]);
      const float gg = tanh_fast(acc[2][reg]);
      const float og = gate_act<ACT>(acc[3][reg]);
      const float cn = __builtin_fmaf(fg, cc[reg], ig * gg);
      cc[reg] = cn;
      hb[(4 * q + reg) * 16 + c] = og * tanh_fast(cn);
    }
    // h_t[row c][4q..4q+3]: next step's A fragment (k-step k uses unit 4q+k) AND this lane's output
    wave_lds_fence();                                     // other lanes' stores above -> this lane's load
    hprev = *(const f32x4*)(hb + c * 16 + 4 * q);
    wave_lds_fence();                                     // ... and this load before the next step's stores
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = hprev[k] * bsc[k] + bsh[k];
    if constexpr (OUT_SPLIT) {
      f16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (_Float16)o[k];
        lo[k] = (_Float16)(o[k] - (float)hi[k]);
      }
      float* d = out_base + (size_t)t * 8 * 128;
      *(f16x4*)d = hi;                                    // term 0 (hi): chunk 4*kb + half
      *(f16x4*)(d + 2 * 128) = lo;                        // term 1 (lo): chunk 4*kb + 2 + half
    } else {
      *(f32x4*)(out_base + (size_t)t * 8 * 128) = o;
    }
    x0 = nx0; x1 = nx1;
  }
}

template <int ACT, bool OUT_SPLIT = false>
__global__ void __launch_bounds__(256) lstm1_kernel(const Lstm1Args args) {
  __shared__ __attribute__((aligned(16))) float hbuf[4][16 * 16 + 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lstm1_unit<ACT, OUT_SPLIT>(args.m[blockIdx.z], args.T, args.n_rows, blockIdx.y, blockIdx.x * 4 + wave, lane, hbuf[wave]);
}


}  // namespace nrv
