// Head: per-timestep MLP (f32 and bf16x3 forms) and the per-window tail.
#pragma once
#include "nrv_lstm_bf16x3.h"   // Split3, split3, mfma_bf16

namespace nrv {

// ---------------------------------------------------------------------------------------
// Head: per timestep Dense(128,relu) -> Dense(32,relu) -> Dense(6,relu); Flatten(6T);
// Dense(16,relu); Dense(C,softmax); argmax.     output_handeler.py:230-237 / :282-289
// Workgroup = one 32-row tile; wave w runs timesteps t = w, w+4, ... through the three
// per-timestep layers on MFMA with a wave-private LDS image, then the block finishes the
// tiny per-window layers on the VALU.
// ---------------------------------------------------------------------------------------
struct HeadModelParams {
  const float* d1pack;    // [nt 4][kg 16][64][4]
  const float* d1bias;    // [128]
  const float* d2pack;    // [kg 16][64][4]
  const float* d2bias;    // [32]
  const float* mopack;    // [kg 4][64][4]   (6 columns padded to 32 with zeros)
  const float* mobias;    // [32] (padded with zeros)
  const float* featw;     // [6T][16]
  const float* featb;     // [16]
  const float* outw;      // [16][C]
  const float* outb;      // [C]
  const float* in;        // LSTM4 output, tiled window-major KQ=32
  float* mo;              // main_out scratch [tile][t][32 rows][8]  (6 used)
  float* prob;            // [n][C]
  int8_t* argmax;         // [n]
  int n_class;
};
struct HeadArgs {
  HeadModelParams m[2];
  int T;
  int n_rows;
};

constexpr int kHeadMaxT = 32;

// Stage 1: one WAVE per (row tile, timestep, model): three chained per-timestep layers on MFMA,
// intermediate activations through a wave-private LDS image (no workgroup barrier at all), weights
// streamed in B-fragment order with a two-group register ring.  grid = (tiles*T, 2), block = 64.
__global__ void __launch_bounds__(64) head_mlp_kernel(const HeadArgs args) {
  constexpr int PLANE = 32 * 4 + 4;
  __shared__ __attribute__((aligned(16))) float im[32 * PLANE];      // up to 128 features
  const HeadModelParams& P = args.m[blockIdx.y];
  const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
  const int bt = blockIdx.x;                                          // tile*T + t

  // dense1: 128 -> 128, A straight from the tiled LSTM4 output
  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = splat16(P.d1bias[nt * 32 + l31]);
  {
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(P.in + (size_t)bt * 32 * 128, 32 * 128 * 4);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(P.d1pack, 4 * 16 * 256 * 4);
    const unsigned av = (half * 128 + l31 * 4) * 4, wv = lane * 16;
    f32x4 a[3], b[3][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = buf_load16(ars, av, i * 1024);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) b[i][nt] = buf_load16(wrs, wv, (nt * 16 + i) * 1024);
    }
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      if (kg + 2 < 16) {
        a[(kg + 2) % 3] = buf_load16(ars, av, (kg + 2) * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[(kg + 2) % 3][nt] = buf_load16(wrs, wv, (nt * 16 + kg + 2) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma32(a[kg % 3][j], b[kg % 3][nt][j], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // first weights of the next two layers, requested before the LDS round trip
  f32x4 w2[16], w3[4];
#pragma unroll
  for (int kg = 0; kg < 16; ++kg) w2[kg] = *(const f32x4*)(P.d2pack + kg * 256 + lane * 4);
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) w3[kg] = *(const f32x4*)(P.mopack + kg * 256 + lane * 4);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int u = nt * 32 + l31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      im[(u >> 2) * PLANE + acc_row(reg, lane) * 4 + (u & 3)] = relu_nan(acc[nt][reg]);
  }
  // dense2: 128 -> 32 (the same wave wrote the image; DS operations of one wave complete in order)
  wave_lds_fence();
  f32x16 a2 = splat16(P.d2bias[l31]);
  {
    const float* hp = im + half * PLANE + l31 * 4;
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      const f32x4 a = *(const f32x4*)(hp + kg * 2 * PLANE);
#pragma unroll
      for (int j = 0; j < 4; ++j) a2 = mfma32(a[j], w2[kg][j], a2);
    }
  }
  wave_lds_fence();                                       // dense2's reads of the image before it is overwritten
#pragma unroll
  for (int reg = 0; reg < 16; ++reg)
    im[(l31 >> 2) * PLANE + acc_row(reg, lane) * 4 + (l31 & 3)] = relu_nan(a2[reg]);
  wave_lds_fence();
  // main_out: 32 -> 6 (padded to 32 columns)
  f32x16 a3 = splat16(P.mobias[l31]);
  {
    const float* hp = im + half * PLANE + l31 * 4;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 a = *(const f32x4*)(hp + kg * 2 * PLANE);
#pragma unroll
      for (int j = 0; j < 4; ++j) a3 = mfma32(a[j], w3[kg][j], a3);
    }
  }
  if (l31 < 8) {
    float* dst = P.mo + (size_t)bt * 32 * 8 + l31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) dst[acc_row(reg, lane) * 8] = relu_nan(a3[reg]);
  }
}

// Stage 1, bf16x3 form (nrv_set_precision): the same three layers on v_mfma_f32_32x32x16_bf16 with
// the exact three-term split of lstm_split_kernel, computed TRANSPOSED so that the chain never leaves
// the registers:  Out^T[n][row] = sum_k W[k][n] * X[row][k]  makes the weights the A operand and the
// activations the B operand (lane = data row), and the C layout of one layer - lane (row, half h)
// holds features (reg&3) + 8*(reg>>2) + 4h - is, eight registers at a time, exactly a B operand of
// the next layer once the host packs that layer's weights in the same permuted k order.  No LDS
// image, no barrier in the loop.  The split weights of all three layers (126 KB) are staged once per
// workgroup in LDS and shared by its four waves; a wave walks over (row tile, timestep) units.
// grid = (min(units, 128), 2 models), block = 256.
struct HeadSplitModelParams {
  const void* wsplit;     // [126 fragments][64 lanes][8 bf16]: dense1 [mt 4][kb 8][term 3], dense2 [kb 8][term 3], main_out [kb 2][term 3]
  const float* bias;      // [128 | 32 | 32] (main_out padded with zeros)
  const float* in;        // LSTM4 output, tiled window-major KQ=32
  float* mo;              // main_out scratch [unit][32 rows][8]
};
struct HeadSplitArgs {
  HeadSplitModelParams m[2];
  int n_units;            // tiles * T
};

__global__ void __launch_bounds__(256) head_mlp_split_kernel(const HeadSplitArgs args) {
  constexpr int NFRAG = 126;
  __shared__ __attribute__((aligned(16))) unsigned short wl[NFRAG * 512];
  __shared__ __attribute__((aligned(16))) float bl[192];
  const HeadSplitModelParams& P = args.m[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    // 126 KB, 8 x 16 B in flight per thread (a plain copy loop waits out every load)
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(P.wsplit, NFRAG * 1024);
    for (int base = 0; base < NFRAG * 64; base += 8 * 256) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load16(srs, (unsigned)(base + j * 256 + tid) * 16, 0);   // out of range -> 0
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (base + j * 256 + tid < NFRAG * 64) ((f32x4*)wl)[base + j * 256 + tid] = v[j];
    }
  }
  if (tid < 192) bl[tid] = P.bias[tid];
  __syncthreads();

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // (weight term, activation term)
  auto frag = [&](int f) __attribute__((always_inline)) {
    return *(const bf16x8*)(wl + f * 512 + lane * 8);
  };
  auto bias_tile = [&](int off) __attribute__((always_inline)) {       // C-layout bias of 32 features at off
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(const f32x4*)(bl + off + 8 * q + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[4 * q + j] = v[j];
    }
    return z;
  };
  auto relu8 = [&](const f32x16& z, int base, f32x4& lo, f32x4& hi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] = relu_nan(z[base + j]);
      hi[j] = relu_nan(z[base + 4 + j]);
    }
  };
  const unsigned av = l31 * 16 + half * 1024;
  auto load_x = [&](int u, f32x4 (&x)[8][2]) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(P.in + (size_t)u * 32 * 128, 32 * 128 * 4);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      x[kb][0] = buf_load16(ars, av, kb * 2048);
      x[kb][1] = buf_load16(ars, av, kb * 2048 + 512);
    }
  };

  const int stride = gridDim.x * 4;
  int u = blockIdx.x * 4 + wave;
  if (u >= args.n_units) return;
  f32x4 x[8][2];
  load_x(u, x);
  for (; u < args.n_units; u += stride) {
    // dense1: 128 -> 128 (four 32-feature tiles)
    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = bias_tile(mt * 32);
    Split3 S[2];
    S[0] = split3(x[0][0], x[0][1]);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      bf16x8 w[4][3];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) w[mt][tm] = frag((mt * 8 + kb) * 3 + tm);
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < 8) S[(kb + 1) & 1] = split3(x[kb + 1][0], x[kb + 1][1]);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(w[mt][PA[pr]], S[kb & 1].t[PB[pr]], acc[mt]);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next unit's inputs travel while the two small layers run
    const int un = u + stride;
    if (un < args.n_units) load_x(un, x);
    // dense2: 128 -> 32; k-block kb takes registers 8*(kb&1).. of tile kb>>1 (two accumulators: the
    // 48 products would otherwise form one dependent chain)
    f32x16 a2[2];
    a2[0] = bias_tile(128);
    a2[1] = splat16(0.f);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      f32x4 lo, hi;
      relu8(acc[kb >> 1], (kb & 1) * 8, lo, hi);
      const Split3 s2 = split3(lo, hi);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
        a2[kb & 1] = mfma_bf16(frag(96 + kb * 3 + PA[pr]), s2.t[PB[pr]], a2[kb & 1]);
    }
    f32x16 h2;
#pragma unroll
    for (int i = 0; i < 16; ++i) h2[i] = a2[0][i] + a2[1][i];
    // main_out: 32 -> 6 (padded to 32)
    f32x16 a3 = bias_tile(160);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 lo, hi;
      relu8(h2, kb * 8, lo, hi);
      const Split3 s3 = split3(lo, hi);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) a3 = mfma_bf16(frag(120 + kb * 3 + PA[pr]), s3.t[PB[pr]], a3);
    }
    // lane (row, half h) holds output features 4h..4h+3 in registers 0..3
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = relu_nan(a3[j]);
    *(f32x4*)(P.mo + ((size_t)u * 32 + l31) * 8 + 4 * half) = o;
  }
}

// Stage 2: Flatten(6T) -> Dense(16,relu) -> Dense(C,softmax) -> argmax, one row tile per workgroup.
// 96T+96 MAC per window: VALU, operands as 16-byte LDS reads.  grid = (tiles, 2), block = 256.
__global__ void __launch_bounds__(256) head_final_kernel(const HeadArgs args) {
  constexpr int KPMAX = 6 * kHeadMaxT;           // 192, multiple of 4
  constexpr int FSM = KPMAX + 4;                 // row stride (floats): 16-byte aligned, bank-skewed
  __shared__ __attribute__((aligned(16))) float flatv[32 * FSM];
  __shared__ __attribute__((aligned(16))) float fwT[16 * FSM];    // feature kernel, transposed [f][k]
  __shared__ float featv[32 * 17];
  __shared__ float logit[32 * 8];
  const HeadModelParams& P = args.m[blockIdx.y];
  const int T = args.T, tile = blockIdx.x, tid = threadIdx.x;
  const int K = 6 * T, KP = (K + 3) & ~3;
  for (int i = tid; i < 16 * KP; i += 256) {
    const int f = i / KP, k = i % KP;
    fwT[f * FSM + k] = k < K ? P.featw[k * 16 + f] : 0.f;
  }
  for (int i = tid; i < 32 * (KP - K); i += 256) flatv[(i / (KP - K)) * FSM + K + i % (KP - K)] = 0.f;
  for (int i = tid; i < T * 32 * 8; i += 256) {                    // coalesced read of [t][row][8]
    const int k = i & 7, r = (i >> 3) & 31, t = i >> 8;
    const float v = P.mo[(size_t)(tile * T) * 256 + i];
    if (k < 6) flatv[r * FSM + t * 6 + k] = v;
  }
  __syncthreads();
  for (int it = tid; it < 32 * 16; it += 256) {
    const int r = it >> 4, f = it & 15;
    float v = P.featb[f];
    const f32x4* fr = (const f32x4*)(flatv + r * FSM);
    const f32x4* fw = (const f32x4*)(fwT + f * FSM);
#pragma unroll 4
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      const f32x4 x = fr[k4], w = fw[k4];
      v = __builtin_fmaf(x[0], w[0], v);
      v = __builtin_fmaf(x[1], w[1], v);
      v = __builtin_fmaf(x[2], w[2], v);
      v = __builtin_fmaf(x[3], w[3], v);
    }
    featv[r * 17 + f] = relu_nan(v);
  }
  __syncthreads();
  const int C = P.n_class;
  {
    const int r = tid >> 3, cc = tid & 7;
    if (cc < C) {
      float v = P.outb[cc];
#pragma unroll
      for (int f = 0; f < 16; ++f) v = __builtin_fmaf(featv[r * 17 + f], P.outw[f * C + cc], v);
      logit[r * 8 + cc] = v;
    }
  }
  __syncthreads();
  if (tid < 32) {
    const int row = tile * 32 + tid;
    if (row < args.n_rows) {
      float mx = logit[tid * 8];
      for (int cc = 1; cc < C; ++cc) mx = __builtin_fmaxf(mx, logit[tid * 8 + cc]);
      float e[8], sum = 0.f;
      for (int cc = 0; cc < C; ++cc) { e[cc] = expf(logit[tid * 8 + cc] - mx); sum += e[cc]; }
      int best = 0; float bv = -1.f;
      for (int cc = 0; cc < C; ++cc) {
        const float p = e[cc] / sum;
        P.prob[(size_t)row * C + cc] = p;
        if (p > bv) { bv = p; best = cc; }     // strict > : ties -> lowest index
      }
      P.argmax[row] = (int8_t)best;
    }
  }
}


}  // namespace nrv
