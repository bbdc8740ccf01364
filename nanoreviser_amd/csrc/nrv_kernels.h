// nrv_kernels.h - gfx950 (MI355X / CDNA4) device kernels of the NanoReviser window reviser.
//
// What is computed: the two Keras graphs of the reference,
//   nanorevutils/output_handeler.py:206-255 (model1) and :258-307 (model2),
//   CNN block nanorevutils/nanorevcnn.py:17-38,
// with Keras-2.2.4 inference semantics.  Everything is IEEE f32.
//
// How it is mapped (MI355X-first, not a translation of TF ops):
//   * Every contraction runs on the exact-f32 matrix pipe, v_mfma_f32_32x32x2_f32
//     (bitwise an fmaf chain, 64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak).
//   * Activations between kernels live in an MFMA-native tiled layout
//         act[tile32][t][kq][32 rows][4]      (kq = feature/4)
//     so that one wave-wide 16-byte load IS the A fragment of four MFMA k-steps
//     (lanes 0-31: features 8g..8g+3 of rows 0..31, lanes 32-63: features 8g+4..8g+7) and is a
//     single contiguous 1 KiB request.  Weights are pre-packed on the host into the matching
//     B-fragment order, so B operands stream L2 -> VGPR with no LDS staging at all.
//   * One Bi-LSTM layer = one launch; a wave owns 32 hidden units x 4 gates x R row tiles, so
//     i,f,g,o of one (window, unit) sit in the same lane/register and the cell update is
//     register-local; c never leaves registers, h_t goes through a double-buffered LDS image
//     (one barrier per step) and is written out coalesced with the following BatchNorm fused.
//   * Rows (windows) are independent, so a launch is (row tiles) x (2 directions) x (2 models)
//     workgroups with no inter-workgroup communication.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSig = 50;        // samples per event window   (output_handeler.py:202)
constexpr int kFeat = 6;        // features per event         (output_handeler.py:203)
constexpr int kTile = 32;       // rows per MFMA tile

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v;
  return r;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register `reg` for this lane (C/D map of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// Keras 2.2.4 `hard_sigmoid`: clip(0.2 x + 0.5, 0, 1)  
__device__ __forceinline__ float hard_sigmoid(float x) {
  return __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(x, 0.2f, 0.5f), 0.0f), 1.0f);
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int ACT>
__device__ __forceinline__ float gate_act(float x) {
  if constexpr (ACT == 0) return hard_sigmoid(x);
  else return sigmoid_exact(x);
}

// Address (in floats) of the 16-byte chunk (row, t, kq) of a tiled activation buffer.
//   window-major  : ev_stride = 0, tt = T, tm = 1  -> [row/32][t][kq][row%32][4]
//   event-major   : ev_stride = 1, tt = 1, tm = 0  -> [e/32][kq][e%32][4] with e = row + t
struct ActView {
  const float* p;
  int kq_total;     // KQ of the buffer
  int ev_stride;    // 0 window-major, 1 event-major (row index advances with t)
  int tt;           // T (window-major) or 1
  int tm;           // 1 (window-major) or 0
  __device__ __forceinline__ const float* chunk(int row, int t, int kq) const {
    int e = row + t * ev_stride;
    long off = ((long)((e >> 5) * tt + t * tm) * kq_total + kq) * 128 + (e & 31) * 4;
    return p + off;
  }
};

// ---------------------------------------------------------------------------------------
// Bi-LSTM layer kernel
// ---------------------------------------------------------------------------------------
// tanh, branch-free.  |x| < 0.625: x + x^3 Q(x^2) (own degree-4 fit, rel. err 1.1e-7 in f32);
// otherwise 1 - 2/(2^(2|x| log2 e) + 1) on the hardware exp2/rcp (1 ulp each).
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = __builtin_fabsf(x);
  const float u = x * x;
  float q = -0.005731194745749235f;
  q = __builtin_fmaf(q, u, 0.020664723590016365f);
  q = __builtin_fmaf(q, u, -0.053748443722724915f);
  q = __builtin_fmaf(q, u, 0.13331560790538788f);
  q = __builtin_fmaf(q, u, -0.3333328664302826f);
  const float small = __builtin_fmaf(ax * u, q, ax);
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);   // exp(2|x|); +inf is fine
  const float big = __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
  return __builtin_copysignf(ax < 0.625f ? small : big, x);
}

struct LstmModelParams {
  // packed [dir][hg][kg][gate][64][4]; kg runs over input k-groups then recurrent k-groups
  const float* wpack;
  const float* bias;      // [dir][hg][gate][32]
  const float* bn_scale;  // [2H] (1 / 0 arrays when the layer has no BatchNorm after it)
  const float* bn_shift;  // [2H]
  ActView in0;            // first input segment (tiled)  - unused when PLAIN_IN
  ActView in1;            // second input segment (tiled) - only when KQ1 > 0
  const float* plain_in;  // PLAIN_IN: [n][T][6] (ev_stride 0) or [N][6] (ev_stride 1)
  int plain_ev_stride;
  float* out;             // tiled window-major [tiles][T][2H/4][32][4]
};

struct LstmArgs {
  LstmModelParams m[2];
  int T;
  int n_rows;             // valid rows (windows)
};

// KQ0/KQ1: input segments in 4-feature chunks (K = 4*KQ, K multiple of 8).  H: hidden units per
// direction, NG = ceil(H/32) hidden groups.  A wave owns one hidden group (32 units x 4 gates) for
// R row tiles; a workgroup is NG x WR waves covering 32*R*WR rows.
// grid = (ceil(tiles/(R*WR)), 2 directions, 2 models), block = 64*NG*WR.
//
// Per step the k-groups run as ONE software pipeline, recurrent groups first (A from the LDS image
// of h_{t-1}), then the input groups (A from the tiled activations in global/L2); the fragments
// of group g+1 are requested before the 16*R MFMAs of group g issue, so the L2 latency of the
// weight stream hides under ~1000*R cycles of matrix work even at one wave per SIMD.
template <int KQ0, int KQ1, int H, int R, int WR, bool PLAIN_IN, int ACT>
__global__ void __launch_bounds__(64 * ((H + 31) / 32) * WR)
lstm_layer_kernel(const LstmArgs args) {
  constexpr int NG = (H + 31) / 32;
  constexpr int HP = NG * 32;
  constexpr int KG_IN = PLAIN_IN ? 1 : (KQ0 + KQ1) / 2;
  constexpr int KG_REC = H / 8;
  constexpr int KG = KG_IN + KG_REC;
  constexpr int ROWS = 32 * R * WR;
  constexpr int PLANE = ROWS * 4 + 4;          // floats per kq plane (+4 pad: conflict-free writes)
  constexpr int HBUF = (HP / 4) * PLANE;       // floats per h buffer
  constexpr int NTHREADS = 64 * NG * WR;
  static_assert(H % 8 == 0, "H must be a multiple of 8");
  static_assert(PLAIN_IN || ((KQ0 % 2 == 0) && (KQ1 % 2 == 0)), "segments must be 8-aligned");

  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hg = wave % NG;
  const int wr = wave / NG;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int dir = blockIdx.y;
  const LstmModelParams& P = args.m[blockIdx.z];
  const int T = args.T;
  const int row0 = blockIdx.x * ROWS + wr * (32 * R);     // first row of this wave
  const int lrow0 = wr * (32 * R);                         // same, block-local

  const float* wp = P.wpack + ((size_t)(dir * NG + hg) * KG) * (4 * 64 * 4) + lane * 4;
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * 32 + l31;
  const float bias_i = bp[0], bias_f = bp[32], bias_g = bp[64], bias_o = bp[96];

  f32x16 c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = splat16(0.0f);

  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const float* hcur = hbuf + (s & 1) * HBUF;
    float* hnxt = hbuf + ((s + 1) & 1) * HBUF;

    f32x16 acc[4][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[0][r] = splat16(bias_i);
      acc[1][r] = splat16(bias_f);
      acc[2][r] = splat16(bias_g);
      acc[3][r] = splat16(bias_o);
    }

    const float* ap0[R];
    const float* ap1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (PLAIN_IN) {
        int row = row0 + r * 32 + l31;
        ap0[r] = P.plain_in +
                 (P.plain_ev_stride ? (size_t)(row + t) * kFeat : ((size_t)row * T + t) * kFeat) + 4 * half;
        ap1[r] = nullptr;
      } else {
        ap0[r] = P.in0.chunk(row0 + r * 32 + l31, t, half);
        if constexpr (KQ1 > 0) ap1[r] = P.in1.chunk(row0 + r * 32 + l31, t, half);
        else ap1[r] = nullptr;
      }
    }
    const float* hp = hcur + half * PLANE + (lrow0 + l31) * 4;

    // fragment loaders; kgG: 0..KG_REC-1 recurrent, then the input groups
    auto loadA = [&](int kgG, f32x4 (&a)[R]) {
      if (kgG < KG_REC) {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = *(const f32x4*)(hp + kgG * 2 * PLANE + r * 128);
      } else {
        const int kgi = kgG - KG_REC;
        if constexpr (PLAIN_IN) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r * 32 + l31 < args.n_rows) {
              v[0] = ap0[r][0];
              v[1] = ap0[r][1];
              if (!half) { v[2] = ap0[r][2]; v[3] = ap0[r][3]; }
            }
            a[r] = v;
          }
        } else if (KQ1 == 0 || kgi < KQ0 / 2) {
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] = *(const f32x4*)(ap0[r] + kgi * 256);
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] = *(const f32x4*)(ap1[r] + (kgi - KQ0 / 2) * 256);
        }
      }
    };
    auto loadB = [&](int kgG, f32x4 (&b)[4]) {
      const int kgw = kgG < KG_REC ? KG_IN + kgG : kgG - KG_REC;
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = *(const f32x4*)(wp + (kgw * 4 + g) * 256);
    };
    auto mma = [&](const f32x4 (&a)[R], const f32x4 (&b)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[g][r] = mfma32(a[r][j], b[g][j], acc[g][r]);
    };

    {
      f32x4 a0[R], b0[4], a1[R], b1[4];
      int kg = (s == 0) ? KG_REC : 0;            // h_0 = 0: the recurrent groups are skipped
      loadA(kg, a0);
      loadB(kg, b0);
      for (; kg + 2 < KG; kg += 2) {
        loadA(kg + 1, a1);
        loadB(kg + 1, b1);
        mma(a0, b0);
        loadA(kg + 2, a0);
        loadB(kg + 2, b0);
        mma(a1, b1);
      }
      if (kg + 1 < KG) {
        loadA(kg + 1, a1);
        loadB(kg + 1, b1);
        mma(a0, b0);
        mma(a1, b1);
      } else {
        mma(a0, b0);
      }
    }

    // ---- gates (register-local) and h_t -> LDS -------------------------------------------
    {
      const int u = hg * 32 + l31;                 // hidden unit of this lane's column
      float* hw = hnxt + (u >> 2) * PLANE + (u & 3);
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float ig = gate_act<ACT>(acc[0][r][reg]);
          float fg = gate_act<ACT>(acc[1][r][reg]);
          float gg = tanh_fast(acc[2][r][reg]);
          float og = gate_act<ACT>(acc[3][r][reg]);
          float cn = __builtin_fmaf(fg, c[r][reg], ig * gg);
          c[r][reg] = cn;
          float hn = og * tanh_fast(cn);
          hw[(lrow0 + r * 32 + acc_row(reg, lane)) * 4] = hn;
        }
      }
    }
    __syncthreads();

    // ---- h_t (+ fused BatchNorm) -> global, coalesced 16-byte stores --------------------------
    {
      constexpr int KQH = H / 4;                   // real chunks of this direction
      constexpr int ITEMS = KQH * ROWS;            // float4 items
      const int kq_total = 2 * KQH;
      for (int it = threadIdx.x; it < ITEMS; it += NTHREADS) {
        int kq = it / ROWS, rr = it % ROWS;
        f32x4 v = *(const f32x4*)(hnxt + kq * PLANE + rr * 4);
        const f32x4 sc = *(const f32x4*)(P.bn_scale + dir * H + kq * 4);
        const f32x4 sh = *(const f32x4*)(P.bn_shift + dir * H + kq * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * sc[q] + sh[q];
        int tile = blockIdx.x * (R * WR) + rr / 32;
        float* dst = P.out + ((size_t)(tile * T + t) * kq_total + dir * KQH + kq) * 128 + (rr & 31) * 4;
        *(f32x4*)dst = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Signal branch: conv1d(1->8,k3)+ReLU+BN, conv1d(8->8,k3)+ReLU+BN, + signal, flatten(400),
// dense(400->64).  nanorevcnn.py:17-38, output_handeler.py:209-215.
// One workgroup = 32 events (one MFMA row tile).  The convolutions run on the VALU (f32 VALU
// rate == f32 MFMA rate on gfx950, and N=8 would waste 3/4 of an MFMA tile); the 400->64 dense
// runs on MFMA out of an LDS image in A-fragment order.
// ---------------------------------------------------------------------------------------
struct CnnModelParams {
  const float* conv;      // 24 w1[k][o], 8 b1, 8 s1, 8 h1, 192 w2[k][c][o], 8 b2, 8 s2, 8 h2  (=264)
  const float* dpack;     // dense 400->64 packed [ntile 2][kg 50][64][4]
  const float* dbias;     // [64]
  float* out;             // tiled, KQ=16: window-major [wtile][t][16][32][4] or event-major [etile][16][32][4]
};
struct CnnArgs {
  CnnModelParams m[2];
  const float* signal;    // [n][T][50] (window mode) or [N][50] (event mode)
  int T;                  // window mode: T; event mode: 1
  int n_rows;             // windows (window mode) or events (event mode)
};

constexpr int kCnnThreads = 320;   // 32 events x 10 chunks of 5 positions; waves 0-3 also run the MFMAs

__global__ void __launch_bounds__(kCnnThreads) cnn_kernel(const CnnArgs args) {
  constexpr int XS = 56;                    // staged signal row stride (floats): [2 zero | 50 | 4 zero]
  constexpr int PLANE = 32 * 4 + 4;         // floats per kq plane of the flat image
  __shared__ __attribute__((aligned(16))) float xs[32 * XS];
  __shared__ __attribute__((aligned(16))) float flat[100 * PLANE];
  __shared__ __attribute__((aligned(16))) float red[2 * 32 * 33];

  const CnnModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int wt = b / T, t = b % T;

  // ---- stage 32 signal rows (zero padded) -------------------------------------------------
  for (int i = tid; i < 32 * XS; i += kCnnThreads) {
    int r = i / XS, p = i % XS - 2;
    int row = wt * 32 + r;
    float v = 0.f;
    if (row < args.n_rows && p >= 0 && p < kSig) v = args.signal[((size_t)row * T + t) * kSig + p];
    xs[i] = v;
  }
  __syncthreads();

  // ---- convolutions: thread = (event r, positions p0..p0+4) ---------------------------------
  {
    const float* cw = P.conv;
    const int r = tid & 31, p0 = (tid >> 5) * 5;
    const float* x = xs + r * XS + 2 + p0;        // x[-2..6] readable
    float b1v[7][8];                               // bn1 at positions p0-1 .. p0+5
    {
      float w1[48];                                // w1[3][8], b1[8], bn1 scale[8], shift[8]
#pragma unroll
      for (int i = 0; i < 48; ++i) w1[i] = cw[i];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        int p = p0 - 1 + q;
        bool inside = (p >= 0) && (p < kSig);
        float xm = x[q - 2], xc = x[q - 1], xp = x[q];
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
    // conv2, weight-stationary: each (tap k, in-channel ci) row of 8 weights is fetched once
    // (one scalar dwordx8 load) and applied to all 5 positions of this thread: 40 FMAs per fetch.
    const float* w2 = cw + 48;
    float o5[5][8];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int o = 0; o < 8; ++o) o5[q][o] = w2[192 + o];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) {
        float wrow[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) wrow[o] = w2[(k * 8 + ci) * 8 + o];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          float a = b1v[q + k][ci];
#pragma unroll
          for (int o = 0; o < 8; ++o) o5[q][o] = __builtin_fmaf(a, wrow[o], o5[q][o]);
        }
      }
    float s2[8], h2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) { s2[o] = w2[200 + o]; h2[o] = w2[208 + o]; }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float xc = x[q];
      float o8[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float v = __builtin_fmaxf(o5[q][o], 0.f);
        v = v * s2[o] + h2[o];
        o8[o] = v + xc;                            // Add(): broadcast the raw signal over channels
      }
      int p = p0 + q;                              // flat index p*8+o -> kq = 2p, 2p+1
      *(f32x4*)(flat + (2 * p) * PLANE + r * 4) = f32x4{o8[0], o8[1], o8[2], o8[3]};
      *(f32x4*)(flat + (2 * p + 1) * PLANE + r * 4) = f32x4{o8[4], o8[5], o8[6], o8[7]};
    }
  }
  __syncthreads();

  // ---- dense 400 -> 64 on MFMA: wave w (<4): n-tile = w&1, k-half = w>>1 ----------------------
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  f32x16 acc = splat16(0.f);
  if (wave < 4) {
    const int nt = wave & 1, kh = wave >> 1;
    const float* wp = P.dpack + ((size_t)nt * 50 + kh * 25) * 256 + lane * 4;
    const float* ap = flat + (kh * 50 + half) * PLANE + l31 * 4;
#pragma unroll 5
    for (int kg = 0; kg < 25; ++kg) {
      f32x4 a = *(const f32x4*)(ap + kg * 2 * PLANE);
      f32x4 bq = *(const f32x4*)(wp + kg * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = mfma32(a[j], bq[j], acc);
    }
    if (kh == 1) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) red[(nt * 32 + acc_row(reg, lane)) * 33 + l31] = acc[reg];
    }
  }
  __syncthreads();            // all reads of `flat` are done; reuse it as the output image
  if (wave < 2) {
    const int nt = wave;
    const float bias = P.dbias[nt * 32 + l31];
    const int u = nt * 32 + l31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int row = acc_row(reg, lane);
      float v = (acc[reg] + red[(nt * 32 + row) * 33 + l31]) + bias;
      flat[(u >> 2) * PLANE + row * 4 + (u & 3)] = v;
    }
  }
  __syncthreads();
  for (int it = tid; it < 16 * 32; it += kCnnThreads) {
    int kq = it >> 5, rr = it & 31;
    f32x4 v = *(const f32x4*)(flat + kq * PLANE + rr * 4);
    *(f32x4*)(P.out + ((size_t)b * 16 + kq) * 128 + rr * 4) = v;
  }
}

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

__global__ void __launch_bounds__(256) head_kernel(const HeadArgs args) {
  constexpr int PLANE = 32 * 4 + 4;
  __shared__ __attribute__((aligned(16))) float img[4][32 * PLANE];   // per-wave A image, up to 128 features
  __shared__ float flatv[32 * (6 * kHeadMaxT + 1)];
  __shared__ float featv[32 * 17];
  __shared__ float logit[32 * 8];

  const HeadModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int FS = 6 * T + 1;
  float* im = img[wave];

  for (int t = wave; t < T; t += 4) {
    // dense1: 128 -> 128, A from global
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = splat16(P.d1bias[nt * 32 + l31]);
    const float* ap = P.in + ((size_t)(tile * T + t) * 32 + half) * 128 + l31 * 4;
    const float* wp = P.d1pack + lane * 4;
#pragma unroll 2
    for (int kg = 0; kg < 16; ++kg) {
      f32x4 a = *(const f32x4*)(ap + kg * 256);
      f32x4 b[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) b[nt] = *(const f32x4*)(wp + (nt * 16 + kg) * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma32(a[j], b[nt][j], acc[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      int u = nt * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        im[(u >> 2) * PLANE + acc_row(reg, lane) * 4 + (u & 3)] = __builtin_fmaxf(acc[nt][reg], 0.f);
    }
    // dense2: 128 -> 32 (same wave wrote the image; LDS ops of one wave complete in order)
    f32x16 a2 = splat16(P.d2bias[l31]);
    {
      const float* hp = im + half * PLANE + l31 * 4;
      const float* w2 = P.d2pack + lane * 4;
#pragma unroll 4
      for (int kg = 0; kg < 16; ++kg) {
        f32x4 a = *(const f32x4*)(hp + kg * 2 * PLANE);
        f32x4 b = *(const f32x4*)(w2 + kg * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) a2 = mfma32(a[j], b[j], a2);
      }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      im[(l31 >> 2) * PLANE + acc_row(reg, lane) * 4 + (l31 & 3)] = __builtin_fmaxf(a2[reg], 0.f);
    // main_out: 32 -> 6 (padded to 32 columns)
    f32x16 a3 = splat16(P.mobias[l31]);
    {
      const float* hp = im + half * PLANE + l31 * 4;
      const float* w3 = P.mopack + lane * 4;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        f32x4 a = *(const f32x4*)(hp + kg * 2 * PLANE);
        f32x4 b = *(const f32x4*)(w3 + kg * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) a3 = mfma32(a[j], b[j], a3);
      }
    }
    if (l31 < 6) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        flatv[acc_row(reg, lane) * FS + t * 6 + l31] = __builtin_fmaxf(a3[reg], 0.f);
    }
  }
  __syncthreads();

  // feature: (6T) -> 16, relu
  for (int it = tid; it < 32 * 16; it += 256) {
    int r = it >> 4, f = it & 15;
    float v = P.featb[f];
    const float* fr = flatv + r * FS;
    for (int k = 0; k < 6 * T; ++k) v = __builtin_fmaf(fr[k], P.featw[k * 16 + f], v);
    featv[r * 17 + f] = __builtin_fmaxf(v, 0.f);
  }
  __syncthreads();
  const int C = P.n_class;
  if (tid < 32 * 8) {
    int r = tid >> 3, cc = tid & 7;
    if (cc < C) {
      float v = P.outb[cc];
#pragma unroll
      for (int f = 0; f < 16; ++f) v = __builtin_fmaf(featv[r * 17 + f], P.outw[f * C + cc], v);
      logit[r * 8 + cc] = v;
    }
  }
  __syncthreads();
  if (tid < 32) {
    int row = tile * 32 + tid;
    if (row < args.n_rows) {
      float mx = logit[tid * 8];
      for (int cc = 1; cc < C; ++cc) mx = __builtin_fmaxf(mx, logit[tid * 8 + cc]);
      float e[8], sum = 0.f;
      for (int cc = 0; cc < C; ++cc) { e[cc] = expf(logit[tid * 8 + cc] - mx); sum += e[cc]; }
      int best = 0; float bv = -1.f;
      for (int cc = 0; cc < C; ++cc) {
        float p = e[cc] / sum;
        P.prob[(size_t)row * C + cc] = p;
        if (p > bv) { bv = p; best = cc; }     // strict > : ties -> lowest index
      }
      P.argmax[row] = (int8_t)best;
    }
  }
}

}  // namespace nrv
