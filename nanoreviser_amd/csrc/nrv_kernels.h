// nrv_kernels.h - gfx950 (MI355X / CDNA4) device kernels of the NanoReviser window reviser.
//
// What is computed: the two Keras graphs of the reference,
//   nanorevutils/output_handeler.py:206-255 (model1) and :258-307 (model2),
//   CNN block nanorevutils/nanorevcnn.py:17-38,
// with Keras-2.2.4 inference semantics.  Everything is IEEE f32.
//
// How it is mapped (MI355X-first, not a translation of TF ops):
//   * Every contraction runs on the exact-f32 matrix pipe, v_mfma_f32_32x32x2_f32 / 16x16x4
//     (bitwise an fmaf chain, 64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak).
//   * Activations between kernels live in an MFMA-native tiled layout
//         act[tile32][t][kq][32 rows][4]      (kq = feature/4)
//     so that one wave-wide 16-byte load IS the A fragment of four MFMA k-steps
//     (lanes 0-31: features 8g..8g+3 of rows 0..31, lanes 32-63: features 8g+4..8g+7) and is a
//     single contiguous 1 KiB request.  Weights are pre-packed on the host into the matching
//     B-fragment order, so B operands stream L2 -> VGPR with no LDS staging at all.
//   * Seven launches per group of windows: cnn_kernel (signal branch), lstm1_kernel,
//     lstm_layer_kernel x3 (lstm2..4), head_mlp_kernel, head_final_kernel.  Rows (windows) are
//     independent, so every launch is (row tiles) x (directions) x (2 models) workgroups with no
//     inter-workgroup communication.
//   * One Bi-LSTM layer = one launch; a wave owns 32 hidden units x 4 gates x R row tiles, so
//     i,f,g,o of one (window, unit) sit in the same lane/register and the cell update is
//     register-local; c never leaves registers, h_t goes through a double-buffered LDS image
//     (one barrier per step) and is written out coalesced with the following BatchNorm fused;
//     the next step's input projection is issued while the VALU does this step's gates.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace nrv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSig = 50;        // samples per event window   (output_handeler.py:202)
constexpr int kFeat = 6;        // features per event         (output_handeler.py:203)

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

// Buffer addressing: address = descriptor base (SGPR, wave-uniform) + per-lane voffset (VGPR, 32
// bit) + soffset (SGPR / immediate).  Used for every streamed operand so that stepping through
// k-groups is scalar arithmetic; with plain 64-bit pointers hipcc materialises (and spills) one
// VGPR address pair per k-group.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0);
  return __builtin_bit_cast(f32x4, v);
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
  // Same address split for scalar-base addressing: chunk(rowbase + l31, t, half + kq) ==
  // ubase(rowbase, t) + kq*128 + voff(rowbase, t, l31, half), with ubase wave-uniform (rowbase is)
  // and voff a small per-lane offset (floats).
  __device__ __forceinline__ const float* ubase(int rowbase, int t) const {
    int eu = rowbase + t * ev_stride;
    return p + ((long)((eu >> 5) * tt + t * tm) * kq_total) * 128;
  }
  __device__ __forceinline__ unsigned voff(int rowbase, int t, int l31, int half) const {
    int x = ((rowbase + t * ev_stride) & 31) + l31;
    return (unsigned)((x >> 5) * (tt * kq_total * 128) + (x & 31) * 4 + half * 128);
  }
};

// ---------------------------------------------------------------------------------------
// Bi-LSTM layer kernel
// ---------------------------------------------------------------------------------------
// tanh for the LSTM cell: 1 - 2/(2^(2x log2 e) + 1) on the hardware exp2/rcp (1 ulp each): five
// instructions, no branch, exact limits at +-inf.  Absolute error <= ~1.5e-7 everywhere (for
// |x| -> 0 the RELATIVE error grows, which is immaterial here: the argument is a 200-500-term f32
// dot product whose own rounding noise is ~1e-6 absolute, and tanh' <= 1).
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
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

// XCD-aware block map for the Bi-LSTM layer kernels (guide T1).  Workgroups are dealt round-robin
// over the 8 XCDs, so blocks b and b + 8 share an XCD and its 4 MiB L2.  With a (rows, dir, model)
// grid every XCD streams all four (direction, model) weight sets - 3.8 MB of split-bf16 weights in
// the 192->128 layer, as much as the whole L2 - next to the activations.  Instead the launch is 1-D
// and the XCD group g = b % 8 fixes the weight set: (dir, model) = (g>>1 & 1, g>>2), two XCDs per
// set, each keeping under 1 MB of weights resident.  Row blocks: 2*(b/8) + (g&1); a block past
// n_blk exits.  Placement is a speed matter only.
struct LstmBlock { int rowblk, dir, model; };
__device__ __forceinline__ LstmBlock lstm_block() {
  const int b = blockIdx.x, g = b & 7;
  return LstmBlock{((b >> 3) << 1) | (g & 1), (g >> 1) & 1, g >> 2};
}
__host__ __device__ constexpr int lstm_grid(int n_blk) { return 8 * ((n_blk + 1) / 2); }

struct LstmArgs {
  LstmModelParams m[2];
  int T;
  int n_rows;             // valid rows (windows)
  int n_blk;              // row blocks (workgroups per direction and model)
};

// KQ0/KQ1: input segments in 4-feature chunks (K = 4*KQ, K multiple of 8).  H: hidden units per
// direction, NG = ceil(H/32) hidden groups.  A wave owns one hidden group (32 units x 4 gates) for
// R row tiles; a workgroup is NG x WR waves covering 32*R*WR rows.
// grid = lstm_grid(ceil(tiles/(R*WR))) (see lstm_block), block = 64*NG*WR.
//
// Schedule of one step s (time index t):
//     Z  = b + x_t W            (already there: computed during step s-1)
//     Z += h_{t-1} U            recurrent k-groups, A fragments from the LDS image of h_{t-1}
//     N  = b + x_{t+1} W        input k-groups of the NEXT step - independent of h - issued on the
//                               matrix pipe while the VALU turns Z into (c_t, h_t): the gate code
//                               is cut into per-element pieces placed between MFMA sub-batches
//     barrier; h_t (+BatchNorm) -> HBM; Z = N
// so the only serial section is recurrent MFMAs -> last gate pieces -> barrier.  k-groups are
// fully unrolled (static LDS/global address spaces, counted waits) and software-pipelined: the
// fragments of group g+1 are requested before group g's 16*R MFMAs issue, and the first fragments
// of each phase are requested one phase early.
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
  // Prefetch depths in k-groups (one k-group = 16*R MFMAs = 1024*R cycles).  vmcnt retires in
  // issue order, so each iteration requests the (L2-resident) weights first and the activations
  // (Infinity-Cache / HBM latency) last: a wait for B(g) then leaves the younger A requests in flight.
  constexpr int PB = (R == 1) ? 2 : 1;         // weight fragments:      PB groups ahead
  constexpr int PA = PB + 1;                   // activation fragments:  PB+1 groups ahead
  constexpr int NE = 16 * R;                   // gate elements per lane per step
  constexpr int NSLOT = 4 * KG_IN;             // MFMA sub-batches of the input projection
  static_assert(H % 8 == 0, "H must be a multiple of 8");
  static_assert(PLAIN_IN || ((KQ0 % 2 == 0) && (KQ1 % 2 == 0)), "segments must be 8-aligned");

  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];
  __shared__ __attribute__((aligned(16))) float bnl[2 * H];     // BatchNorm scale | shift of this direction

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
  const int hg = wave % NG;
  const int wr = wave / NG;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS + wr * (32 * R);     // first row of this wave (uniform)
  const int lrow0 = wr * (32 * R);                         // same, block-local

  // weights: wave-uniform base (SGPR pair) + per-lane 32-bit offset -> saddr addressing, so the
  // k-group addresses are scalar adds instead of one 64-bit VGPR pair per group
  const __amdgpu_buffer_rsrc_t wrs =
      make_rsrc(P.wpack + ((size_t)(dir * NG + hg) * KG) * (4 * 64 * 4), KG * 4 * 64 * 4 * 4);
  const unsigned wlane = lane * 16;                        // bytes
  constexpr int WREC = KG_IN;                              // first recurrent k-group in the pack
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * 32 + l31;
  const float bias4[4] = {bp[0], bp[32], bp[64], bp[96]};
  const int u = hg * 32 + l31;                             // hidden unit of this lane's column
  const int hw_off = (u >> 2) * PLANE + (u & 3) + (lrow0 + 4 * half) * 4;
  const int hr_off = half * PLANE + (lrow0 + l31) * 4;

  f32x16 c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = splat16(0.0f);
  for (int i = threadIdx.x; i < 2 * H; i += NTHREADS)
    bnl[i] = i < H ? P.bn_scale[dir * H + i] : P.bn_shift[dir * H + i - H];
  __syncthreads();

  // ---- h_t (+ fused BatchNorm) -> HBM: LDS image -> 16-byte coalesced stores.  Split in two so the
  // LDS reads are issued ahead of, and the stores behind, the first recurrent MFMAs of the next step.
  constexpr int KQH = H / 4;                   // real 4-feature chunks of this direction
  // NG > 1: the whole workgroup copies the whole image.  NG == 1: every wave copies its own rows.
  constexpr int CROWS = (NG > 1) ? ROWS : 32 * R;
  constexpr int CTHREADS = (NG > 1) ? NTHREADS : 64;
  constexpr int ITEMS = KQH * CROWS;           // float4 items per step
  constexpr int NIT = (ITEMS + CTHREADS - 1) / CTHREADS;
  const int ctid = (NG > 1) ? threadIdx.x : lane;
  const int crow0 = (NG > 1) ? 0 : lrow0;
  f32x4 cov[NIT];
  auto copyout_read = [&](const float* himg) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = ctid + i * CTHREADS;
      if (ITEMS % CTHREADS == 0 || it < ITEMS)
        cov[i] = *(const f32x4*)(himg + (it / CROWS) * PLANE + (crow0 + it % CROWS) * 4);
    }
  };
  auto copyout_write = [&](int t) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = ctid + i * CTHREADS;
      if (ITEMS % CTHREADS == 0 || it < ITEMS) {
        const int kq = it / CROWS, rr = crow0 + it % CROWS;
        const f32x4 sc = *(const f32x4*)(bnl + kq * 4);
        const f32x4 sh = *(const f32x4*)(bnl + H + kq * 4);
        f32x4 v = cov[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * sc[q] + sh[q];
        const int tile = blk.rowblk * (R * WR) + rr / 32;
        float* dst = P.out + ((size_t)(tile * T + t) * (2 * KQH) + dir * KQH + kq) * 128 + (rr & 31) * 4;
        *(f32x4*)dst = v;
      }
    }
  };

  // ---- fragment loaders ---------------------------------------------------------------------
  const float* ap0[R];                 // PLAIN_IN only: per-lane pointers
  __amdgpu_buffer_rsrc_t ar0[R], ar1[R];  // tiled inputs: descriptors rebased per (step, row tile)
  unsigned av0[R], av1[R];             // per-lane offsets (bytes)
  auto set_t = [&](int t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (PLAIN_IN) {
        const int row = row0 + r * 32 + l31;
        ap0[r] = P.plain_in +
                 (P.plain_ev_stride ? (size_t)(row + t) * kFeat : ((size_t)row * T + t) * kFeat) + 4 * half;
      } else {
        const int rb = row0 + r * 32;
        ar0[r] = make_rsrc(P.in0.ubase(rb, t), 0xffffffffu);
        av0[r] = P.in0.voff(rb, t, l31, half) * 4;
        if constexpr (KQ1 > 0) {
          ar1[r] = make_rsrc(P.in1.ubase(rb, t), 0xffffffffu);
          av1[r] = P.in1.voff(rb, t, l31, half) * 4;
        }
      }
    }
  };
  auto loadA_in = [&](int kgi, f32x4 (&a)[R]) {
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
      for (int r = 0; r < R; ++r) a[r] = buf_load16(ar0[r], av0[r], kgi * 1024);
    } else {
      if constexpr (KQ1 > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = buf_load16(ar1[r], av1[r], (kgi - KQ0 / 2) * 1024);
      }
    }
  };
  auto loadB = [&](int kg, f32x4 (&b)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = buf_load16(wrs, wlane, (kg * 4 + g) * 1024);
  };

  // ---- one gate element: (row tile r, accumulator register reg) ---------------------------------
  auto gate = [&](const f32x16 (&Z)[4][R], float* hw, int r, int reg) {
    float ig = gate_act<ACT>(Z[0][r][reg]);
    float fg = gate_act<ACT>(Z[1][r][reg]);
    float gg = tanh_fast(Z[2][r][reg]);
    float og = gate_act<ACT>(Z[3][r][reg]);
    float cn = __builtin_fmaf(fg, c[r][reg], ig * gg);
    c[r][reg] = cn;
    hw[(r * 32 + (reg & 3) + 8 * (reg >> 2)) * 4] = og * tanh_fast(cn);
  };

  // ---- N = b + x_t W, optionally with the gates of Z spread between the MFMA sub-batches ---------
  auto preload = [&](f32x4 (&pa)[PA][R], f32x4 (&pb)[PB][4]) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (i < PB && i < KG_IN) loadB(i, pb[i]);
      if (i < KG_IN) loadA_in(i, pa[i]);
    }
  };
  auto inproj = [&](f32x16 (&N)[4][R], const f32x4 (&pa)[PA][R], const f32x4 (&pb)[PB][4], auto hook) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < R; ++r) N[g][r] = splat16(bias4[g]);
    f32x4 a[PA + 1][R], b[PB + 1][4];
#pragma unroll
    for (int i = 0; i < PA; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r) a[i][r] = pa[i][r];
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) b[i][g] = pb[i][g];
#pragma unroll
    for (int kg = 0; kg < KG_IN; ++kg) {
      if (kg + PB < KG_IN) loadB(kg + PB, b[(kg + PB) % (PB + 1)]);
      if (kg + PA < KG_IN) loadA_in(kg + PA, a[(kg + PA) % (PA + 1)]);
      __builtin_amdgcn_sched_barrier(0);         // requests first; PB / PA k-groups of MFMAs cover them
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < R; ++r)
            N[g][r] = mfma32(a[kg % (PA + 1)][r][j], b[kg % (PB + 1)][g][j], N[g][r]);
        hook(kg * 4 + j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16 acc[4][R];                      // the one accumulator set (matrix pipe)
  f32x16 zv[4][R];                       // z of the current step, read out for the VALU
  f32x4 preA[PA][R], preB[PB][4], brec0[4];

  // prologue: acc = b + x_{t0} W
  set_t(dir ? T - 1 : 0);
  preload(preA, preB);
  inproj(acc, preA, preB, [](int) {});

  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const float* hcur = hbuf + (s & 1) * HBUF;
    float* hnxt = hbuf + ((s + 1) & 1) * HBUF;
    float* hw = hnxt + hw_off;
    const bool more = s + 1 < T;

    if (more) {                                  // first fragments of the next input projection
      set_t(dir ? t - 1 : t + 1);
      preload(preA, preB);
    }

    // ---- acc += h_{t-1} U   (h_0 = 0: nothing to add on the first step) -------------------------
    // The previous step's h image (hcur) is also what still has to go out to HBM: its LDS reads are
    // issued here, its BatchNorm + stores after the first k-group's MFMAs are in the pipe.
    if (s > 0) {
      copyout_read(hcur);
      const float* hp = hcur + hr_off;
      f32x4 a[2][R], b[2][4];
#pragma unroll
      for (int r = 0; r < R; ++r) a[0][r] = *(const f32x4*)(hp + r * 128);
#pragma unroll
      for (int g = 0; g < 4; ++g) b[0][g] = brec0[g];
#pragma unroll
      for (int kg = 0; kg < KG_REC; ++kg) {
        const int cur = kg & 1;
        if (kg + 1 < KG_REC) {
#pragma unroll
          for (int r = 0; r < R; ++r) a[cur ^ 1][r] = *(const f32x4*)(hp + (kg + 1) * 2 * PLANE + r * 128);
          loadB(WREC + kg + 1, b[cur ^ 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[g][r] = mfma32(a[cur][r][j], b[cur][g][j], acc[g][r]);
        if (kg == 0) copyout_write(dir ? t + 1 : t - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- z -> VGPRs; gates of step s hidden under the input projection of step s+1 -----------------
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < R; ++r) zv[g][r] = acc[g][r];
    if (more) {
      inproj(acc, preA, preB, [&](int slot) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if ((e * NSLOT) / NE == slot) gate(zv, hw, e / 16, e % 16);
      });
      loadB(WREC, brec0);                        // first recurrent weights of step s+1, ahead of the barrier
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) gate(zv, hw, e / 16, e % 16);
    }
    // h_t must be visible to the other hidden groups of this row block before the next recurrent
    // product.  With a single hidden group (NG == 1) the wave only ever reads its own rows (the
    // copy-out is per wave too), and DS operations of one wave complete in order: no barrier.
    if constexpr (NG > 1) __syncthreads();

  }
  // last step's h
  copyout_read(hbuf + (T & 1) * HBUF);
  copyout_write(dir ? 0 : T - 1);
}

// ---------------------------------------------------------------------------------------
// Split-bf16 variant of the Bi-LSTM layer kernel.
// Every f32 operand x is written as hi + mid + lo with three bf16 terms (24 mantissa bits, i.e. the
// whole f32 value) and the product a*b is formed from the six term pairs with i + j <= 2,
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid,
// each on v_mfma_f32_32x32x16_bf16 with f32 accumulation: the dropped pairs are below 2^-24 of the
// product, so the result is as accurate as the f32 pipe (tools/bf16_split_study.py: max |dp| vs
// fp64 equal to the f32 path, no argmax flips) at 6/16 of its matrix time.  Weights are split on
// the host (exact); activations are split in registers as they are loaded (the f32 tiled layouts
// and every other kernel are untouched).  Gates run after the matrix phase (no read-out copy): the
// register budget goes to R = 2 row tiles per wave, which is what keeps the 1.5x larger operand
// stream inside the CU's 64 B/clk vector-memory path.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Split3 { bf16x8 t[3]; };

__device__ __forceinline__ float bf16_to_f32(__bf16 v) {
  return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, v) << 16);
}

// x[0..7] -> three bf16x8 terms, round-to-nearest-even at every level
__device__ __forceinline__ Split3 split3(const f32x4& lo4, const f32x4& hi4) {
  Split3 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? lo4[j] : hi4[j - 4];
    const __bf16 h = (__bf16)x;
    const float r1 = x - bf16_to_f32(h);
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - bf16_to_f32(m);
    o.t[0][j] = h;
    o.t[1][j] = m;
    o.t[2][j] = (__bf16)r2;
  }
  return o;
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct LstmSplitModelParams {
  const void* wsplit;     // [dir][hg][kb][gate][term 3][64 lanes][8 bf16]; kb: input k-blocks then recurrent
  const float* bias;      // [dir][hg][gate][32]
  const float* bn_scale;  // [2H]
  const float* bn_shift;  // [2H]
  ActView in0, in1;
  float* out;             // tiled window-major [tiles][T][2H/4][32][4]
};
struct LstmSplitArgs {
  LstmSplitModelParams m[2];
  int T;
  int n_rows;
  int n_blk;
};

// grid = lstm_grid(ceil(tiles/(R*WR))), block = 64*NG*WR.  K0 = 4*KQ0, K1 = 4*KQ1, H all multiples of 16.
template <int KQ0, int KQ1, int H, int R, int WR, int ACT>
__global__ void __launch_bounds__(64 * ((H + 31) / 32) * WR)
lstm_split_kernel(const LstmSplitArgs args) {
  constexpr int NG = (H + 31) / 32;
  constexpr int KB0 = KQ0 / 4, KB1 = KQ1 / 4, KB_IN = KB0 + KB1, KB_REC = H / 16, KB = KB_IN + KB_REC;
  constexpr int ROWS = 32 * R * WR;
  constexpr int PLANE = ROWS * 4 + 4;
  constexpr int HBUF = (NG * 32 / 4) * PLANE;
  constexpr int NTHREADS = 64 * NG * WR;
  static_assert(KQ0 % 4 == 0 && KQ1 % 4 == 0 && H % 16 == 0, "K must come in blocks of 16");

  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];
  __shared__ __attribute__((aligned(16))) float bnl[2 * H];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hg = wave % NG, wr = wave / NG;
  const int half = lane >> 5, l31 = lane & 31;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmSplitModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS + wr * (32 * R);
  const int lrow0 = wr * (32 * R);

  // weights: [kb][gate][term] x 1 KiB, buffer-addressed
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + hg) * KB) * (4 * 3 * 1024), KB * 4 * 3 * 1024);
  const unsigned wlane = lane * 16;
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * 32 + l31;
  const float bias4[4] = {bp[0], bp[32], bp[64], bp[96]};
  const int u = hg * 32 + l31;
  const int hw_off = (u >> 2) * PLANE + (u & 3) + (lrow0 + 4 * half) * 4;

  for (int i = threadIdx.x; i < 2 * H; i += NTHREADS)
    bnl[i] = i < H ? P.bn_scale[dir * H + i] : P.bn_shift[dir * H + i - H];
  __syncthreads();

  f32x16 c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = splat16(0.0f);

  // Input addressing of one timestep: per row tile a buffer resource and a lane offset per segment.
  struct ABase {
    __amdgpu_buffer_rsrc_t r0[R], r1[R];
    unsigned v0[R], v1[R];
  };
  auto mk_base = [&](int t) {
    ABase ab;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ab.r0[r] = make_rsrc(P.in0.ubase(row0 + r * 32, t), 0xffffffffu);
      ab.v0[r] = P.in0.voff(row0 + r * 32, t, l31, 0) * 4 + half * 1024;    // chunk kq = 4kb + 2*half
      if constexpr (KQ1 > 0) {
        ab.r1[r] = make_rsrc(P.in1.ubase(row0 + r * 32, t), 0xffffffffu);
        ab.v1[r] = P.in1.voff(row0 + r * 32, t, l31, 0) * 4 + half * 1024;
      } else {
        ab.r1[r] = ab.r0[r];
        ab.v1[r] = 0;
      }
    }
    return ab;
  };
  // B terms of k-block kb: 4 gates x 3 terms, 1 KiB each
  auto loadB = [&](int kb, bf16x8 (&bb)[4][3]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
        bb[g][tm] = __builtin_bit_cast(bf16x8, buf_load16(wrs, wlane, ((kb * 4 + g) * 3 + tm) * 1024));
  };
  // raw f32 A chunks (two float4 per k-block and row tile) from input segment 0 / 1
  auto loadA0 = [&](const ABase& ab, int kb, int r, f32x4 (&a)[2]) {
    a[0] = buf_load16(ab.r0[r], ab.v0[r], kb * 2048);
    a[1] = buf_load16(ab.r0[r], ab.v0[r], kb * 2048 + 512);
  };
  auto loadA1 = [&](const ABase& ab, int kb, int r, f32x4 (&a)[2]) {
    a[0] = buf_load16(ab.r1[r], ab.v1[r], (kb - KB0) * 2048);
    a[1] = buf_load16(ab.r1[r], ab.v1[r], (kb - KB0) * 2048 + 512);
  };

  // Operand pipeline.  A "unit" is one (k-block, row tile): 24 MFMAs (6 term pairs x 4 gates).
  // Two k-blocks (2R units) run per loop trip on static ring slots:
  //   b[slot]     weights of the k-block, loaded one k-block (R units) ahead;
  //   a[slot][r]  raw f32 activations, loaded two k-blocks ahead;
  //   S[q & 1]    the three bf16 terms of a unit's activations.  They are produced DURING the previous
  //               unit: ~40 VALU ops that sched_group_barrier interleaves two per MFMA, in the shadow
  //               of the matrix pipe (done serially in front of each unit they cost 25 % of it).
  // A trip is one basic block: where its A refills come from (input segment 0, segment 1, the LDS
  // image of h_{t-1}, or - in the last trip of a step - blocks 0/1 of the NEXT step, which do not
  // depend on h_t) is a compile-time parameter and the k loop is cut into one rolled loop per source
  // (branches between units would also let LLVM sink each split down to its use).  The pipeline thus
  // runs across timesteps and the matrix pipe restarts warm after the gates.
  static_assert(KB0 % 2 == 0 && KB1 % 2 == 0 && KB_REC % 2 == 0 && KB0 >= 2, "k-block counts must be even");
  f32x4 a[2][R][2];
  bf16x8 b[2][4][3];
  Split3 S[2];
  ABase cur = mk_base(dir ? T - 1 : 0);
  loadB(0, b[0]);
#pragma unroll
  for (int r = 0; r < R; ++r) loadA0(cur, 0, r, a[0][r]);
#pragma unroll
  for (int r = 0; r < R; ++r) loadA0(cur, 1, r, a[1][r]);
  S[0] = split3(a[0][0][0], a[0][0][1]);

  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const float* hcur = hbuf + (s & 1) * HBUF;
    float* hnxt = hbuf + ((s + 1) & 1) * HBUF;
    const float* hp = hcur + (2 * half) * PLANE + (lrow0 + l31) * 4;
    // the last step "prefetches" its own inputs again (harmless, keeps the trip branch-free)
    const ABase nxt = mk_base(s + 1 < T ? (dir ? t - 1 : t + 1) : t);

    f32x16 acc[4][R];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[g][r] = splat16(bias4[g]);

    // SRC: 0 = segment 0, 1 = segment 1, 2 = recurrent (LDS), 3 = next step's blocks 0/1
    auto trip = [&](auto src_tag, int kb) {
      constexpr int SRC = decltype(src_tag)::value;
#pragma unroll
      for (int q = 0; q < 2 * R; ++q) {
        const int slot = q / R, r = q % R;
        const int qn = (q + 1) % (2 * R), slot_n = qn / R, r_n = qn % R;
        // refill the registers the preceding units have released
        if (r == 0) loadB(slot == 0 ? kb + 1 : (SRC == 3 ? 0 : kb + 2), b[1 - slot]);
        const int kbA = kb + slot + 2;
        if constexpr (SRC == 0) loadA0(cur, kbA, r, a[slot][r]);
        if constexpr (SRC == 1) loadA1(cur, kbA, r, a[slot][r]);
        if constexpr (SRC == 2) {
          const float* qh = hp + (kbA - KB_IN) * 4 * PLANE + r * 128;
          a[slot][r][0] = *(const f32x4*)(qh);
          a[slot][r][1] = *(const f32x4*)(qh + PLANE);
        }
        if constexpr (SRC == 3) loadA0(nxt, slot, r, a[slot][r]);
        __builtin_amdgcn_sched_barrier(0);
        S[(q + 1) & 1] = split3(a[slot_n][r_n][0], a[slot_n][r_n][1]);
        const Split3& as = S[q & 1];
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            acc[g][r] = mfma_bf16(as.t[PA[pr]], b[slot][g][PB[pr]], acc[g][r]);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two VALU ops of the next unit's split
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const int kb_end = (s == 0) ? KB_IN : KB;            // h_0 = 0: no recurrent blocks on the first step
    int kb = 0;
#pragma unroll 1
    for (; kb + 2 < KB0; kb += 2) trip(std::integral_constant<int, 0>{}, kb);
    if constexpr (KB1 > 0) {
#pragma unroll 1
      for (; kb + 2 < KB_IN; kb += 2) trip(std::integral_constant<int, 1>{}, kb);
    }
#pragma unroll 1
    for (; kb + 2 < kb_end; kb += 2) trip(std::integral_constant<int, 2>{}, kb);
    trip(std::integral_constant<int, 3>{}, kb);
    cur = nxt;

    // gates, h_t -> LDS
    {
      float* hw = hnxt + hw_off;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float ig = gate_act<ACT>(acc[0][r][reg]);
          const float fg = gate_act<ACT>(acc[1][r][reg]);
          const float gg = tanh_fast(acc[2][r][reg]);
          const float og = gate_act<ACT>(acc[3][r][reg]);
          const float cn = __builtin_fmaf(fg, c[r][reg], ig * gg);
          c[r][reg] = cn;
          hw[(r * 32 + (reg & 3) + 8 * (reg >> 2)) * 4] = og * tanh_fast(cn);
          // keep the accumulator read-out local to each group of elements (hipcc otherwise hoists
          // all 64*R v_accvgpr_read to the top: 128 live VGPRs at R=2 and a 15-minute compile)
          if ((reg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    // h_t (+BatchNorm) -> global
    {
      constexpr int KQH = H / 4;
      constexpr int ITEMS = KQH * ROWS;
      for (int it = threadIdx.x; it < ITEMS; it += NTHREADS) {
        const int kq = it / ROWS, rr = it % ROWS;
        f32x4 v = *(const f32x4*)(hnxt + kq * PLANE + rr * 4);
        const f32x4 sc = *(const f32x4*)(bnl + kq * 4);
        const f32x4 sh = *(const f32x4*)(bnl + H + kq * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * sc[q] + sh[q];
        const int tile = blk.rowblk * (R * WR) + rr / 32;
        float* dst = P.out + ((size_t)(tile * T + t) * (2 * KQH) + dir * KQH + kq) * 128 + (rr & 31) * 4;
        *(f32x4*)dst = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// lstm_pair_kernel: lstm_split_kernel with the timesteps taken in PAIRS.
// The input blocks of steps s and s+1 use the same weights and neither depends on h, so their
// products are formed together: one weight fetch feeds 2R row tiles (R of each step) and the weight
// bytes per MFMA halve for the input part of the layer (60 % of the k-blocks of the 192->128 layer,
// 80 % of 256->64).  tools/microbench/mfma_rate.hip: the L2->L1 operand feed is what holds the
// bf16 pipe below its register-only rate.  Schedule of a pair (s, s+1):
//     in(s, s+1) -> X[0..R), X[R..2R)      2R units per k-block
//     rec(s)     -> X[0..R);   gates(s)   -> h_s   (LDS);  barrier;  h_s -> global
//     rec(s+1)   -> X[R..2R);  gates(s+1) -> h_s+1 (LDS);  barrier;  h_s+1 -> global
// An odd last step runs alone (the R-tile forms of the same code).  Units, rings and the
// software-pipelined operand split are those of lstm_split_kernel; a trip's refill source, tile
// count and accumulator base are compile-time parameters.
// ---------------------------------------------------------------------------------------
template <int KQ0, int KQ1, int H, int R, int WR, int ACT>
__global__ void __launch_bounds__(64 * ((H + 31) / 32) * WR)
lstm_pair_kernel(const LstmSplitArgs args) {
  constexpr int NG = (H + 31) / 32;
  constexpr int KB0 = KQ0 / 4, KB1 = KQ1 / 4, KB_IN = KB0 + KB1, KB_REC = H / 16, KB = KB_IN + KB_REC;
  constexpr int ROWS = 32 * R * WR;
  constexpr int PLANE = ROWS * 4 + 4;
  constexpr int HBUF = (NG * 32 / 4) * PLANE;
  constexpr int NTHREADS = 64 * NG * WR;
  constexpr int R2 = 2 * R;
  static_assert(KQ0 % 4 == 0 && KQ1 % 4 == 0 && H % 16 == 0, "K must come in blocks of 16");
  static_assert(KB0 % 2 == 0 && KB1 % 2 == 0 && KB_REC % 2 == 0 && KB0 >= 2 && KB_REC >= 2, "k-block counts must be even");

  // At R = 2 the two accumulator sets take all 256 AGPRs and c (32 registers) is what pushes the
  // VGPR side over: every spill reload sits in the in-order vmcnt queue behind the operand prefetches
  // and drains it.  The cell state then lives in LDS ([cell][thread], conflict-free; 32 KB).
  constexpr bool CLDS = R >= 2;
  __shared__ __attribute__((aligned(16))) float hbuf[2 * HBUF];
  __shared__ __attribute__((aligned(16))) float bnl[2 * H];
  __shared__ float cl[CLDS ? 16 * R * NTHREADS : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hg = wave % NG, wr = wave / NG;
  const int half = lane >> 5, l31 = lane & 31;
  const LstmBlock blk = lstm_block();
  if (blk.rowblk >= args.n_blk) return;
  const int dir = blk.dir;
  const LstmSplitModelParams& P = args.m[blk.model];
  const int T = args.T;
  const int row0 = blk.rowblk * ROWS + wr * (32 * R);
  const int lrow0 = wr * (32 * R);

  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(
      (const char*)P.wsplit + ((size_t)(dir * NG + hg) * KB) * (4 * 3 * 1024), KB * 4 * 3 * 1024);
  const unsigned wlane = lane * 16;
  const float* bp = P.bias + (size_t)(dir * NG + hg) * 4 * 32 + l31;
  const float bias4[4] = {bp[0], bp[32], bp[64], bp[96]};
  const int u = hg * 32 + l31;
  const int hw_off = (u >> 2) * PLANE + (u & 3) + (lrow0 + 4 * half) * 4;

  for (int i = threadIdx.x; i < 2 * H; i += NTHREADS)
    bnl[i] = i < H ? P.bn_scale[dir * H + i] : P.bn_shift[dir * H + i - H];
  __syncthreads();

  f32x16 c[CLDS ? 1 : R];
  if constexpr (CLDS) {
#pragma unroll
    for (int i = 0; i < 16 * R; ++i) cl[i * NTHREADS + threadIdx.x] = 0.f;
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) c[r] = splat16(0.0f);
  }

  // Input addressing: ONE buffer resource per segment for the whole workgroup (anchored at its first
  // row tile, t = 0); a timestep is a wave-uniform byte offset per row tile (SGPR) + a lane offset.
  const float* const base0 = P.in0.ubase(blk.rowblk * ROWS, 0);
  const float* const base1 = KQ1 > 0 ? P.in1.ubase(blk.rowblk * ROWS, 0) : base0;
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(base0, 0xffffffffu);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(base1, 0xffffffffu);
  struct ABase {
    unsigned s0[R], s1[R];   // uniform byte offsets
    unsigned v0[R], v1[R];   // lane byte offsets (chunk kq = 4kb + 2*half)
  };
  auto t_of = [&](int s) __attribute__((always_inline)) { return dir ? (T - 1 - s) : s; };
  auto mk_base = [&](int s) __attribute__((always_inline)) {
    const int t = t_of(s < T ? s : T - 1);              // steps past the end alias the last one (harmless prefetch)
    ABase ab;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ab.s0[r] = (unsigned)((P.in0.ubase(row0 + r * 32, t) - base0) * 4);
      ab.v0[r] = P.in0.voff(row0 + r * 32, t, l31, 0) * 4 + half * 1024;
      if constexpr (KQ1 > 0) {
        ab.s1[r] = (unsigned)((P.in1.ubase(row0 + r * 32, t) - base1) * 4);
        ab.v1[r] = P.in1.voff(row0 + r * 32, t, l31, 0) * 4 + half * 1024;
      } else {
        ab.s1[r] = 0;
        ab.v1[r] = 0;
      }
    }
    return ab;
  };
  auto loadB = [&](int kb, bf16x8 (&bb)[4][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
        bb[g][tm] = __builtin_bit_cast(bf16x8, buf_load16(wrs, wlane, ((kb * 4 + g) * 3 + tm) * 1024));
  };
  // raw f32 A chunks of input k-block kb, row tile r of the step described by ab.  SEG (0 / 1) is a
  // compile-time parameter: a run-time segment test inside a trip would split its basic block.
  auto loadAin = [&](auto seg_tag, const ABase& ab, int kb, int r, f32x4 (&a)[2]) __attribute__((always_inline)) {
    if constexpr (decltype(seg_tag)::value == 0) {
      a[0] = buf_load16(rs0, ab.v0[r], ab.s0[r] + kb * 2048);
      a[1] = buf_load16(rs0, ab.v0[r], ab.s0[r] + kb * 2048 + 512);
    } else {
      a[0] = buf_load16(rs1, ab.v1[r], ab.s1[r] + (kb - KB0) * 2048);
      a[1] = buf_load16(rs1, ab.v1[r], ab.s1[r] + (kb - KB0) * 2048 + 512);
    }
  };

  f32x4 a[2][R2][2];
  bf16x8 b[2][4][3];
  Split3 S[2];
  f32x16 acc[4][R2];
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // term pairs, small ones first

  // One trip = two k-blocks (kb, kb+1) x RE row tiles; accumulator tiles T0 .. T0+RE-1.
  // SRC (where the A registers released by a unit are refilled from, two k-blocks ahead):
  //   0 input segment 0, 1 input segment 1   (tile rr belongs to step rr / R: bases ba[rr / R])
  //   2 the LDS image hp (recurrent blocks; tiles rr < R only)
  //   3 blocks 0/1 of the NEXT input phase (bases ba[0], ba[1]): a unit refills its tile for both steps
  //   4 nothing (the phase that follows restarts the A ring after a barrier)
  // nextB: the k-block whose weights the slot-1 unit requests (kb + 2 inside a phase).
  auto trip = [&](auto re_tag, auto t0_tag, auto src_tag, int kb, const ABase (&ba)[2], const float* hp,
                  int nextB) __attribute__((always_inline)) {
    constexpr int RE = decltype(re_tag)::value, T0 = decltype(t0_tag)::value, SRC = decltype(src_tag)::value;
#pragma unroll
    for (int q = 0; q < 2 * RE; ++q) {
      const int slot = q / RE, rr = q % RE;
      const int qn = (q + 1) % (2 * RE), slot_n = qn / RE, rr_n = qn % RE;
      if (rr == 0) loadB(slot == 0 ? kb + 1 : nextB, b[1 - slot]);
      const int kbA = kb + slot + 2;
      if constexpr (SRC == 0 || SRC == 1) loadAin(src_tag, ba[rr / R], kbA, rr % R, a[slot][rr]);
      if constexpr (SRC == 2) {
        if (rr < R) {
          const float* qh = hp + (kbA - KB_IN) * 4 * PLANE + rr * 128;
          a[slot][rr][0] = *(const f32x4*)(qh);
          a[slot][rr][1] = *(const f32x4*)(qh + PLANE);
        }
      }
      if constexpr (SRC == 3) {
        static_assert(SRC != 3 || RE == R, "next-input refills come from single-step trips");
        loadAin(std::integral_constant<int, 0>{}, ba[0], slot, rr, a[slot][rr]);       // blocks 0/1 lie in segment 0
        loadAin(std::integral_constant<int, 0>{}, ba[1], slot, rr, a[slot][R + rr]);
      }
      __builtin_amdgcn_sched_barrier(0);
      S[(q + 1) & 1] = split3(a[slot_n][rr_n][0], a[slot_n][rr_n][1]);
      const Split3& as = S[q & 1];
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g][T0 + rr] = mfma_bf16(as.t[PA[pr]], b[slot][g][PB[pr]], acc[g][T0 + rr]);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two VALU ops of the next unit's split
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using IR = std::integral_constant<int, R>;
  using IR2 = std::integral_constant<int, R2>;

  // input blocks of one step (RE = R) or of a pair (RE = 2R); the last trip refills from hp (h_{s-1})
  auto input_phase = [&](auto re_tag, const ABase (&ba)[2], const float* hp) __attribute__((always_inline)) {
    int kb = 0;
#pragma unroll 1
    for (; kb + 2 < KB0; kb += 2) trip(re_tag, I0{}, I0{}, kb, ba, hp, kb + 2);
    if constexpr (KB1 > 0) {
#pragma unroll 1
      for (; kb + 2 < KB_IN; kb += 2) trip(re_tag, I0{}, I1{}, kb, ba, hp, kb + 2);
    }
    trip(re_tag, I0{}, I2{}, kb, ba, hp, kb + 2);
  };
  // recurrent blocks of one step into tiles T0..; the last trip either hands over to the next input
  // phase (last_src 3: bases bn) or to a recurrent phase behind a barrier (last_src 4)
  auto rec_phase = [&](auto t0_tag, auto last_src, const float* hp, const ABase (&bn)[2], int lastB)
                       __attribute__((always_inline)) {
    int kb = KB_IN;
#pragma unroll 1
    for (; kb + 2 < KB; kb += 2) trip(IR{}, t0_tag, I2{}, kb, bn, hp, kb + 2);
    trip(IR{}, t0_tag, last_src, kb, bn, hp, lastB);
  };
  auto restart_rec = [&](const float* hp) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float* qh = hp + sl * 4 * PLANE + r * 128;
        a[sl][r][0] = *(const f32x4*)(qh);
        a[sl][r][1] = *(const f32x4*)(qh + PLANE);
      }
    S[0] = split3(a[0][0][0], a[0][0][1]);
  };
  auto gates = [&](auto t0_tag, float* hw) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float ig = gate_act<ACT>(acc[0][T0 + r][reg]);
        const float fg = gate_act<ACT>(acc[1][T0 + r][reg]);
        const float gg = tanh_fast(acc[2][T0 + r][reg]);
        const float og = gate_act<ACT>(acc[3][T0 + r][reg]);
        float cprev;
        if constexpr (CLDS) cprev = cl[(r * 16 + reg) * NTHREADS + threadIdx.x];
        else cprev = c[r][reg];
        const float cn = __builtin_fmaf(fg, cprev, ig * gg);
        if constexpr (CLDS) cl[(r * 16 + reg) * NTHREADS + threadIdx.x] = cn;
        else c[r][reg] = cn;
        hw[(r * 32 + (reg & 3) + 8 * (reg >> 2)) * 4] = og * tanh_fast(cn);
        if ((reg & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the accumulator read-out local
      }
  };
  auto copyout = [&](const float* hsrc, int t) __attribute__((always_inline)) {
    constexpr int KQH = H / 4;
    constexpr int ITEMS = KQH * ROWS;
    for (int it = threadIdx.x; it < ITEMS; it += NTHREADS) {
      const int kq = it / ROWS, rr = it % ROWS;
      f32x4 v = *(const f32x4*)(hsrc + kq * PLANE + rr * 4);
      const f32x4 sc = *(const f32x4*)(bnl + kq * 4);
      const f32x4 sh = *(const f32x4*)(bnl + H + kq * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] * sc[q] + sh[q];
      const int tile = blk.rowblk * (R * WR) + rr / 32;
      float* dst = P.out + ((size_t)(tile * T + t) * (2 * KQH) + dir * KQH + kq) * 128 + (rr & 31) * 4;
      *(f32x4*)dst = v;
    }
  };
  auto himg = [&](int s) __attribute__((always_inline)) { return hbuf + ((s + 1) & 1) * HBUF; };   // image of h_s
  const int hp_off = (2 * half) * PLANE + (lrow0 + l31) * 4;

  // h_{-1} = 0: the recurrent blocks of step 0 run against a zeroed image (8 of 20 k-blocks of one
  // step in the 192->128 layer) - that keeps the loop body free of first-iteration branches, whose
  // merges cost more in register moves than the products do
  for (int i = threadIdx.x; i < HBUF; i += NTHREADS) hbuf[i] = 0.f;          // image of h_{-1} is buffer 0
  __syncthreads();

  // pipeline prologue: weights of block 0, inputs of blocks 0/1 of the first pair
  {
    const ABase b0[2] = {mk_base(0), mk_base(1)};
    loadB(0, b[0]);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int rr = 0; rr < R2; ++rr) loadAin(std::integral_constant<int, 0>{}, b0[rr / R], sl, rr % R, a[sl][rr]);
    S[0] = split3(a[0][0][0], a[0][0][1]);
  }

  int s = 0;
#pragma unroll 1
  for (; s + 1 < T; s += 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int rr = 0; rr < R2; ++rr) acc[g][rr] = splat16(bias4[g]);
    const float* hp0 = himg(s - 1) + hp_off;
    {                                                        // (address sets live only where they are used)
      const ABase ba[2] = {mk_base(s), mk_base(s + 1)};
      input_phase(IR2{}, ba, hp0);
      rec_phase(I0{}, I4{}, hp0, ba, KB_IN);
    }
    gates(I0{}, himg(s) + hw_off);
    __syncthreads();
    copyout(himg(s), t_of(s));
    const float* hp1 = himg(s) + hp_off;
    restart_rec(hp1);
    {
      const ABase bn[2] = {mk_base(s + 2), mk_base(s + 3)};
      rec_phase(IR{}, I3{}, hp1, bn, 0);
    }
    gates(IR{}, himg(s + 1) + hw_off);
    __syncthreads();
    copyout(himg(s + 1), t_of(s + 1));
  }
  if (s < T) {                                               // odd T: the last step alone
    const ABase ba[2] = {mk_base(s), mk_base(s)};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[g][r] = splat16(bias4[g]);
    const float* hp0 = himg(s - 1) + hp_off;
    input_phase(IR{}, ba, hp0);
    rec_phase(I0{}, I4{}, hp0, ba, KB_IN);
    gates(I0{}, himg(s) + hw_off);
    __syncthreads();
    copyout(himg(s), t_of(s));
  }
}

// ---------------------------------------------------------------------------------------
// Device-side signal segmentation (preprocessing.py:103-131 through
// hoststage.segment_windows_f32): per base the 50 samples [st-25, st+25) clipped to the read,
// (x - shift)/scale in IEEE f64 then rounded to f32, symmetric zero padding (the odd sample goes in
// front).  Integer / exact work: the output is bit-identical to the host stage, which is pinned to
// the reference's own function.  One thread per output sample; stores are fully coalesced, the
// int16 gathers hit a ~100-byte neighbourhood per event.
// ---------------------------------------------------------------------------------------
struct SegRead {            // = nrv_read_desc (include/nanorev.h)
  long long raw_off, raw_len, ev_off, ev_len;
  double shift, scale;
};
struct SegArgs {
  const short* raw;         // all reads' samples, concatenated
  const int* starts;        // [N] event starts, relative to the start of their own read's samples
  const SegRead* reads;
  int n_reads;
  long long ev0;            // first event of this launch
  int n_ev;
  float* out;               // [n_ev][50]
};
__global__ void __launch_bounds__(256) segment_kernel(const SegArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.n_ev * 50) return;
  const int e = (int)(idx / 50), j = (int)(idx % 50);
  const long long E = a.ev0 + e;
  int lo_r = 0, hi_r = a.n_reads - 1;                   // last read with ev_off <= E
  while (lo_r < hi_r) {
    const int mid = (lo_r + hi_r + 1) >> 1;
    if (a.reads[mid].ev_off <= E) lo_r = mid; else hi_r = mid - 1;
  }
  const SegRead rd = a.reads[lo_r];
  float v = 0.f;
  if (E < rd.ev_off + rd.ev_len) {
    const long long st = a.starts[E], L = rd.raw_len;
    const long long lo = st - 25 <= 0 ? 0 : st - 25;
    const long long hi = st + 25 >= L ? L : st + 25;
    const long long seg = hi - lo, pad = 50 - seg;
    const long long left = pad > 0 ? pad / 2 + pad % 2 : 0;
    if (j >= left && j < left + seg)
      v = (float)(((double)a.raw[rd.raw_off + lo + j - left] - rd.shift) / rd.scale);
  }
  a.out[idx] = v;
}

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
  float* out;             // tiled window-major, KQ = 8: [tile32][T][8][32][4]
};
struct Lstm1Args {
  Lstm1ModelParams m[2];
  int T;
  int n_rows;
};

template <int ACT>
__global__ void __launch_bounds__(256) lstm1_kernel(const Lstm1Args args) {
  __shared__ __attribute__((aligned(16))) float hbuf[4][16 * 16 + 16];
  const Lstm1ModelParams& P = args.m[blockIdx.z];
  const int T = args.T, dir = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int rb = blockIdx.x * 4 + wave;                  // 16-row block of this wave
  const int row = rb * 16 + c;                           // the row this lane feeds as A operand
  float* hb = hbuf[wave];

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
    if (row < args.n_rows) {
      const float* src = P.plain_in +
          (P.plain_ev_stride ? (size_t)(row + t) * kFeat : ((size_t)row * T + t) * kFeat);
      x0 = src[q];                                       // k = q        (k-step 0)
      if (q < 2) x1 = src[4 + q];                        // k = 4 + q    (k-step 1; k = 6,7 are padding)
    }
  };
  float* out_base = P.out + ((size_t)(rb >> 1) * T * 8 + dir * 4 + q) * 128 + (16 * (rb & 1) + c) * 4;

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
    hprev = *(const f32x4*)(hb + c * 16 + 4 * q);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = hprev[k] * bsc[k] + bsh[k];
    *(f32x4*)(out_base + (size_t)t * 8 * 128) = o;
    x0 = nx0; x1 = nx1;
  }
}

// ---------------------------------------------------------------------------------------
// Signal branch: conv1d(1->8,k3)+ReLU+BN, conv1d(8->8,k3)+ReLU+BN, + signal, flatten(400),
// dense(400->64).  nanorevcnn.py:17-38, output_handeler.py:209-215.
//
// Persistent, wave-specialised workgroups (one per CU): five CONV waves turn the next 32-event tile
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

__global__ void __launch_bounds__(kCnnThreads) cnn_kernel(const CnnArgs args) {
  constexpr int PLANE = 32 * 4 + 4;         // floats per kq plane of the image (+4: conflict-free)
  constexpr int IMG = 100 * PLANE;
  __shared__ __attribute__((aligned(16))) float img[2 * IMG];

  const CnnModelParams& P = args.m[blockIdx.y];
  const int T = args.T;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ntile = args.n_tiles, G = gridDim.x;
  const int nloc = (ntile - (int)blockIdx.x + G - 1) / G;     // tiles of this workgroup: b = blockIdx.x + i*G

  if (wave >= kCnnMatWaves) {
    // ================================ CONV role ==============================================
    // wave cw0 holds the two 7-position chunks, the others 6-position chunks (wave-uniform count)
    const int cwv = wave - kCnnMatWaves;
    const int chunk = cwv * 2 + (lane >> 5);
    const int r = lane & 31;
    const int p0 = chunk < 2 ? 7 * chunk : 14 + 6 * (chunk - 2);
    float x[11];                                   // samples p0-2 .. p0+8 of the tile being convolved
    auto load_x = [&](int b, float (&xo)[11]) {
      const int row = (b / T) * 32 + r, t = b % T;
      const bool ok = b < ntile && row < args.n_rows;
      const float* src = args.signal + ((size_t)row * T + t) * kSig;
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        const int p = p0 - 2 + i;
        xo[i] = (ok && p >= 0 && p < kSig) ? src[p] : 0.f;
      }
    };
    load_x(blockIdx.x, x);
    for (int i = 0; i <= nloc; ++i) {
      // iteration i builds the image of local tile i (the matrix waves consume tile i-1)
      if (i < nloc) {
        float xn[11];
        load_x(blockIdx.x + (i + 1) * G, xn);        // next tile's samples: a whole iteration of lead
        float* flat = img + (i & 1) * IMG;
        if (cwv == 0) conv_positions<7>(P.conv, x, p0, r, flat, PLANE);
        else conv_positions<6>(P.conv, x, p0, r, flat, PLANE);
#pragma unroll
        for (int k = 0; k < 11; ++k) x[k] = xn[k];
      }
      __syncthreads();
    }
  } else {
    // ================================ MATRIX role ============================================
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
      im[(u >> 2) * PLANE + acc_row(reg, lane) * 4 + (u & 3)] = __builtin_fmaxf(acc[nt][reg], 0.f);
  }
  // dense2: 128 -> 32 (the same wave wrote the image; DS operations of one wave complete in order)
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
#pragma unroll
  for (int reg = 0; reg < 16; ++reg)
    im[(l31 >> 2) * PLANE + acc_row(reg, lane) * 4 + (l31 & 3)] = __builtin_fmaxf(a2[reg], 0.f);
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
    for (int reg = 0; reg < 16; ++reg) dst[acc_row(reg, lane) * 8] = __builtin_fmaxf(a3[reg], 0.f);
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
      lo[j] = __builtin_fmaxf(z[base + j], 0.f);
      hi[j] = __builtin_fmaxf(z[base + 4 + j], 0.f);
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
    for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaxf(a3[j], 0.f);
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
    featv[r * 17 + f] = __builtin_fmaxf(v, 0.f);
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
