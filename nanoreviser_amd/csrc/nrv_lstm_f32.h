// Bi-LSTM layer kernel, f32 matrix instructions (NRV_PREC_F32), and the launch geometry shared by all
// Bi-LSTM layer kernels.
#pragma once
#include "nrv_common.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// Bi-LSTM layer kernel
// ---------------------------------------------------------------------------------------
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


}  // namespace nrv
