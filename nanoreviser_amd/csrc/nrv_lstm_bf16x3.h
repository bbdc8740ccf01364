// Bi-LSTM layer kernels on the bf16 matrix pipe with the exact three-term split (NRV_PREC_BF16X3):
// lstm_split_kernel (one timestep at a time) and lstm_pair_kernel (timesteps in pairs).
#pragma once
#include "nrv_lstm_f32.h"

namespace nrv {

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


}  // namespace nrv
