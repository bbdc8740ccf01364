// Bi-LSTM layer kernel on the f16 matrix pipe with the scaled two-term split (NRV_PREC_F16X2).
#pragma once
#include "nrv_lstm_f32.h"

// -DNRV_EXP=<bits> (tools/lstm_exp.sh D:...; 0 in the product): timing experiments that compile parts of cnn_r_kernel
// out (nrv_cnn_r.h) - results WRONG by construction, only time means something, and even that only with care: the
// chip's clock follows the kernel's power, which follows the DATA.
namespace nrv {

// ---------------------------------------------------------------------------------------
// f16x2: f32-grade products from THREE matrix products instead of bf16x3's six.
//
// A value x, first multiplied by a power of two 2^s chosen so that the tensor's largest magnitude
// sits near 2^14 (exact; the f16 range is what made the unscaled form lose bits), is written as
// hi + lo with two f16 terms, both rounded to nearest even: |x 2^s - hi - lo| <= 2^-22 |x 2^s| for
// values whose lo term is a normal f16, and <= 2^-25 absolute (the f16 subnormal spacing) for the
// small ones - i.e. <= 2^-39 of the tensor's maximum, far below the rounding of an f32 dot product.
// a*b is formed as  lo*hi + hi*lo + hi*hi  on v_mfma_f32_32x32x16_f16 with f32 accumulation; the
// dropped lo*lo is below 2^-22 of the product.  tools/bf16_split_study.py (modes h3s / h3sz): max
// |dp| vs fp64 equal to the f32 pipe's on the fixture windows of both species, no argmax flips, with
// or without subnormal flushing.
//
// What makes it cheap on the device, beyond the halved MFMA count:
//   * activations travel BETWEEN kernels already split: every producer writes, instead of the f32
//     tile [kq][32 rows][4 f32], the same 512-byte chunks as [k-block][term hi|lo][half][32 rows][8 f16]
//     (same bytes, same chunk addressing: chunk index 4*kb + 2*term + half takes the place of kq), so
//     one wave-wide 16-byte load IS an A fragment of one term and the consumer splits nothing.  The
//     split happens once per element in the producer's epilogue instead of once per (element, hidden
//     group) in every consumer;
//   * the power-of-two scales are folded into constants that exist anyway: the producer's BatchNorm
//     scale/shift (x 2^s_out), the packed weights (x 2^(E - s_in)), the bias (x 2^E) and the gate
//     constants (x 2^-E), E being the layer's accumulator exponent (nrv_api.hip: plan_f16x2);
//   * only the recurrent operand h_{t-1} (|h| < 1, kept in LDS as f32 x 2^13) is split in the loop:
//     24 VALU ops per 12 MFMAs, in the shadow of the previous unit.
//
// Structure: as lstm_split_kernel (a wave owns 32 hidden units x 4 gates x R row tiles, c in
// registers, h through a double-buffered LDS image, one barrier per step), but the k-blocks of a
// step are FULLY unrolled: ring slots, operand types (pre-split input block / raw recurrent block)
// and the counted waits are static, weights and activations are requested a fixed number of entries
// ahead, and the requests run across the step boundary (the first blocks of step s+1 do not depend on h_s).
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Split2 { f16x8 t[2]; };

// x[0..7] (already scaled) -> hi + lo
__device__ __forceinline__ Split2 split2(const f32x4& lo4, const f32x4& hi4) {
  Split2 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? lo4[j] : hi4[j - 4];
    const _Float16 h = (_Float16)x;
    o.t[0][j] = h;
    o.t[1][j] = (_Float16)(x - (float)h);
  }
  return o;
}

// The same split with the low terms as v_fma_mix_f32(hi, -1, x): four instructions per pair instead of six.  (With the
// low terms converted by the same instruction, v_fma_mixlo / mixhi_f16 - three per pair - the signal branch and the
// 32->64 layer were measured SLOWER, r03.)  m1 is
// -1.0f in a register the compiler cannot see through (neg_one_opaque(), once per kernel): written as x - (float)h
// the subtraction needs v_cvt_f32_f16 first.
__device__ __forceinline__ float neg_one_opaque() {
  float m1;
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  return m1;
}
__device__ __forceinline__ Split2 split2(const f32x4& lo4, const f32x4& hi4, const float m1) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  Split2 o;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const float x0 = j < 4 ? lo4[j] : hi4[j - 4], x1 = j < 4 ? lo4[j + 1] : hi4[j - 3];
    const h2_t h = __builtin_convertvector(f2_t{x0, x1}, h2_t);
    const h2_t l = __builtin_convertvector(f2_t{__builtin_fmaf((float)h[0], m1, x0), __builtin_fmaf((float)h[1], m1, x1)}, h2_t);
    o.t[0][j] = h[0]; o.t[0][j + 1] = h[1];
    o.t[1][j] = l[0]; o.t[1][j + 1] = l[1];
  }
  return o;
}

__device__ __forceinline__ f32x16 mfma_f16(const f16x8& a, const f16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// gate nonlinearities of an accumulator that holds z * 2^E: the descale rides on constants
template <int ACT>
__device__ __forceinline__ float gate_act_scaled(float acc, float d, float d02) {
  if constexpr (ACT == 0) return __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(acc, d02, 0.5f), 0.0f), 1.0f);
  else return sigmoid_exact(acc * d);
}
__device__ __forceinline__ float tanh_fast_scaled(float acc, float d2) {    // d2 = 2 log2(e) * 2^-E
  const float e = __builtin_amdgcn_exp2f(acc * d2);
  return __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
}

constexpr float kHScale = 8192.0f;      // h_{t-1} in LDS is h * 2^13 (|h| < 1)

struct LstmH2ModelParams {
  const void* wsplit;     // [dir][hg][kb][gate][term 2][64 lanes][8 f16], already x 2^(E - s_in); kb: input then recurrent
  const float* bias;      // [dir][hg][gate][32], x 2^E
  const float* out_scale; // [2H]  BatchNorm scale x 2^(s_out - 13)   (the image holds h * 2^13)
  const float* out_shift; // [2H]  BatchNorm shift x 2^s_out
  ActView in0, in1;       // split-plane buffers (chunk index = 4*kb + 2*term + half)
  float* out;             // split planes [tiles][T][2H/16][term][half][32][8 f16], or f32 tiles when OUT_F32
  float descale;          // 2^-E
};
struct LstmH2Args {
  LstmH2ModelParams m[2];
  int T;
  int n_rows;
  int n_blk;
};

}  // namespace nrv
