// Shared types and helpers of the MI355X kernels (see nrv_kernels.h for the map).
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
__device__ __forceinline__ float buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
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

// ReLU of the dense layers: NaN in, NaN out, as NumPy / the oracle (and IEEE comparison semantics) have it -
// v_max_f32 would return the non-NaN operand and turn a poisoned window into finite garbage.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

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


// Cross-lane exchange through LDS inside ONE wave (no workgroup barrier): the hardware completes a
// wave's DS operations in order, but the compiler must be told that the stores of this lane and the
// loads of data written by other lanes are ordered - otherwise it may legally reorder or forward them.
// No instruction is emitted besides what the fences require at wavefront scope.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tanh for the LSTM cell: 1 - 2/(2^(2x log2 e) + 1) on the hardware exp2/rcp (1 ulp each): five
// instructions, no branch, exact limits at +-inf.  Absolute error <= ~1.5e-7 everywhere (for
// |x| -> 0 the RELATIVE error grows, which is immaterial here: the argument is a 200-500-term f32
// dot product whose own rounding noise is ~1e-6 absolute, and tanh' <= 1).
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
}

}  // namespace nrv
