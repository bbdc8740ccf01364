// f16_mfma_probe.hip - what the f16 matrix pipe of gfx950 does with the operands the f16x2 mode
// (include/nanorev.h NRV_PREC_F16X2) feeds it: are f16 SUBNORMAL inputs kept or flushed, does the
// f32 -> f16 conversion round to nearest even and saturate or overflow to inf, and is the product of
// two f16 terms accumulated exactly in f32.  Prints one line per probe; no timing.
//   hipcc --offload-arch=gfx950 -O2 -o f16_mfma_probe f16_mfma_probe.hip && ./f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* av, const float* bv, float* out) {
  // every A element = av[0] (as f16), every B element = bv[0]: each C element = 16 * a * b
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av[0]; b[j] = (_Float16)bv[0]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  if (threadIdx.x == 0) out[1] = d[0];
  if (threadIdx.x == 0) { out[2] = (float)(_Float16)av[1]; out[3] = (float)(_Float16)av[2]; out[4] = (float)(_Float16)av[3]; }
}

int main() {
  float *da, *db, *dout;
  hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 64);
  struct { float a, b; const char* what; } cases[] = {
      {ldexpf(1.f, -20), 1024.f, "A = 2^-20 (f16 subnormal), B = 1024: 16ab = 2^-6 if subnormals are kept, 0 if flushed"},
      {ldexpf(1.f, -24), 16384.f, "A = 2^-24 (smallest f16 subnormal), B = 2^14: 16ab = 2^-6"},
      {ldexpf(3.f, -16), ldexpf(5.f, -16), "A = 3*2^-16, B = 5*2^-16 (both subnormal): 16ab = 240*2^-32"},
      {1.0009765625f, 1.0009765625f, "A = B = 1 + 2^-10: 16ab = 16(1 + 2^-9 + 2^-20) exact in f32"},
      {60000.f, 60000.f, "A = B = 60000: 16ab = 5.76e10 (f32 accumulate, no f16 overflow)"},
  };
  for (auto& cs : cases) {
    float ha[4] = {cs.a, 1.00048828125f /* 1 + 2^-11: tie -> 1.0 under RNE */, 1.00146484375f /* 1 + 3*2^-11: tie -> 1 + 2^-9 */,
                   70000.f /* > 65504 */};
    hipMemcpy(da, ha, 16, hipMemcpyHostToDevice);
    hipMemcpy(db, &cs.b, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
    float o[5];
    hipMemcpy(o, dout, 20, hipMemcpyDeviceToHost);
    printf("%s\n   32x32x16: %.10e   16x16x32: %.10e   expected %.10e\n", cs.what, o[0], o[1], 16.0 * (double)(float)(_Float16)cs.a * (double)(float)(_Float16)cs.b);
    if (&cs == &cases[0])
      printf("cvt f32->f16: 1+2^-11 -> %.10f (RNE: 1.0)   1+3*2^-11 -> %.10f (RNE: 1.001953125)   70000 -> %f\n", o[2], o[3], o[4]);
  }
  return 0;
}
