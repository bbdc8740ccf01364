// Does the VGPR bank of the operands change the issue rate of v_fma_f32 / v_fmac_f32 / v_mul_f32?
// One wave per SIMD, 24 independent accumulators, registers named explicitly.  bank = register % 4.
//   hipcc --offload-arch=gfx950 -O3 -o valu_banks valu_banks.hip && ./valu_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

// body: 24 instructions "OP vACC, vA, vB[, vACC]" with ACC taken from a list
#define RUN(NAME, BODY)                                                                       \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                          \
    for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: "v1", "v2", "v3", "v5", "v6", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68", "v72", "v76", "v80", "v84", "v88", "v92", "v96", "v100", "v104", "v108"); } \
    if (iters < 0) out[threadIdx.x] = 1.f;                                                        \
  }

// all accumulators in bank 0 (v16, v20, ...), sources v1 (bank 1), v2 (bank 2): no two operands share a bank
#define L3(acc) "v_fma_f32 v" #acc ", v1, v2, v" #acc "\n"
RUN(k_fma_nobank, L3(16) L3(20) L3(24) L3(28) L3(32) L3(36) L3(40) L3(44) L3(48) L3(52) L3(56) L3(60) L3(64) L3(68) L3(72) L3(76) L3(80) L3(84) L3(88) L3(92) L3(96) L3(100) L3(104) L3(108))
// sources v5 and v1 (both bank 1) and accumulators in bank 1 as well (v17, v21, ...): all three reads hit one bank
#define L3c(acc) "v_fma_f32 v" #acc ", v1, v5, v" #acc "\n"
RUN(k_fma_samebank, L3c(17) L3c(21) L3c(25) L3c(29) L3c(33) L3c(37) L3c(17) L3c(21) L3c(25) L3c(29) L3c(33) L3c(37) L3c(17) L3c(21) L3c(25) L3c(29) L3c(33) L3c(37) L3c(17) L3c(21) L3c(25) L3c(29) L3c(33) L3c(37))
// consecutive accumulators v16..v39 (what a compiler allocates): banks rotate
RUN(k_fma_seq, L3(16) L3(17) L3(18) L3(19) L3(20) L3(21) L3(22) L3(23) L3(24) L3(25) L3(26) L3(27) L3(28) L3(29) L3(30) L3(31) L3(32) L3(33) L3(34) L3(35) L3(36) L3(37) L3(38) L3(39))
// two-source forms
#define L2(acc) "v_mul_f32 v" #acc ", v1, v2\n"
RUN(k_mul, L2(16) L2(20) L2(24) L2(28) L2(32) L2(36) L2(40) L2(44) L2(48) L2(52) L2(56) L2(60) L2(64) L2(68) L2(72) L2(76) L2(80) L2(84) L2(88) L2(92) L2(96) L2(100) L2(104) L2(108))
#define LM(acc) "v_fmac_f32 v" #acc ", v1, v2\n"
RUN(k_fmac, LM(16) LM(20) LM(24) LM(28) LM(32) LM(36) LM(40) LM(44) LM(48) LM(52) LM(56) LM(60) LM(64) LM(68) LM(72) LM(76) LM(80) LM(84) LM(88) LM(92) LM(96) LM(100) LM(104) LM(108))

template <class K>
void run(K kern, const char* name, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms  -> %.2f ns per instruction (one wave per SIMD)\n", name, ms, ms * 1e6 / (iters * 24.0));
  }
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  run(k_fma_nobank, "v_fma v,v,v  three banks", d);
  run(k_fma_samebank, "v_fma v,v,v  one bank", d);
  run(k_fma_seq, "v_fma v,v,v  consecutive acc", d);
  run(k_mul, "v_mul v,v", d);
  run(k_fmac, "v_fmac v,v (acc in bank 0)", d);
  return 0;
}
