// H2D copy rate of page-locked host memory, by how it was page-locked: what bounds nrv_predict's window mode
// (2958 B per base over PCIe).   hipcc --offload-arch=gfx950 -O2 -o h2d_rate h2d_rate.hip && ./h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MB = 1 << 20;
  const size_t sizes[3] = {12 * MB, 48 * MB, 192 * MB};
  char* dev; CK(hipMalloc(&dev, 256 * MB));
  hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  for (int kind = 0; kind < 4; ++kind) {
    char* host = nullptr; void* raw = nullptr;
    const char* name = kind == 0 ? "hipHostMalloc default" : kind == 1 ? "hipHostMalloc non-coherent" : kind == 2 ? "malloc + hipHostRegister" : "aligned_alloc(2MB) + madvise-free hipHostRegister";
    if (kind == 0) CK(hipHostMalloc((void**)&host, 256 * MB, hipHostMallocDefault));
    else if (kind == 1) CK(hipHostMalloc((void**)&host, 256 * MB, hipHostMallocNonCoherent));
    else { raw = kind == 2 ? malloc(256 * MB + 4096) : aligned_alloc(2 * MB, 256 * MB); host = (char*)raw; memset(host, 1, 256 * MB); CK(hipHostRegister(host, 256 * MB, hipHostRegisterDefault)); }
    memset(host, 2, 256 * MB);
    for (size_t sz : sizes) {
      for (int streams = 1; streams <= 2; ++streams) {
        const int reps = (int)(1536 * MB / sz);
        for (int w = 0; w < 2; ++w) CK(hipMemcpyAsync(dev, host, sz, hipMemcpyHostToDevice, s[0]));
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; ++r) {
          if (streams == 1) CK(hipMemcpyAsync(dev, host + (r & 1) * sz % (64 * MB), sz, hipMemcpyHostToDevice, s[0]));
          else { CK(hipMemcpyAsync(dev, host, sz / 2, hipMemcpyHostToDevice, s[0])); CK(hipMemcpyAsync(dev + sz / 2, host + sz / 2, sz / 2, hipMemcpyHostToDevice, s[1])); }
        }
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        printf("%-52s %4zu MB x %3d, %d stream(s): %6.1f GB/s\n", name, sz / MB, reps, streams, sz * (double)reps / dt / 1e9);
      }
    }
    // D2H of a small result block, for the record
    { const double t0 = now(); for (int r = 0; r < 200; ++r) CK(hipMemcpyAsync(host, dev, 188 * 1024, hipMemcpyDeviceToHost, s[0])); CK(hipDeviceSynchronize());
      printf("%-52s D2H 188 KB x 200: %.1f us each\n", name, (now() - t0) / 200 * 1e6); }
    if (kind < 2) CK(hipHostFree(host)); else { CK(hipHostUnregister(host)); free(raw); }
  }
  return 0;
}
