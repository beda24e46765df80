// Diagnostic: does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (out) *out = 1; }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4);
  for (int flags : {0, 1}) {
    hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 8; ++i) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags, 200000ll /* 2 ms */, d);
    hipStreamSynchronize(st);
    printf("8 x 2 ms single-block kernels on one stream, flags=%d: %.2f ms\n", flags, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  return 0;
}
