// Does a hipGraph shorten the gap between two DEPENDENT kernels of one stream on this stack?  (tests/micro, GPU box)
//   hipcc --offload-arch=gfx950 -O2 tests/micro/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_tiny(int* p, int spin) { if (threadIdx.x == 0) { int v = *p; for (int i = 0; i < spin; ++i) v = v * 3 + 1; *p = v; } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  int* d; CK(hipMalloc(&d, 256)); CK(hipMemset(d, 0, 256));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 1000;
  for (int spin : {0, 2000}) {
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(a, st));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d, spin);
      CK(hipEventRecord(b, st));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("spin %4d stream: %.2f us per kernel on the GPU, host enqueue %.2f us per launch\n", spin, ms * 1000 / N, std::chrono::duration<double>(t1 - t0).count() * 1e6 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(a, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("spin %4d graph : %.2f us per kernel on the GPU, host launch of the whole graph %.1f us\n", spin, ms * 1000 / N, std::chrono::duration<double>(t1 - t0).count() * 1e6);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
