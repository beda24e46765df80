// Hand-off latency between two workgroups through a word in HBM (agent-scope relaxed atomics, the sweep's granule pattern):
// does it matter whether the two workgroups sit on the same XCD?   (tests/micro, GPU box)
//   hipcc --offload-arch=gfx950 -O2 tests/micro/xcd_pingpong.hip -o tests/micro/xcd_pingpong && tests/micro/xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_pp(unsigned long long* word, int* xcc, long long* ticks, int a, int b, int iters) {
  // workgroup a and workgroup b play ping-pong on word[0]; every other workgroup just reports its XCD
  const int wg = blockIdx.x;
  if (threadIdx.x == 0) {
    int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[wg] = id & 0xf;
  }
  if (threadIdx.x != 0 || (wg != a && wg != b)) return;
  const bool first = wg == a;
  long long t0 = 0;
  for (int i = 0; i < iters; ++i) {
    const unsigned long long want = 2ull * i + (first ? 0 : 1);        // value this side waits for
    if (first && i == 0) { t0 = wall_clock64(); }
    else { int spins = 0; while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) { if (++spins > (1 << 26)) return; } }
    __hip_atomic_store(word, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (first) ticks[0] = wall_clock64() - t0;
}
int main() {
  unsigned long long* word; int* xcc; long long* ticks;
  const int nwg = 64, iters = 20000;
  CK(hipMalloc(&word, 256)); CK(hipMalloc(&xcc, nwg * sizeof(int))); CK(hipMalloc(&ticks, 64));
  int hx[64];
  for (int b : {8, 16, 1, 2, 4, 9}) {          // with round-robin dispatch, workgroup 8 and 16 share workgroup 0's XCD
    CK(hipMemset(word, 0, 256)); CK(hipMemset(ticks, 0, 64));
    hipLaunchKernelGGL(k_pp, dim3(nwg), dim3(64), 0, 0, word, xcc, ticks, 0, b, iters);
    CK(hipDeviceSynchronize());
    long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost));
    printf("workgroup 0 (XCD %d) <-> workgroup %2d (XCD %d): %.0f ns per one-way hand-off\n", hx[0], b, hx[b], t * 10.0 / (2.0 * iters));
  }
  return 0;
}
