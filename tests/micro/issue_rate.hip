// Microbenchmark (diagnostic, not a test): VALU issue behaviour of a lone wave vs co-resident waves on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int n) {
  float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  const float m = 1.0000001f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 32; ++u) a = a * m + 0.5f; }                       // 64 dependent (mul,add)
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { a = a * m + 0.5f; b = b * m + 0.5f; c = c * m + 0.5f; d = d * m + 0.5f; } }     // 64, 4 independent chains
    if (MODE == 6) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { a = (a < b) ? a * m : b; b = (b < a) ? b + 0.5f : a; } }     // cmp+cndmask chains: 16*(cmp,mul,cnd,cmp,add,cnd)=96
    if (MODE == 2) { a = sqrtf(a) + 1.0f; a = sqrtf(a) + 1.0f; }                                                       // IEEE sqrt chain
    if (MODE == 3) { a = a / b + 1.0f; a = a / b + 1.0f; }                                                             // IEEE div chain
    if (MODE == 4) { a = __builtin_amdgcn_sqrtf(a) + 1.0f; a = __builtin_amdgcn_sqrtf(a) + 1.0f; a = __builtin_amdgcn_sqrtf(a) + 1.0f; a = __builtin_amdgcn_sqrtf(a) + 1.0f; }  // raw v_sqrt chain
    if (MODE == 5) { a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x101, 0xf, 0xf, false)) + 1.0f;
                     a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x101, 0xf, 0xf, false)) + 1.0f; }  // dpp chain
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int instr_per_iter) {
  float* out; long long* cyc; hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
  const int n = 4096;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, n); hipDeviceSynchronize();
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, n); hipDeviceSynchronize();
  long long h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("%-28s waves=%2d  cycles/iter(wave0)=%7.1f  per-instr=%5.2f\n", name, threads / 64, (double)h[0] / n, (double)h[0] / n / instr_per_iter);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int th : {64, 512}) {
    run<0>("dependent mul+add (64)", th, 64);
    run<1>("4 independent chains (64)", th, 64);
    run<6>("cmp/cndmask chain (96)", th, 96);
    run<2>("IEEE sqrtf+add x2", th, 2);
    run<3>("IEEE div+add x2", th, 2);
    run<4>("raw v_sqrt+add x4 (8)", th, 8);
    run<5>("dpp+add x2 (4)", th, 4);
  }
  return 0;
}
