// Is there a real read-after-write hazard between a packed-fp32 VALU instruction and an immediately dependent VALU
// instruction on gfx950?  The compiler (ROCm 7.2 LLVM) puts an s_nop 0 between them: its "dst_sel forwarding hazard" test
// reads bit 3 of src0_modifiers, which is DST_OP_SEL for VOP3 but op_sel_hi[0] for VOP3P -- set by default on every v_pk_*.
// The ISA's hazard is about PARTIAL (16-bit) register writes; v_pk_*_f32 writes two whole registers.
// This probe runs the same dependent chain twice -- back to back, and with s_nop 3 after every instruction -- on random
// data, one wave alone and 16 waves per SIMD, and compares all results bit for bit.
// Build: hipcc --offload-arch=gfx950 -O2 -o pk_hazard_probe pk_hazard_probe.hip ; run: ./pk_hazard_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f2p __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// fixed registers: x = v[10:11], y = v[12:13], a = v[14:15], b = v[16:17], scalars v18..v20
#define CHAIN(N)                                                                         \
  "v_mov_b32 v10, %7\n v_mov_b32 v11, %8\n v_mov_b32 v12, %9\n v_mov_b32 v13, %10\n s_nop 4\n" \
  "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n" N                                         \
  "v_pk_fma_f32 v[16:17], v[14:15], v[14:15], v[10:11]\n" N                               \
  "v_pk_add_f32 v[14:15], v[16:17], v[14:15]\n" N                                         \
  "v_pk_fma_f32 v[16:17], v[14:15], v[12:13], v[16:17] neg_lo:[1,0,0] neg_hi:[1,0,0]\n" N \
  "v_pk_mul_f32 v[14:15], v[16:17], v[16:17]\n" N                                         \
  "v_pk_add_f32 v[16:17], v[14:15], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n" N            \
  "v_pk_fma_f32 v[14:15], v[16:17], v[10:11], v[12:13]\n" N                               \
  "v_pk_fma_f32 v[16:17], v[14:15], v[16:17], v[14:15]\n" N                               \
  "v_pk_mul_f32 v[14:15], v[16:17], v[14:15]\n" N                                         \
  "v_mul_f32 v18, v14, v15\n" N                                                           \
  "v_pk_add_f32 v[16:17], v[14:15], v[16:17] neg_lo:[0,1] neg_hi:[0,1]\n" N               \
  "v_cvt_i32_f32 v19, v17\n" N                                                            \
  "v_pk_fma_f32 v[14:15], v[16:17], v[12:13], v[10:11]\n" N                               \
  "v_med3_f32 v20, v14, v18, v15\n" N                                                     \
  "v_pk_mul_f32 v[16:17], v[14:15], v[12:13]\n" N                                         \
  "v_frexp_exp_i32_f32 v18, v16\n" N                                                      \
  "v_pk_add_f32 v[14:15], v[16:17], v[10:11]\n" N                                         \
  "v_rsq_f32 v10, v14\n s_nop 4\n"                                                       \
  "v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n v_mov_b32 %2, v16\n v_mov_b32 %3, v17\n v_mov_b32 %4, v18\n v_mov_b32 %5, v19\n v_mov_b32 %6, v20\n"

template <bool NOPS>
__global__ void k_chain(const f2p* __restrict__ in, float* __restrict__ out, int rounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  f2p x = in[2 * i], y = in[2 * i + 1];
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    float a0, a1, b0, b1, c, d, e;
    if (NOPS) asm volatile(CHAIN("s_nop 3\n") : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y)
                           : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
    else      asm volatile(CHAIN("") : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y)
                           : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
    auto tame = [](float v) { return (v == v && __builtin_fabsf(v) < 1e30f) ? v : 1.0f; };
    acc += tame(a0) + tame(a1) + tame(b0) + tame(b1) + float(__float_as_int(c) & 0xffff) + float(__float_as_int(d) & 0xffff) + tame(e);
    const float nx = __builtin_fminf(__builtin_fmaxf(y.y + 0.25f * float(r & 3) + tame(b0) * 1e-6f, -4.f), 4.f);
    const float ny = __builtin_fminf(__builtin_fmaxf(x.y * 0.5f + tame(a1) * 1e-6f, -4.f), 4.f);
    x = f2p{nx, x.x}; y = f2p{ny, y.x};
  }
  out[i] = acc;
}

static unsigned long long compare(int blocks, int threads, int rounds, const char* what) {
  const int n = blocks * threads;
  std::vector<float> h(4 * n);
  unsigned s = 12345u + blocks;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = float(int(s >> 8) % 20001 - 10000) / 4096.0f; }
  f2p* d_in; float *d_a, *d_b;
  CK(hipMalloc(&d_in, 4 * n * sizeof(float))); CK(hipMalloc(&d_a, n * sizeof(float))); CK(hipMalloc(&d_b, n * sizeof(float)));
  CK(hipMemcpy(d_in, h.data(), 4 * n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_chain<false>, dim3(blocks), dim3(threads), 0, 0, d_in, d_a, rounds);
  hipLaunchKernelGGL(k_chain<true>, dim3(blocks), dim3(threads), 0, 0, d_in, d_b, rounds);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> a(n), b(n);
  CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  unsigned long long bad = 0, nonfinite = 0;
  for (int i = 0; i < n; ++i) { bad += a[i] != b[i]; nonfinite += (a[i] & 0x7f800000u) == 0x7f800000u; }
  printf("%-34s threads %8d rounds %5d  differing results %llu (non-finite %llu)\n", what, n, rounds, bad, nonfinite);
  CK(hipFree(d_in)); CK(hipFree(d_a)); CK(hipFree(d_b));
  return bad;
}

int main() {
  unsigned long long bad = 0;
  bad += compare(1, 64, 20000, "one wave alone");
  bad += compare(256, 64, 2000, "one wave per CU");
  bad += compare(256 * 4, 1024, 200, "16 waves per SIMD");
  bad += compare(256 * 8, 256, 500, "4 waves per SIMD, 2 blocks/CU");
  printf("packed-fp32 back-to-back dependent issue: %s\n", bad ? "RESULTS DIFFER (hazard is real)" : "identical with and without s_nop");
  return bad ? 1 : 0;
}
