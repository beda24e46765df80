// Do the sticky IEEE exception flags of a wave (TRAPSTS.EXCP) accumulate on gfx950 with traps disabled, and how soon after a
// VALU instruction can s_getreg_b32 see them?  (Candidate for the sweep's range guard: "no underflow / overflow / invalid /
// input-denormal flag raised during the step" instead of exponent tests on every operand of the exact fast forms.)
// Build: hipcc --offload-arch=gfx950 -O2 -o trapsts_probe trapsts_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
// EXCP bits: 0 invalid, 1 input denormal, 2 float div0, 3 overflow, 4 underflow, 5 inexact, 6 int div0
#define CLR "s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 7), 0\n s_nop 3\n"
#define GET(n) "s_nop " #n "\n s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 7)\n"
#define GET0 "s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 7)\n"

__global__ void k(unsigned* out) {
  const int lane = threadIdx.x;
  unsigned r; float t; int i = 0;
  typedef float f2p __attribute__((ext_vector_type(2)));
  const float tiny = 1e-30f, one = 1.0f, two = 2.0f, big = 1e30f, den = 1e-40f;
  // 0: exact op -> expect 0
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(one), "v"(two)); if (lane == 0) out[i] = r; ++i;
  // 1: underflow in every lane, read after 8 wait states
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(tiny), "v"(tiny)); if (lane == 0) out[i] = r; ++i;
  // 2: the same, read in the very next slot
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET0 : "=s"(r), "=v"(t) : "v"(tiny), "v"(tiny)); if (lane == 0) out[i] = r; ++i;
  // 3: one wait state
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(0) : "=s"(r), "=v"(t) : "v"(tiny), "v"(tiny)); if (lane == 0) out[i] = r; ++i;
  // 4: three wait states
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(2) : "=s"(r), "=v"(t) : "v"(tiny), "v"(tiny)); if (lane == 0) out[i] = r; ++i;
  // 5: underflow in lane 37 only
  { const float a = lane == 37 ? tiny : one; asm volatile(CLR "v_mul_f32 %1, %2, %2\n" GET(7) : "=s"(r), "=v"(t) : "v"(a)); if (lane == 0) out[i] = r; ++i; }
  // 6: overflow
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(big), "v"(big)); if (lane == 0) out[i] = r; ++i;
  // 7: packed multiply, underflow in the high half only
  { f2p a = {one, tiny}, d; asm volatile(CLR "v_pk_mul_f32 %1, %2, %2\n" GET(7) : "=s"(r), "=v"(d) : "v"(a)); if (lane == 0) out[i] = r; ++i; }
  // 8: packed fma whose tiny result is exact (2^-140 representable: 2^-70 * 2^-70 + 0) -> no underflow flag expected (IEEE: tiny AND inexact)
  { f2p a = {0x1p-70f, 0x1p-70f}, z = {0.f, 0.f}, d; asm volatile(CLR "v_pk_fma_f32 %1, %2, %2, %3\n" GET(7) : "=s"(r), "=v"(d) : "v"(a), "v"(z)); if (lane == 0) out[i] = r; ++i; }
  // 9: fma whose tiny result is inexact (1.0000001 * 2^-75)^2 -> underflow expected
  { const float a = 0x1.000002p-75f; asm volatile(CLR "v_fma_f32 %1, %2, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(a), "v"(0.f)); if (lane == 0) out[i] = r; ++i; }
  // 10: denormal input operand (1e-40 * 1.0): input-denormal flag?
  asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(den), "v"(one)); if (lane == 0) out[i] = r; ++i;
  // 11: inf * 0 -> invalid
  { const float inf = __builtin_inff(); asm volatile(CLR "v_mul_f32 %1, %2, %3\n" GET(7) : "=s"(r), "=v"(t) : "v"(inf), "v"(0.f)); if (lane == 0) out[i] = r; ++i; }
  // 12: v_rsq_f32 of a denormal / of zero
  asm volatile(CLR "v_rsq_f32 %1, %2\n" GET(7) : "=s"(r), "=v"(t) : "v"(den)); if (lane == 0) out[i] = r; ++i;
  asm volatile(CLR "v_rsq_f32 %1, %2\n" GET(7) : "=s"(r), "=v"(t) : "v"(0.f)); if (lane == 0) out[i] = r; ++i;
  // 14: flags survive unrelated instructions and accumulate (underflow, then overflow, read once)
  asm volatile(CLR "v_mul_f32 %1, %2, %2\n v_mov_b32 %1, %3\n v_mul_f32 %1, %3, %3\n" GET(7) : "=s"(r), "=&v"(t) : "v"(tiny), "v"(big)); if (lane == 0) out[i] = r; ++i;
  // 15: inactive lanes do not raise flags (only lane 0 active computes an exact product; lane 37's operands would underflow)
  { const float a = lane == 37 ? tiny : one; r = 0;
    asm volatile(CLR "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 1\n v_mul_f32 %1, %2, %2\n s_mov_b64 exec, s[20:21]\n" GET(7) : "=s"(r), "=v"(t) : "v"(a) : "s20", "s21"); if (lane == 0) out[i] = r; ++i; }
  if (lane == 0) out[31] = i;
}

int main() {
  unsigned* d; CK(hipMalloc(&d, 128)); CK(hipMemset(d, 0xff, 128));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); CK(hipDeviceSynchronize());
  unsigned h[32]; CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
  const char* names[] = {"exact product", "underflow, 8 wait states", "underflow, read in the next slot", "underflow, 1 wait state", "underflow, 3 wait states",
                         "underflow in one lane of 64", "overflow", "packed mul, underflow in the high half", "packed fma, tiny exact result", "fma, tiny inexact result",
                         "denormal input", "inf * 0", "v_rsq_f32(denormal)", "v_rsq_f32(0)", "underflow then overflow, one read", "underflowing lane inactive"};
  for (unsigned i = 0; i < h[31] && i < 16; ++i)
    printf("%-42s EXCP = 0x%02x  (%s%s%s%s%s%s)\n", names[i], h[i], h[i] & 1 ? "invalid " : "", h[i] & 2 ? "denorm-in " : "", h[i] & 4 ? "div0 " : "", h[i] & 8 ? "overflow " : "",
           h[i] & 16 ? "underflow " : "", h[i] & 32 ? "inexact" : "");
  return 0;
}
