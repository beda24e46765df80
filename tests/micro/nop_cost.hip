// What does an s_nop cost a LONE wave (the sweep's regime: one compute wave per SIMD, issue-bound)?  hipcc puts 8 of them into every step of
// the latency-form sweep (profiles/r05_sweep_step_isa.txt): wait states for DPP reads and behind packed-fp32 results.
// One wave per CU on a few CUs; per kind: cycles per loop body / instructions.      hipcc --offload-arch=gfx950 -O3 -o nop_cost nop_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
enum { ADD_DEP, ADD_DEP_NOP, ADD_IND, ADD_IND_NOP, PK_DEP, PK_DEP_NOP, PK_THEN_ADD, PK_NOP_THEN_ADD, DPP_DEP_NOP, DPP_DEP_NOP2X, MIX, MIX_NOP, NK };
static const char* kNames[NK] = {"v_add dependent", "v_add dependent + s_nop 0 each", "v_add x8 independent", "v_add x8 independent + s_nop 0 each", "v_pk_add dependent",
  "v_pk_add dependent + s_nop 0 each", "v_pk_mul ; v_add (reads it)", "v_pk_mul ; s_nop 0 ; v_add (reads it)", "v_mov_dpp dependent + s_nop 1", "v_mov_dpp dependent + s_nop 0 x2",
  "step-like mix, 16 instr", "step-like mix, 16 instr + 4 s_nop 0"};
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(float* out, unsigned long long* rec, int iters) {
  float a[8]; f2 p = {1.f + threadIdx.x, 2.f};
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + threadIdx.x * 0.001f + i;
  const float c = 1.0000001f; const f2 pc = {c, c};
  asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0" ::: "v10", "v11");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == ADD_DEP) { REP8(REP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));)) }
    if (KIND == ADD_DEP_NOP) { REP8(REP8(asm volatile("v_add_f32 %0, %0, %1\n s_nop 0" : "+v"(a[0]) : "v"(c));)) }
    if (KIND == ADD_IND) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (KIND == ADD_IND_NOP) { REP8(asm volatile("v_add_f32 %0, %0, %8\n s_nop 0\n v_add_f32 %1, %1, %8\n s_nop 0\n v_add_f32 %2, %2, %8\n s_nop 0\n v_add_f32 %3, %3, %8\n s_nop 0\n v_add_f32 %4, %4, %8\n s_nop 0\n v_add_f32 %5, %5, %8\n s_nop 0\n v_add_f32 %6, %6, %8\n s_nop 0\n v_add_f32 %7, %7, %8\n s_nop 0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (KIND == PK_DEP) { REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pc));)) }
    if (KIND == PK_DEP_NOP) { REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n s_nop 0" : "+v"(p) : "v"(pc));)) }
    if (KIND == PK_THEN_ADD) { REP8(REP8(asm volatile("v_pk_mul_f32 v[10:11], v[10:11], %1\n v_add_f32 %0, v10, %0" : "+v"(a[0]) : "v"(pc) : "v10", "v11");)) }
    if (KIND == PK_NOP_THEN_ADD) { REP8(REP8(asm volatile("v_pk_mul_f32 v[10:11], v[10:11], %1\n s_nop 0\n v_add_f32 %0, v10, %0" : "+v"(a[0]) : "v"(pc) : "v10", "v11");)) }
    if (KIND == DPP_DEP_NOP) { REP8(REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[0]));)) }
    if (KIND == DPP_DEP_NOP2X) { REP8(REP8(asm volatile("s_nop 0\n s_nop 0\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[0]));)) }
    if (KIND == MIX) { REP8(asm volatile("v_pk_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n v_fma_f32 %1, %1, %2, %2\n v_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %3\n v_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n"
                               "v_add_f32 %1, %1, %2\n v_fma_f32 %1, %1, %2, %2\n v_pk_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(p), "+v"(a[0]) : "v"(c), "v"(pc));) }
    if (KIND == MIX_NOP) { REP8(asm volatile("v_pk_add_f32 %0, %0, %3\n s_nop 0\n v_add_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n v_fma_f32 %1, %1, %2, %2\n s_nop 0\n v_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %3\n v_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n"
                               "s_nop 0\n v_add_f32 %1, %1, %2\n v_fma_f32 %1, %1, %2, %2\n v_pk_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %2\n s_nop 0\n v_pk_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(p), "+v"(a[0]) : "v"(c), "v"(pc));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = p.x + p.y;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) rec[blockIdx.x] = t1 - t0;
}
template <int KIND> static void run(int body, int nops) {
  const int grid = 8, iters = 4000;
  float* out; unsigned long long* rec; hipMalloc(&out, grid * 64 * 4); hipMalloc(&rec, grid * 8);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, rec, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), rec, grid * 8, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += double(v); c /= grid;
  printf("%-40s %4d VALU + %3d s_nop per body: %8.2f cycles per body = %5.2f per VALU instruction\n", kNames[KIND], body, nops, c / iters, c / iters / body);
  hipFree(out); hipFree(rec);
}
int main() {
  printf("# lone wave (one 64-thread workgroup per CU on 8 CUs), cycles by s_memtime\n");
  run<ADD_DEP>(64, 0); run<ADD_DEP_NOP>(64, 64); run<ADD_IND>(64, 0); run<ADD_IND_NOP>(64, 64); run<PK_DEP>(64, 0); run<PK_DEP_NOP>(64, 64);
  run<PK_THEN_ADD>(128, 0); run<PK_NOP_THEN_ADD>(128, 64); run<DPP_DEP_NOP>(64, 64); run<DPP_DEP_NOP2X>(64, 128); run<MIX>(128, 0); run<MIX_NOP>(128, 32);
  return 0;
}
