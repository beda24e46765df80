// Does a DPP move whose bank / row masks leave some lanes unwritten need wait states behind a VALU instruction that wrote its DESTINATION
// (the "old" value those lanes keep)?  LLVM's hazard recognizer counts the tied old operand like the DPP source (2 wait states); the hardware
// hazard is about the SOURCE, which is read through the cross-lane network a stage early -- lanes a mask disables are simply not written.
// The latency-form step builds its proposals with three partial writes of one register pair in a row (row_newbcast:0 banks 0/3,
// row_newbcast:8 bank 2, row_bcast:15 bank 1: kernels_sweep2.hip compute_band), so with the rule it needs 4 wait states per step, without
// it none.  This probe runs exactly that sequence -- a VALU producer of the pair (packed add / two v_cndmask), then the 64-bit and 32-bit
// partial DPP writes back to back -- with and without s_nop 4 between the instructions, on changing data, one wave alone up to 16 waves per
// SIMD, and compares every lane bit for bit.  (The DPP SOURCES are written >= 3 slots earlier in both variants: that rule is not in question.)
// Build: hipcc --offload-arch=gfx950 -O2 -o dpp_old_probe dpp_old_probe.hip ; run: ./dpp_old_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// v[10:11] = the pair that is built up (old), v[12:13] = prev (DPP source), v[14:15] = t15, v[16:17] / v18, v19 = producer inputs
#define SEQ(N, PRODUCER)                                                                                      \
  "v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n s_nop 4\n" \
  "v_mov_b64_dpp v[14:15], v[12:13] row_newbcast:8 row_mask:0xf bank_mask:0x8 bound_ctrl:1\n s_nop 4\n"      \
  PRODUCER N                                                                                                  \
  "v_mov_b64_dpp v[10:11], v[12:13] row_newbcast:0 row_mask:0xf bank_mask:0x9\n" N                            \
  "v_mov_b64_dpp v[10:11], v[12:13] row_newbcast:8 row_mask:0xf bank_mask:0x4\n" N                            \
  "v_mov_b32_dpp v10, v14 row_bcast:15 row_mask:0xe bank_mask:0x2\n" N                                        \
  "v_mov_b32_dpp v11, v15 row_bcast:15 row_mask:0xe bank_mask:0x2\n s_nop 4\n"                               \
  "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n"
#define PROD_PK "v_pk_add_f32 v[10:11], v[16:17], v[18:19]\n"
#define PROD_CND "v_cmp_lt_f32 vcc, v16, v17\n s_nop 4\n v_cndmask_b32 v10, v18, v19, vcc\n v_cndmask_b32 v11, v19, v16, vcc\n"

template <int NOPS, int PROD>
__global__ void k_seq(const float* __restrict__ in, unsigned* __restrict__ out, int rounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float p0 = in[6 * i], p1 = in[6 * i + 1], a0 = in[6 * i + 2], a1 = in[6 * i + 3], b0 = in[6 * i + 4], b1 = in[6 * i + 5];
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    float x, y;
#define RUN(N, P) asm volatile(SEQ(N, P) : "=&v"(x), "=&v"(y) : "v"(p0), "v"(p1), "v"(a0), "v"(a1), "v"(b0), "v"(b1) \
                               : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "vcc")
    if (PROD == 0) { if (NOPS) RUN("s_nop 4\n", PROD_PK); else RUN("", PROD_PK); }
    else { if (NOPS) RUN("s_nop 4\n", PROD_CND); else RUN("", PROD_CND); }
#undef RUN
    acc = acc * 1664525u + (__float_as_uint(x) ^ (__float_as_uint(y) * 2654435761u));
    // next round's inputs depend on this round's results and on the lane (every lane and round a different pattern, all finite)
    const float fx = float(int(__float_as_uint(x) >> 9) % 4001 - 2000) / 64.0f, fy = float(int(__float_as_uint(y) >> 9) % 4001 - 2000) / 64.0f;
    p0 = a1 + fx; p1 = b0 - fy; a0 = fy + float(r & 7); a1 = fx * 0.5f + 1.0f; b0 = p0 * 0.25f + 2.0f; b1 = fabsf(fy) + 0.5f;
  }
  out[i] = acc;
}

template <int PROD>
static unsigned long long compare(int blocks, int threads, int rounds, const char* what) {
  const int n = blocks * threads;
  std::vector<float> h(6 * n);
  unsigned s = 4711u + blocks;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = float(int(s >> 8) % 20001 - 10000) / 256.0f; }
  float* d_in; unsigned *d_a, *d_b;
  CK(hipMalloc(&d_in, 6 * n * sizeof(float))); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4));
  CK(hipMemcpy(d_in, h.data(), 6 * n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_seq<0, PROD>), dim3(blocks), dim3(threads), 0, 0, d_in, d_a, rounds);
  hipLaunchKernelGGL((k_seq<1, PROD>), dim3(blocks), dim3(threads), 0, 0, d_in, d_b, rounds);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> a(n), b(n);
  CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  unsigned long long bad = 0;
  for (int i = 0; i < n; ++i) bad += a[i] != b[i];
  printf("%-52s threads %8d rounds %5d  differing results %llu\n", what, n, rounds, bad);
  CK(hipFree(d_in)); CK(hipFree(d_a)); CK(hipFree(d_b));
  return bad;
}

int main() {
  unsigned long long bad = 0;
  bad += compare<0>(1, 64, 20000, "packed-add producer, one wave alone");
  bad += compare<0>(256, 64, 4000, "packed-add producer, one wave per CU");
  bad += compare<0>(256 * 4, 256, 1000, "packed-add producer, 4 waves per SIMD");
  bad += compare<0>(256 * 4, 1024, 500, "packed-add producer, 16 waves per SIMD");
  bad += compare<1>(1, 64, 20000, "v_cndmask producers, one wave alone");
  bad += compare<1>(256, 64, 4000, "v_cndmask producers, one wave per CU");
  bad += compare<1>(256 * 4, 1024, 500, "v_cndmask producers, 16 waves per SIMD");
  if (bad == 0) printf("partial DPP writes: identical with and without wait states behind the writer of the old value\n");
  printf("mismatches %llu\n", bad);
  return bad ? 1 : 0;
}
