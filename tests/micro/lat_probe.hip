// Micro-benchmark (diagnostic, not a test): dependent-chain latency and independent issue cost of the instruction kinds that make up
// one step of the sweep (csrc/kernels_sweep2.hip, compute_band), for ONE wave alone on a SIMD of gfx950 -- the inputs of
// tests/micro/isa_chain.py's in-order model where MI355X_MICROARCH.md gives no figure.
//   hipcc --offload-arch=gfx950 -O2 -o lat_probe lat_probe.hip && ./lat_probe
// Each chain is N links of inline assembly inside a loop; cycles come from s_memtime (shader clock) and, for the effective clock,
// from the 100 MHz wall clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

enum Mode {
  ADD_DEP, ADD_IND4, MUL_ADD_DEP, PK_ADD_DEP, PK_FMA_DEP, PK_MUL_DEP, PK_ADD_IND4, SQRT_DEP, SQRT_ADD_DEP, CMP64_CND_DEP, CMPVCC_CND_DEP, DPP_SHR8_DEP, DPP_BCAST_DEP,
  DPP_QUAD_DEP, DPP_SHL1_DEP, DPP_IND4, MED3_DEP, CVT_I32_DEP, FRACT_DEP, FREXP_DEP, LSHL_ADD_DEP, MIN3I_DEP, RFL_SALU_DEP, RFL_CMP_BRANCH, DS_B32_DEP, DS_B64_DEP, DS_B128_DEP,
  DS_R2ST64_DEP, DS_WRITE_READ, DS_2XR2ST64_WAIT, SAVEEXEC_PAIR, DPP_NEWBCAST_DEP, DPP_NEWBCAST_IND4, DS_WRITE64_SAME8, DS_WRITE64_DISTINCT, NMODES
};
static const char* kNames[NMODES] = {
    "v_add_f32 dependent", "v_add_f32 4 independent chains (per instr)", "v_mul_f32 + v_add_f32 dependent (per instr)", "v_pk_add_f32 dependent", "v_pk_fma_f32 dependent",
    "v_pk_mul_f32 dependent", "v_pk_add_f32 4 independent chains (per instr)", "v_sqrt_f32 dependent", "v_sqrt_f32 + v_add_f32 dependent (per pair)",
    "v_cmp_lt_f32_e64 s[a:b] -> v_cndmask_e64 (per pair)", "v_cmp_lt_f32_e32 vcc -> v_cndmask_e32 (per pair)", "v_mov_b32_dpp row_shr:8 dependent", "v_mov_b32_dpp row_bcast:15 dependent",
    "v_mov_b32_dpp quad_perm dependent", "v_mov_b32_dpp row_shl:1 bound_ctrl dependent", "v_mov_b32_dpp 4 independent (per instr)", "v_med3_f32 dependent", "v_cvt_i32_f32 + v_cvt_f32_i32 (per pair)",
    "v_fract_f32 dependent", "v_frexp_exp_i32_f32 + v_cvt_f32_i32 (per pair)", "v_lshl_add_u32 dependent", "v_min3_i32 dependent", "v_readfirstlane -> v_mov from SGPR (per pair)",
    "v_readfirstlane -> s_cmp -> s_cbranch (not taken) -> v_add (per group)", "ds_read_b32 address-dependent chain", "ds_read_b64 address-dependent chain", "ds_read_b128 address-dependent chain",
    "ds_read2st64_b64 address-dependent chain", "ds_write_b64 + ds_write_b32 + ds_read_b32 of it + wait (per group)", "2 x ds_read2st64_b64 + s_waitcnt lgkmcnt(0) + use (per group)",
    "s_and_saveexec_b64 + s_or_b64 exec pair + v_add (per group)", "v_mov_b32_dpp row_newbcast:0 dependent", "v_mov_b32_dpp row_newbcast 4 independent (per instr)",
    "ds_write_b64, 8 distinct addresses (8 lanes each) (per instr)", "ds_write_b64, 64 distinct addresses (per instr)"};

template <int MODE>
__global__ void k(float* out, long long* res, int n) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0;   // every chained address stays 0 + lane offset
  __syncthreads();
  float a = threadIdx.x * 1e-3f + 1.5f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  float2 pa = make_float2(a, b), pb = make_float2(c, d), pc = make_float2(1.f, 1.f), pd = make_float2(2.f, 2.f);
  const float m = 1.0000001f, h = 0.5f;
  const float2 pm = make_float2(m, m);
  int ia = threadIdx.x, ib = 3;
  unsigned addr = (threadIdx.x & 63) * 16;
  const long long w0 = wall_clock64();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    if (MODE == ADD_DEP) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(h));) }
    if (MODE == ADD_IND4) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(h));) }
    if (MODE == MUL_ADD_DEP) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(m), "v"(h));) }
    if (MODE == PK_ADD_DEP) { REP32(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pm));) }
    if (MODE == PK_FMA_DEP) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pm), "v"(pc));) }
    if (MODE == PK_MUL_DEP) { REP32(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pm));) }
    if (MODE == PK_ADD_IND4) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pm));) }
    if (MODE == SQRT_DEP) { REP32(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));) }
    if (MODE == SQRT_ADD_DEP) { REP32(asm volatile("v_sqrt_f32 %0, %0\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == CMP64_CND_DEP) { REP32(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");) }
    if (MODE == CMPVCC_CND_DEP) { REP32(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %1, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
    if (MODE == DPP_SHR8_DEP) { REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(a));) }
    if (MODE == DPP_BCAST_DEP) { REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xe bank_mask:0x3" : "+v"(a));) }
    if (MODE == DPP_QUAD_DEP) { REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(a));) }
    if (MODE == DPP_SHL1_DEP) { REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));) }
    if (MODE == DPP_IND4) { REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                          "v_mov_b32_dpp %2, %4 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(h));) }
    if (MODE == DPP_NEWBCAST_DEP) { REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_newbcast:0 row_mask:0xf bank_mask:0x9" : "+v"(a));) }
    if (MODE == DPP_NEWBCAST_IND4) { REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0x9\n v_mov_b32_dpp %1, %4 row_newbcast:8 row_mask:0xf bank_mask:0x4\n"
                                          "v_mov_b32_dpp %2, %4 row_newbcast:8 row_mask:0xf bank_mask:0x8\n v_mov_b32_dpp %3, %4 row_newbcast:0 row_mask:0xf bank_mask:0x9"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(h));) }
    if (MODE == DS_WRITE64_SAME8) { const unsigned ad = (threadIdx.x >> 3) * 8u; REP32(asm volatile("ds_write_b64 %0, %1 offset:1024" :: "v"(ad), "v"(pa) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (MODE == DS_WRITE64_DISTINCT) { const unsigned ad = threadIdx.x * 8u; REP32(asm volatile("ds_write_b64 %0, %1 offset:1024" :: "v"(ad), "v"(pa) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (MODE == MED3_DEP) { REP32(asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(a) : "v"(b));) }
    if (MODE == CVT_I32_DEP) { REP32(asm volatile("v_cvt_i32_f32 %1, %0\n v_cvt_f32_i32 %0, %1" : "+v"(a), "+v"(ia));) }
    if (MODE == FRACT_DEP) { REP32(asm volatile("v_fract_f32 %0, %0" : "+v"(a));) }
    if (MODE == FREXP_DEP) { REP32(asm volatile("v_frexp_exp_i32_f32 %1, %0\n v_cvt_f32_i32 %0, %1" : "+v"(a), "+v"(ia));) }
    if (MODE == LSHL_ADD_DEP) { REP32(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(ia) : "v"(ib));) }
    if (MODE == MIN3I_DEP) { REP32(asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(ia) : "v"(ib));) }
    if (MODE == RFL_SALU_DEP) { REP32(asm volatile("v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20" : "+v"(ia) : : "s20");) }
    if (MODE == RFL_CMP_BRANCH) { REP32(asm volatile("v_readfirstlane_b32 s20, %0\n s_cmp_lt_i32 s20, 0\n s_cbranch_scc1 1f\n v_add_u32 %0, 1, %0\n1:" : "+v"(ia) : : "s20", "scc");) }
    if (MODE == DS_B32_DEP) { REP32(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));) }
    if (MODE == DS_B64_DEP) { uint2 v; REP32(asm volatile("ds_read_b64 %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "=&v"(v)); addr += v.x;) }
    if (MODE == DS_B128_DEP) { uint4 v; REP32(asm volatile("ds_read_b128 %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "=&v"(v)); addr += v.x;) }
    if (MODE == DS_R2ST64_DEP) { uint4 v; REP32(asm volatile("ds_read2st64_b64 %1, %0 offset0:0 offset1:1\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "=&v"(v)); addr += v.x;) }
    if (MODE == DS_WRITE_READ) { const uint2 w = make_uint2(0, 0); REP32(asm volatile("ds_write_b64 %0, %2 offset:8192\n ds_write_b32 %0, %1 offset:12288\n ds_read_b32 %1, %0 offset:12288\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, 0"
                                                                                : "+v"(addr), "+v"(ia) : "v"(w));) }
    if (MODE == DS_2XR2ST64_WAIT) { uint4 v, v2; REP32(asm volatile("ds_read2st64_b64 %1, %0 offset0:0 offset1:1\n ds_read2st64_b64 %2, %0 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)"
                                                                    : "+v"(addr), "=&v"(v), "=&v"(v2)); addr += v.x + v2.x;) }
    if (MODE == SAVEEXEC_PAIR) { REP32(asm volatile("v_cmp_gt_f32_e32 vcc, 0, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execnz 1f\n2:\n s_or_b64 exec, exec, s[20:21]\n v_add_f32 %0, %0, %1\n s_branch 3f\n1:\n s_branch 2b\n3:" : "+v"(a) : "v"(h) : "s20", "s21", "vcc");) }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  out[threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pb.y + pc.x + pd.x + float(ia) + float(addr);
  if (threadIdx.x == 0) { res[0] = t1 - t0; res[1] = w1 - w0; }
}

static const int kPerIter[NMODES] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};

template <int MODE>
void run(float* out, long long* res) {
  const int n = 2048;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, res, n); hipDeviceSynchronize(); }
  long long h[2]; hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost);
  const double cyc = (double)h[0] / n / kPerIter[MODE];
  const double ghz = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0;   // 100 MHz wall ticks -> ns
  printf("%-78s %7.2f cycles per link   (s_memtime ticks / wall: %.3f GHz)\n", kNames[MODE], cyc, ghz);
  if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
}

template <int M>
struct Runner { static void go(float* out, long long* res) { run<M>(out, res); Runner<M + 1>::go(out, res); } };
template <>
struct Runner<NMODES> { static void go(float*, long long*) {} };

int main() {
  float* out; long long* res;
  hipMalloc(&out, 4096); hipMalloc(&res, 64);
  printf("lone wave on one SIMD, gfx950; a 'link' is the instruction (or the group named) repeated 32 x 2048 times; loop overhead (~3 SALU per 32 links) included\n");
  Runner<0>::go(out, res);
  return 0;
}
