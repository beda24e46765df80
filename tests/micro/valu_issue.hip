// Microbenchmark (diagnostic, not a test): AGGREGATE vector-issue rate of one SIMD of gfx950 with 1 / 2 / 4 / 8 resident waves.
//
// Question (round-4 review, weak #3): how many cycles does a wave64 VALU instruction occupy a CDNA4 SIMD for -- 2 (the guide's
// "SIMD-32" reading) or 4 (what SQ_ACTIVE_INST_VALU and the wide-sweep A/B suggested)?  tests/micro/issue_rate.hip only timed wave 0.
// Here every CU of the chip holds exactly B workgroups of 4 * Wv waves (a 60 / 100 KB LDS request pins the number of workgroups
// per CU; waves of a workgroup land on SIMD (wave % 4)), i.e. N = B * Wv waves on every SIMD, each running the same loop of 64
// instructions in 8 independent dependency chains (inline asm: the exact opcode, no compiler rewriting).  Reported per kind and N:
//   cycles per instruction of ONE wave (s_memtime), the SIMD's aggregate rate = N * instrs / (last end - first start of its waves: they
//   do not finish together), and the clock the cycles were counted at (s_memtime against the 100 MHz s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f2v __attribute__((ext_vector_type(2)));

enum Kind { K_FMA, K_MUL, K_ADD, K_PKFMA, K_PKMUL, K_DPP, K_RSQ, K_SQRT, K_RCP, K_CNDMASK, K_CMP_CND, K_MED3, K_MAD_U24, K_LDS_B64, K_PERMSWAP, K_MIX_SWEEP,
            K_MAX, K_MIN3, K_CMP, K_MOV, K_AND, K_LSHL_ADD, K_CVT_I2F, K_CVT_F2I, K_FRACT, K_FREXP, K_ADD_U32, K_PKADD, K_FMAC, K_MUL_LIT, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_mov_b32 dpp(row_shr:1)", "v_rsq_f32", "v_sqrt_f32",
                                       "v_rcp_f32", "v_cndmask_b32", "v_cmp_lt + v_cndmask", "v_med3_f32", "v_mad_u32_u24", "ds_read_b64", "v_permlane32_swap",
                                       "mix 5 plain : 3 packed", "v_max_f32", "v_min3_f32", "v_cmp_lt_f32 (vcc)", "v_mov_b32", "v_and_b32", "v_lshl_add_u32",
                                       "v_cvt_f32_i32", "v_cvt_i32_f32", "v_fract_f32", "v_frexp_exp_i32_f32", "v_add_u32", "v_pk_add_f32", "v_fmac_f32 (VOP2)", "v_mul_f32 x, 0.5 (inline const)"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void k_issue(float* out, unsigned long long* rec, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  float a[8]; f2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 1e-3f * float(threadIdx.x + i); p[i] = f2v{a[i], a[i] + 0.5f}; }
  const float m = 0.99999f, c = 1e-6f;
  const f2v pm = f2v{m, m}, pc = f2v{c, c};
  lds[threadIdx.x] = a[0];
  __syncthreads();
  const unsigned ldsAddr = (unsigned)(size_t)(&lds[0]) + (lane & 31) * 8;
  __syncthreads();
  if (KIND == K_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(a[1]) : "vcc");
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
        REP8(X)
#undef X
      } else if (KIND == K_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
        REP8(X)
#undef X
      } else if (KIND == K_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_RSQ) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == K_CMP_CND) {   // 4 pairs = 8 instructions
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 4)]) : "vcc");
        X(0) X(1) X(2) X(3)
#undef X
      } else if (KIND == K_MED3) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_MAD_U24) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_LDS_B64) {
#define X(i) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(ldsAddr) : "memory");
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (KIND == K_PERMSWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 4)]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if (KIND == K_MAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_MIN3) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_CMP) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
        REP8(X)
#undef X
      } else if (KIND == K_MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (KIND == K_AND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == K_LSHL_ADD) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == K_CVT_I2F) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_CVT_F2I) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_FRACT) {
#define X(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_FREXP) {
#define X(i) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == K_PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        REP8(X)
#undef X
      } else if (KIND == K_FMAC) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (KIND == K_MUL_LIT) {
#define X(i) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == K_MIX_SWEEP) {   // 5 plain + 3 packed per 8: roughly the throughput-form step's VALU mix
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[1]) : "v"(pm), "v"(pc));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[2]) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[3]) : "v"(c));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[4]) : "v"(pm));
        asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[5]) : "v"(m), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[6]) : "v"(pm), "v"(pc));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[7]) : "v"(m), "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF;
    unsigned long long* r = rec + (size_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64) * 6;
    r[0] = t1 - t0; r[1] = w1 - w0; r[2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32); r[3] = w0; r[4] = t0; r[5] = t1;
  }
}

template <int KIND>
static void run(int wavesPerWgPerSimd, int wgPerCu, int ncu, int iters, int instrPerIter) {
  const int threads = 64 * 4 * wavesPerWgPerSimd;
  const int grid = ncu * wgPerCu;
  const size_t ldsBytes = wgPerCu == 1 ? 100 * 1024 : 60 * 1024;   // pins the number of resident workgroups per CU (160 KB of LDS)
  const int nw = grid * threads / 64;
  float* out; unsigned long long* rec;
  hipMalloc(&out, size_t(grid) * threads * 4); hipMalloc(&rec, size_t(nw) * 48);
  hipFuncSetAttribute((const void*)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), ldsBytes, 0, out, rec, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(size_t(nw) * 6);
  hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
  // per SIMD (xcc, se, sh, cu, simd): the waves it held, the first start and the last end on its own cycle counter.  A SIMD's waves do
  // not finish together (the arbiter favours the oldest), so its throughput is N * instructions / (last end - first start), NOT N / one
  // wave's own duration.
  const int nkeys = 16 * 8 * 2 * 16 * 4;
  std::vector<int> cnt(nkeys, 0);
  std::vector<unsigned long long> first(nkeys, ~0ull), last(nkeys, 0);
  double cyc = 0, wall = 0;
  for (int i = 0; i < nw; ++i) {
    cyc += double(h[6 * i]); wall += double(h[6 * i + 1]);
    const unsigned hw = unsigned(h[6 * i + 2]), xcc = unsigned(h[6 * i + 2] >> 32);
    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const int key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd;
    cnt[key]++; first[key] = std::min(first[key], h[6 * i + 4]); last[key] = std::max(last[key], h[6 * i + 5]);
  }
  int used = 0, lo = 1 << 30, hi = 0; double span = 0; int nfull = 0;
  const int N = wavesPerWgPerSimd * wgPerCu;
  for (int k = 0; k < nkeys; ++k) if (cnt[k]) {
    ++used; lo = std::min(lo, cnt[k]); hi = std::max(hi, cnt[k]);
    if (cnt[k] == N) { span += double(last[k] - first[k]); ++nfull; }
  }
  cyc /= nw; wall /= nw; span /= (nfull ? nfull : 1);
  const double instrs = double(iters) * instrPerIter;
  printf("%-32s N=%d waves/SIMD (%4d SIMDs, %d..%d waves each)  one wave: %6.2f cycles/instr   SIMD: one instr per %5.2f cycles = %5.3f wave-instr/cycle   clock %4.0f MHz\n",
         kNames[KIND], N, used, lo, hi, cyc / instrs, span / (N * instrs), N * instrs / span, cyc / (wall * 10.0) * 1000.0);
  hipFree(out); hipFree(rec);
}

template <int KIND>
static void sweep(int ncu, int ipi = 64) {
  const int iters = 2000;
  run<KIND>(1, 1, ncu, iters, ipi);
  run<KIND>(2, 1, ncu, iters, ipi);
  run<KIND>(4, 1, ncu, iters, ipi);
  run<KIND>(4, 2, ncu, iters, ipi);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("# %s, %d CUs, clock %d kHz; every CU holds the same number of workgroups; 64 instructions per loop iteration in 8 independent chains\n", p.gcnArchName, ncu, p.clockRate);
  sweep<K_FMA>(ncu); sweep<K_MUL>(ncu); sweep<K_ADD>(ncu); sweep<K_PKFMA>(ncu); sweep<K_PKMUL>(ncu); sweep<K_MIX_SWEEP>(ncu);
  sweep<K_DPP>(ncu); sweep<K_PERMSWAP>(ncu); sweep<K_CNDMASK>(ncu); sweep<K_CMP_CND>(ncu); sweep<K_MED3>(ncu); sweep<K_MAD_U24>(ncu);
  sweep<K_RSQ>(ncu); sweep<K_SQRT>(ncu); sweep<K_RCP>(ncu); sweep<K_LDS_B64>(ncu);
  sweep<K_FMAC>(ncu); sweep<K_MUL_LIT>(ncu); sweep<K_MAX>(ncu); sweep<K_MIN3>(ncu); sweep<K_CMP>(ncu); sweep<K_MOV>(ncu); sweep<K_AND>(ncu); sweep<K_ADD_U32>(ncu); sweep<K_LSHL_ADD>(ncu);
  sweep<K_CVT_I2F>(ncu); sweep<K_CVT_F2I>(ncu); sweep<K_FRACT>(ncu); sweep<K_FREXP>(ncu); sweep<K_PKADD>(ncu);
  return 0;
}
