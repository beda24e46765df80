// Exact, cheaper forms of the IEEE square root and of the division by a constant, as the sweep kernels use them
// (kernels_sweep2.hip).  Own header so that tools/exact_forms_check.hip tests exactly this code, exhaustively, on the GPU.
#pragma once
#include <hip/hip_runtime.h>
namespace pf {
typedef float f2p __attribute__((ext_vector_type(2)));
// PF_PK_ASM: the serial packed-fp32 chains as single inline-asm blocks (below).  That form makes an assumption about the HARDWARE
// (no wait state between a v_pk_*_f32 and a VALU instruction that reads its result), so it only exists for the one target it was
// established on -- gfx950 -- and only where the probe vouches for it: pf_create runs a short form of tests/micro/pk_hazard_probe.hip
// on the device and refuses to create a context on a mismatch.  Everywhere else, and with -DPF_SAFE_PK, the same arithmetic is plain
// packed-vector C++: the compiler emits the sequences with its own wait states (identical bits, 3 more issue slots per step).
#if defined(__gfx950__) && !defined(PF_SAFE_PK)
#define PF_PK_ASM 1
#else
#define PF_PK_ASM 0
#endif
// ---- chains of packed fp32 instructions as ONE inline-asm block ----------------------------------------------
// The compiler (ROCm 7.2 LLVM) separates every v_pk_*_f32 from an immediately dependent VALU instruction with an s_nop:
// its "dst_sel forwarding hazard" test reads bit 3 of src0_modifiers, which is DST_OP_SEL for VOP3 but op_sel_hi[0] for
// VOP3P -- set on every ordinary packed instruction (and it assumes the same of any asm statement).  The hardware hazard
// is about partial (16-bit) register writes; a packed fp32 result is two whole registers, and
// tests/micro/pk_hazard_probe.hip shows dependent packed chains give the same bits with and without the wait states
// (1 wave alone .. 16 waves per SIMD).  The sweep's step is issue-bound (every slot ~0.7 % of it) and these chains are
// serial, so they are issued as one asm block each: no wait states inside.  The one REAL hazard inside is kept by hand:
// a VALU instruction that reads the result of a transcendental one (v_rsq_f32) needs one wait state.  The blocks'
// results are read by ordinary compiler-generated VALU instructions (never directly by a DPP instruction, whose
// 2-wait-state hazard after a VALU write the compiler cannot see through an asm statement).

// ---- exact, cheaper forms of the two IEEE operations that dominate a lone wave's step ------------------
// (measured on MI355X: correctly rounded sqrtf ~118 cycles, division ~78 cycles per dependent use)
// sqrt: the core of LLVM's correctly rounded f32 sqrt (v_sqrt_f32 is within 1 ulp; test the two neighbours
// with exact FMA residuals) without its denormal pre-scaling: valid for x == 0 or x >= 2^-96, finite.
__device__ __forceinline__ float sqrt_core(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = (rm <= 0.0f) ? sm : s;
  s = (rp > 0.0f) ? sp : s;
  return s;
}
// a / c for a constant c with y = (float)(1.0 / (double)c), the correctly rounded reciprocal: ONE FMA refinement step
// (Markstein).  q0 = RN(a*y) is within 1.5 ulp of a/c, so the residual r0 = a - q0*c is exact and
// q0 + r0*y = a/c * (1 + d) with |d| <= 1.5 * 2^-24 ulp.  That rounds to RN(a/c) unless a/c lies that close to a midpoint
// between two floats:
//  * c = an image width (integer of k <= 21 bits): a/c - midpoint = (integer != 0) / (c * 2^(k+1)), i.e. at least
//    2^-(k+1) ulp away -- never that close;
//  * c = 0.001f (24 significant bits, the bound above does not help): checked bit-exact against IEEE division for
//    EVERY float a with |a| in [2^-100, 2^100] (3.36e9 inputs), plus 4.8e9 random (a, c) with c = 2..12000:
//    tests/micro/divtest.c, 0 mismatches.  a == 0 is exact.
//    Both again, exhaustively and on the device, by tools/exact_forms_check (tests/test_gpu_exact_forms.py).
// Below 2^-100 the residual underflows: the callers' range guard (emin) keeps every numerator above 2^-95.
// a == -0 gives +0 where IEEE gives -0: no caller can pass it (the numerators are c*|x| and differences e1 - e0 of
// energies that are sums of square roots, i.e. never -0)
__device__ __forceinline__ float div_core(float a, float c, float y) {
  const float q0 = a * y;
  const float r0 = __builtin_fmaf(-q0, c, a);
  return __builtin_fmaf(r0, y, q0);
}
// two quotients by the same constant at once (packed fp32 FMA)
__device__ __forceinline__ f2p div_core2(f2p a, float c, float y) {
  const f2p cc = {c, c}, yy = {y, y};
  const f2p q0 = a * yy;
  const f2p r0 = __builtin_elementwise_fma(-q0, cc, a);
  return __builtin_elementwise_fma(r0, yy, q0);
}
// Two correctly rounded square roots at once: the reciprocal-square-root form (v_rsq_f32 is within 1 ulp; Newton step on
// (s, h = 1/(2s)) and a final exact-residual correction, all packed FMAs): 10 instructions for the pair instead of 16
// for two sqrt_core.  x == 0 gives 0 (the 2^-126 added to the v_rsq operand changes no x >= 2^-102 and keeps rsq(0)
// finite).  Verified on the MI355X against the correctly rounded sqrtf for EVERY float in [2^-96, 2^100] and 0
// (tests/micro/sqrt_exhaust.hip, run by tests/test_gpu_exact_forms.py).
// Also returns frexp_exp(x.x) (the callers' range guard wants it): it is the independent instruction that fills the wait
// state a VALU instruction needs after the v_rsq_f32 whose result it reads.
// the eight operations as plain packed-vector C++: the compiler schedules them and inserts its wait states
__device__ __forceinline__ f2p sqrt_core2_safe(f2p x, int& exp_x0) {
  const f2p xt = x + f2p{0x1p-126f, 0x1p-126f};
  const f2p r = {__builtin_amdgcn_rsqf(xt.x), __builtin_amdgcn_rsqf(xt.y)};
  const f2p half = {0.5f, 0.5f};
  exp_x0 = __builtin_amdgcn_frexp_expf(x.x);
  f2p s = x * r, h = r * half;
  const f2p e = __builtin_elementwise_fma(-h, s, half);
  h = __builtin_elementwise_fma(h, e, h);
  s = __builtin_elementwise_fma(s, e, s);
  const f2p d = __builtin_elementwise_fma(-s, s, x);
  return __builtin_elementwise_fma(d, h, s);
}
__device__ __forceinline__ f2p sqrt_core2(f2p x, int& exp_x0) {
#if !PF_PK_ASM
  return sqrt_core2_safe(x, exp_x0);
#else
  const f2p xt = x + f2p{0x1p-126f, 0x1p-126f};
  const f2p r = {__builtin_amdgcn_rsqf(xt.x), __builtin_amdgcn_rsqf(xt.y)};
  f2p s, h, e, out;
  const float x0 = x.x;
  asm("v_frexp_exp_i32_f32 %4, %7\n\t"                  // (r.y comes from the transcendental unit one slot ago)
      "v_pk_mul_f32 %0, %5, %6\n\t"                     // s = x * r
      "v_pk_mul_f32 %1, %6, 0.5 op_sel_hi:[1,0]\n\t"    // h = r / 2
      "v_pk_fma_f32 %2, %1, %0, 0.5 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // e = 0.5 - h * s
      "v_pk_fma_f32 %1, %1, %2, %1\n\t"                 // h = h + h * e
      "v_pk_fma_f32 %0, %0, %2, %0\n\t"                 // s = s + s * e
      "v_pk_fma_f32 %2, %0, %0, %5 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // d = x - s * s (exact)
      "v_pk_fma_f32 %3, %2, %1, %0"                      // s + d * h, correctly rounded
      : "=&v"(s), "=&v"(h), "=&v"(e), "=&v"(out), "=&v"(exp_x0) : "v"(x), "v"(r), "v"(x0));
  return out;
#endif
}
// Sum of the squares of a packed difference, (a - b).x^2 + (a - b).y^2, as ONE block: subtraction, product, and the sum of the product's two halves as a
// packed add that reads it with swapped halves (lo = x^2 + y^2, hi = y^2 + x^2: the same addition as the scalar v_add_f32 of the two halves).  Three
// instructions as before, but no wait state between them (round 5, tests/micro/nop_cost.hip: an s_nop costs a lone wave 4 cycles -- as much as an instruction).
__device__ __forceinline__ float sumsq_diff2(f2p a, f2p b) {
#if PF_PK_ASM
  f2p r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
  return r.x;
#else
  const f2p d = a - b, d2 = d * d;
  return d2.x + d2.y;
#endif
}
__device__ __forceinline__ float sumsq_diff2_safe(f2p a, f2p b) { const f2p d = a - b, d2 = d * d; return d2.x + d2.y; }
// The same with the subtrahend still to be summed: (a - (p + l)).x^2 + (a - (p + l)).y^2 -- the bilinear sample's last addition rides in the block.
__device__ __forceinline__ float sumsq_diff2_sum(f2p a, f2p p, f2p l) {
#if PF_PK_ASM
  f2p r;
  asm("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]"
      : "=&v"(r) : "v"(a), "v"(p), "v"(l));
  return r.x;
#else
  const f2p d = a - (p + l), d2 = d * d;
  return d2.x + d2.y;
#endif
}
__device__ __forceinline__ float sumsq_diff2_sum_safe(f2p a, f2p p, f2p l) { const f2p d = a - (p + l), d2 = d * d; return d2.x + d2.y; }
}  // namespace pf
