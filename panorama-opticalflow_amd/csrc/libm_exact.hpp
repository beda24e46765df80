// Bit-exact device restatements of the two libm functions on the blend path (combineNovelViews,
// CPU/OpticalFlow.cpp:69-76: `tanhf(colorDiff * kColorDiffCoef)` and the two fp64 `exp(...)` of the softmax).
//
// The reference gets these from the host's libm; the output panorama is bytes and a last-place difference in either
// function flips `(unsigned char)(c * wL + c * wR)` wherever the two warped colours agree (the sum is then within an
// ulp of the integer c).  A result that is byte-identical to the CPU path therefore needs the SAME roundings, not
// merely a <1 ulp function.  These are the algorithms of glibc 2.35 on x86-64 (the libm of the reference's platform
// and of this image, host and GPU box alike), operation for operation:
//
//   * tanhf / expm1f: fdlibm's float algorithms (sysdeps/ieee754/flt-32/s_tanhf.c, s_expm1f.c), plain fp32
//     add / mul / div in source order (the x86-64 build has no FMA variant of them; checked in the disassembly);
//   * exp (fp64): sysdeps/ieee754/dbl-64/e_exp.c (Szabolcs Nagy's table-driven algorithm, N = 128), in the form the
//     `__exp_fma` ifunc variant executes on every x86-64 with FMA3 (all current EPYC / Xeon): the contractions below
//     are the ones in that function's machine code (z + Shift, both reduction steps, the polynomial, scale + scale*tmp).
//
// tests/cpp/libm_exact_test.cpp holds both against the host libm: every one of the 2^32 float bit patterns for
// tanhf, 2^31 doubles spread over the whole argument range (incl. the subnormal / overflow paths) for exp.
// No libm function is called here; hipcc's fp32/fp64 +, *, /, fma are IEEE-754 correctly rounded, so the same source
// gives the same bits on the device.  Compiled with -ffp-contract=off: every fma below is explicit.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif

namespace pf_libm {

PF_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PF_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
PF_HD uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
PF_HD double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

// ---- expm1f (glibc 2.35 s_expm1f.c; fdlibm) ----
PF_HD float expm1f_exact(float x) {
  const float one = 1.0f, huge = 1.0e+30f, tiny = 1.0e-30f;
  const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  float y, hi, lo, c = 0.f, t, e, hxs, hfx, r1;
  int32_t k;
  uint32_t hx = f2u(x);
  const uint32_t xsb = hx & 0x80000000u;
  hx &= 0x7fffffffu;
  if (hx >= 0x4195b844u) {                 // |x| >= 27 ln2
    if (hx >= 0x42b17218u) {               // |x| >= 88.721...
      if (hx > 0x7f800000u) return x + x;  // NaN
      if (hx == 0x7f800000u) return xsb == 0 ? x : -1.0f;
      if (x > o_threshold) return huge * huge;   // overflow -> +inf
    }
    if (xsb != 0) return tiny - one;       // x < -27 ln2: -1 (inexact)
  }
  if (hx > 0x3eb17218u) {                  // |x| > 0.5 ln2
    if (hx < 0x3F851592u) {                // |x| < 1.5 ln2
      if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
      else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
    } else {
      k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5f : -0.5f));
      t = (float)k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x33000000u) {           // |x| < 2^-25: x (inexact)
    t = huge + x;
    return x - (t - (huge + x));
  } else k = 0;
  hfx = 0.5f * x;
  hxs = x * hfx;
  r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  t = 3.0f - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0f - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5f * (x - e) - 0.5f;
  if (k == 1) {
    if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
    return one + 2.0f * (x - e);
  }
  if (k <= -2 || k > 56) {                 // suffices to return exp(x) - 1
    y = one - (e - x);
    y = u2f(f2u(y) + ((uint32_t)k << 23));
    return y - one;
  }
  if (k < 23) {
    t = u2f(0x3f800000u - (0x1000000u >> k));   // 1 - 2^-k
    y = t - (e - x);
    y = u2f(f2u(y) + ((uint32_t)k << 23));
  } else {
    t = u2f((uint32_t)(0x7f - k) << 23);        // 2^-k
    y = x - (e + t);
    y += one;
    y = u2f(f2u(y) + ((uint32_t)k << 23));
  }
  return y;
}

// ---- tanhf (glibc 2.35 s_tanhf.c; fdlibm) ----
PF_HD float tanhf_exact(float x) {
  const float one = 1.0f, two = 2.0f, tiny = 1.0e-30f;
  float t, z;
  const uint32_t jx = f2u(x);
  const uint32_t ix = jx & 0x7fffffffu;
  if (ix >= 0x7f800000u) {                 // inf / NaN
    if ((int32_t)jx >= 0) return one / x + one;
    return one / x - one;
  }
  if (ix < 0x41b00000u) {                  // |x| < 22
    if (ix == 0) return x;
    if (ix < 0x24000000u) return x * (one + x);   // |x| < 2^-55
    const float ax = u2f(ix);
    if (ix >= 0x3f800000u) {               // |x| >= 1
      t = expm1f_exact(two * ax);
      z = one - two / (t + two);
    } else {
      t = expm1f_exact(-two * ax);
      z = -t / (t + two);
    }
  } else z = one - tiny;                   // |x| >= 22: 1 (inexact)
  return (int32_t)jx >= 0 ? z : -z;
}

// ---- exp, fp64 (glibc 2.35 e_exp.c, N = 128, as executed by the x86-64 FMA variant) ----
// kExpTab[2k] = bits of the tail of 2^(k/128), kExpTab[2k+1] = bits of 2^(k/128) minus k << 45 (e_exp_data.c)
#if defined(__HIPCC__)
#define PF_EXP_TAB_QUAL __device__ static const
#else
#define PF_EXP_TAB_QUAL static const
#endif
#include "libm_exact_tab.inl"

PF_HD double exp_specialcase(double tmp, uint64_t sbits, uint64_t ki, const uint64_t* tab) {
  (void)tab;
  double scale, y;
  if ((ki & 0x80000000u) == 0) {           // k > 0: the exponent of scale might have overflowed by <= 460
    sbits -= 1009ull << 52;
    scale = u2d(sbits);
    y = 0x1p1009 * __builtin_fma(scale, tmp, scale);
    return y;                              // +inf on overflow, like __math_check_oflow's value
  }
  sbits += 1022ull << 52;                  // k < 0: take care in the subnormal range
  scale = u2d(sbits);
  y = scale + scale * tmp;
  if (y < 1.0) {
    // round y to the right precision before scaling it into the subnormal range
    double hi, lo;
    lo = scale - y + scale * tmp;
    hi = 1.0 + y;
    lo = 1.0 - hi + y + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;                 // avoid -0.0 with downward rounding (not reachable in round-to-nearest)
  }
  return 0x1p-1022 * y;
}

PF_HD double exp_exact(double x, const uint64_t* tab) {
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const uint64_t xb = d2u(x);
  uint32_t abstop = (uint32_t)(xb >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {          // |x| < 2^-54, |x| >= 512, inf or NaN
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;   // tiny: exp(x) rounds to 1 (+x keeps the rounding-mode behaviour)
    if (abstop >= 0x409u) {                // |x| >= 1024
      if (xb == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;               // +inf, NaN
      if (xb >> 63) return 0x1p-767 * 0x1p-767;           // underflow: +0 in round-to-nearest
      return 0x1p769 * 0x1p769;                           // overflow: +inf
    }
    abstop = 0;                            // 512 <= |x| < 1024: large, handled by exp_specialcase
  }
  double kd = __builtin_fma(x, InvLn2N, Shift);
  const uint64_t ki = d2u(kd);
  kd -= Shift;
  double r = __builtin_fma(kd, NegLn2hiN, x);
  r = __builtin_fma(kd, NegLn2loN, r);
  const uint64_t idx = 2 * (ki % 128);
  const uint64_t top = ki << 45;
  const double tail = u2d(tab[idx]);
  const uint64_t sbits = tab[idx + 1] + top;
  const double r2 = r * r;
  const double p23 = __builtin_fma(r, C3, C2);
  const double p45 = __builtin_fma(r, C5, C4);
  double tmp = __builtin_fma(p23, r2, tail + r);
  tmp = __builtin_fma(r2 * r2, p45, tmp);
  if (abstop == 0) return exp_specialcase(tmp, sbits, ki, tab);
  const double scale = u2d(sbits);
  return __builtin_fma(scale, tmp, scale);
}

}  // namespace pf_libm
