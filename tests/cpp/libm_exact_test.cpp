// Holds csrc/libm_exact.hpp (the restatements the blend kernel uses on the device) against THIS host's libm, bit for bit:
//   tanhf : all 2^32 float bit patterns (mode "full") or every 16th + the 766 arguments the blend can produce (default);
//   exp   : N doubles spread over the whole argument range (uniform in bit pattern per binade range, both signs, plus
//           dense samples of [0, 1100] where the blend's arguments live, and the special values).
// Usage: libm_exact_test [full] [threads]      exit code 0 = identical everywhere.
// g++ -O2 -ffp-contract=off -mfma  (fma() must be a real fused operation; -mfma makes __builtin_fma one instruction).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../panorama-opticalflow_amd/csrc/libm_exact.hpp"

using namespace pf_libm;

static std::atomic<long> g_bad{0};

static bool same_f(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }
static bool same_d(double a, double b) { return d2u(a) == d2u(b) || (a != a && b != b); }

static void tanh_range(uint32_t lo, uint64_t hi, uint32_t stride) {
  for (uint64_t u = lo; u < hi; u += stride) {
    const float x = u2f((uint32_t)u);
    const float a = tanhf(x), b = tanhf_exact(x);
    if (!same_f(a, b)) { if (g_bad++ < 10) fprintf(stderr, "tanhf(%a): libm %a, restatement %a\n", x, a, b); }
  }
}

static uint64_t splitmix(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static void exp_check(double x) {
  const double a = exp(x), b = exp_exact(x, kExpTab);
  if (!same_d(a, b)) { if (g_bad++ < 10) fprintf(stderr, "exp(%a): libm %a, restatement %a\n", x, a, b); }
}

static void exp_random(uint64_t seed, long n) {
  uint64_t s = seed;
  for (long i = 0; i < n; ++i) {
    const uint64_t r = splitmix(s);
    switch (i & 3) {
      case 0: exp_check(u2d(r)); break;                                            // any bit pattern (incl. NaN, inf, subnormals)
      case 1: exp_check((double)(r >> 11) * (1100.0 / 9007199254740992.0)); break;   // [0, 1100): the blend's arguments
      case 2: exp_check(-(double)(r >> 11) * (1100.0 / 9007199254740992.0)); break;
      default: {                                                                   // float-valued products like the blend's
        const float bl = (float)((r >> 40) & 0xffffff) / 16777216.0f, al = (float)((r >> 8) & 0xff) / 255.0f;
        const float mag = (float)((r >> 16) & 0xffffff) / 16777216.0f * 0.05f;
        exp_check(10.0f * bl * al * (1.0 + 100.0f * mag));
      }
    }
  }
}

int main(int argc, char** argv) {
  const bool full = argc > 1 && strcmp(argv[1], "full") == 0;
  int nth = argc > 2 ? atoi(argv[2]) : (int)std::thread::hardware_concurrency();
  if (nth < 1) nth = 1;
  if (nth > 64) nth = 64;
  // the blend's tanhf arguments: (n / 255.0f) * 10.0f, n = 0..765
  for (int n = 0; n <= 765; ++n) {
    const float x = (float)n / 255.0f * 10.0f;
    if (!same_f(tanhf(x), tanhf_exact(x))) { ++g_bad; fprintf(stderr, "tanhf blend argument n=%d differs\n", n); }
  }
  const double specials[] = {0.0, -0.0, 1.0, -1.0, 0x1p-54, 0x1p-55, -0x1p-54, 511.999, 512.0, 709.78, 709.79, 710.0, 1023.9, 1024.0, 1e308, -708.4, -745.1, -745.2, -746.0,
                             -1024.0, -1e308, INFINITY, -INFINITY, NAN, 0x1p-1074, -0x1p-1074};
  for (double x : specials) exp_check(x);
  std::vector<std::thread> th;
  const uint32_t stride = full ? 1 : 16;
  const long nexp = full ? (1L << 31) : (1L << 26);
  for (int t = 0; t < nth; ++t) {
    th.emplace_back([=] {
      const uint64_t span = (1ull << 32) / nth;
      tanh_range((uint32_t)(t * span), t == nth - 1 ? (1ull << 32) : (t + 1) * span, stride);
      exp_random(0x1234 + 977 * t, nexp / nth);
    });
  }
  for (auto& t : th) t.join();
  printf("libm_exact_test: tanhf %s, exp %ld samples, %ld mismatches\n", full ? "all 2^32 floats" : "every 16th float + blend arguments", nexp, g_bad.load());
  return g_bad ? 1 : 0;
}
