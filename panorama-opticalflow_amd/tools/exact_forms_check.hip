// Exhaustive on-device check of csrc/exact_forms.hpp (the sweep's cheap square root and division by a constant) against the
// correctly rounded operations, for EVERY float inside the range the sweep's guard admits (zero, or 2^-96 .. 2^100):
//   sqrt_core(x), both halves of sqrt_core2        vs  sqrtf(x) (correctly rounded in HIP) and (float)sqrt((double)x)
//   div_core(a, 0.001f, y), both halves of div_core2  vs  a / 0.001f, for a of both signs (not -0, see exact_forms.hpp)
//   div_core(a, W, y) for the image widths given on the command line (default: a spread of widths up to 2^20)
// v_rsq_f32 / v_sqrt_f32 are hardware approximations: only the GPU itself can vouch for the forms built on them.
// Prints one line per check and "mismatches N"; exit code 1 if any.  ~4e9 inputs per pass, well under a second each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../csrc/exact_forms.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline bool in_range(float ax) { return ax == 0.0f || (ax >= 0x1p-96f && ax <= 0x1p100f); }

struct Report { unsigned long long tested, bad; unsigned first_bad[4]; };

__device__ inline void note_bad(Report* r, unsigned bits) {
  const unsigned long long k = atomicAdd(&r->bad, 1ull);
  if (k < 4) r->first_bad[k] = bits;
}

// mode 0: sqrt forms; mode 1: division by c
__global__ void k_check(int mode, float c, float y, Report* rep) {
  unsigned long long tested = 0;
  const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned bits = (unsigned)i;
    const float x = __uint_as_float(bits);
    if (mode == 0) {
      if (!(x >= 0.0f) || (bits >> 31) || !in_range(x)) continue;
      // the partner in the other half of the packed form: same exponent, other mantissa (also in range)
      const unsigned pb = bits == 0 ? 0x3f9e3779u : (bits ^ 0x002aaaaau);
      const float p = __uint_as_float(pb);
      const float want = sqrtf(x), wantp = sqrtf(p), want64 = (float)sqrt((double)x);
      int ea, eb;
      const pf::f2p a = pf::sqrt_core2(pf::f2p{x, p}, ea), b = pf::sqrt_core2(pf::f2p{p, x}, eb);
      const float s1 = pf::sqrt_core(x);
      const bool ok = ea == __builtin_amdgcn_frexp_expf(x) && eb == __builtin_amdgcn_frexp_expf(p) &&
                      __float_as_uint(want) == __float_as_uint(want64) && __float_as_uint(s1) == __float_as_uint(want) &&
                      __float_as_uint(a.x) == __float_as_uint(want) && __float_as_uint(b.y) == __float_as_uint(want) &&
                      __float_as_uint(a.y) == __float_as_uint(wantp) && __float_as_uint(b.x) == __float_as_uint(wantp);
      ++tested;
      if (!ok) note_bad(rep, bits);
    } else {
      if (!in_range(fabsf(x)) || bits == 0x80000000u) continue;   // -0 / c is -0, the refinement gives +0: no caller passes -0 (exact_forms.hpp)
      const unsigned pb = (bits & 0x7fffffffu) == 0 ? 0xbf9e3779u : (bits ^ 0x802aaaaau);
      const float p = __uint_as_float(pb);
      const float want = x / c, wantp = p / c;
      const float q1 = pf::div_core(x, c, y);
      const pf::f2p a = pf::div_core2(pf::f2p{x, p}, c, y);
      const bool ok = __float_as_uint(q1) == __float_as_uint(want) && __float_as_uint(a.x) == __float_as_uint(want) &&
                      __float_as_uint(a.y) == __float_as_uint(wantp);
      ++tested;
      if (!ok) note_bad(rep, bits);
    }
  }
  atomicAdd(&rep->tested, tested);
}

static unsigned long long run(int mode, float c, const char* what, Report* d_rep) {
  Report h = {};
  CK(hipMemcpy(d_rep, &h, sizeof h, hipMemcpyHostToDevice));
  const float y = (float)(1.0 / (double)c);
  hipLaunchKernelGGL(k_check, dim3(256 * 16), dim3(256), 0, 0, mode, c, y, d_rep);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h, d_rep, sizeof h, hipMemcpyDeviceToHost));
  printf("%-28s tested %llu  mismatches %llu", what, h.tested, h.bad);
  for (unsigned long long k = 0; k < h.bad && k < 4; ++k) printf("  0x%08x", h.first_bad[k]);
  printf("\n");
  return h.bad;
}

int main(int argc, char** argv) {
  Report* d_rep; CK(hipMalloc(&d_rep, sizeof(Report)));
  unsigned long long bad = 0;
  bad += run(0, 1.0f, "sqrt_core / sqrt_core2", d_rep);
  bad += run(1, 0.001f, "div by kGradEpsilon 0.001f", d_rep);
  std::vector<int> widths;
  for (int i = 1; i < argc; ++i) widths.push_back(atoi(argv[i]));
  if (widths.empty()) widths = {2, 3, 5, 7, 13, 63, 125, 250, 251, 500, 563, 1000, 1125, 2000, 2250, 4000, 4500, 8191, 9000, 12000, 16383, 65535, 999983, 1048575};
  for (int w : widths) { char name[64]; snprintf(name, sizeof name, "div by width %d", w); bad += run(1, (float)w, name, d_rep); }
  printf("mismatches %llu\n", bad);
  return bad ? 1 : 0;
}
