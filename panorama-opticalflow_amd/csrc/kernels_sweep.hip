// K6: the two order-dependent Gauss-Seidel raster sweeps of PixFlow (CPU/PixFlow.hpp:315-324 forward,
// :328-337 backward; proposeFlowUpdate :342-362, errorGradient :364-386, errorFunction :427-456,
// getPixBilinear32FExtend :407-425) reproduced EXACTLY on the GPU.
//
// Pixel (x,y) consumes the already-updated flow of its left and top neighbour (right/bottom for the
// backward sweep) and nothing else that changes during the sweep: errorFunction reads the frozen
// blurredFlow and the static gradient planes only.  So an anti-diagonal wavefront is exact.
//
// Mapping (CDNA4, wave64): one wavefront owns a band of 64 consecutive rows; lane r processes row
// y0+r at column s-r in step s (skew 1 column per row).  Left neighbour = the lane's own result of
// the previous step (register); top neighbour = lane r-1's result of the previous step (cross-lane
// move); lane 0's top neighbour comes from the previous band through an 8-byte granule per column in
// HBM whose data is its own "ready" flag (agent-scope relaxed atomic store/load, no fences: the
// pattern cdna_hip_programming.md G16 calls R2).  Bands take their index from an atomic ticket so a
// band only ever waits for one that is already running (no dispatch-order assumption), every spin is
// bounded, and a timeout raises ctrl[1] instead of hanging the GPU.
//
// The backward sweep is the same kernel in mirrored coordinates (x -> W-1-x, y -> H-1-y).
//
// LAB BUILD ONLY (-DPF_EXPERIMENTS, libpanoflow_exp.so): this v1 kernel is the independent GPU cross-check of the product's
// sweep (kernels_sweep2.hip); it is not compiled into libpanoflow.so.
#include <algorithm>

#include "pf_common.hpp"

namespace pf {

constexpr int kBandRows = 64;
constexpr int kSpinLimit = 1 << 22;

struct F2x2 { float a, b, c, d; } __attribute__((aligned(8)));  // two adjacent float2 texels

// errorFunction (PixFlow.hpp:427-456); every operation in the reference's order, no FMA.
__device__ __forceinline__ float d_error(const float2* __restrict__ g1, int W, float wm2, float hm2, float fW, const SolverCoef& cf, int x, int y, float i0x, float i0y,
                                         float bx, float by, float fdx, float fdy) {
  const float matchX = float(x) + fdx, matchY = float(y) + fdy;
  float cx = (0.0f < matchX) ? matchX : 0.0f; cx = (cx < wm2) ? cx : wm2;   // min(w-2, max(0,x)) with std::min/max semantics
  float cy = (0.0f < matchY) ? matchY : 0.0f; cy = (cy < hm2) ? cy : hm2;
  const int x0 = int(cx), y0 = int(cy);
  const float xR = cx - float(x0), yR = cy - float(y0);
  const float2* p = g1 + size_t(y0) * W + x0;
  const F2x2 t0 = *reinterpret_cast<const F2x2*>(p);      // (f00x,f00y,f10x,f10y)
  const F2x2 t1 = *reinterpret_cast<const F2x2*>(p + W);  // (f01x,f01y,f11x,f11y)
  float i1x, i1y;
  {
    const float f00 = t0.a, f10 = t0.c, f01 = t1.a, f11 = t1.c;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1x = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  {
    const float f00 = t0.b, f10 = t0.d, f01 = t1.b, f11 = t1.d;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1y = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  const float dfx = bx - fdx, dfy = by - fdy;
  const float smoothness = sqrtf(dfx * dfx + dfy * dfy);
  const float err = sqrtf((i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y)) + smoothness * cf.smooth +
                    cf.vreg * fabsf(fdy) / fW + cf.hreg * fabsf(fdx) / fW;
  return err;
}

__device__ __forceinline__ unsigned long long d_pack(float2 f) {
  return (unsigned long long)__float_as_uint(f.x) | ((unsigned long long)__float_as_uint(f.y) << 32);
}
__device__ __forceinline__ float2 d_unpack(unsigned long long v) {
  return make_float2(__uint_as_float((unsigned)(v & 0xffffffffu)), __uint_as_float((unsigned)(v >> 32)));
}

__global__ __launch_bounds__(64) void k_sweep(SweepArgs a) {
  const int lane = threadIdx.x;
  int band = 0;
  if (lane == 0) band = atomicAdd(&a.ctrl[0], 1);
  band = __builtin_amdgcn_readfirstlane(band);
  const int W = a.W, H = a.H;
  const int ry = band * kBandRows + lane;  // row in sweep order
  const bool rowActive = ry < H;
  const int y = a.forward ? ry : H - 1 - ry;
  const bool publishes = rowActive && (lane == kBandRows - 1) && (ry + 1 < H);
  const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W);
  const unsigned long long* bnd_in = a.boundary + size_t(band > 0 ? band - 1 : 0) * W;
  unsigned long long* bnd_out = a.boundary + size_t(band) * W;
  const size_t rowBase = size_t(rowActive ? y : 0) * W;
  bool aborted = false;

  float2 prev = make_float2(0.f, 0.f);  // this lane's final flow of the previous step
  const int nsteps = W + kBandRows - 1;
  for (int s = 0; s < nsteps; ++s) {
    const int cx = s - lane;
    float2 up;
    up.x = __shfl_up(prev.x, 1);
    up.y = __shfl_up(prev.y, 1);
    if (lane == 0 && band > 0 && cx < W) {
      unsigned long long v = kNotReady;
      if (!aborted) {
        v = __hip_atomic_load(bnd_in + cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (v == kNotReady) {
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(bnd_in + cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((++spins & 1023) == 0) {
            if (spins >= kSpinLimit || __hip_atomic_load(&a.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
              __hip_atomic_store(&a.ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              aborted = true;
              break;
            }
          }
        }
      }
      up = d_unpack(v);
    }
    const bool act = rowActive && cx >= 0 && cx < W;
    if (act) {
      const int x = a.forward ? cx : W - 1 - cx;
      const size_t idx = rowBase + x;
      float2 f = a.flow[idx];
      if (a.gate[idx]) {
        const float2 g0 = a.g0[idx];
        const float2 bl = a.blurred[idx];
        // proposals: previous column (left / right), then previous row (top / bottom).  A missing
        // neighbour proposes the current flow, which can never be strictly better.
        const float2 pl = (cx > 0) ? prev : f;
        const float2 pt = (ry > 0) ? up : f;
        float currErr = d_error(a.g1, W, wm2, hm2, fW, a.cf, x, y, g0.x, g0.y, bl.x, bl.y, f.x, f.y);
        const float eL = d_error(a.g1, W, wm2, hm2, fW, a.cf, x, y, g0.x, g0.y, bl.x, bl.y, pl.x, pl.y);
        const float eT = d_error(a.g1, W, wm2, hm2, fW, a.cf, x, y, g0.x, g0.y, bl.x, bl.y, pt.x, pt.y);
        if (eL < currErr) { f = pl; currErr = eL; }
        if (eT < currErr) { f = pt; currErr = eT; }
        const float ex = d_error(a.g1, W, wm2, hm2, fW, a.cf, x, y, g0.x, g0.y, bl.x, bl.y, f.x + kGradEpsilon, f.y + 0.0f);
        const float ey = d_error(a.g1, W, wm2, hm2, fW, a.cf, x, y, g0.x, g0.y, bl.x, bl.y, f.x + 0.0f, f.y + kGradEpsilon);
        const float gx = (ex - currErr) / kGradEpsilon, gy = (ey - currErr) / kGradEpsilon;
        f.x = f.x - a.cf.step * gx;
        f.y = f.y - a.cf.step * gy;
        a.flow[idx] = f;
      }
      if (publishes) __hip_atomic_store(bnd_out + cx, d_pack(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prev = f;
    }
  }
}

size_t sweep1_boundary_elems(int W, int H) { return size_t((H + kBandRows - 1) / kBandRows) * W; }

void launch_sweep(hipStream_t st, const SweepArgs& a) {
  hipLaunchKernelGGL(k_sweep, dim3((a.H + kBandRows - 1) / kBandRows), dim3(64), 0, st, a);
}

}  // namespace pf
