// Per-level kernels of PixFlow::patchMatchPropagationAndSearch (CPU/PixFlow.hpp:272-340) except the
// sweeps, plus the inter-level and final upsampling (:122-134).  All stencil/streaming, HBM-bound.
#include <stdio.h>
#include <stdlib.h>
#include "pf_common.hpp"

namespace pf {

// ------------------------------------------------------------------------------------------------
// K3 gradients: Sobel(ksize=1) central difference with BORDER_REPLICATE, then Gaussian 3x3 s0.5 with
// BORDER_REFLECT_101 (PixFlow.hpp:281-294).  Output interleaved (Ix,Iy) so the sweep's bilinear
// gather fetches both with one 8-byte load per texel.
// ------------------------------------------------------------------------------------------------
// Sobel(ksize 1) + Gauss3 sigma 0.5 at one pixel (PixFlow.hpp:281-294); shared by the per-level and the all-levels kernel
__device__ __forceinline__ float2 d_gradient_px(const float* __restrict__ img, int w, int h, int x, int y, const Gauss& g) {
  const float k0 = g.k[1], k1 = g.k[2];
  const int xm = d_reflect101(x - 1, w), xp = d_reflect101(x + 1, w);
  const int xs[3] = {xm, x, xp};
  float tx[3], ty[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int yy = d_reflect101(y - 1 + j, h);
    const float* r = img + size_t(yy) * w;
    const float* ru = img + size_t(d_replicate(yy - 1, h)) * w;
    const float* rd = img + size_t(d_replicate(yy + 1, h)) * w;
    float sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int xx = xs[i];
      sx[i] = r[d_replicate(xx + 1, w)] - r[d_replicate(xx - 1, w)];
      sy[i] = rd[xx] - ru[xx];
    }
    tx[j] = sx[1] * k0 + (sx[0] + sx[2]) * k1;
    ty[j] = sy[1] * k0 + (sy[0] + sy[2]) * k1;
  }
  float ox = k0 * tx[1] + 0.0f; ox += k1 * (tx[2] + tx[0]);
  float oy = k0 * ty[1] + 0.0f; oy += k1 * (ty[2] + ty[0]);
  return make_float2(ox, oy);
}
// The same pixel away from the border (2 <= x < w - 2, 2 <= y < h - 2): every reflect / replicate is the identity, so the 21 values of
// the 5 x 5 neighbourhood (minus its corners) are loaded once and the same expressions applied in the same order -- identical bits,
// ~90 instead of ~250 instructions (the index clamps and the 33 partly repeated loads were most of the general form; measured with
// SQ_INSTS_VALU: this kernel took 8.7 % of a batch's vector instructions, profiles/r04_batch_instruction_counts.txt).
__device__ __forceinline__ float2 d_gradient_px_interior(const float* __restrict__ img, int w, int x, int y, const Gauss& g) {
  const float k0 = g.k[1], k1 = g.k[2];
  const float* c = img + size_t(y) * w + x;
  float v[5][5];   // v[r][q] = I(y - 2 + r, x - 2 + q); the four corners are not needed
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (!((r == 0 || r == 4) && (q == 0 || q == 4))) v[r][q] = c[(r - 2) * w + (q - 2)];
  float tx[3], ty[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      sx[i] = v[j + 1][i + 2] - v[j + 1][i];       // I(yy, xx + 1) - I(yy, xx - 1)
      sy[i] = v[j + 2][i + 1] - v[j][i + 1];       // I(yy + 1, xx) - I(yy - 1, xx)
    }
    tx[j] = sx[1] * k0 + (sx[0] + sx[2]) * k1;
    ty[j] = sy[1] * k0 + (sy[0] + sy[2]) * k1;
  }
  float ox = k0 * tx[1] + 0.0f; ox += k1 * (tx[2] + tx[0]);
  float oy = k0 * ty[1] + 0.0f; oy += k1 * (ty[2] + ty[0]);
  return make_float2(ox, oy);
}
// Four horizontally adjacent interior pixels (x .. x + 3, all with 2 <= x, x + 3 < w - 2, 2 <= y < h - 2): the 5 x 8 neighbourhood is
// loaded once (36 values instead of 4 x 21) and the per-pixel expressions of d_gradient_px_interior are written out with the column
// offset q -- the central differences two neighbours share are the same subtraction of the same operands, so computing them once
// changes no bit.
__device__ __forceinline__ void d_gradient_quad_interior(const float* __restrict__ img, int w, int x, int y, const Gauss& g, float2 (&o)[4]) {
  const float k0 = g.k[1], k1 = g.k[2];
  const float* c = img + size_t(y) * w + x;
  float v[5][8];   // v[r][q] = I(y - 2 + r, x - 2 + q); rows 0 and 4 only need columns 1..6
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (!((r == 0 || r == 4) && (q == 0 || q == 7))) v[r][q] = c[(r - 2) * w + (q - 2)];
  float dx[3][6], dy[3][6];   // central differences at rows y - 1 .. y + 1, columns x - 1 .. x + 4
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dx[j][i] = v[j + 1][i + 2] - v[j + 1][i];       // I(yy, xx + 1) - I(yy, xx - 1)
      dy[j][i] = v[j + 2][i + 1] - v[j][i + 1];       // I(yy + 1, xx) - I(yy - 1, xx)
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float tx[3], ty[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      tx[j] = dx[j][q + 1] * k0 + (dx[j][q] + dx[j][q + 2]) * k1;
      ty[j] = dy[j][q + 1] * k0 + (dy[j][q] + dy[j][q + 2]) * k1;
    }
    float ox = k0 * tx[1] + 0.0f; ox += k1 * (tx[2] + tx[0]);
    float oy = k0 * ty[1] + 0.0f; oy += k1 * (ty[2] + ty[0]);
    o[q] = make_float2(ox, oy);
  }
}
__device__ __forceinline__ float2 d_gradient_any(const float* __restrict__ img, int w, int h, int x, int y, const Gauss& g) {
  return (x >= 2 && x < w - 2 && y >= 2 && y < h - 2) ? d_gradient_px_interior(img, w, x, y, g) : d_gradient_px(img, w, h, x, y, g);
}
__global__ __launch_bounds__(256) void k_gradients(const float* __restrict__ img, int w, int h, float2* __restrict__ gxy, Gauss g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  gxy[size_t(y) * w + x] = d_gradient_any(img, w, h, x, y, g);
}
// All pyramid levels of both images in ONE launch (the per-level launches of the small levels are pure launch latency):
// a thread's flat index inside the pyramid plane -> level by binary search in the offset table -> (x, y).
__global__ __launch_bounds__(256) void k_gradients_all(const float* __restrict__ img0, const float* __restrict__ img1, float2* __restrict__ g0,
                                                       float2* __restrict__ g1, LevelTable t, unsigned first, unsigned total, Gauss g, size_t bstride) {
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(img0, bo); PF_BOFF(img1, bo); PF_BOFF(g0, bo); PF_BOFF(g1, bo); }
  // elements [first, total) of the pyramid plane; grid-stride, so that a launch can be made with few blocks on purpose.  The level
  // is searched once per block and chunk (scalar instructions on the kernel arguments); only a chunk that runs into the next level
  // makes its threads step on.
  // A thread takes FOUR consecutive plane elements (level offsets and `first` are multiples of 64: a group never straddles levels):
  // inside a row and away from the border they are one d_gradient_quad_interior and two 16-byte stores, otherwise four single pixels.
  for (unsigned base = first + blockIdx.x * (blockDim.x * 4); base < total; base += gridDim.x * (blockDim.x * 4)) {
    const unsigned i = base + 4 * threadIdx.x;
    if (i >= total) continue;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (base >= t.off[mid]) lo = mid; else hi = mid - 1; }
    while (lo + 1 < t.n && i >= t.off[lo + 1]) ++lo;
    const int w = t.w[lo], h = t.h[lo];
    const unsigned local = i - t.off[lo], npx = unsigned(w) * unsigned(h);
    if (local >= npx) continue;   // padding between levels
    const int y = int(local / unsigned(w)), x = int(local - unsigned(y) * unsigned(w));
    const float* img = (blockIdx.y ? img1 : img0) + t.off[lo];
    float2* out = (blockIdx.y ? g1 : g0) + t.off[lo];
    if (x >= 2 && x + 3 < w - 2 && y >= 2 && y < h - 2) {
      float2 o[4];
      d_gradient_quad_interior(img, w, x, y, g, o);
      float4* o4 = reinterpret_cast<float4*>(out + local);   // local is a multiple of 4: 32-byte aligned
      o4[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
      o4[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
    } else {
      int xx = x, yy = y;
      for (int q = 0; q < 4 && local + q < npx; ++q) {
        out[local + q] = d_gradient_any(img, w, h, xx, yy, g);
        if (++xx == w) { xx = 0; ++yy; }
      }
    }
  }
}
void launch_gradients(hipStream_t st, const float* img, int w, int h, float* gxy, const Gauss& g3) {
  dim3 grid((w + 255) / 256, h);
  hipLaunchKernelGGL(k_gradients, grid, dim3(256), 0, st, img, w, h, reinterpret_cast<float2*>(gxy), g3);
}
// max_blocks > 0 caps the blocks per image: a launch that runs BESIDE latency-critical kernels (the finest levels' gradients
// next to the coarse levels' sweeps) is made narrow so that it takes a few wave slots per CU instead of all of them.
void launch_gradients_all(hipStream_t st, const float* pyr0, const float* pyr1, float* grad0, float* grad1, const LevelTable& t, size_t first,
                          size_t total, const Gauss& g3, int max_blocks, Batch bt) {
  if (total <= first) return;
  // k_gradients_all takes four adjacent elements per thread and stores them as two float4: every level must start on a multiple of four
  // elements and so must `first` (pf_api.hip pads the levels to multiples of 64).  A table from anywhere else is refused loudly.
  bool aligned = first % 4 == 0;
  for (int l = 0; l < t.n; ++l) aligned = aligned && t.off[l] % 4 == 0;
  if (!aligned) { fprintf(stderr, "[panoflow] launch_gradients_all: level offsets must be multiples of 4 elements\n"); abort(); }
  size_t blocks = (total - first + 1023) / 1024;   // four elements per thread
  if (max_blocks > 0 && blocks > size_t(max_blocks)) blocks = size_t(max_blocks);
  hipLaunchKernelGGL(k_gradients_all, dim3((unsigned)blocks, 2, bt.n), dim3(256), 0, st, pyr0, pyr1, reinterpret_cast<float2*>(grad0),
                     reinterpret_cast<float2*>(grad1), t, (unsigned)first, (unsigned)total, g3, bt.stride);
}

// update gate of the sweeps (PixFlow.hpp:317,330): alpha0 > 0.9 && alpha1 > 0.9
__global__ __launch_bounds__(256) void k_gate(const float* __restrict__ a0, const float* __restrict__ a1, int n, uint8_t* __restrict__ gate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gate[i] = (a0[i] > kUpdateAlphaThreshold && a1[i] > kUpdateAlphaThreshold) ? 1 : 0;
}
void launch_gate(hipStream_t st, const float* a0, const float* a1, int n, uint8_t* gate) {
  hipLaunchKernelGGL(k_gate, dim3((n + 255) / 256), dim3(256), 0, st, a0, a1, n, gate);
}

// Bounding box of the gated pixels of every pyramid level in one launch: box[4*l + {0,1,2,3}] = (min x, min y, max x, max y),
// initialised by the caller to (INT_MAX, INT_MAX, -1, -1).  A block scans 16 Ki consecutive plane elements, reduces in
// registers / LDS and issues at most four atomics per level it touched (a chunk spans at most a few of the small levels).
__global__ __launch_bounds__(256) void k_gate_bbox(const uint8_t* __restrict__ gate, LevelTable t, unsigned total, int* __restrict__ box) {
  __shared__ int sbox[4];
  const unsigned base = blockIdx.x * 16384u;
  int lvl = -1, w = 1, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
  unsigned off = 0, cnt = 0;
  auto flush = [&]() {   // block-level reduction of one level's partial box, then the atomics
    if (threadIdx.x < 4) sbox[threadIdx.x] = (threadIdx.x < 2) ? 0x7fffffff : -1;
    __syncthreads();
    if (mxx >= 0) { atomicMin(&sbox[0], mnx); atomicMin(&sbox[1], mny); atomicMax(&sbox[2], mxx); atomicMax(&sbox[3], mxy); }
    __syncthreads();
    if (threadIdx.x == 0 && sbox[2] >= 0) {
      atomicMin(&box[4 * lvl + 0], sbox[0]); atomicMin(&box[4 * lvl + 1], sbox[1]);
      atomicMax(&box[4 * lvl + 2], sbox[2]); atomicMax(&box[4 * lvl + 3], sbox[3]);
    }
    __syncthreads();
    mnx = 0x7fffffff; mny = 0x7fffffff; mxx = -1; mxy = -1;
  };
  // all threads of the block walk the levels the chunk intersects in the same order (block-uniform control flow)
  int l0 = 0, hi = t.n - 1;
  while (l0 < hi) { const int mid = (l0 + hi + 1) >> 1; if (base >= t.off[mid]) l0 = mid; else hi = mid - 1; }
  const unsigned end = (base + 16384u < total) ? base + 16384u : total;
  for (int l = l0; l < t.n && t.off[l] < end; ++l) {
    lvl = l; w = t.w[l]; off = t.off[l]; cnt = unsigned(t.w[l]) * unsigned(t.h[l]);
    const unsigned lo = off > base ? off : base, hiE = (off + cnt < end) ? off + cnt : end;
    for (unsigned i = lo + threadIdx.x; i < hiE; i += 256) {
      if (gate[i]) {
        const unsigned local = i - off;
        const int y = int(local / unsigned(w)), x = int(local - unsigned(y) * unsigned(w));
        mnx = min(mnx, x); mny = min(mny, y); mxx = max(mxx, x); mxy = max(mxy, y);
      }
    }
    flush();
  }
}
void launch_gate_bbox(hipStream_t st, const uint8_t* gate, const LevelTable& t, size_t total, int* box) {
  hipLaunchKernelGGL(k_gate_bbox, dim3((unsigned)((total + 16383) / 16384)), dim3(256), 0, st, gate, t, (unsigned)total, box);
}

// Gate + bounding boxes + level-0 count of ALL levels in ONE launch, published straight into mapped pinned host memory:
//   gate[i] = alpha0 > 0.9 && alpha1 > 0.9 (PixFlow.hpp:317,330); work[4*l..] = bounding box of level l's gated pixels,
//   work[4*kLevelTableMax] = number of gated pixels of level 0, work[4*kLevelTableMax + 1] = blocks finished.
// The last block to finish copies boxes + count to `host` (system-scope stores), stores the call's epoch behind them as the
// "ready" flag, and resets the work area for the next call -- the host polls that flag instead of synchronising the stream
// (no pageable copies, no blocking wait: the round trip costs microseconds).
#ifndef PF_GATE_PER
#define PF_GATE_PER 16384
#endif
constexpr unsigned kGatePer = PF_GATE_PER;   // level-pixels per block
__global__ __launch_bounds__(256) void k_gate_bbox_all(const float* __restrict__ a0, const float* __restrict__ a1, uint8_t* __restrict__ gate, LevelTable t,
                                                       unsigned total, int* __restrict__ work, int* __restrict__ host, int epoch, size_t bstride, size_t hstride) {
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(a0, bo); PF_BOFF(a1, bo); PF_BOFF(gate, bo); PF_BOFF(work, bo); PF_BOFF(host, size_t(blockIdx.z) * hstride); }
  __shared__ int sbox[4];
  __shared__ int scnt, slast;
  constexpr unsigned kPer = kGatePer;
  const unsigned base = blockIdx.x * kPer;
  if (threadIdx.x == 0) scnt = 0;
  int lvl = -1, w = 1, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1, cnt0 = 0;
  unsigned off = 0, cnt = 0;
  auto flush = [&]() {   // block-level reduction of one level's partial box, then the atomics
    if (threadIdx.x < 4) sbox[threadIdx.x] = (threadIdx.x < 2) ? 0x7fffffff : -1;
    for (int o = 32; o > 0; o >>= 1) {   // wave-level reduction first: four LDS atomics per wave, not per thread
      mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o)); mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && mxx >= 0) { atomicMin(&sbox[0], mnx); atomicMin(&sbox[1], mny); atomicMax(&sbox[2], mxx); atomicMax(&sbox[3], mxy); }
    __syncthreads();
    if (threadIdx.x == 0 && sbox[2] >= 0) {
      // Look before the atomic: all but the first few blocks of a level lie inside the box found so far, and thousands of atomics on
      // the same four words serialise in L2 (they were most of this kernel's 250 us on a dense pair).  The box only grows, so a stale
      // value read here can cause a superfluous atomic, never a missing one.
      int* wl = &work[4 * lvl];
      if (sbox[0] < __hip_atomic_load(&wl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&wl[0], sbox[0]);
      if (sbox[1] < __hip_atomic_load(&wl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&wl[1], sbox[1]);
      if (sbox[2] > __hip_atomic_load(&wl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&wl[2], sbox[2]);
      if (sbox[3] > __hip_atomic_load(&wl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&wl[3], sbox[3]);
    }
    __syncthreads();
    mnx = 0x7fffffff; mny = 0x7fffffff; mxx = -1; mxy = -1;
  };
  int l0 = 0, hi = t.n - 1;
  while (l0 < hi) { const int mid = (l0 + hi + 1) >> 1; if (base >= t.off[mid]) l0 = mid; else hi = mid - 1; }
  const unsigned end = (base + kPer < total) ? base + kPer : total;
  for (int l = l0; l < t.n && t.off[l] < end; ++l) {
    lvl = l; w = t.w[l]; off = t.off[l]; cnt = unsigned(t.w[l]) * unsigned(t.h[l]);
    const unsigned lo = off > base ? off : base, hiE = (off + cnt < end) ? off + cnt : end;
    // four consecutive level-pixels per thread and iteration (16-byte loads, one 4-byte gate store); `lo` is a multiple of 4
    // (block bases and level offsets are multiples of 64).  One division per thread and level: the following groups (1024
    // apart) advance (x, y) by a per-level quotient / remainder.
    const int qStep = 1024 / w, rStep = 1024 - qStep * w;
    unsigned i = lo + 4 * threadIdx.x;
    int y = 0, x = 0;
    if (i < hiE) { const unsigned local = i - off; y = int(local / unsigned(w)); x = int(local - unsigned(y) * unsigned(w)); }
    for (; i < hiE; i += 1024) {
      bool gt[4];
      if (i + 3 < hiE) {
        const float4 va = *reinterpret_cast<const float4*>(a0 + i), vb = *reinterpret_cast<const float4*>(a1 + i);
        gt[0] = va.x > kUpdateAlphaThreshold && vb.x > kUpdateAlphaThreshold; gt[1] = va.y > kUpdateAlphaThreshold && vb.y > kUpdateAlphaThreshold;
        gt[2] = va.z > kUpdateAlphaThreshold && vb.z > kUpdateAlphaThreshold; gt[3] = va.w > kUpdateAlphaThreshold && vb.w > kUpdateAlphaThreshold;
        *reinterpret_cast<uchar4*>(gate + i) = make_uchar4(gt[0], gt[1], gt[2], gt[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gt[j] = (i + j < hiE) && a0[i + j] > kUpdateAlphaThreshold && a1[i + j] > kUpdateAlphaThreshold;
          if (i + j < hiE) gate[i + j] = gt[j] ? 1 : 0;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (gt[j]) {
          int xj = x + j, yj = y;
          while (xj >= w) { xj -= w; ++yj; }
          mnx = min(mnx, xj); mny = min(mny, yj); mxx = max(mxx, xj); mxy = max(mxy, yj);
          if (l == 0) ++cnt0;
        }
      }
      x += rStep; y += qStep;
      if (x >= w) { x -= w; ++y; }
    }
    flush();
  }
  // padding between levels: keep the gate defined (0) there
  for (int o = 32; o > 0; o >>= 1) cnt0 += __shfl_down(cnt0, o);
  if ((threadIdx.x & 63) == 0 && cnt0) atomicAdd(&scnt, cnt0);
  __syncthreads();
  constexpr int kCnt = 4 * kLevelTableMax, kDone = kCnt + 1;
  if (threadIdx.x == 0) {
    if (scnt) atomicAdd(&work[kCnt], scnt);
    __threadfence();
    slast = (atomicAdd(&work[kDone], 1) == int(gridDim.x) - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!slast) return;
  __threadfence();
  for (int i = threadIdx.x; i <= kCnt; i += 256) {
    const int v = __hip_atomic_load(&work[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&host[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    work[i] = (i == kCnt) ? 0 : (((i & 3) < 2) ? 0x7fffffff : -1);   // reset for the next call
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    work[kDone] = 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);   // the boxes are visible to the host before the flag
    __hip_atomic_store(&host[kDone], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_gate_bbox_all(hipStream_t st, const float* a0, const float* a1, uint8_t* gate, const LevelTable& t, size_t total, int* work, int* host_mapped, int epoch,
                          Batch bt, size_t host_stride) {
  hipLaunchKernelGGL(k_gate_bbox_all, dim3((unsigned)((total + kGatePer - 1) / kGatePer), 1, bt.n), dim3(256), 0, st, a0, a1, gate, t, (unsigned)total, work, host_mapped,
                     epoch, bt.stride, host_stride);
}

__global__ __launch_bounds__(256) void k_count_gate(const uint8_t* __restrict__ gate, int n, unsigned* __restrict__ count) {
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += gate[i];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned)c);
}
void launch_count_gate(hipStream_t st, const uint8_t* gate, int n, unsigned* count) {
  const int blocks = (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256;
  hipLaunchKernelGGL(k_count_gate, dim3(blocks), dim3(256), 0, st, gate, n, count);
}

// ------------------------------------------------------------------------------------------------
// K7 medianBlur(5) on float2, per channel, BORDER_REPLICATE, out of place (PixFlow.hpp:325,338).
// A median is a pure selection: the element of rank 13 of the window, whatever finds it.  Two horizontally adjacent outputs per
// thread: their 5x5 windows share four columns (20 values), and an element of that shared set can only be the median of either
// window if its rank inside the set is 8..13.  The selection network is GENERATED (tools/gen_median_net.py -> median_net.inl):
// columns sorted with v_min3 / v_med3 / v_max3 (12 instructions per column), Batcher odd-even merges of the sorted columns, ranks
// 8..13 of the shared twenty merged with each output's own fifth column, and everything the two medians do not depend on pruned
// by a liveness pass -- 180 instructions per channel for the two outputs (rounds 1-3: hand-written "forgetful selection", ~345),
// verified exhaustively with the 0-1 principle (2^30 inputs; tests/test_median_net.py).  All in registers.
// ------------------------------------------------------------------------------------------------
// (Measured and rejected: the five operations as inline-asm instructions, to shed the 60 v_max x, x, x with which the compiler
// canonicalises the values loaded from memory before an fminf / fmaxf -- the hazard recogniser then puts an s_nop behind every asm
// statement whose result is read next, 66 of them: 571 instead of 524 instructions per thread.)
__device__ __forceinline__ float d_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }   // v_min3_f32
__device__ __forceinline__ float d_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32
#include "median_net.inl"
// medians of the two horizontally adjacent outputs whose 6 x 5 neighbourhood is col[0..5][0..4] (output 0: columns 0-4, output 1: 1-5)
__device__ __forceinline__ void d_median_pair(const float2 (&col)[6][5], float2& m0, float2& m1) {
  float in[30], a, b;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) in[i * 5 + j] = col[i][j].x;
  d_median_pair_net(in, a, b); m0.x = a; m1.x = b;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) in[i * 5 + j] = col[i][j].y;
  d_median_pair_net(in, a, b); m0.y = a; m1.y = b;
}
// One output pixel of medianBlur(5), with exactly the operations k_median5 performs for it (that kernel computes outputs xp = x & ~1
// and xp + 1 together; which four columns are shared depends on the parity of x) -- used by the Gaussian that takes the median
// while it loads (small levels in the throughput mode).
__device__ __forceinline__ float2 d_median5_px(const float2* __restrict__ src, int w, int h, int x, int y) {
  const int o = x & 1, xp = x - o;
  float2 col[6][5];   // columns xp-2 .. xp+3 (replicate border), rows y-2 .. y+2; output o does not use column (o ? 0 : 5)
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float2* r = src + size_t(d_replicate(y + j - 2, h)) * w;
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i][j] = r[d_replicate(xp + i - 2, w)];
  }
  float2 m0, m1;
  d_median_pair(col, m0, m1);
  return o ? m1 : m0;
}

// one output pixel of the inter-level upsample (K9 below: resize INTER_CUBIC on float2, then *= 1/0.9f); shared by
// k_upsample_cubic and by the Gauss15 kernel that upsamples while it loads (small levels)
__device__ __forceinline__ float2 d_upsample_cubic_px(const float2* __restrict__ src, int sw, int sh, int dx, int dy, double scale_x, double scale_y, float mul) {
  int sx, sy; float fx, fy;
  d_src_coord(dx, scale_x, sx, fx);
  d_src_coord(dy, scale_y, sy, fy);
  float a[4], b[4];
  d_cubic_coeffs(fx, a);
  d_cubic_coeffs(fy, b);
  const int x0 = d_replicate(sx - 1, sw), x1 = d_replicate(sx, sw), x2 = d_replicate(sx + 1, sw), x3 = d_replicate(sx + 2, sw);
  float hx[4], hy[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float2* r = src + size_t(d_replicate(sy - 1 + j, sh)) * sw;
    const float2 p0 = r[x0], p1 = r[x1], p2 = r[x2], p3 = r[x3];
    hx[j] = p0.x * a[0] + p1.x * a[1] + p2.x * a[2] + p3.x * a[3];
    hy[j] = p0.y * a[0] + p1.y * a[1] + p2.y * a[2] + p3.y * a[3];
  }
  const float ox = hx[0] * b[0] + hx[1] * b[1] + hx[2] * b[2] + hx[3] * b[3];
  const float oy = hy[0] * b[0] + hy[1] * b[1] + hy[2] * b[2] + hy[3] * b[3];
  return make_float2(ox * mul + 0.0f, oy * mul + 0.0f);
}

// ------------------------------------------------------------------------------------------------
// K5 Gaussian 15x15 s8 on the float2 flow, BORDER_REFLECT_101 (PixFlow.hpp:306-311, :389-394).
// [OpenCV filter.cpp] row pass = RowFilter (plain left-to-right accumulation), column pass =
// SymmColumnFilter (centre, then symmetric pairs outward).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gauss15_row(const float2* __restrict__ src, float2* __restrict__ tmp, int w, int h, Gauss g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const float2* r = src + size_t(y) * w;
  float2 v = r[d_reflect101(x - 7, w)];
  float sx = g.k[0] * v.x, sy = g.k[0] * v.y;
#pragma unroll
  for (int j = 1; j < 15; ++j) {
    v = r[d_reflect101(x - 7 + j, w)];
    sx += g.k[j] * v.x; sy += g.k[j] * v.y;
  }
  tmp[size_t(y) * w + x] = make_float2(sx, sy);
}
__device__ __forceinline__ float2 d_gauss15_col(const float2* __restrict__ tmp, int w, int h, int x, int y, const Gauss& g) {
  float2 c = tmp[size_t(y) * w + x];
  float sx = g.k[7] * c.x + 0.0f, sy = g.k[7] * c.y + 0.0f;
#pragma unroll
  for (int j = 1; j <= 7; ++j) {
    const float2 a = tmp[size_t(d_reflect101(y + j, h)) * w + x], b = tmp[size_t(d_reflect101(y - j, h)) * w + x];
    sx += g.k[7 + j] * (a.x + b.x); sy += g.k[7 + j] * (a.y + b.y);
  }
  return make_float2(sx, sy);
}
__global__ __launch_bounds__(256) void k_gauss15_col(const float2* __restrict__ tmp, float2* __restrict__ dst, int w, int h, Gauss g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  dst[size_t(y) * w + x] = d_gauss15_col(tmp, w, h, x, y, g);
}
// Row + column pass in ONE launch, staged through LDS: a block owns a 64 x 32 output tile.  (1) the source tile with its
// 7-pixel ring (reflect-101 on both indices) is loaded once, coalesced, into LDS; (2) the row pass (RowFilter: plain
// left-to-right accumulation) of the 46 rows the column pass needs goes from LDS to LDS; (3) the column pass
// (SymmColumnFilter: centre, then symmetric pairs outward) reads LDS.  Same operations in the same order as the two-kernel
// form => identical bits.  Both passes are register-blocked along their filter axis -- a thread reads a run of 16 + 14 (row
// pass) or 8 + 14 (column pass) values once and produces 16 / 8 outputs from registers, 1.9 / 2.75 LDS reads per output
// instead of 15 -- because the first fused version was bound by LDS bandwidth.  In the row pass a wave's lanes run along y
// (one source row each), so the LDS row strides are odd numbers of float2 (79, 65): at most 2-way bank conflicts.
// MIX fuses lowAlphaFlowDiffusion's alpha mix (PixFlow.hpp:396-403) into the epilogue.
constexpr int kG15TX = 64, kG15TY = 32, kG15R = 7;
constexpr int kG15SW = kG15TX + 2 * kG15R, kG15SH = kG15TY + 2 * kG15R;   // 78 x 46 source texels per tile
// Which texels of the 78 x 46 source tile a thread carries (the same map for the prefetch and for the LDS store).  AFFINE in the
// prefetch index k, so that an interior tile's addresses are "tile base + constant": k < 12: row = wave + 4k, column = lane (the 64
// left columns, whole-wave 512-byte runs); k = 12..14: the 14 right-hand columns of all rows, 644 texels dealt out 256 at a time
// (row = i / 14, column = 64 + i % 14 with i = thread + 256 (k - 12)).  Before round 4 the map was t = thread + 256 k -> (t / 78, t % 78):
// a division, two reflections and a bounds test per texel, ~600 of the kernel's ~1,800 vector instructions per tile.
constexpr int kG15Main = (kG15SH + 3) / 4;                                 // 12
constexpr int kG15TailW = kG15SW - 64;                                     // 14
constexpr int kG15Tail = (kG15TailW * kG15SH + 255) / 256;                 // 3
constexpr int kG15Pre = kG15Main + kG15Tail;                               // 15 texels per thread
// The blocks are persistent (at most three per CU, the LDS limit) and walk the tiles with a stride of gridDim.x: the global
// loads of a block's NEXT tile are issued into registers before it computes the current one, so that with only three blocks
// per CU the load latency hides behind the two passes instead of in front of them.
// A tile whose 78 x 46 footprint lies inside the plane (all but the rim) takes the INTERIOR form of every step: no reflection, no
// bounds tests, no partial rows -- the same loads, products and sums in the same order, so the same bits.
// UPS (small levels): the source plane does not exist yet -- it is the bicubic upsample of the coarser level's result
// (PixFlow.hpp:122-125); the tile loader computes it on the fly (same expressions as k_upsample_cubic) and writes the tile's own
// 64 x 32 pixels of it to `up_out`, which saves the separate upsample launch where a launch costs more than its work.
struct UpsSrc { const float2* src; int sw, sh; double scale_x, scale_y; float mul; float2* up_out; };
template <bool V> struct G15Flag { static constexpr bool value = V; };
template <bool MIX, bool UPS, bool MED>
__global__ __launch_bounds__(256, 3) void k_gauss15_fused(const float2* __restrict__ src, float2* __restrict__ dst, int w, int h, Gauss g,
                                                        const float* __restrict__ a0, const float* __restrict__ a1, int ntx, int ntiles, UpsSrc ups, size_t bstride) {
  {   // blockIdx.z = pair of a batched launch (every pointer is pair 0's)
    const size_t bo = size_t(blockIdx.z) * bstride;
    PF_BOFF(src, bo); PF_BOFF(dst, bo); PF_BOFF(a0, bo); PF_BOFF(a1, bo); PF_BOFF(ups.src, bo); PF_BOFF(ups.up_out, bo);
  }
  constexpr int SW = kG15SW, SH = kG15SH;
  constexpr int SS = SW + 1, RS = kG15TX + 1;                        // LDS row strides in float2: 79, 65 (odd); 53 KB in all: three blocks per CU
  __shared__ float2 srct[SH * SS];
  __shared__ float2 rowp[SH * RS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float2 pre[kG15Pre];
  float coef[8];   // MIX: diffusion coefficients of this thread's eight outputs of the prefetched tile
  // tail texels of this thread: i = thread + 256 j -> (i / 14, 64 + i % 14); 256 = 18 * 14 + 4
  const int tr0 = int(threadIdx.x) / kG15TailW, tc0 = int(threadIdx.x) - tr0 * kG15TailW;
  auto tail_rc = [&](int j, int& r, int& c) __attribute__((always_inline)) {
    c = tc0 + 4 * j; r = tr0 + 18 * j;
    if (c >= kG15TailW) { c -= kG15TailW; r += 1; }
    c += 64;
  };
  auto is_interior = [&](int tile) __attribute__((always_inline)) {
    const int x0 = (tile % ntx) * kG15TX, y0 = (tile / ntx) * kG15TY;
    return x0 >= kG15R && x0 + kG15TX + kG15R <= w && y0 >= kG15R && y0 + kG15TY + kG15R <= h;
  };
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int x0 = (tile % ntx) * kG15TX, y0 = (tile / ntx) * kG15TY;
    if (is_interior(tile)) {   // block-uniform
      // (MIX: an interior tile loads its alphas at the start of its own column pass -- a tile ahead they cost 16 registers across the
      // row pass, which spilled)
      const float2* p = src + size_t(y0 - kG15R) * w + (x0 - kG15R);
      const float2* pm = p + size_t(wv) * w + lane;
#pragma unroll
      for (int k = 0; k < kG15Main; ++k)
        if (4 * k + 3 < SH || wv + 4 * k < SH) pre[k] = pm[size_t(4 * k) * w];
#pragma unroll
      for (int j = 0; j < kG15Tail; ++j) {
        int r, c; tail_rc(j, r, c);
        if (256 * (j + 1) <= kG15TailW * SH || r < SH) pre[kG15Main + j] = p[size_t(r) * w + c];
      }
      return;
    }
    const int rowsNeeded = min(SH, h + kG15R - (y0 - kG15R));   // rows past (h - 1) + 7 are read by no output of this tile
    if (MIX) {
      const int xq = x0 + lane, yq = y0 + wv * 8;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const size_t i = size_t(min(yq + o, h - 1)) * w + min(xq, w - 1);
        coef[o] = 1.0f - a0[i] * a1[i];
      }
    }
#pragma unroll
    for (int k = 0; k < kG15Pre; ++k) {
      int r, c;
      if (k < kG15Main) { r = wv + 4 * k; c = lane; } else tail_rc(k - kG15Main, r, c);
      if (r < rowsNeeded) pre[k] = src[size_t(d_reflect101(y0 - kG15R + r, h)) * w + d_reflect101(min(x0 - kG15R + c, w - 1 + kG15R), w)];
    }
  };
  int tile = blockIdx.x;
  constexpr bool DIRECT = UPS || MED;   // the loader computes its texels: no register prefetch of the next tile
  if (!DIRECT && tile < ntiles) fetch(tile);
  while (tile < ntiles) {
    const int next = tile + int(gridDim.x);
    auto do_tile = [&](auto interiorFlag) __attribute__((always_inline)) {
      constexpr bool INT = decltype(interiorFlag)::value;
      const int x0 = (tile % ntx) * kG15TX, y0 = (tile / ntx) * kG15TY;
      const int rowsNeeded = INT ? SH : min(SH, h + kG15R - (y0 - kG15R));
      // ---- (1) source tile: registers -> LDS ----
      if (MED) {
        // the source is the median-filtered plane (medianBlur 5, PixFlow.hpp:338), computed here instead of by its own launch
        for (int t = threadIdx.x; t < rowsNeeded * SW; t += 256) {
          const int r = t / SW, cidx = t - r * SW;
          srct[r * SS + cidx] = d_median5_px(src, w, h, d_reflect101(min(x0 - kG15R + cidx, w - 1 + kG15R), w), d_reflect101(y0 - kG15R + r, h));
        }
      } else if (UPS) {
        for (int t = threadIdx.x; t < rowsNeeded * SW; t += 256) {
          const int r = t / SW, cidx = t - r * SW;
          const int yy = d_reflect101(y0 - kG15R + r, h), xx = d_reflect101(min(x0 - kG15R + cidx, w - 1 + kG15R), w);
          const float2 v = d_upsample_cubic_px(ups.src, ups.sw, ups.sh, xx, yy, ups.scale_x, ups.scale_y, ups.mul);
          srct[r * SS + cidx] = v;
          const int oy = r - kG15R, ox = cidx - kG15R;   // the tile's own pixels are unreflected: every pixel of the plane is written once
          if (oy >= 0 && oy < kG15TY && ox >= 0 && ox < kG15TX && y0 + oy < h && x0 + ox < w) ups.up_out[size_t(y0 + oy) * w + x0 + ox] = v;
        }
      } else {
        float2* sm = srct + wv * SS + lane;
#pragma unroll
        for (int k = 0; k < kG15Main; ++k)
          if ((INT && 4 * k + 3 < SH) || wv + 4 * k < rowsNeeded) sm[4 * k * SS] = pre[k];
#pragma unroll
        for (int j = 0; j < kG15Tail; ++j) {
          int r, c; tail_rc(j, r, c);
          if ((INT && 256 * (j + 1) <= kG15TailW * SH) || r < rowsNeeded) srct[r * SS + c] = pre[kG15Main + j];
        }
      }
      __syncthreads();
      float cf[8];
      if (!INT) {
#pragma unroll
        for (int o = 0; o < 8; ++o) cf[o] = coef[o];
      }
      if (!DIRECT && next < ntiles) fetch(next);
      // ---- (2) row pass: lane = source row (46 of 64 lanes), wave = 16 output columns; 30 values -> 16 outputs ----
      if (lane < rowsNeeded) {
        const float2* sr = srct + lane * SS + wv * 16;
        float2 v[30];
#pragma unroll
        for (int t = 0; t < 30; ++t) v[t] = sr[t];
        float2* rp = rowp + lane * RS + wv * 16;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          float sx = g.k[0] * v[o].x, sy = g.k[0] * v[o].y;
#pragma unroll
          for (int t = 1; t < 15; ++t) { sx += g.k[t] * v[o + t].x; sy += g.k[t] * v[o + t].y; }
          rp[o] = make_float2(sx, sy);
        }
      }
      __syncthreads();
      // ---- (3) column pass: lane = column, wave = 8 output rows; 22 values -> 8 outputs.  The row pass of source row
      // reflect101(y + d) sits at LDS row (y - y0) + 7 + d ----
      const int x = x0 + lane;
      const int oy0 = wv * 8;
      if (INT || (x < w && y0 + oy0 < h)) {
        float al0[8], al1[8];
        if (MIX && INT && !DIRECT) {
          const size_t i0 = size_t(y0 + oy0) * w + x;
#pragma unroll
          for (int o = 0; o < 8; ++o) { al0[o] = a0[i0 + size_t(o) * w]; al1[o] = a1[i0 + size_t(o) * w]; }
        }
        float2 cv[22];
#pragma unroll
        for (int t = 0; t < 22; ++t) cv[t] = (INT || oy0 + t < rowsNeeded) ? rowp[(oy0 + t) * RS + lane] : make_float2(0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const int y = y0 + oy0 + o;
          if (!INT && y >= h) break;
          const float2 c = cv[o + kG15R];
          float sx = g.k[7] * c.x + 0.0f, sy = g.k[7] * c.y + 0.0f;
#pragma unroll
          for (int j = 1; j <= 7; ++j) {
            const float2 a = cv[o + kG15R + j], b = cv[o + kG15R - j];
            sx += g.k[7 + j] * (a.x + b.x); sy += g.k[7 + j] * (a.y + b.y);
          }
          const size_t i = size_t(y) * w + x;
          if (MIX) {
            const float2 f = srct[(oy0 + o + kG15R) * SS + lane + kG15R];
            const float diffusionCoef = DIRECT ? 1.0f - a0[i] * a1[i] : (INT ? 1.0f - al0[o] * al1[o] : cf[o]);
            dst[i] = make_float2(diffusionCoef * sx + (1.0f - diffusionCoef) * f.x, diffusionCoef * sy + (1.0f - diffusionCoef) * f.y);
          } else {
            dst[i] = make_float2(sx, sy);
          }
        }
      }
    };
    if (!DIRECT && is_interior(tile)) do_tile(G15Flag<true>{}); else do_tile(G15Flag<false>{});
    tile = next;
    if (tile < ntiles) __syncthreads();   // every read of this tile's LDS is done before the next one is stored
  }
}
static inline void gauss15_grid(int w, int h, int& ntx, int& ntiles, unsigned& blocks, int nbatch = 1) {
  ntx = (w + kG15TX - 1) / kG15TX;
  ntiles = ntx * ((h + kG15TY - 1) / kG15TY);
  static const int cap = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 768; return 3 * p.multiProcessorCount; }();
  int per = cap / (nbatch > 0 ? nbatch : 1);   // the persistent blocks of all pairs of a batch share the chip
  if (per < 1) per = 1;
  blocks = (unsigned)(ntiles < per ? ntiles : per);
}
void launch_gauss15(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, const Gauss& g15, Batch bt) {
  (void)tmp;
  int ntx, ntiles; unsigned blocks;
  gauss15_grid(w, h, ntx, ntiles, blocks, bt.n);
  hipLaunchKernelGGL((k_gauss15_fused<false, false, false>), dim3(blocks, 1, bt.n), dim3(256), 0, st, reinterpret_cast<const float2*>(src), reinterpret_cast<float2*>(dst), w, h, g15, nullptr, nullptr, ntx, ntiles, UpsSrc{}, bt.stride);
}

// upsample (coarse sw x sh -> w x h, times mul) + Gauss15 of the upsampled plane in one launch: `up` receives the upsampled flow
void launch_gauss15_upsample(hipStream_t st, const float* coarse, int sw, int sh, float mul, float* up, float* dst, int w, int h, const Gauss& g15, Batch bt) {
  int ntx, ntiles; unsigned blocks;
  gauss15_grid(w, h, ntx, ntiles, blocks, bt.n);
  const UpsSrc u{reinterpret_cast<const float2*>(coarse), sw, sh, 1. / ((double)w / sw), 1. / ((double)h / sh), mul, reinterpret_cast<float2*>(up)};
  hipLaunchKernelGGL((k_gauss15_fused<false, true, false>), dim3(blocks, 1, bt.n), dim3(256), 0, st, nullptr, reinterpret_cast<float2*>(dst), w, h, g15, nullptr, nullptr, ntx, ntiles, u, bt.stride);
}

// K8 lowAlphaFlowDiffusion (PixFlow.hpp:388-405): column pass fused with the alpha mix.
__global__ __launch_bounds__(256) void k_gauss15_col_mix(const float2* __restrict__ tmp, const float2* __restrict__ flow, const float* __restrict__ a0,
                                                         const float* __restrict__ a1, float2* __restrict__ out, int w, int h, Gauss g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t i = size_t(y) * w + x;
  const float2 b = d_gauss15_col(tmp, w, h, x, y, g);
  const float2 f = flow[i];
  const float diffusionCoef = 1.0f - a0[i] * a1[i];
  out[i] = make_float2(diffusionCoef * b.x + (1.0f - diffusionCoef) * f.x, diffusionCoef * b.y + (1.0f - diffusionCoef) * f.y);
}
void launch_gauss15_mix(hipStream_t st, float* flow, float* tmp, const float* a0, const float* a1, int w, int h, const Gauss& g15, float* out, Batch bt) {
  (void)tmp;
  int ntx, ntiles; unsigned blocks;
  gauss15_grid(w, h, ntx, ntiles, blocks, bt.n);
  hipLaunchKernelGGL((k_gauss15_fused<true, false, false>), dim3(blocks, 1, bt.n), dim3(256), 0, st, reinterpret_cast<const float2*>(flow), reinterpret_cast<float2*>(out), w, h, g15, a0, a1, ntx, ntiles, UpsSrc{}, bt.stride);
}

// Direct form: every thread loads the 6 x 5 neighbourhood of its output pair straight from memory (the 15x re-reads are served by
// L1 / L2).  Lowest latency per launch: used for the levels where a launch is latency-bound (below kMedTiledMinPx pixels).
__global__ __launch_bounds__(256) void k_median5(const float2* __restrict__ src, float2* __restrict__ dst, int w, int h, size_t bstride) {
  const int xp = (blockIdx.x * blockDim.x + threadIdx.x) * 2, y = blockIdx.y;   // outputs xp and xp + 1
  if (xp >= w) return;
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(src, bo); PF_BOFF(dst, bo); }
  float2 col[6][5];   // columns xp-2 .. xp+3 (replicate border), rows y-2 .. y+2
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float2* r = src + size_t(d_replicate(y + j - 2, h)) * w;
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i][j] = r[d_replicate(xp + i - 2, w)];
  }
  float2 m0, m1;
  d_median_pair(col, m0, m1);
  dst[size_t(y) * w + xp] = m0;
  if (xp + 1 < w) dst[size_t(y) * w + xp + 1] = m1;
}
// A block of 256 threads owns a 32 x 16 output tile.  The tile + its 2-pixel halo (replicate border) is staged in LDS once -- 1.41
// global loads per output instead of 15 through L1 -- and every thread then selects 2 horizontally adjacent outputs of one row
// from LDS (16-byte reads: the six columns of an output pair are three float4).  Same generated selection network (d_median_pair).
// Measured per launch beside the other direction's kernels (tests/micro/kern_by_grid.sh): 4950x2000 123 vs 139 us, 4455x1800 112 vs
// 124 (1024-thread blocks; they lost below ~2.3 Mpix): levels >= kMedTiledMinPx.
// Tile shape: until round 4 a block was 1024 threads (128 x 16) -- at 71 VGPRs ONE such block fits a CU, so every block's HBM round trip
// stood in front of its network with nothing beside it.  Small blocks interleave (profiles/r04_median_tile_ab.txt: median family of a
// dense pair 3.11 -> 2.78 ms, 8 pairs in flight +1.5 %).
#ifndef PF_MEDX
#define PF_MEDX 32
#define PF_MEDY 16
#endif
constexpr int kMedX = PF_MEDX, kMedY = PF_MEDY, kMedSX = kMedX + 4, kMedSY = kMedY + 4, kMedT = (kMedX / 2) * kMedY;
__global__ __launch_bounds__(kMedT) void k_median5_tiled(const float2* __restrict__ src, float2* __restrict__ dst, int w, int h, size_t bstride) {
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(src, bo); PF_BOFF(dst, bo); }
  __shared__ __attribute__((aligned(16))) float2 tile[kMedSY][kMedSX];
  const int x0 = blockIdx.x * kMedX, y0 = blockIdx.y * kMedY;
  {   // all of a thread's loads are issued before the first LDS store: one HBM round trip per block
    constexpr int kN = (kMedSY * kMedSX + kMedT - 1) / kMedT;
    float2 v[kN];
#pragma unroll
    for (int u = 0; u < kN; ++u) {
      const int i = threadIdx.x + u * kMedT, ty = i / kMedSX, tx = i - ty * kMedSX;
      v[u] = i < kMedSY * kMedSX ? src[size_t(d_replicate(y0 + ty - 2, h)) * w + d_replicate(x0 + tx - 2, w)] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kN; ++u) {
      const int i = threadIdx.x + u * kMedT;
      if (i < kMedSY * kMedSX) (&tile[0][0])[i] = v[u];
    }
  }
  __syncthreads();
  const int lx = (threadIdx.x % (kMedX / 2)) * 2, xp = x0 + lx;   // outputs xp and xp + 1
  const int ly = threadIdx.x / (kMedX / 2), y = y0 + ly;
  if (xp >= w || y >= h) return;
  float2 col[6][5];   // columns xp-2 .. xp+3, rows y-2 .. y+2
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float4* rp = reinterpret_cast<const float4*>(&tile[ly + j][lx]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float4 v = rp[i]; col[2 * i][j] = make_float2(v.x, v.y); col[2 * i + 1][j] = make_float2(v.z, v.w); }
  }
  float2 m0, m1;
  d_median_pair(col, m0, m1);
  if (xp + 1 < w && ((w & 1) == 0)) *reinterpret_cast<float4*>(&dst[size_t(y) * w + xp]) = make_float4(m0.x, m0.y, m1.x, m1.y);
  else {
    dst[size_t(y) * w + xp] = m0;
    if (xp + 1 < w) dst[size_t(y) * w + xp + 1] = m1;
  }
}
// (threshold re-measured with the small blocks, same file: 3 M -> 20 k pixels; the direct form only keeps the levels of a few blocks)
#ifndef PF_MED_MINPX
#define PF_MED_MINPX 20000
#endif
constexpr long kMedTiledMinPx = PF_MED_MINPX;
void launch_median5_form(hipStream_t st, const float* src, float* dst, int w, int h, bool tiled, Batch bt) {
  if (tiled) {
    dim3 grid((w + kMedX - 1) / kMedX, (h + kMedY - 1) / kMedY, bt.n);
    hipLaunchKernelGGL(k_median5_tiled, grid, dim3(kMedT), 0, st, reinterpret_cast<const float2*>(src), reinterpret_cast<float2*>(dst), w, h, bt.stride);
  } else {
    dim3 grid(((w + 1) / 2 + 255) / 256, h, bt.n);
    hipLaunchKernelGGL(k_median5, grid, dim3(256), 0, st, reinterpret_cast<const float2*>(src), reinterpret_cast<float2*>(dst), w, h, bt.stride);
  }
}
void launch_median5(hipStream_t st, const float* src, float* dst, int w, int h, Batch bt) { launch_median5_form(st, src, dst, w, h, (long)w * h >= kMedTiledMinPx, bt); }

// medianBlur(5) + lowAlphaFlowDiffusion in one launch: `flow` is the backward sweep's output, `out` a different plane
void launch_median_gauss15_mix(hipStream_t st, const float* flow, const float* a0, const float* a1, int w, int h, const Gauss& g15, float* out, Batch bt) {
  int ntx, ntiles; unsigned blocks;
  gauss15_grid(w, h, ntx, ntiles, blocks, bt.n);
  hipLaunchKernelGGL((k_gauss15_fused<true, false, true>), dim3(blocks, 1, bt.n), dim3(256), 0, st, reinterpret_cast<const float2*>(flow), reinterpret_cast<float2*>(out), w, h, g15, a0, a1, ntx, ntiles, UpsSrc{}, bt.stride);
}

// ------------------------------------------------------------------------------------------------
// K9 inter-level upsample: resize INTER_CUBIC on float2 then *= 1/0.9f (PixFlow.hpp:122-125).
// [OpenCV imgwarp.cpp] HResizeCubic (taps clamped to the row) then VResizeCubic (rows clipped).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample_cubic(const float2* __restrict__ src, int sw, int sh, float2* __restrict__ dst, int dw, int dh,
                                                        double scale_x, double scale_y, float mul, size_t bstride) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(src, bo); PF_BOFF(dst, bo); }
  dst[size_t(dy) * dw + dx] = d_upsample_cubic_px(src, sw, sh, dx, dy, scale_x, scale_y, mul);
}
// Tiled form: a block owns a 64 x 16 output tile.  The horizontal pass (HResizeCubic) of every source row the tile's vertical
// pass touches (at most kUpRows: 16 * 0.9 + 4 at the pyramid's scale) is computed once per output column into LDS, then the
// vertical pass (VResizeCubic) reads four LDS rows: ~5 global loads per output instead of 16, same expressions.
constexpr int kUpX = 64, kUpY = 16, kUpRows = 24;
__global__ __launch_bounds__(256) void k_upsample_cubic_tiled(const float2* __restrict__ src, int sw, int sh, float2* __restrict__ dst, int dw, int dh,
                                                              double scale_x, double scale_y, float mul, size_t bstride) {
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(src, bo); PF_BOFF(dst, bo); }
  __shared__ float2 hp[kUpRows][kUpX];
  const int x0 = blockIdx.x * kUpX, y0 = blockIdx.y * kUpY;
  const int tx = threadIdx.x & (kUpX - 1), ty4 = threadIdx.x >> 6;
  const int dx = min(x0 + tx, dw - 1);
  int syLo, syHi; float fdummy;
  d_src_coord(y0, scale_y, syLo, fdummy);
  d_src_coord(min(y0 + kUpY, dh) - 1, scale_y, syHi, fdummy);
  const int rowLo = syLo - 1, nrows = syHi + 2 - rowLo + 1;   // block-uniform; the host only launches this kernel when nrows <= kUpRows
  int sx; float fx;
  d_src_coord(dx, scale_x, sx, fx);
  float a[4];
  d_cubic_coeffs(fx, a);
  const int xa = d_replicate(sx - 1, sw), xb = d_replicate(sx, sw), xc = d_replicate(sx + 1, sw), xd = d_replicate(sx + 2, sw);
  for (int j = ty4; j < nrows; j += 4) {
    const float2* r = src + size_t(d_replicate(rowLo + j, sh)) * sw;
    const float2 p0 = r[xa], p1 = r[xb], p2 = r[xc], p3 = r[xd];
    hp[j][tx] = make_float2(p0.x * a[0] + p1.x * a[1] + p2.x * a[2] + p3.x * a[3], p0.y * a[0] + p1.y * a[1] + p2.y * a[2] + p3.y * a[3]);
  }
  __syncthreads();
  if (x0 + tx >= dw) return;
  for (int oy = ty4; oy < kUpY; oy += 4) {
    const int dy = y0 + oy;
    if (dy >= dh) break;
    int sy; float fy;
    d_src_coord(dy, scale_y, sy, fy);
    float b[4];
    d_cubic_coeffs(fy, b);
    const int j0 = sy - 1 - rowLo;
    const float2 h0 = hp[j0][tx], h1 = hp[j0 + 1][tx], h2 = hp[j0 + 2][tx], h3 = hp[j0 + 3][tx];
    const float ox = h0.x * b[0] + h1.x * b[1] + h2.x * b[2] + h3.x * b[3];
    const float oy2 = h0.y * b[0] + h1.y * b[1] + h2.y * b[2] + h3.y * b[3];
    dst[size_t(dy) * dw + x0 + tx] = make_float2(ox * mul + 0.0f, oy2 * mul + 0.0f);
  }
}
void launch_upsample_cubic(hipStream_t st, const float* src, int sw, int sh, float* dst, int dw, int dh, float mul, Batch bt) {
  const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
  if (kUpY * sy + 5.0 <= kUpRows) {   // source rows a 16-row output tile can touch: floor(15 * sy) + 4 (+1 for rounding)
    dim3 grid((dw + kUpX - 1) / kUpX, (dh + kUpY - 1) / kUpY, bt.n);
    hipLaunchKernelGGL(k_upsample_cubic_tiled, grid, dim3(256), 0, st, reinterpret_cast<const float2*>(src), sw, sh, reinterpret_cast<float2*>(dst), dw, dh, sx, sy, mul, bt.stride);
    return;
  }
  dim3 grid((dw + 255) / 256, dh, bt.n);
  hipLaunchKernelGGL(k_upsample_cubic, grid, dim3(256), 0, st, reinterpret_cast<const float2*>(src), sw, sh, reinterpret_cast<float2*>(dst), dw, dh, sx,
                     sy, mul, bt.stride);
}

// ------------------------------------------------------------------------------------------------
// K10 final: resize INTER_LINEAR to the padded full-res size, *= 1/0.5f, GaussianBlur 3x3 s1.0
// (PixFlow.hpp:128-134), then crop `pad` columns each side (OpticalFlow.cpp:143-144).  Fused: each
// output evaluates the 3x3 neighbourhood of the (virtual) upsampled image.
// ------------------------------------------------------------------------------------------------
// Tiled: a block owns a 64 x 16 output tile; the (virtual) upsampled image is evaluated ONCE per pixel of the tile plus a
// one-pixel ring (reflect-101) into LDS, then the 3x3 Gaussian (row pass SymmRowSmallFilter, column pass SymmColumnFilter,
// same expressions as before) reads LDS -- 1.2 bilinear samples per output instead of 9.
constexpr int kFFX = 64, kFFY = 16;
__global__ __launch_bounds__(256) void k_final_flow(const float* __restrict__ flow0, int sw, int sh, int pad_cols, int rows, int pad, double scale_x,
                                                    double scale_y, float mul, Gauss g, ExtPtrs outs, size_t bstride) {
  PF_BOFF(flow0, size_t(blockIdx.z) * bstride);
  float2* __restrict__ out = static_cast<float2*>(const_cast<void*>(outs.p[blockIdx.z]));   // caller-owned (or internal) output plane of this pair
  __shared__ float2 up[kFFY + 2][kFFX + 2];
  const int cols = pad_cols - 2 * pad;
  const int x0 = blockIdx.x * kFFX, y0 = blockIdx.y * kFFY;
  for (int t = threadIdx.x; t < (kFFY + 2) * (kFFX + 2); t += 256) {
    const int r = t / (kFFX + 2), cc = t - r * (kFFX + 2);
    const int yy = d_reflect101(y0 - 1 + r, rows), xx = d_reflect101(x0 + pad - 1 + cc, pad_cols);
    float v[2];
    d_resize_linear_px<2>(flow0, sw, sh, pad_cols, rows, scale_x, scale_y, xx, yy, v);
    up[r][cc] = make_float2(v[0] * mul + 0.0f, v[1] * mul + 0.0f);
  }
  __syncthreads();
  const float k0 = g.k[1], k1 = g.k[2];
  const int tx = threadIdx.x & (kFFX - 1);
  const int x = x0 + tx;
  if (x >= cols) return;
  for (int oy = threadIdx.x >> 6; oy < kFFY; oy += 4) {
    const int y = y0 + oy;
    if (y >= rows) break;
    float tx3[3], ty3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float2 u0 = up[oy + j][tx], u1 = up[oy + j][tx + 1], u2 = up[oy + j][tx + 2];
      tx3[j] = u1.x * k0 + (u0.x + u2.x) * k1;
      ty3[j] = u1.y * k0 + (u0.y + u2.y) * k1;
    }
    float ox = k0 * tx3[1] + 0.0f; ox += k1 * (tx3[2] + tx3[0]);
    float oy2 = k0 * ty3[1] + 0.0f; oy2 += k1 * (ty3[2] + ty3[0]);
    out[size_t(y) * cols + x] = make_float2(ox, oy2);
  }
}
void launch_final_flow(hipStream_t st, const float* flow0, int sw, int sh, int pad_cols, int rows, int pad, float mul, const Gauss& g3, float* out, Batch bt,
                       const ExtPtrs* outs) {
  const double sx = 1. / ((double)pad_cols / sw), sy = 1. / ((double)rows / sh);
  const int cols = pad_cols - 2 * pad;
  dim3 grid((cols + kFFX - 1) / kFFX, (rows + kFFY - 1) / kFFY, bt.n);
  ExtPtrs e{}; if (outs) e = *outs; else e.p[0] = out;
  hipLaunchKernelGGL(k_final_flow, grid, dim3(256), 0, st, flow0, sw, sh, pad_cols, rows, pad, sx, sy, mul, g3, e, bt.stride);
}

}  // namespace pf
