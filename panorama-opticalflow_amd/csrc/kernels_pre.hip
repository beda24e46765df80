// Pre-processing and resampling kernels:
//   K1  wrap-pad + 8-bit fixed-point bicubic downscale + BGRA->gray + /255   (CPU/OpticalFlow.cpp:113-126, CPU/PixFlow.hpp:78-100)
//   K1b small symmetric Gaussian (5x5 s0.25 pre-blur, PixFlow.hpp:102-103)
//   K2  bilinear 0.9x pyramid step for the 4 planes (PixFlow.hpp:137-151)
// All are streaming, HBM-bound kernels: one thread per output element, coalesced along x.
#include "pf_common.hpp"

namespace pf {

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// [OpenCV imgwarp.cpp] INTER_CUBIC 8UC4: HResizeCubic<uchar,int,short> + VResizeCubic/FixedPtCast<int,uchar,22>.
// The padded image [last `pad` cols | image | first `pad` cols] is virtual: taps are re-mapped.
__global__ __launch_bounds__(256) void k_downscale_gray(ExtPtrs imgs, int cols, int rows, int pad, float* __restrict__ gray,
                                                        float* __restrict__ alpha, int dw, int dh, double scale_x, double scale_y, size_t bstride) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  const uint8_t* __restrict__ bgra = static_cast<const uint8_t*>(imgs.p[blockIdx.z]);   // caller-owned: one pointer per pair
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(gray, bo); PF_BOFF(alpha, bo); }
  const int ce = cols + 2 * pad;
  int sx, sy; float fx, fy;
  d_src_coord(dx, scale_x, sx, fx);
  d_src_coord(dy, scale_y, sy, fy);
  float cx[4], cy[4];
  d_cubic_coeffs(fx, cx);
  d_cubic_coeffs(fy, cy);
  int ia[4], ib[4], xs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ia[i] = sat_short(__float2int_rn(cx[i] * 2048.f));
    ib[i] = sat_short(__float2int_rn(cy[i] * 2048.f));
    int xp = d_replicate(sx - 1 + i, ce) - pad;
    if (xp < 0) xp += cols; else if (xp >= cols) xp -= cols;
    xs[i] = xp;
  }
  int acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = d_replicate(sy - 1 + j, rows);
    const uchar4* row = reinterpret_cast<const uchar4*>(bgra) + size_t(yy) * cols;
    int h[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uchar4 p = row[xs[i]];
      h[0] += int(p.x) * ia[i]; h[1] += int(p.y) * ia[i]; h[2] += int(p.z) * ia[i]; h[3] += int(p.w) * ia[i];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += h[c] * ib[j];
  }
  int px[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { int v = (acc[c] + (1 << 21)) >> 22; px[c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
  const int g = (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14;  // [OpenCV color.cpp] BGRA2GRAY, 8u
  const float inv255 = (float)(1.0 / 255.0f);
  gray[size_t(dy) * dw + dx] = float(g) * inv255 + 0.0f;
  alpha[size_t(dy) * dw + dx] = float(px[3]) * inv255 + 0.0f;
}

void launch_downscale_gray(hipStream_t st, const uint8_t* bgra, int cols, int rows, int pad, float* gray, float* alpha, int dw, int dh, Batch bt,
                           const ExtPtrs* imgs) {
  const int ce = cols + 2 * pad;
  const double sx = 1. / ((double)dw / ce), sy = 1. / ((double)dh / rows);
  dim3 grid((dw + 255) / 256, dh, bt.n);
  ExtPtrs e{}; if (imgs) e = *imgs; else e.p[0] = bgra;
  hipLaunchKernelGGL(k_downscale_gray, grid, dim3(256), 0, st, e, cols, rows, pad, gray, alpha, dw, dh, sx, sy, bt.stride);
}

// [OpenCV filter.cpp] separable symmetric Gaussian, ksize 3 or 5, BORDER_REFLECT_101:
// row pass SymmRowSmallFilter (centre*k0 + (l+r)*k1 + ...), column pass SymmColumnFilter.
template <int R, int CN>
__global__ __launch_bounds__(256) void k_gauss_small(const float* __restrict__ src, float* __restrict__ dst, int w, int h, Gauss g, size_t bstride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(src, bo); PF_BOFF(dst, bo); }
  int xi[2 * R + 1];
#pragma unroll
  for (int i = -R; i <= R; ++i) xi[i + R] = d_reflect101(x + i, w);
  const float* kc = &g.k[R];
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    float t[2 * R + 1];
#pragma unroll
    for (int j = -R; j <= R; ++j) {
      const float* row = src + size_t(d_reflect101(y + j, h)) * w * CN;
      float s = row[xi[R] * CN + c] * kc[0];
#pragma unroll
      for (int i = 1; i <= R; ++i) s = s + (row[xi[R - i] * CN + c] + row[xi[R + i] * CN + c]) * kc[i];
      t[j + R] = s;
    }
    float s = kc[0] * t[R] + 0.0f;
#pragma unroll
    for (int j = 1; j <= R; ++j) s += kc[j] * (t[R + j] + t[R - j]);
    dst[(size_t(y) * w + x) * CN + c] = s;
  }
}

void launch_gauss_small(hipStream_t st, const float* src, float* dst, int w, int h, int cn, const Gauss& g, Batch bt) {
  dim3 grid((w + 255) / 256, h, bt.n);
  if (g.ksize == 3 && cn == 1) hipLaunchKernelGGL((k_gauss_small<1, 1>), grid, dim3(256), 0, st, src, dst, w, h, g, bt.stride);
  else if (g.ksize == 3 && cn == 2) hipLaunchKernelGGL((k_gauss_small<1, 2>), grid, dim3(256), 0, st, src, dst, w, h, g, bt.stride);
  else if (g.ksize == 5 && cn == 1) hipLaunchKernelGGL((k_gauss_small<2, 1>), grid, dim3(256), 0, st, src, dst, w, h, g, bt.stride);
  else hipLaunchKernelGGL((k_gauss_small<2, 2>), grid, dim3(256), 0, st, src, dst, w, h, g, bt.stride);
}

template <int CN>
__global__ __launch_bounds__(256) void k_resize_linear(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw, int dh,
                                                       double scale_x, double scale_y, float mul, int do_mul) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  float v[CN];
  d_resize_linear_px<CN>(src, sw, sh, dw, dh, scale_x, scale_y, dx, dy, v);
#pragma unroll
  for (int c = 0; c < CN; ++c) dst[(size_t(dy) * dw + dx) * CN + c] = do_mul ? v[c] * mul + 0.0f : v[c];
}

void launch_resize_linear(hipStream_t st, const float* src, int sw, int sh, float* dst, int dw, int dh, int cn, float mul, bool do_mul) {
  const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
  dim3 grid((dw + 255) / 256, dh);
  if (cn == 1) hipLaunchKernelGGL((k_resize_linear<1>), grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh, sx, sy, mul, int(do_mul));
  else hipLaunchKernelGGL((k_resize_linear<2>), grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh, sx, sy, mul, int(do_mul));
}

struct Ptr4 { const float* s[4]; float* d[4]; };
__global__ __launch_bounds__(256) void k_pyr_down4(Ptr4 p, int sw, int sh, int dw, int dh, double scale_x, double scale_y, int nplanes, size_t bstride) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, pl = blockIdx.z % nplanes;
  if (dx >= dw) return;
  const size_t bo = size_t(blockIdx.z / nplanes) * bstride;   // z = plane + nplanes * pair
  const float* src = p.s[pl]; float* dst = p.d[pl];
  PF_BOFF(src, bo); PF_BOFF(dst, bo);
  float v[1];
  d_resize_linear_px<1>(src, sw, sh, dw, dh, scale_x, scale_y, dx, dy, v);
  dst[size_t(dy) * dw + dx] = v[0];
}

void launch_pyr_down4(hipStream_t st, const float* s0, const float* s1, const float* s2, const float* s3, int sw, int sh, float* d0,
                      float* d1, float* d2, float* d3, int dw, int dh, Batch bt) {
  Ptr4 p{{s0, s1, s2, s3}, {d0, d1, d2, d3}};
  const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
  dim3 grid((dw + 255) / 256, dh, 4 * bt.n);
  hipLaunchKernelGGL(k_pyr_down4, grid, dim3(256), 0, st, p, sw, sh, dw, dh, sx, sy, 4, bt.stride);
}

// two planes per launch: the alpha pyramids and the grey pyramids are built on different streams (pf_api.hip: solve)
void launch_pyr_down2(hipStream_t st, const float* s0, const float* s1, int sw, int sh, float* d0, float* d1, int dw, int dh) {
  Ptr4 p{{s0, s1, s0, s1}, {d0, d1, d0, d1}};
  const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
  dim3 grid((dw + 255) / 256, dh, 2);
  hipLaunchKernelGGL(k_pyr_down4, grid, dim3(256), 0, st, p, sw, sh, dw, dh, sx, sy, 2, size_t(0));
}

// Several pyramid levels per launch.  The levels form a dependency chain (level l+1 is a bilinear resize of level l), and for the
// small levels a launch costs ~5 us whatever it does -- 36 dependent launches are ~0.2 ms in front of everything else.  A pixel
// of level l+2 only needs four pixels of level l+1, which are four pixels of level l each: computing them on the fly with the
// very same expressions gives the same bits, so one launch can write levels l+1 .. l+K from level l (4^K loads per pixel of the
// last one: only worth it where the levels are small).
struct PyrChain { int n; int w[4], h[4]; double sx[4], sy[4]; size_t off[4]; };   // [0] = source level, [1..n] = levels written
template <int D>
__device__ __forceinline__ float d_pyr_virtual(const float* __restrict__ base, const PyrChain& c, int x, int y) {
  // value of pixel (x, y) of chain level D, computed from the stored level 0
  if constexpr (D == 0) {
    return base[c.off[0] + size_t(y) * c.w[0] + x];
  } else {
    const int sw = c.w[D - 1], sh = c.h[D - 1];
    int sx, sy; float fx, fy;
    d_src_coord(x, c.sx[D], sx, fx);
    d_src_coord(y, c.sy[D], sy, fy);
    if (sx < 0) { fx = 0; sx = 0; }
    const bool tail = (sx + 1 >= sw);
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int y0 = d_replicate(sy, sh), y1 = d_replicate(sy + 1, sh);
    float h0, h1;
    if (tail) { h0 = d_pyr_virtual<D - 1>(base, c, sx, y0) * 1.0f; h1 = d_pyr_virtual<D - 1>(base, c, sx, y1) * 1.0f; }
    else {
      h0 = d_pyr_virtual<D - 1>(base, c, sx, y0) * a0 + d_pyr_virtual<D - 1>(base, c, sx + 1, y0) * a1;
      h1 = d_pyr_virtual<D - 1>(base, c, sx, y1) * a0 + d_pyr_virtual<D - 1>(base, c, sx + 1, y1) * a1;
    }
    return h0 * b0 + h1 * b1;
  }
}
struct Ptr4P { float* p[4]; };
__global__ __launch_bounds__(256) void k_pyr_chain(Ptr4P planes, PyrChain c, int rows1, int rows2, size_t bstride) {
  // blockIdx.y walks the rows of level 1, then of level 2, then of level 3 of the chain; blockIdx.z = plane + 4 * pair
  float* base = planes.p[blockIdx.z & 3];
  PF_BOFF(base, size_t(blockIdx.z >> 2) * bstride);
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y, lvl = 1;
  if (y >= rows1) { y -= rows1; lvl = 2; if (y >= rows2) { y -= rows2; lvl = 3; } }
  if (x >= c.w[lvl]) return;
  float v;
  if (lvl == 1) v = d_pyr_virtual<1>(base, c, x, y);
  else if (lvl == 2) v = d_pyr_virtual<2>(base, c, x, y);
  else v = d_pyr_virtual<3>(base, c, x, y);
  base[c.off[lvl] + size_t(y) * c.w[lvl] + x] = v;
}
// planes p0..p3 hold every level at offset off[l]; writes levels first+1 .. first+k (k = 1..3) from level `first`
void launch_pyr_chain4(hipStream_t st, float* p0, float* p1, float* p2, float* p3, const int* ws, const int* hs, const size_t* off, int first, int k, Batch bt) {
  PyrChain c; c.n = k;
  for (int i = 0; i <= 3; ++i) {
    const int l = first + (i <= k ? i : k);
    c.w[i] = ws[l]; c.h[i] = hs[l]; c.off[i] = off[l];
    c.sx[i] = i ? 1. / ((double)ws[l] / ws[l - 1]) : 1.; c.sy[i] = i ? 1. / ((double)hs[l] / hs[l - 1]) : 1.;
  }
  Ptr4P pl{{p0, p1, p2, p3}};
  const int rows1 = c.h[1], rows2 = k >= 2 ? c.h[2] : 0, rows3 = k >= 3 ? c.h[3] : 0;
  dim3 grid((c.w[1] + 255) / 256, rows1 + rows2 + rows3, 4 * bt.n);
  hipLaunchKernelGGL(k_pyr_chain, grid, dim3(256), 0, st, pl, c, rows1, rows2, bt.stride);
}

template <class T>
__global__ void k_fill(T* p, size_t n, T v, size_t bstride) {
  PF_BOFF(p, size_t(blockIdx.z) * bstride);
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
void launch_fill_u64(hipStream_t st, unsigned long long* p, size_t n, unsigned long long v, Batch bt) {
  if (n == 0) return;
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL((k_fill<unsigned long long>), dim3(blocks, 1, bt.n), dim3(256), 0, st, p, n, v, bt.stride);
}
void launch_fill_u32(hipStream_t st, unsigned* p, size_t n, unsigned v, Batch bt) {
  if (n == 0) return;
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL((k_fill<unsigned>), dim3(blocks, 1, bt.n), dim3(256), 0, st, p, n, v, bt.stride);
}

// Did any sweep launch of a solve give up (ctrl[4*l+1], ctrl[4*l+3])?  One word in mapped pinned host memory per
// context: the host reads it after the stream sync the call ends with -- no copy, no extra sync.
// 64-bit content checksum of a device buffer: sum over 8-byte words of mix(word ^ index * K) (splitmix64 finaliser), so that
// it is order-independent (one atomic add per wave) yet position-sensitive.  Lets a caller compare results that live on
// different GPUs -- or verify a gather -- without moving them to the host (tools/pano_batch).
__global__ __launch_bounds__(256) void k_checksum64(const unsigned long long* __restrict__ p, size_t nwords, const uint8_t* __restrict__ tail, int ntail,
                                                    unsigned long long* __restrict__ acc) {
  unsigned long long h = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nwords + (ntail ? 1 : 0); i += size_t(gridDim.x) * blockDim.x) {
    unsigned long long w = 0;
    if (i < nwords) w = p[i];
    else for (int k = 0; k < ntail; ++k) w |= (unsigned long long)tail[k] << (8 * k);
    unsigned long long z = w ^ ((i + 1) * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    h += z ^ (z >> 31);
  }
  for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, h);
}
void launch_checksum64(hipStream_t st, const void* p, size_t bytes, unsigned long long* acc /* zeroed by the caller */) {
  const size_t nwords = bytes / 8;
  size_t blocks = (nwords + 1 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_checksum64, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const unsigned long long*>(p), nwords,
                     static_cast<const uint8_t*>(p) + nwords * 8, int(bytes - nwords * 8), acc);
}

__global__ __launch_bounds__(256) void k_count_diff_u32(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, int* __restrict__ count) {
  int d = 0;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) d += a[i] != b[i];
  if (d) atomicAdd(count, d);
}
void launch_count_diff_u32(hipStream_t st, const uint32_t* a, const uint32_t* b, size_t n, int* count) {
  size_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_count_diff_u32, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n, count);
}

__global__ void k_collect_status(const int* __restrict__ ctrl, int nwords, int* __restrict__ status, int bit, size_t bstride) {
  PF_BOFF(ctrl, size_t(blockIdx.z) * bstride);   // every pair of a batch reports into the same word
  int bad = 0;
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) if ((i & 1) && ctrl[i]) bad = 1;
  if (__any(bad) && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_collect_status(hipStream_t st, const int* ctrl, int nwords, int* status, int bit, Batch bt) {
  hipLaunchKernelGGL(k_collect_status, dim3(1, 1, bt.n), dim3(64), 0, st, ctrl, nwords, status, bit, bt.stride);
}

}  // namespace pf
