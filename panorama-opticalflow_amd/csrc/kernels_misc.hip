// K4 coarsest-level search (CPU/PixFlow.hpp:153-270), K11 novel-view blend (CPU/OpticalFlow.cpp:9-92),
// K12-K15 StitchTool kernels (CPU/StitchTool.cpp).
#include "pf_common.hpp"
#include "libm_exact.hpp"

namespace pf {

// ------------------------------------------------------------------------------------------------
// K4.  computeIntensityRatio (PixFlow.hpp:190-205) is a sequential fp32 accumulation in row-major
// order; the coarsest level has <= a few thousand pixels, so one lane adds them in that order (the
// products are staged through LDS by the whole block so the serial part only reads LDS).
// ------------------------------------------------------------------------------------------------
// (a device function: k_adjust_initial_flow computes the ratio itself where the level is small -- every block the same serial sum,
// side by side -- which saves the launch in front of it; pl / pr = 2 x 1024 floats of LDS)
__device__ __forceinline__ float d_intensity_ratio(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ a0,
                                                   const float* __restrict__ a1, int n, float* pl, float* pr) {
  float sumL = 0.f, sumR = 0.f;
  for (int base = 0; base < n; base += 1024) {
    for (int i = threadIdx.x; i < 1024 && base + i < n; i += blockDim.x) {
      const float alpha = a0[base + i] * a1[base + i];
      pl[i] = alpha * i0[base + i];
      pr[i] = alpha * i1[base + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // ONE lane adds in the reference's order; eight products of each plane are fetched per step (two 16-byte LDS reads each), so
      // that what paces the loop is the chain of dependent additions (two independent chains side by side), not LDS round trips
      const int m = (n - base) < 1024 ? (n - base) : 1024;
      int i = 0;
      for (; i + 8 <= m; i += 8) {
        const float4 l0 = *reinterpret_cast<const float4*>(&pl[i]), l1 = *reinterpret_cast<const float4*>(&pl[i + 4]);
        const float4 r0 = *reinterpret_cast<const float4*>(&pr[i]), r1 = *reinterpret_cast<const float4*>(&pr[i + 4]);
        sumL += l0.x; sumR += r0.x; sumL += l0.y; sumR += r0.y; sumL += l0.z; sumR += r0.z; sumL += l0.w; sumR += r0.w;
        sumL += l1.x; sumR += r1.x; sumL += l1.y; sumR += r1.y; sumL += l1.z; sumR += r1.z; sumL += l1.w; sumR += r1.w;
      }
      for (; i < m; ++i) { sumL += pl[i]; sumR += pr[i]; }
    }
    __syncthreads();
  }
  return sumL / sumR;   // (thread 0's value is the ratio)
}
__global__ __launch_bounds__(256) void k_intensity_ratio(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ a0,
                                                         const float* __restrict__ a1, int n, float* __restrict__ ratio, size_t bstride) {
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(i0, bo); PF_BOFF(i1, bo); PF_BOFF(a0, bo); PF_BOFF(a1, bo); PF_BOFF(ratio, bo); }
  __shared__ __attribute__((aligned(16))) float pl[1024], pr[1024];
  const float r = d_intensity_ratio(i0, i1, a0, a1, n, pl, pr);
  if (threadIdx.x == 0) *ratio = r;
}

// computePatchError (PixFlow.hpp:157-188) on LDS tiles.  A block owns a segment of 64 pixels of one row; everything its 19 candidates x 25
// taps read is staged once: rows y-2 .. y+2 of I0 and alpha0 (out-of-image taps are SKIPPED by the reference, so the bounds test stays
// with the tap), and the window of the equalised I1 = I1 * ratio + 0 ([OpenCV] Mat * scalar, PixFlow.hpp:235) and of alpha1 that the 5x5
// patch sweeps while it slides over the search box, with the reference's clamped coordinates applied at staging time (a clamped tap of
// candidate (i1x, i1y) is the texel at the clamped ABSOLUTE coordinate, whichever candidate asks).  The sums keep the reference's tap
// order (dy outer, dx inner, sequential fp32), so the bits are the same; round 3 read every tap from global memory -- 4 dependent-latency
// loads x 475 taps per pixel, ~130 us per launch for <= 1.8 k pixels -- which this brings to ~10 us.
struct PatchTiles {
  const float* i0; const float* a0;   // [5][kSegW + 4]: rows y-2 .. y+2, columns x0-2 .. x0+kSegW+1 (unclamped: taps outside the image are skipped)
  const float* i1; const float* a1;   // [th][tw]: rows y-2+by .. , columns x0-2+bx .. (clamped coordinates)
  int tw;
};
constexpr int kSegW = 64;
__device__ __forceinline__ float d_patch_error_lds(const PatchTiles& t, int lx, int i0x, int i0y, int ox, int oy, int w, int h, int dist) {
  // (ox, oy) = candidate offset relative to the box origin: its taps are t.i1[(oy + dy + 2) * tw + lx + ox + dx + 2]
  float sad = 0.f, alpha = 0.f;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const int d0y = i0y + dy;
    const bool rowIn = 0 <= d0y && d0y < h;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int d0x = i0x + dx;
      const bool in = rowIn && 0 <= d0x && d0x < w;
      const int k0 = (dy + 2) * (kSegW + 4) + lx + dx + 2, k1 = (oy + dy + 2) * t.tw + lx + ox + dx + 2;
      const float difference = t.i0[k0] - t.i1[k1];
      const float s1 = sad + fabsf(difference), a1 = alpha + t.a0[k0] * t.a1[k1];
      sad = in ? s1 : sad; alpha = in ? a1 : alpha;   // a skipped tap leaves both sums untouched, as in the reference's loop
    }
  }
  return sad / alpha;
}
__device__ __forceinline__ float d_patch_penalty(float sad, int fxI, int fyI, int dist) {
  const float fx = float(fxI), fy = float(fyI);
  const float length = (float)sqrt((double)fx * fx + (double)fy * fy);   // [OpenCV] cv::norm(Point2f): sqrt in double
  return sad * (1 + length / dist);
}

// adjustInitialFlow (PixFlow.hpp:226-270)
__global__ __launch_bounds__(kSegW) void k_adjust_initial_flow(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ a0,
                                                               const float* __restrict__ a1, int w, int h, int bx, int by, int bw, int bh, int dist,
                                                               const float* __restrict__ ratio_p, float2* __restrict__ flow, size_t bstride) {
  const bool ownRatio = ratio_p == nullptr;   // (before the batch offset is added)
  { const size_t bo = size_t(blockIdx.z) * bstride; PF_BOFF(i0, bo); PF_BOFF(i1, bo); PF_BOFF(a0, bo); PF_BOFF(a1, bo); PF_BOFF(ratio_p, bo); PF_BOFF(flow, bo); }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int x0 = blockIdx.x * kSegW, i0y = blockIdx.y, lx = threadIdx.x, i0x = x0 + lx;
  const int tw = kSegW + 4 + (bw - 1), th = 5 + (bh - 1);
  float ratio;
  if (ownRatio) {
    // small level: the intensity ratio here instead of by a launch of its own (the same sequential sums, in every block)
    float* pl = lds; float* pr = lds + 1024;
    __shared__ float ratio_sh;
    const float r = d_intensity_ratio(i0, i1, a0, a1, w * h, pl, pr);
    if (threadIdx.x == 0) ratio_sh = r;
    __syncthreads();
    ratio = ratio_sh;
    __syncthreads();   // pl / pr are reused by the tiles below
  } else {
    ratio = *ratio_p;
  }
  float* t_i0 = lds; float* t_a0 = t_i0 + 5 * (kSegW + 4); float* t_i1 = t_a0 + 5 * (kSegW + 4); float* t_a1 = t_i1 + th * tw;
  for (int k = lx; k < 5 * (kSegW + 4); k += kSegW) {
    const int ty = k / (kSegW + 4), tx = k - ty * (kSegW + 4);
    const int Y = i0y - 2 + ty, X = x0 - 2 + tx;
    const bool in = 0 <= Y && Y < h && 0 <= X && X < w;
    t_i0[k] = in ? i0[size_t(Y) * w + X] : 0.f; t_a0[k] = in ? a0[size_t(Y) * w + X] : 0.f;
  }
  for (int k = lx; k < th * tw; k += kSegW) {
    const int ty = k / tw, tx = k - ty * tw;
    const size_t o = size_t(d_replicate(i0y - 2 + by + ty, h)) * w + d_replicate(x0 - 2 + bx + tx, w);
    t_i1[k] = i1[o] * ratio + 0.0f; t_a1[k] = a1[o];
  }
  __syncthreads();
  if (i0x >= w) return;
  if (!(t_a0[2 * (kSegW + 4) + lx + 2] > kUpdateAlphaThreshold)) return;
  const PatchTiles t{t_i0, t_a0, t_i1, t_a1, tw};
  const float kFraction = 0.8f;
  // the zero-flow candidate: offset (0, 0) relative to the pixel = (-bx, -by) relative to the box origin (inside the tile for every
  // search box of computeSearchBox: the box always contains the zero flow)
  float errorBest = kFraction * d_patch_penalty(d_patch_error_lds(t, lx, i0x, i0y, -bx, -by, w, h, dist), 0, 0, dist);
  int i1xBest = i0x, i1yBest = i0y;
  for (int dy = by; dy < by + bh; ++dy)
    for (int dx = bx; dx < bx + bw; ++dx) {
      const int i1x = i0x + dx, i1y = i0y + dy;
      if (0 <= i1x && i1x < w && 0 <= i1y && i1y < h) {
        const float error = d_patch_penalty(d_patch_error_lds(t, lx, i0x, i0y, dx - bx, dy - by, w, h, dist), dx, dy, dist);
        if (errorBest > error) { errorBest = error; i1xBest = i1x; i1yBest = i1y; }
      }
    }
  flow[size_t(i0y) * w + i0x] = make_float2(float(i1xBest - i0x), float(i1yBest - i0y));
}

// `flow` must be zero-filled by the caller (PixFlow.hpp:298).  i1eq_tmp: >= 1 float of scratch (the ratio).
void launch_adjust_initial_flow(hipStream_t st, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, int hint,
                                int max_pct, float* ratio_tmp, float* flow, Batch bt) {
  const int dist = (kPyrMinImageSize * max_pct + 50) / 100;  // computeSearchDistance, PixFlow.hpp:153-155
  const int kRatio = 8, ortho = (dist + kRatio / 2) / kRatio, thickness = 2 * ortho + 1;
  int bx, by, bw, bh;  // computeSearchBox, PixFlow.hpp:207-224
  switch (hint) {
    case 1: bx = 0; by = -ortho; bw = dist + 1; bh = thickness; break;       // RIGHT
    case 2: bx = -ortho; by = 0; bw = thickness; bh = dist + 1; break;       // DOWN
    case 3: bx = -dist; by = -ortho; bw = dist + 1; bh = thickness; break;   // LEFT
    case 4: bx = -ortho; by = -dist; bw = thickness; bh = dist + 1; break;   // UP
    default: return;
  }
  // levels up to 4096 pixels (every coarsest level of a pyramid whose images are not extreme strips): the search kernel sums the
  // intensity ratio itself, one launch instead of two (0.13 -> 0.09 ms per stitch step on the 5-step chain)
  const bool fused = w * h <= 4096;
  if (!fused) hipLaunchKernelGGL(k_intensity_ratio, dim3(1, 1, bt.n), dim3(256), 0, st, i0, i1, a0, a1, w * h, ratio_tmp, bt.stride);
  dim3 grid((w + kSegW - 1) / kSegW, h, bt.n);
  size_t lds = (size_t(2) * 5 * (kSegW + 4) + size_t(2) * (5 + bh - 1) * (kSegW + 4 + bw - 1)) * sizeof(float);   // <= ~14 KB at max_percentage 100
  if (fused && lds < 2 * 1024 * sizeof(float)) lds = 2 * 1024 * sizeof(float);
  if (fused) ratio_tmp = nullptr;
  hipLaunchKernelGGL(k_adjust_initial_flow, grid, dim3(kSegW), lds, st, i0, i1, a0, a1, w, h, bx, by, bw, bh, dist, ratio_tmp,
                     reinterpret_cast<float2*>(flow), bt.stride);
}

// ------------------------------------------------------------------------------------------------
// K11 combineNovelViews (OpticalFlow.cpp:30-92).  Streaming: 36 B/pixel + two dependent 4-byte gathers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uchar4 d_novel_view_point(const uchar4* __restrict__ src, float2 flowDir, double t, int x, int y, int cols, int rows) {
  int srcx = int(x + flowDir.x * t);  // fp64, truncation toward zero (OpticalFlow.cpp:17)
  if (srcx > cols - 1) srcx = srcx - cols;
  if (srcx < 0) srcx = srcx + cols;
  // latent single-wrap hazard defined as a true modulo -- behind a test: a flow longer than the image is wide practically never
  // happens, and the emulated integer division cost every pixel ~30 instructions twice
  if (unsigned(srcx) >= unsigned(cols)) { srcx %= cols; if (srcx < 0) srcx += cols; }
  int srcy = int(y + flowDir.y * t);
  if (srcy > rows - 1) srcy = rows - 1;
  if (srcy < 0) srcy = 0;
  return src[size_t(srcy) * cols + srcx];
}
__device__ __forceinline__ float d_lerp(float x0, float x1, float alpha) { return x0 * (1.0f - alpha) + x1 * alpha; }  // util.hpp:93-101

// Two of the blend's per-pixel values only take a few hundred different arguments: the de-ghosting coefficient tanhf(colorDiff * 10)
// with colorDiff = n / 255.0f, n = sum of three byte differences (0..765), and alpha = a / 255.0f, a = 0..255.  Both are tabulated once
// per device with the very expressions the kernel used to evaluate per pixel (tanhf_exact: libm_exact.hpp) -- the same bits, and the
// piecewise tanhf (divergent branches, an IEEE division) and three more IEEE divisions leave the per-pixel path.
__device__ float g_blend_tanh[766];
__device__ float g_blend_alpha[256];
__global__ void k_blend_tables() {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const float kColorDiffCoef = 10.0f;
  if (n < 766) { const float colorDiff = n / 255.0f; g_blend_tanh[n] = pf_libm::tanhf_exact(colorDiff * kColorDiffCoef); }
  if (n < 256) g_blend_alpha[n] = n / 255.0f;
}
void launch_blend_tables(hipStream_t st) { hipLaunchKernelGGL(k_blend_tables, dim3(3), dim3(256), 0, st); }

__device__ __forceinline__ void d_blend_px(const uchar4* __restrict__ L, const uchar4* __restrict__ R, const float2* __restrict__ flowLR,
                                           const float2* __restrict__ flowRL, const float* __restrict__ blend, int cols, int rows,
                                           uchar4* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= cols) return;
  const size_t i = size_t(y) * cols + x;
  const float blendR = blend[i], blendL = 1 - blendR;
  const float2 fLR = flowLR[i], fRL = flowRL[i];
  const uchar4 colorL = d_novel_view_point(L, fRL, (double)blendR, x, y, cols, rows);
  const uchar4 colorR = d_novel_view_point(R, fLR, (double)blendL, x, y, cols, rows);
  uchar4 o;
  if (colorL.w == 0 || colorR.w == 0) {
    o = make_uchar4(0, 0, 0, 0);
  } else {
    const float kSoftmaxSharpness = 10.0f, kFlowMagCoef = 100.0f;
    const float flowMagLR = sqrtf(fLR.x * fLR.x + fLR.y * fLR.y) / float(cols);
    const float flowMagRL = sqrtf(fRL.x * fRL.x + fRL.y * fRL.y) / float(cols);
    const int colorDiffN = abs(int(colorL.x) - int(colorR.x)) + abs(int(colorL.y) - int(colorR.y)) + abs(int(colorL.z) - int(colorR.z));   // colorDiff = N / 255.0f
    // tanhf / exp with the host libm's roundings (libm_exact.hpp): where the two warped colours agree, c*wL + c*wR sits within
    // an ulp of the integer c and a last-place difference in either function flips the truncated byte
    const float deghostCoef = g_blend_tanh[colorDiffN];               // = tanhf(colorDiff * kColorDiffCoef)
    const float alphaL = g_blend_alpha[colorL.w], alphaR = g_blend_alpha[colorR.w];   // = w / 255.0f
    const double expL = pf_libm::exp_exact(kSoftmaxSharpness * blendL * alphaL * (1.0 + kFlowMagCoef * flowMagRL), pf_libm::kExpTab);
    const double expR = pf_libm::exp_exact(kSoftmaxSharpness * blendR * alphaR * (1.0 + kFlowMagCoef * flowMagLR), pf_libm::kExpTab);
    const double sumExp = expL + expR + 0.00001;
    const float softmaxL = float(expL / sumExp), softmaxR = float(expR / sumExp);
    const float wL = d_lerp(blendL, softmaxL, deghostCoef), wR = d_lerp(blendR, softmaxR, deghostCoef);
    o.x = (unsigned char)(int)(float(colorL.x) * wL + float(colorR.x) * wR);
    o.y = (unsigned char)(int)(float(colorL.y) * wL + float(colorR.y) * wR);
    o.z = (unsigned char)(int)(float(colorL.z) * wL + float(colorR.z) * wR);
    o.w = 255;
  }
  out[i] = o;
}
__global__ __launch_bounds__(256) void k_blend(const uchar4* __restrict__ L, const uchar4* __restrict__ R, const float2* __restrict__ flowLR,
                                               const float2* __restrict__ flowRL, const float* __restrict__ blend, int cols, int rows,
                                               uchar4* __restrict__ out) {
  d_blend_px(L, R, flowLR, flowRL, blend, cols, rows, out);
}
__global__ __launch_bounds__(256) void k_blend_batch(BlendPtrs p, int cols, int rows) {
  const int z = blockIdx.z;
  d_blend_px(reinterpret_cast<const uchar4*>(p.L[z]), reinterpret_cast<const uchar4*>(p.R[z]), reinterpret_cast<const float2*>(p.fLR[z]),
             reinterpret_cast<const float2*>(p.fRL[z]), p.blend[z], cols, rows, reinterpret_cast<uchar4*>(p.out[z]));
}
// n pairs in one launch (blockIdx.z = pair): every buffer is caller-owned or per-pair, so the pointers come as tables
void launch_blend_batch(hipStream_t st, const BlendPtrs& p, int n, int cols, int rows) {
  dim3 grid((cols + 255) / 256, rows, n);
  hipLaunchKernelGGL(k_blend_batch, grid, dim3(256), 0, st, p, cols, rows);
}
void launch_blend(hipStream_t st, const uint8_t* L, const uint8_t* R, const float* flowLR, const float* flowRL, const float* blend, int cols,
                  int rows, uint8_t* out) {
  dim3 grid((cols + 255) / 256, rows);
  hipLaunchKernelGGL(k_blend, grid, dim3(256), 0, st, reinterpret_cast<const uchar4*>(L), reinterpret_cast<const uchar4*>(R),
                     reinterpret_cast<const float2*>(flowLR), reinterpret_cast<const float2*>(flowRL), blend, cols, rows, reinterpret_cast<uchar4*>(out));
}

// ------------------------------------------------------------------------------------------------
// K12 MatchImages + overlap masking (StitchTool.cpp:17-33, :38-50)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_images(const uchar4* __restrict__ L, const uchar4* __restrict__ R, int n, uint8_t* __restrict__ map,
                                                      uchar4* __restrict__ ovL, uchar4* __restrict__ ovR) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 l = L[i], r = R[i];
  const uint8_t m = (uint8_t)((l.w > 0 ? 100 : 0) + (r.w > 0 ? 50 : 0));
  map[i] = m;
  const bool ov = m > 140;
  ovL[i] = ov ? l : make_uchar4(0, 0, 0, 0);
  ovR[i] = ov ? r : make_uchar4(0, 0, 0, 0);
}
void launch_match_images(hipStream_t st, const uint8_t* L, const uint8_t* R, int cols, int rows, uint8_t* map, uint8_t* ovL, uint8_t* ovR) {
  const int n = cols * rows;
  hipLaunchKernelGGL(k_match_images, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uchar4*>(L), reinterpret_cast<const uchar4*>(R), n,
                     map, reinterpret_cast<uchar4*>(ovL), reinterpret_cast<uchar4*>(ovR));
}

// ------------------------------------------------------------------------------------------------
// K13 GenerateBlend's per-pixel part + countblend (StitchTool.cpp:98-128, :148-191).  The map extended
// by cols/5 wrapped columns each side (:102-111) is virtual.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_countblend(const uint8_t* __restrict__ map, int cols, int rows, int length, int step, float* __restrict__ blend,
                                                    float* __restrict__ mergedDis) {
  const int x0 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x0 >= cols) return;
  const int MW = cols + 2 * length, MH = rows;
  auto M = [&](int yy, int xe) -> int {
    int sx = xe - length;
    if (sx < 0) sx += cols; else if (sx >= cols) sx -= cols;
    return map[size_t(yy) * cols + sx];
  };
  const int x = x0 + length;
  const int m = M(y, x);
  float b, md = 0.f;
  if (m == 100) b = 0;
  else if (m == 50) b = 1;
  else if (m == 150) {
    float minLdis = float(10 * cols), minRdis = float(10 * cols);
    const double sqrt2 = sqrt(2.0);
    for (int i = 0; i < cols / 2; i = i + step) {
      int v;
      if (x + i < MW) { v = M(y, x + i); if (v == 100 && i < minLdis) minLdis = float(i); if (v == 50 && i < minRdis) minRdis = float(i); }
      if (x - i > 0) { v = M(y, x - i); if (v == 100 && i < minLdis) minLdis = float(i); if (v == 50 && i < minRdis) minRdis = float(i); }
      if (y + i < MH) { v = M(y + i, x); if (v == 100 && i < minLdis) minLdis = float(i); if (v == 50 && i < minRdis) minRdis = float(i); }
      if (y - i > 0) { v = M(y - i, x); if (v == 100 && i < minLdis) minLdis = float(i); if (v == 50 && i < minRdis) minRdis = float(i); }
      const double di = i * sqrt2;
      if (x + i < MW && y + i < MH) { v = M(y + i, x + i); if (v == 100 && di < minLdis) minLdis = float(di); if (v == 50 && di < minRdis) minRdis = float(di); }
      if (x - i > 0 && y - i > 0) { v = M(y - i, x - i); if (v == 100 && di < minLdis) minLdis = float(di); if (v == 50 && di < minRdis) minRdis = float(di); }
      if (x + i < MW && y - i > 0) { v = M(y - i, x + i); if (v == 100 && di < minLdis) minLdis = float(di); if (v == 50 && di < minRdis) minRdis = float(di); }
      if (x - i > 0 && y + i < MH) { v = M(y + i, x - i); if (v == 100 && di < minLdis) minLdis = float(di); if (v == 50 && di < minRdis) minRdis = float(di); }
    }
    b = minLdis / (minRdis + minLdis);
    md = minLdis < minRdis ? minLdis : minRdis;
  } else b = 0.5f;
  blend[size_t(y) * cols + x0] = b;
  mergedDis[size_t(y) * cols + x0] = md;
}
void launch_countblend(hipStream_t st, const uint8_t* map, int cols, int rows, float* blend, float* mergedDis) {
  int step = cols <= rows ? cols / 200 : rows / 200;
  if (step < 1) step = 1;  // reference never terminates for step==0 (inputs < 200 px); defined as 1
  dim3 grid((cols + 255) / 256, rows);
  hipLaunchKernelGGL(k_countblend, grid, dim3(256), 0, st, map, cols, rows, cols / 5, step, blend, mergedDis);
}

// ------------------------------------------------------------------------------------------------
// K14 blend-ramp smoothing (StitchTool.cpp:130-143).  [OpenCV] blur() on CV_32F = RowSum<float,double>
// + ColumnSum<double,float>: sliding sums in double, anchor k/2, BORDER_REFLECT_101 w.r.t. the whole
// image.  The sliding sums are order-dependent along a row/column, so one lane walks each row (row
// pass) and each column (column pass) -- exactly the O(1)/pixel schedule one wants anyway.
// ------------------------------------------------------------------------------------------------
// any kernel width, one lane per row straight from memory (only used beyond kBoxKMax)
__global__ __launch_bounds__(64) void k_box_rows_wide(const float* __restrict__ src, double* __restrict__ rs, int cols, int rows, int k) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= rows) return;
  const int a = k / 2;
  const float* r = src + size_t(y) * cols;
  double s = 0;
  for (int i = 0; i < k; ++i) s += (double)r[d_reflect101(-a + i, cols)];
  rs[size_t(y) * cols] = s;
  for (int x = 1; x < cols; ++x) {
    s += (double)r[d_reflect101(x - a - 1 + k, cols)] - (double)r[d_reflect101(x - a - 1, cols)];
    rs[size_t(y) * cols + x] = s;
  }
}
// Row pass.  A block owns 64 rows and walks them in chunks of 64 columns: the chunk (+ the k-1 taps beside it, reflected) is staged
// in LDS with coalesced loads, lane y then advances ITS row's sliding sum through the chunk in the reference's order -- the
// dependent chain is one fp64 add per pixel, everything else comes from LDS -- and the 64 x 64 sums leave through LDS as coalesced
// 512-byte rows.  (Before: every lane read its own row straight from HBM, 64 cache lines per load: ~150 GB/s.)
constexpr int kBoxR = 64, kBoxC = 64, kBoxKMax = 32;   // 58 KB of LDS (24.8 in + 33.3 out)
__global__ __launch_bounds__(256) void k_box_rows(const float* __restrict__ src, double* __restrict__ rs, int cols, int rows, int k) {
  __shared__ float tin[kBoxR][kBoxC + kBoxKMax + 1];   // odd row stride: the 64 lanes (rows) of the walking wave hit different banks
  __shared__ double tout[kBoxR][kBoxC + 1];
  const int y0 = blockIdx.x * kBoxR, tid = threadIdx.x;
  const int a = k / 2;
  const int wIn = kBoxC + k;   // columns c0 - a - 1 .. c0 + 63 - a - 1 + k: the subtracted tap of the chunk's first pixel .. the added tap of its last
  double s = 0;
  for (int c0 = 0; c0 < cols; c0 += kBoxC) {
    // stage: all four waves, consecutive threads = consecutive columns of one row, 8 independent loads in flight per thread
    const int total = kBoxR * wIn;
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int r = idx / wIn, i = idx - r * wIn;
        const int y = y0 + r < rows ? y0 + r : rows - 1;
        v[u] = idx < total ? src[size_t(y) * cols + d_reflect101(c0 - a - 1 + i, cols)] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int r = idx / wIn, i = idx - r * wIn;
        if (idx < total) tin[r][i] = v[u];
      }
    }
    __syncthreads();
    const int n = cols - c0 < kBoxC ? cols - c0 : kBoxC;
    if (tid < kBoxR) {   // one wave walks the 64 rows: lane = row, the reference's order along x
      const float* t = tin[tid];
      int x = 0;
      if (c0 == 0) {   // RowSum: the first k taps summed left to right, then the sliding update
        for (int i = 0; i < k; ++i) s += (double)t[1 + i];
        tout[tid][0] = s;
        x = 1;
      }
      for (; x < n; ++x) {
        s += (double)t[x + k] - (double)t[x];
        tout[tid][x] = s;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < kBoxR * kBoxC; idx += 256) {
      const int r = idx / kBoxC, x = idx - r * kBoxC;
      if (x < n && y0 + r < rows) rs[size_t(y0 + r) * cols + c0 + x] = tout[r][x];
    }
    // the next chunk's staging writes tin only (read by the walking wave before the barrier above); tout is rewritten after the next barrier
  }
}
// Column pass: lanes are adjacent columns (coalesced already); the two taps of the next 8 rows are loaded before the 8 dependent
// updates, so that a wave has 16 loads in flight instead of one round trip per row.
__global__ __launch_bounds__(64) void k_box_cols(const double* __restrict__ rs, float* __restrict__ dst, int cols, int rows, int k) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= cols) return;
  const int a = k / 2;
  const double scale = 1. / ((double)k * k);
  double sum = 0;
  for (int j = 0; j < k - 1; ++j) sum += rs[size_t(d_reflect101(-a + j, rows)) * cols + x];
  constexpr int U = 8;
  for (int y0 = 0; y0 < rows; y0 += U) {
    double add[U], sub[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int y = y0 + u < rows ? y0 + u : rows - 1;
      add[u] = rs[size_t(d_reflect101(y - a + k - 1, rows)) * cols + x];
      sub[u] = rs[size_t(d_reflect101(y - a, rows)) * cols + x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (y0 + u < rows) {
        const double s0 = sum + add[u];
        dst[size_t(y0 + u) * cols + x] = (float)(s0 * scale);
        sum = s0 - sub[u];
      }
    }
  }
}
void launch_box_blur(hipStream_t st, const float* src, float* dst, double* rowsum_tmp, int cols, int rows, int k) {
  if (k <= kBoxKMax) hipLaunchKernelGGL(k_box_rows, dim3((rows + kBoxR - 1) / kBoxR), dim3(256), 0, st, src, rowsum_tmp, cols, rows, k);
  else hipLaunchKernelGGL(k_box_rows_wide, dim3((rows + 63) / 64), dim3(64), 0, st, src, rowsum_tmp, cols, rows, k);   // canvases beyond 13000 rows
  hipLaunchKernelGGL(k_box_cols, dim3((cols + 63) / 64), dim3(64), 0, st, rowsum_tmp, dst, cols, rows, k);
}

// Conditional per-tile box blur, in place, raster order (StitchTool.cpp:134-141).  A tile reads the
// CURRENT image in a (step+k-1)^2 window, so it depends on raster-earlier tiles within d =
// ceil(max(a, k-1-a)/step) tiles and must precede raster-later ones in that range.  Tiles with equal
// t = tx + (d+1)*ty are mutually independent: a wavefront over tiles, diagonal by diagonal, is exact.
//
// ONE persistent launch (round 2: one launch per diagonal, ~850 of them per 9000x4000 canvas, ~9 us each).  Only tiles well
// inside the overlap are blurred (MergedDis > step), typically a few percent; a first pass counts the active tiles of every
// diagonal, and the blocks then walk the diagonals together, skipping the empty ones WITHOUT synchronising and meeting at a
// grid barrier (agent-scope release / acquire, guide G16; every spin is bounded by wall-clock time) after each non-empty one.
// A tile is done by one 256-thread block: its window is staged in LDS with coalesced independent loads, then RowSum /
// ColumnSum exactly as blur() computes them (sliding fp64 sums, one lane per row, then one lane per column).
struct TileBlurWork { int bar; int err; int cnt[1]; };   // cnt[tmax + 1]; the whole struct is zeroed before every launch
__device__ __forceinline__ bool d_grid_barrier(int* bar, int* err, int target, long long deadline) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores first (G16)
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1;
    for (unsigned spins = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
      __builtin_amdgcn_s_sleep(8);
      if ((spins & 255) == 255 && ((long long)wall_clock64() > deadline || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
__global__ __launch_bounds__(256) void k_tile_blur(float* __restrict__ img, const float* __restrict__ mergedDis, int cols, int rows, int step, int k,
                                                   int dskew, int ntx, int nty, TileBlurWork* __restrict__ wk, long long budget_ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int a = k / 2, nr = step + k - 1;
  double* sm = reinterpret_cast<double*>(smraw);                 // row sums: nr x step
  float* win = reinterpret_cast<float*>(sm + size_t(nr) * step);  // the tile's input window: nr x nr
  const int tid = threadIdx.x, nblk = gridDim.x;
  const long long deadline = (long long)wall_clock64() + budget_ticks;
  const int tmax = (ntx - 1) + dskew * (nty - 1);
  auto active = [&](int tx, int ty) { return mergedDis[size_t(ty) * step * cols + size_t(tx) * step] > step; };
  // pass 0: active tiles per diagonal
  for (int i = blockIdx.x * 256 + tid; i < ntx * nty; i += nblk * 256) {
    const int ty = i / ntx, tx = i - ty * ntx;
    if (active(tx, ty)) __hip_atomic_fetch_add(&wk->cnt[tx + dskew * ty], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int phase = 1;
  if (!d_grid_barrier(&wk->bar, &wk->err, phase * nblk, deadline)) return;
  const double scale = 1. / ((double)k * k);
  for (int t = 0; t <= tmax; ++t) {
    if (__hip_atomic_load(&wk->cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) continue;   // the same answer in every block: no barrier needed
    const int lo = t - (ntx - 1);
    const int ty_min = lo > 0 ? (lo + dskew - 1) / dskew : 0;
    const int ty_max = t / dskew < nty - 1 ? t / dskew : nty - 1;
    for (int ty = ty_min + blockIdx.x; ty <= ty_max; ty += nblk) {
      const int tx = t - dskew * ty;
      if (tx < 0 || tx >= ntx || !active(tx, ty)) continue;   // block-uniform
      const int x0 = tx * step, y0 = ty * step;
      __syncthreads();   // the previous tile's LDS is no longer read
      for (int base = 0; base < nr * nr; base += 256 * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 256 + tid, j = idx / nr, i = idx - j * nr;
          v[u] = idx < nr * nr ? img[size_t(d_reflect101(y0 - a + j, rows)) * cols + d_reflect101(x0 - a + i, cols)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int idx = base + u * 256 + tid; if (idx < nr * nr) win[idx] = v[u]; }
      }
      __syncthreads();
      for (int j = tid; j < nr; j += 256) {
        const float* r = win + j * nr;
        double s = 0;
        for (int i = 0; i < k; ++i) s += (double)r[i];
        sm[j * step] = s;
        for (int x = 1; x < step; ++x) {
          s += (double)r[x - 1 + k] - (double)r[x - 1];
          sm[j * step + x] = s;
        }
      }
      __syncthreads();
      for (int x = tid; x < step; x += 256) {
        double sum = 0;
        for (int j = 0; j < k - 1; ++j) sum += sm[j * step + x];
        for (int y = 0; y < step; ++y) {
          const double s0 = sum + sm[(y + k - 1) * step + x];
          img[size_t(y0 + y) * cols + x0 + x] = (float)(s0 * scale);
          sum = s0 - sm[y * step + x];
        }
      }
    }
    ++phase;
    if (!d_grid_barrier(&wk->bar, &wk->err, phase * nblk, deadline)) return;
  }
}
size_t tile_blur_work_bytes(int cols, int rows, int step, int k) {
  if (step < 1 || k < 1) return 0;
  int nty = 0, ntx = 0;
  for (int y = 0; y + step < rows; y += step) ++nty;
  for (int x = 0; x + step < cols; x += step) ++ntx;
  if (nty <= 0 || ntx <= 0) return 0;
  const int a = k / 2, reach = a > (k - 1 - a) ? a : (k - 1 - a);
  const int dskew = (reach + step - 1) / step + 1;
  return sizeof(TileBlurWork) + sizeof(int) * size_t((ntx - 1) + dskew * (nty - 1) + 1);
}
size_t tile_blur_lds_bytes(int step, int k) { const size_t nr = size_t(step) + k - 1; return nr * step * sizeof(double) + nr * nr * sizeof(float); }
// work: tile_blur_work_bytes() of device memory (zeroed here); work[1] != 0 afterwards = a grid barrier timed out (never expected)
void launch_tile_blur(hipStream_t st, float* blend, const float* mergedDis, int cols, int rows, int step, int k, void* work) {
  if (step < 1 || k < 1) return;
  // tiles: y = 0, step, ... while y+step < rows  (StitchTool.cpp:134-135)
  int nty = 0, ntx = 0;
  for (int y = 0; y + step < rows; y += step) ++nty;
  for (int x = 0; x + step < cols; x += step) ++ntx;
  if (nty <= 0 || ntx <= 0) return;
  const int a = k / 2, reach = a > (k - 1 - a) ? a : (k - 1 - a);
  const int d = (reach + step - 1) / step, dskew = d + 1;
  const size_t shmem = tile_blur_lds_bytes(step, k);
  // large canvases need more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
  if (shmem > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_blur), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipMemsetAsync(work, 0, tile_blur_work_bytes(cols, rows, step, k), st);
  // every block must be resident (grid barrier): one block per CU at most, and no more than the longest diagonal has tiles
  // (the CU count of the device this launch goes to -- contexts on different GPU models may live in one process -- and what the
  // occupancy calculator says fits beside nothing else: a block that can never be resident would leave the others spinning)
  int dev = 0, ncu = 64, per_cu = 1;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_tile_blur), 256, shmem) != hipSuccess || per_cu < 1) per_cu = 1;
  // Few blocks: a diagonal rarely holds more than a dozen active tiles, a barrier among 32 blocks is cheaper than among 128, and
  // in pf_stitch_step this launch runs BESIDE the two flow solves, whose latency-bound sweeps should not share their CUs and
  // the L2 channel of the barrier word with a crowd of pollers.
  int blocks = nty < 32 ? nty : 32;
  if (blocks > ncu * per_cu / 2) blocks = ncu * per_cu / 2;
  if (blocks < 1) blocks = 1;
  const long long budget = 200000000ll * 5;   // 10 s of 100 MHz ticks: the launch may queue behind other work of the process
  hipLaunchKernelGGL(k_tile_blur, dim3(blocks), dim3(256), shmem, st, blend, mergedDis, cols, rows, step, k, dskew, ntx, nty, static_cast<TileBlurWork*>(work), budget);
}

// ------------------------------------------------------------------------------------------------
// K15 Gather (StitchTool.cpp:52-96).  Out-of-range probes (latent OOB reads) are defined as "no match".
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const uchar4* __restrict__ L, const uchar4* __restrict__ R, const uchar4* __restrict__ merged,
                                                const uint8_t* __restrict__ map, int cols, int rows, uchar4* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= cols) return;
  auto M = [&](int yy, int xx) -> int {
    if (xx < 0 || xx >= cols || yy < 0 || yy >= rows) return -1;
    const int v = map[size_t(yy) * cols + xx] + (merged[size_t(yy) * cols + xx].w > 0 ? 75 : 0);
    return v > 255 ? 255 : v;
  };
  const size_t i = size_t(y) * cols + x;
  const int m = M(y, x);
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (m == 100) o = L[i];
  else if (m == 50) o = R[i];
  else if (m == 225 || m == 125 || m == 175) o = merged[i];
  else if (m == 150) {
    for (int k = 1; k < 100; k++) {
      const int n0 = M(y, x + k), n1 = M(y, x - k), n2 = M(y + k, x), n3 = M(y - k, x), n4 = M(y - k, x - k), n5 = M(y - k, x + k),
                n6 = M(y + k, x - k), n7 = M(y + k, x + k);
      const bool any100 = n0 == 100 || n1 == 100 || n2 == 100 || n3 == 100 || n4 == 100 || n5 == 100 || n6 == 100 || n7 == 100;
      const bool any50 = n0 == 50 || n1 == 50 || n2 == 50 || n3 == 50 || n4 == 50 || n5 == 50 || n6 == 50 || n7 == 50;
      if (any100) { o = L[i]; break; }
      else if (any50) { o = R[i]; break; }
      else o = make_uchar4(0, 0, 0, 255);
    }
  }
  out[i] = o;
}
void launch_gather(hipStream_t st, const uint8_t* L, const uint8_t* R, const uint8_t* merged, const uint8_t* map, int cols, int rows, uint8_t* out) {
  dim3 grid((cols + 255) / 256, rows);
  hipLaunchKernelGGL(k_gather, grid, dim3(256), 0, st, reinterpret_cast<const uchar4*>(L), reinterpret_cast<const uchar4*>(R),
                     reinterpret_cast<const uchar4*>(merged), map, cols, rows, reinterpret_cast<uchar4*>(out));
}

}  // namespace pf
