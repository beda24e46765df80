// =============================================================================================
// oracle/pixflow_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the reference's asymmetric bidirectional optical-flow blending path:
//   /root/reference/CPU/PixFlow.hpp (all), CPU/OpticalFlow.cpp:9-145, CPU/StitchTool.cpp (all),
//   CPU/util.hpp:78-81,93-101.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/link/call this
// file.  The product (panorama-opticalflow_amd/) never does.
//
// PARITY UNPINNED: the reference has no tests / golden vectors (Test_data is absent), and all its
// pixel arithmetic lives in OpenCV 3.2 which is neither vendored nor installed here, so the
// reference cannot be compiled in this image.  The OpenCV primitives below are restated from the
// published OpenCV-3.2 algorithms (imgproc/src/{imgwarp,smooth,filter,deriv,color}.cpp, scalar
// non-IPP paths); each one says which.  They are cross-checked against independent
// implementations (scipy.ndimage, torch.nn.functional.interpolate) in tests/test_oracle_primitives.py.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared  (see oracle/Makefile).
//   -ffp-contract=off: the reference's build line (README.md:53, plain g++/nvcc host, baseline
//   x86-64) has no FMA, and the sweeps make strict '<' decisions on nearly-equal floats.
//
// SENSITIVITY VARIANTS (tests/golden/oracle_sensitivity.py only; never defined by oracle/Makefile, the default build has none of
// them): -DORC_VAR_GAUSS_SYMM (k > 5 Gaussian row pass as centre + symmetric pairs, the order a vectorised OpenCV row filter may use),
// -DORC_VAR_BOX_FLOAT (box-filter sliding sums in float instead of double), -DORC_VAR_LIBM_ULP (exp / tanhf results moved one ulp up),
// and the compiler flags -mfma -ffp-contract=fast.  They measure how far a real OpenCV build COULD sit from this restatement.
// =============================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

// ------------------------------------------------------------------------------------------
// tiny image container (row-major, interleaved channels)
// ------------------------------------------------------------------------------------------
template <typename T>
struct Img {
  int w = 0, h = 0, c = 1;
  std::vector<T> d;
  Img() {}
  Img(int w_, int h_, int c_ = 1) : w(w_), h(h_), c(c_), d(size_t(w_) * h_ * c_) {}
  T& at(int y, int x, int k = 0) { return d[(size_t(y) * w + x) * c + k]; }
  const T& at(int y, int x, int k = 0) const { return d[(size_t(y) * w + x) * c + k]; }
  bool empty() const { return d.empty(); }
};
using ImgF = Img<float>;
using ImgU8 = Img<uint8_t>;

// CPU/util.hpp:78-81
template <typename T>
static inline T clampT(const T& x, const T& a, const T& b) { return x < a ? a : x > b ? b : x; }
// CPU/util.hpp:93-101
static inline float lerpf(float x0, float x1, float alpha) { return x0 * (1.0f - alpha) + x1 * alpha; }

// [OpenCV] borderInterpolate
static inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
  return p;
}
static inline int replicate(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); }

// [OpenCV] cvRound == lrint under default rounding mode (round-half-even)
static inline int cvRoundf(float v) { return (int)std::lrintf(v); }
static inline int cvFloorf(float v) { int i = (int)v; return i - (v < (float)i); }

// ------------------------------------------------------------------------------------------
// [OpenCV imgwarp.cpp] resize.  src coord: fx = (float)((dx+0.5)*scale - 0.5), scale = 1/(dst/src) in double
// ------------------------------------------------------------------------------------------
static inline void interpolateCubic(float x, float* coeffs) {
  const float A = -0.75f;
  coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

struct ResizeTab {
  std::vector<int> ofs;        // source index of tap 1 (cubic) / tap 0 (linear)
  std::vector<float> coef;     // ksize per dst
  int xmax;                    // linear: first dst with ofs+1 >= n  (horizontal only)
};

// horizontal==true applies OpenCV's "fx=0" fix-ups for INTER_LINEAR at the borders; the vertical
// table has none (rows are clipped per tap instead).
static ResizeTab make_tab(int ssize, int dsize, bool cubic, bool horizontal) {
  ResizeTab t;
  const int ksize = cubic ? 4 : 2;
  t.ofs.resize(dsize);
  t.coef.resize(size_t(dsize) * ksize);
  t.xmax = dsize;
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1. / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cvFloorf(f);
    f -= s;
    if (horizontal) {
      if (s < 0 && !cubic) { f = 0; s = 0; }
      if (s + ksize / 2 >= ssize) {
        t.xmax = std::min(t.xmax, d);
        if (s >= ssize - 1 && !cubic) { f = 0; s = ssize - 1; }
      }
    }
    t.ofs[d] = s;
    float* cb = &t.coef[size_t(d) * ksize];
    if (cubic) interpolateCubic(f, cb);
    else { cb[0] = 1.f - f; cb[1] = f; }
  }
  return t;
}

// INTER_LINEAR, float, cn channels.  HResizeLinear then VResizeLinear.
static void resize_linear_f32(const ImgF& src, ImgF& dst, int dw, int dh) {
  const int cn = src.c;
  dst = ImgF(dw, dh, cn);
  ResizeTab tx = make_tab(src.w, dw, false, true), ty = make_tab(src.h, dh, false, false);
  ImgF tmp(dw, src.h, cn);  // all horizontally resized rows
  for (int sy = 0; sy < src.h; ++sy) {
    const float* S = &src.d[size_t(sy) * src.w * cn];
    float* D = &tmp.d[size_t(sy) * dw * cn];
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = tx.ofs[dx];
      for (int k = 0; k < cn; ++k) {
        if (dx < tx.xmax) D[dx * cn + k] = S[sx * cn + k] * tx.coef[dx * 2] + S[(sx + 1) * cn + k] * tx.coef[dx * 2 + 1];
        else D[dx * cn + k] = S[sx * cn + k] * 1.0f;
      }
    }
  }
  for (int dy = 0; dy < dh; ++dy) {
    const float* S0 = &tmp.d[size_t(replicate(ty.ofs[dy], src.h)) * dw * cn];
    const float* S1 = &tmp.d[size_t(replicate(ty.ofs[dy] + 1, src.h)) * dw * cn];
    const float b0 = ty.coef[dy * 2], b1 = ty.coef[dy * 2 + 1];
    float* D = &dst.d[size_t(dy) * dw * cn];
    for (int i = 0; i < dw * cn; ++i) D[i] = S0[i] * b0 + S1[i] * b1;
  }
}

// INTER_CUBIC, float, cn channels.  HResizeCubic (taps clamped) then VResizeCubic (rows clipped).
static void resize_cubic_f32(const ImgF& src, ImgF& dst, int dw, int dh) {
  const int cn = src.c;
  dst = ImgF(dw, dh, cn);
  ResizeTab tx = make_tab(src.w, dw, true, true), ty = make_tab(src.h, dh, true, false);
  ImgF tmp(dw, src.h, cn);  // all horizontally resized rows
  for (int sy = 0; sy < src.h; ++sy) {
    const float* S = &src.d[size_t(sy) * src.w * cn];
    float* D = &tmp.d[size_t(sy) * dw * cn];
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = tx.ofs[dx];
      const float* a = &tx.coef[size_t(dx) * 4];
      const int x0 = replicate(sx - 1, src.w), x1 = replicate(sx, src.w), x2 = replicate(sx + 1, src.w), x3 = replicate(sx + 2, src.w);
      for (int k = 0; k < cn; ++k)
        D[dx * cn + k] = S[x0 * cn + k] * a[0] + S[x1 * cn + k] * a[1] + S[x2 * cn + k] * a[2] + S[x3 * cn + k] * a[3];
    }
  }
  for (int dy = 0; dy < dh; ++dy) {
    const int sy = ty.ofs[dy];
    const float* b = &ty.coef[size_t(dy) * 4];
    const float* S0 = &tmp.d[size_t(replicate(sy - 1, src.h)) * dw * cn];
    const float* S1 = &tmp.d[size_t(replicate(sy, src.h)) * dw * cn];
    const float* S2 = &tmp.d[size_t(replicate(sy + 1, src.h)) * dw * cn];
    const float* S3 = &tmp.d[size_t(replicate(sy + 2, src.h)) * dw * cn];
    float* D = &dst.d[size_t(dy) * dw * cn];
    for (int i = 0; i < dw * cn; ++i) D[i] = S0[i] * b[0] + S1[i] * b[1] + S2[i] * b[2] + S3[i] * b[3];
  }
}

// INTER_CUBIC, 8UC4 fixed point (HResizeCubic<uchar,int,short>, VResizeCubic + FixedPtCast<int,uchar,22>)
static void resize_cubic_u8(const ImgU8& src, ImgU8& dst, int dw, int dh) {
  const int cn = src.c;
  dst = ImgU8(dw, dh, cn);
  ResizeTab tx = make_tab(src.w, dw, true, true), ty = make_tab(src.h, dh, true, false);
  std::vector<short> ia(size_t(dw) * 4), ib(size_t(dh) * 4);
  for (size_t i = 0; i < ia.size(); ++i) ia[i] = (short)clampT(cvRoundf(tx.coef[i] * 2048.f), -32768, 32767);
  for (size_t i = 0; i < ib.size(); ++i) ib[i] = (short)clampT(cvRoundf(ty.coef[i] * 2048.f), -32768, 32767);
  Img<int> tmp(dw, src.h, cn);
  for (int sy = 0; sy < src.h; ++sy) {
    const uint8_t* S = &src.d[size_t(sy) * src.w * cn];
    int* D = &tmp.d[size_t(sy) * dw * cn];
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = tx.ofs[dx];
      const short* a = &ia[size_t(dx) * 4];
      const int x0 = replicate(sx - 1, src.w), x1 = replicate(sx, src.w), x2 = replicate(sx + 1, src.w), x3 = replicate(sx + 2, src.w);
      for (int k = 0; k < cn; ++k)
        D[dx * cn + k] = S[x0 * cn + k] * a[0] + S[x1 * cn + k] * a[1] + S[x2 * cn + k] * a[2] + S[x3 * cn + k] * a[3];
    }
  }
  for (int dy = 0; dy < dh; ++dy) {
    const int sy = ty.ofs[dy];
    const short* b = &ib[size_t(dy) * 4];
    const int* S0 = &tmp.d[size_t(replicate(sy - 1, src.h)) * dw * cn];
    const int* S1 = &tmp.d[size_t(replicate(sy, src.h)) * dw * cn];
    const int* S2 = &tmp.d[size_t(replicate(sy + 1, src.h)) * dw * cn];
    const int* S3 = &tmp.d[size_t(replicate(sy + 2, src.h)) * dw * cn];
    uint8_t* D = &dst.d[size_t(dy) * dw * cn];
    for (int i = 0; i < dw * cn; ++i) {
      const int v = (S0[i] * b[0] + S1[i] * b[1] + S2[i] * b[2] + S3[i] * b[3] + (1 << 21)) >> 22;
      D[i] = (uint8_t)clampT(v, 0, 255);
    }
  }
}

// ------------------------------------------------------------------------------------------
// [OpenCV smooth.cpp] getGaussianKernel(n, sigma, CV_32F)
// ------------------------------------------------------------------------------------------
static std::vector<float> gaussian_kernel(int n, double sigma) {
  std::vector<float> cf(n);
  const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    const double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) cf[i] = (float)(cf[i] * sum);
  return cf;
}

// [OpenCV filter.cpp] separable symmetric filter, float src/buf/dst, BORDER_REFLECT_101.
// Row pass: ksize<=5 -> SymmRowSmallFilter (centre + symmetric pairs); otherwise RowFilter (plain
// left-to-right accumulation).  Column pass: SymmColumnFilter (centre, then pairs outward).
static void gaussian_blur_f32(const ImgF& src, ImgF& dst, int ksize, double sigma) {
  const int w = src.w, h = src.h, cn = src.c, r = ksize / 2;
  const std::vector<float> k = gaussian_kernel(ksize, sigma);
  const float* kc = &k[r];
  ImgF tmp(w, h, cn);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < cn; ++c) {
        float s;
        if (ksize <= 5) {
          s = src.at(y, x, c) * kc[0];
          for (int j = 1; j <= r; ++j) s = s + (src.at(y, reflect101(x - j, w), c) + src.at(y, reflect101(x + j, w), c)) * kc[j];
        } else {
#ifdef ORC_VAR_GAUSS_SYMM
          s = src.at(y, x, c) * kc[0];
          for (int j = 1; j <= r; ++j) s = s + (src.at(y, reflect101(x - j, w), c) + src.at(y, reflect101(x + j, w), c)) * kc[j];
#else
          s = k[0] * src.at(y, reflect101(x - r, w), c);
          for (int j = 1; j < ksize; ++j) s += k[j] * src.at(y, reflect101(x - r + j, w), c);
#endif
        }
        tmp.at(y, x, c) = s;
      }
  ImgF out(w, h, cn);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < cn; ++c) {
        float s = kc[0] * tmp.at(y, x, c) + 0.0f;
        for (int j = 1; j <= r; ++j) s += kc[j] * (tmp.at(reflect101(y + j, h), x, c) + tmp.at(reflect101(y - j, h), x, c));
        out.at(y, x, c) = s;
      }
  dst = std::move(out);
}

// [OpenCV deriv.cpp] Sobel(ksize=1): [-1,0,1] along the derivative axis, [1] across; BORDER_REPLICATE
static void sobel1(const ImgF& src, ImgF& dst, int dx, int dy) {
  const int w = src.w, h = src.h;
  ImgF out(w, h, 1);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if (dx) out.at(y, x) = src.at(y, replicate(x + 1, w)) - src.at(y, replicate(x - 1, w));
      else out.at(y, x) = src.at(replicate(y + 1, h), x) - src.at(replicate(y - 1, h), x);
    }
  (void)dy;
  dst = std::move(out);
}

// [OpenCV smooth.cpp] medianBlur(ksize=5) on CV_32F: exact per-channel median, replicate border,
// source copied when in place.  (Selection => identical to any correct sorting network.)
static void median5(const ImgF& src, ImgF& dst) {
  const int w = src.w, h = src.h, cn = src.c;
  ImgF out(w, h, cn);
  float v[25];
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < cn; ++c) {
        int n = 0;
        for (int j = -2; j <= 2; ++j)
          for (int i = -2; i <= 2; ++i) v[n++] = src.at(replicate(y + j, h), replicate(x + i, w), c);
        std::nth_element(v, v + 12, v + 25);
        out.at(y, x, c) = v[12];
      }
  dst = std::move(out);
}

// [OpenCV smooth.cpp] blur()/boxFilter normalised, CV_32F: RowSum<float,double> + ColumnSum<double,float>
// (sliding sums in double), anchor k/2, BORDER_REFLECT_101 relative to the WHOLE image (ROI not
// isolated).  Blurs the roi [x0,x0+rw) x [y0,y0+rh) of img "in place" reading a snapshot of img.
#ifdef ORC_VAR_BOX_FLOAT
typedef float box_acc_t;
#else
typedef double box_acc_t;
#endif
static void box_blur_roi(ImgF& img, int x0, int y0, int rw, int rh, int k) {
  const int W = img.w, H = img.h, a = k / 2;
  std::vector<box_acc_t> rows(size_t(rh + k - 1) * rw);  // row sums for source rows y0-a .. y0-a+rh+k-2
  for (int j = 0; j < rh + k - 1; ++j) {
    const int sy = reflect101(y0 - a + j, H);
    box_acc_t s = 0;
    for (int i = 0; i < k; ++i) s += (box_acc_t)img.at(sy, reflect101(x0 - a + i, W));
    rows[size_t(j) * rw] = s;
    for (int x = 1; x < rw; ++x) {
      s += (box_acc_t)img.at(sy, reflect101(x0 - a + x - 1 + k, W)) - (box_acc_t)img.at(sy, reflect101(x0 - a + x - 1, W));
      rows[size_t(j) * rw + x] = s;
    }
  }
  const box_acc_t scale = box_acc_t(1. / ((double)k * k));
  std::vector<box_acc_t> sum(rw, 0.0);
  for (int j = 0; j < k - 1; ++j)
    for (int x = 0; x < rw; ++x) sum[x] += rows[size_t(j) * rw + x];
  for (int y = 0; y < rh; ++y)
    for (int x = 0; x < rw; ++x) {
      const box_acc_t s0 = sum[x] + rows[size_t(y + k - 1) * rw + x];
      img.at(y0 + y, x0 + x) = (float)(s0 * scale);
      sum[x] = s0 - rows[size_t(y) * rw + x];
    }
}

// ------------------------------------------------------------------------------------------
// PixFlow  (CPU/PixFlow.hpp)
// ------------------------------------------------------------------------------------------
enum Hint { UNKNOWN = 0, RIGHT = 1, DOWN = 2, LEFT = 3, UP = 4 };  // PixFlow.hpp:19

struct Params {  // PixFlow.hpp:32-44 and factory :459-497 (both presets share these)
  static constexpr int kPyrMinImageSize = 24;
  static constexpr int kPyrMaxLevels = 1000;
  static constexpr float kGradEpsilon = 0.001f;
  static constexpr float kUpdateAlphaThreshold = 0.9f;
  static constexpr int kPreBlurKernelWidth = 5;
  static constexpr float kPreBlurSigma = 0.25f;
  static constexpr int kFinalFlowBlurKernelWidth = 3;
  static constexpr float kFinalFlowBlurSigma = 1.0f;
  static constexpr int kGradientBlurKernelWidth = 3;
  static constexpr float kGradientBlurSigma = 0.5f;
  static constexpr int kBlurredFlowKernelWidth = 15;
  static constexpr float kBlurredFlowSigma = 8.0f;
  float pyrScaleFactor = 0.9f, smoothnessCoef = 0.001f, verticalRegularizationCoef = 0.01f,
        horizontalRegularizationCoef = 0.01f, gradientStepSize = 0.5f, downscaleFactor = 0.5f;
  int maxPercentage = 0;
};

// PixFlow.hpp:407-425
static inline float getPixBilinear32FExtend(const ImgF& img, float x, float y) {
  x = std::min(img.w - 2.0f, std::max(0.0f, x));
  y = std::min(img.h - 2.0f, std::max(0.0f, y));
  const int x0 = int(x), y0 = int(y);
  const float xR = x - float(x0), yR = y - float(y0);
  const float* p = &img.d[size_t(y0) * img.w];
  const float f00 = p[x0], f01 = p[x0 + img.w], f10 = p[x0 + 1], f11 = p[x0 + img.w + 1];
  const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
  return a1 + a2 * xR + a3 * yR + a4 * xR * yR;
}

struct LevelCtx {
  const ImgF *I0x, *I0y, *I1x, *I1y, *blurred;
  const Params* p;
  int cols;
};

// PixFlow.hpp:427-456
static inline float errorFunction(const LevelCtx& L, int x, int y, float fdx, float fdy) {
  const float matchX = x + fdx, matchY = y + fdy;
  const float i0x = L.I0x->at(y, x), i0y = L.I0y->at(y, x);
  const float i1x = getPixBilinear32FExtend(*L.I1x, matchX, matchY);
  const float i1y = getPixBilinear32FExtend(*L.I1y, matchX, matchY);
  const float dfx = L.blurred->at(y, x, 0) - fdx, dfy = L.blurred->at(y, x, 1) - fdy;
  const float smoothness = sqrtf(dfx * dfx + dfy * dfy);
  float err = sqrtf((i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y)) + smoothness * L.p->smoothnessCoef +
              L.p->verticalRegularizationCoef * fabsf(fdy) / float(L.cols) +
              L.p->horizontalRegularizationCoef * fabsf(fdx) / float(L.cols);
  return err;
}

// PixFlow.hpp:137-151
static void pyramid_sizes(int w0, int h0, float scale, std::vector<int>& ws, std::vector<int>& hs) {
  ws = {w0}; hs = {h0};
  while ((int)ws.size() < Params::kPyrMaxLevels) {
    const int nw = int(ws.back() * scale + 0.5f), nh = int(hs.back() * scale + 0.5f);
    if (nh <= Params::kPyrMinImageSize || nw <= Params::kPyrMinImageSize) break;
    ws.push_back(nw); hs.push_back(nh);
  }
}
static std::vector<ImgF> buildPyramid(const ImgF& src, float scale) {
  std::vector<int> ws, hs;
  pyramid_sizes(src.w, src.h, scale, ws, hs);
  std::vector<ImgF> pyr = {src};
  for (size_t l = 1; l < ws.size(); ++l) { ImgF n; resize_linear_f32(pyr.back(), n, ws[l], hs[l]); pyr.push_back(std::move(n)); }
  return pyr;
}

// PixFlow.hpp:281-294
static void gradients(const ImgF& I, ImgF& Ix, ImgF& Iy) {
  sobel1(I, Ix, 1, 0); sobel1(I, Iy, 0, 1);
  gaussian_blur_f32(Ix, Ix, Params::kGradientBlurKernelWidth, Params::kGradientBlurSigma);
  gaussian_blur_f32(Iy, Iy, Params::kGradientBlurKernelWidth, Params::kGradientBlurSigma);
}

// PixFlow.hpp:153-155
static inline int computeSearchDistance(int maxPct) { return (Params::kPyrMinImageSize * maxPct + 50) / 100; }

// PixFlow.hpp:157-188
static float computePatchError(const ImgF& i0, const ImgF& alpha0, int i0x, int i0y, const ImgF& i1, const ImgF& alpha1,
                               int i1x, int i1y, int maxPct) {
  const int kPatchRadius = 2;
  float sad = 0, alpha = 0;
  for (int dy = -kPatchRadius; dy <= kPatchRadius; ++dy) {
    const int d0y = i0y + dy;
    if (0 <= d0y && d0y < i0.h) {
      const int d1y = clampT(i1y + dy, 0, i1.h - 1);
      for (int dx = -kPatchRadius; dx <= kPatchRadius; ++dx) {
        const int d0x = i0x + dx;
        if (0 <= d0x && d0x < i0.w) {
          const int d1x = clampT(i1x + dx, 0, i1.w - 1);
          const float difference = i0.at(d0y, d0x) - i1.at(d1y, d1x);
          sad += std::abs(difference);
          alpha += alpha0.at(d0y, d0x) * alpha1.at(d1y, d1x);
        }
      }
    }
  }
  sad /= alpha;
  const float fx = float(i1x - i0x), fy = float(i1y - i0y);
  const float length = (float)std::sqrt((double)fx * fx + (double)fy * fy);  // [OpenCV] cv::norm(Point2f) in double
  sad *= 1 + length / computeSearchDistance(maxPct);
  return sad;
}

// PixFlow.hpp:190-205
static float computeIntensityRatio(const ImgF& lhs, const ImgF& lhsAlpha, const ImgF& rhs, const ImgF& rhsAlpha) {
  float sumLhs = 0, sumRhs = 0;
  for (int y = 0; y < lhs.h; ++y)
    for (int x = 0; x < lhs.w; ++x) {
      const float alpha = lhsAlpha.at(y, x) * rhsAlpha.at(y, x);
      sumLhs += alpha * lhs.at(y, x);
      sumRhs += alpha * rhs.at(y, x);
    }
  return sumLhs / sumRhs;
}

// PixFlow.hpp:207-224 ; returns false for UNKNOWN (reference: LOG(FATAL), unreachable from :300)
static bool computeSearchBox(int hint, int maxPct, int& bx, int& by, int& bw, int& bh) {
  const int dist = computeSearchDistance(maxPct), kRatio = 8;
  const int ortho = (dist + kRatio / 2) / kRatio, thickness = 2 * ortho + 1;
  switch (hint) {
    case RIGHT: bx = 0; by = -ortho; bw = dist + 1; bh = thickness; return true;
    case DOWN: bx = -ortho; by = 0; bw = thickness; bh = dist + 1; return true;
    case LEFT: bx = -dist; by = -ortho; bw = dist + 1; bh = thickness; return true;
    case UP: bx = -ortho; by = -dist; bw = thickness; bh = dist + 1; return true;
    default: return false;
  }
}

// PixFlow.hpp:226-270
static void adjustInitialFlow(const ImgF& I0, const ImgF& I1, const ImgF& alpha0, const ImgF& alpha1, ImgF& flow, int hint,
                              int maxPct) {
  const float ratio = computeIntensityRatio(I0, alpha0, I1, alpha1);
  ImgF I1eq(I1.w, I1.h, 1);
  for (size_t i = 0; i < I1.d.size(); ++i) I1eq.d[i] = I1.d[i] * ratio + 0.0f;  // [OpenCV] Mat*scalar = convertTo(alpha=ratio, beta=0)
  int bx, by, bw, bh;
  if (!computeSearchBox(hint, maxPct, bx, by, bw, bh)) return;
  for (int i0y = 0; i0y < I0.h; ++i0y)
    for (int i0x = 0; i0x < I0.w; ++i0x)
      if (alpha0.at(i0y, i0x) > Params::kUpdateAlphaThreshold) {
        const float kFraction = 0.8f;
        float errorBest = kFraction * computePatchError(I0, alpha0, i0x, i0y, I1eq, alpha1, i0x, i0y, maxPct);
        int i1xBest = i0x, i1yBest = i0y;
        for (int dy = by; dy < by + bh; ++dy)
          for (int dx = bx; dx < bx + bw; ++dx) {
            const int i1x = i0x + dx, i1y = i0y + dy;
            if (0 <= i1x && i1x < I1.w && 0 <= i1y && i1y < I1.h) {
              const float error = computePatchError(I0, alpha0, i0x, i0y, I1eq, alpha1, i1x, i1y, maxPct);
              if (errorBest > error) { errorBest = error; i1xBest = i1x; i1yBest = i1y; }
            }
          }
        flow.at(i0y, i0x, 0) = float(i1xBest - i0x);
        flow.at(i0y, i0x, 1) = float(i1yBest - i0y);
      }
}

// One raster sweep (PixFlow.hpp:315-324 forward, :328-337 backward), incl. proposeFlowUpdate
// (:342-362) and errorGradient (:364-386).
static void sweep(const LevelCtx& L, const ImgF& alpha0, const ImgF& alpha1, ImgF& flow, bool forward) {
  const int W = flow.w, H = flow.h;
  const float eps = Params::kGradEpsilon, thr = Params::kUpdateAlphaThreshold;
  auto body = [&](int x, int y, bool hasA, int ax, int ay, bool hasB, int bx, int by) {
    if (!(alpha0.at(y, x) > thr && alpha1.at(y, x) > thr)) return;
    float fx = flow.at(y, x, 0), fy = flow.at(y, x, 1);
    float currErr = errorFunction(L, x, y, fx, fy);
    if (hasA) {
      const float px = flow.at(ay, ax, 0), py = flow.at(ay, ax, 1);
      const float e = errorFunction(L, x, y, px, py);
      if (e < currErr) { fx = px; fy = py; currErr = e; }
    }
    if (hasB) {
      const float px = flow.at(by, bx, 0), py = flow.at(by, bx, 1);
      const float e = errorFunction(L, x, y, px, py);
      if (e < currErr) { fx = px; fy = py; currErr = e; }
    }
    const float ex = errorFunction(L, x, y, fx + eps, fy + 0.0f);
    const float ey = errorFunction(L, x, y, fx + 0.0f, fy + eps);
    const float gx = (ex - currErr) / eps, gy = (ey - currErr) / eps;
    flow.at(y, x, 0) = fx - L.p->gradientStepSize * gx;
    flow.at(y, x, 1) = fy - L.p->gradientStepSize * gy;
  };
  if (forward) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) body(x, y, x > 0, x - 1, y, y > 0, x, y - 1);
  } else {
    for (int y = H - 1; y >= 0; --y)
      for (int x = W - 1; x >= 0; --x) body(x, y, x < W - 1, x + 1, y, y < H - 1, x, y + 1);
  }
}

// PixFlow.hpp:388-405
static void lowAlphaFlowDiffusion(const ImgF& alpha0, const ImgF& alpha1, ImgF& flow) {
  ImgF blurred;
  gaussian_blur_f32(flow, blurred, Params::kBlurredFlowKernelWidth, Params::kBlurredFlowSigma);
  for (int y = 0; y < flow.h; ++y)
    for (int x = 0; x < flow.w; ++x) {
      const float diffusionCoef = 1.0f - alpha0.at(y, x) * alpha1.at(y, x);
      for (int c = 0; c < 2; ++c) flow.at(y, x, c) = diffusionCoef * blurred.at(y, x, c) + (1.0f - diffusionCoef) * flow.at(y, x, c);
    }
}

// PixFlow.hpp:272-340.  stage_mask lets tests stop after a stage (bit0 fwd sweep, bit1 median,
// bit2 bwd sweep, bit3 median, bit4 diffusion); 0x1f = everything.
static void patchMatchPropagationAndSearch(const ImgF& I0, const ImgF& I1, const ImgF& alpha0, const ImgF& alpha1, ImgF& flow,
                                           int hint, const Params& p, int stage_mask = 0x1f) {
  ImgF I0x, I0y, I1x, I1y;
  gradients(I0, I0x, I0y);
  gradients(I1, I1x, I1y);
  if (flow.empty()) {
    flow = ImgF(I0.w, I0.h, 2);
    if (p.maxPercentage > 0 && hint != UNKNOWN) adjustInitialFlow(I0, I1, alpha0, alpha1, flow, hint, p.maxPercentage);
  }
  ImgF blurredFlow;
  gaussian_blur_f32(flow, blurredFlow, Params::kBlurredFlowKernelWidth, Params::kBlurredFlowSigma);
  LevelCtx L{&I0x, &I0y, &I1x, &I1y, &blurredFlow, &p, I0.w};
  if (stage_mask & 1) sweep(L, alpha0, alpha1, flow, true);
  if (stage_mask & 2) median5(flow, flow);
  if (stage_mask & 4) sweep(L, alpha0, alpha1, flow, false);
  if (stage_mask & 8) median5(flow, flow);
  if (stage_mask & 16) lowAlphaFlowDiffusion(alpha0, alpha1, flow);
}

// PixFlow.hpp:78-103 : downscale, gray, alpha, pre-blur
static void preprocess(const ImgU8& bgra, const Params& p, ImgF& I, ImgF& alpha) {
  const int dw = int(bgra.w * p.downscaleFactor), dh = int(bgra.h * p.downscaleFactor);
  ImgU8 small;
  resize_cubic_u8(bgra, small, dw, dh);
  I = ImgF(dw, dh, 1); alpha = ImgF(dw, dh, 1);
  const float inv255 = (float)(1.0 / 255.0f);  // [OpenCV] Mat /= s  ==  convertTo(alpha = 1./s) with float scale
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const uint8_t* px = &small.d[(size_t(y) * dw + x) * 4];
      const int g = (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14;  // [OpenCV color.cpp] BGRA2GRAY 8u
      I.at(y, x) = float(g) * inv255 + 0.0f;
      alpha.at(y, x) = float(px[3]) * inv255 + 0.0f;
    }
  gaussian_blur_f32(I, I, Params::kPreBlurKernelWidth, Params::kPreBlurSigma);
}

// PixFlow.hpp:72-135
static void computeOpticalFlow(const ImgU8& rgba0, const ImgU8& rgba1, ImgF& flow, int hint, const Params& p) {
  ImgF I0, I1, alpha0, alpha1;
  preprocess(rgba0, p, I0, alpha0);
  preprocess(rgba1, p, I1, alpha1);
  std::vector<ImgF> pI0 = buildPyramid(I0, p.pyrScaleFactor), pI1 = buildPyramid(I1, p.pyrScaleFactor),
                    pA0 = buildPyramid(alpha0, p.pyrScaleFactor), pA1 = buildPyramid(alpha1, p.pyrScaleFactor);
  flow = ImgF();
  for (int level = int(pI0.size()) - 1; level >= 0; --level) {
    patchMatchPropagationAndSearch(pI0[level], pI1[level], pA0[level], pA1[level], flow, hint, p);
    if (level > 0) {
      ImgF up;
      resize_cubic_f32(flow, up, pI0[level - 1].w, pI0[level - 1].h);
      const float s = 1.0f / p.pyrScaleFactor;
      for (auto& v : up.d) v = v * s + 0.0f;  // [OpenCV] Mat *= s == convertTo(alpha=s, beta=0)
      flow = std::move(up);
    }
  }
  ImgF full;
  resize_linear_f32(flow, full, rgba0.w, rgba0.h);
  const float s = 1.0f / p.downscaleFactor;
  for (auto& v : full.d) v = v * s + 0.0f;
  gaussian_blur_f32(full, full, Params::kFinalFlowBlurKernelWidth, Params::kFinalFlowBlurSigma);
  flow = std::move(full);
}

// ------------------------------------------------------------------------------------------
// OpticalFlow.cpp
// ------------------------------------------------------------------------------------------
// OpticalFlow.cpp:113-126 : [last `len` cols | image | first `len` cols]
static ImgU8 wrap_pad(const ImgU8& im, int len) {
  ImgU8 out(im.w + 2 * len, im.h, im.c);
  for (int y = 0; y < im.h; ++y)
    for (int x = 0; x < out.w; ++x) {
      int sx = x - len;
      if (sx < 0) sx += im.w; else if (sx >= im.w) sx -= im.w;
      for (int c = 0; c < im.c; ++c) out.at(y, x, c) = im.at(y, sx, c);
    }
  return out;
}

// OpticalFlow.cpp:102-145
static void flow_bidir(const ImgU8& L, const ImgU8& R, const Params& p, ImgF& flowLtoR, ImgF& flowRtoL) {
  const int length = L.w / 20;
  ImgU8 nL = wrap_pad(L, length), nR = wrap_pad(R, length);
  ImgF fLR, fRL;
  computeOpticalFlow(nL, nR, fLR, LEFT, p);
  computeOpticalFlow(nR, nL, fRL, RIGHT, p);
  flowLtoR = ImgF(L.w, L.h, 2); flowRtoL = ImgF(L.w, L.h, 2);
  for (int y = 0; y < L.h; ++y)
    for (int x = 0; x < L.w; ++x)
      for (int c = 0; c < 2; ++c) { flowLtoR.at(y, x, c) = fLR.at(y, x + length, c); flowRtoL.at(y, x, c) = fRL.at(y, x + length, c); }
}

// OpticalFlow.cpp:9-28.  Latent hazard (single wrap => OOB when |flow*t| > cols) is defined here
// as a true modulo.
static inline const uint8_t* novelViewPoint(const ImgU8& src, const ImgF& flow, double t, int x, int y) {
  const float fx = flow.at(y, x, 0), fy = flow.at(y, x, 1);
  int srcx = int(x + fx * t);
  if (srcx > src.w - 1) srcx = srcx - src.w;
  if (srcx < 0) srcx = srcx + src.w;
  srcx %= src.w; if (srcx < 0) srcx += src.w;
  int srcy = int(y + fy * t);
  if (srcy > src.h - 1) srcy = src.h - 1;
  if (srcy < 0) srcy = 0;
  return &src.d[(size_t(srcy) * src.w + srcx) * 4];
}

// OpticalFlow.cpp:30-92
static void combineNovelViews(const ImgU8& imageL, const ImgU8& imageR, const ImgF& flowLtoR, const ImgF& flowRtoL, const ImgF& blend,
                              ImgU8& out) {
  out = ImgU8(imageL.w, imageL.h, 4);
  for (int y = 0; y < imageL.h; ++y)
    for (int x = 0; x < imageL.w; ++x) {
      const float blendR = blend.at(y, x), blendL = 1 - blendR;
      const uint8_t* colorL = novelViewPoint(imageL, flowRtoL, blendR, x, y);
      const uint8_t* colorR = novelViewPoint(imageR, flowLtoR, blendL, x, y);
      uint8_t* o = &out.d[(size_t(y) * out.w + x) * 4];
      if (colorL[3] == 0 || colorR[3] == 0) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
      const float fLRx = flowLtoR.at(y, x, 0), fLRy = flowLtoR.at(y, x, 1), fRLx = flowRtoL.at(y, x, 0), fRLy = flowRtoL.at(y, x, 1);
      const float kColorDiffCoef = 10.0f, kSoftmaxSharpness = 10.0f, kFlowMagCoef = 100.0f;
      const float flowMagLR = sqrtf(fLRx * fLRx + fLRy * fLRy) / float(imageL.w);
      const float flowMagRL = sqrtf(fRLx * fRLx + fRLy * fRLy) / float(imageL.w);
      const float colorDiff =
          (std::abs(colorL[0] - colorR[0]) + std::abs(colorL[1] - colorR[1]) + std::abs(colorL[2] - colorR[2])) / 255.0f;
#ifdef ORC_VAR_LIBM_ULP
      auto tanhf = [](float v) { return std::nextafterf(::tanhf(v), 2.0f); };
      auto exp = [](double v) { return std::nextafter(::exp(v), 1e300); };
#endif
      const float deghostCoef = tanhf(colorDiff * kColorDiffCoef);
      const float alphaL = colorL[3] / 255.0f, alphaR = colorR[3] / 255.0f;
      const double expL = exp(kSoftmaxSharpness * blendL * alphaL * (1.0 + kFlowMagCoef * flowMagRL));
      const double expR = exp(kSoftmaxSharpness * blendR * alphaR * (1.0 + kFlowMagCoef * flowMagLR));
      const double sumExp = expL + expR + 0.00001;
      const float softmaxL = float(expL / sumExp), softmaxR = float(expR / sumExp);
      const float wL = lerpf(blendL, softmaxL, deghostCoef), wR = lerpf(blendR, softmaxR, deghostCoef);
      for (int c = 0; c < 3; ++c) o[c] = (uint8_t)(int)(float(colorL[c]) * wL + float(colorR[c]) * wR);  // float->uchar truncation
      o[3] = 255;
    }
}

// ------------------------------------------------------------------------------------------
// StitchTool.cpp
// ------------------------------------------------------------------------------------------
struct Stitch {
  ImgU8 ImageL, ImageR, OverlappedL, OverlappedR, Mergedmiddle, Map, FinalResult;
  ImgF Blend, MergedDis;
  ImgU8 MapExt;  // the wrap-extended map countblend() searches (StitchTool.cpp:102-111)

  // StitchTool.cpp:38-50
  void MatchImages() {
    Map = ImgU8(ImageL.w, ImageL.h, 1);
    for (int y = 0; y < ImageL.h; ++y)
      for (int x = 0; x < ImageL.w; ++x)
        Map.at(y, x) = (uint8_t)((ImageL.at(y, x, 3) > 0 ? 100 : 0) + (ImageR.at(y, x, 3) > 0 ? 50 : 0));
  }

  // StitchTool.cpp:148-191 (x is in extended-map coordinates)
  float countblend(int x, int y) {
    int step = ImageL.w <= ImageL.h ? ImageL.w / 200 : ImageL.h / 200;
    if (step < 1) step = 1;  // reference: step==0 never terminates; defined as 1
    float minLdis = float(10 * ImageL.w), minRdis = float(10 * ImageL.w);
    const int MW = MapExt.w, MH = MapExt.h;
    const double sqrt2 = std::sqrt(2.0);
    for (int i = 0; i < ImageL.w / 2; i = i + step) {
      if (x + i < MW && MapExt.at(y, x + i) == 100 && i < minLdis) minLdis = float(i);
      if (x + i < MW && MapExt.at(y, x + i) == 50 && i < minRdis) minRdis = float(i);
      if (x - i > 0 && MapExt.at(y, x - i) == 100 && i < minLdis) minLdis = float(i);
      if (x - i > 0 && MapExt.at(y, x - i) == 50 && i < minRdis) minRdis = float(i);
      if (y + i < MH && MapExt.at(y + i, x) == 100 && i < minLdis) minLdis = float(i);
      if (y + i < MH && MapExt.at(y + i, x) == 50 && i < minRdis) minRdis = float(i);
      if (y - i > 0 && MapExt.at(y - i, x) == 100 && i < minLdis) minLdis = float(i);
      if (y - i > 0 && MapExt.at(y - i, x) == 50 && i < minRdis) minRdis = float(i);
      if ((x + i < MW && y + i < MH) && MapExt.at(y + i, x + i) == 100 && i * sqrt2 < minLdis) minLdis = float(i * sqrt2);
      if ((x + i < MW && y + i < MH) && MapExt.at(y + i, x + i) == 50 && i * sqrt2 < minRdis) minRdis = float(i * sqrt2);
      if ((x - i > 0 && y - i > 0) && MapExt.at(y - i, x - i) == 100 && i * sqrt2 < minLdis) minLdis = float(i * sqrt2);
      if ((x - i > 0 && y - i > 0) && MapExt.at(y - i, x - i) == 50 && i * sqrt2 < minRdis) minRdis = float(i * sqrt2);
      if ((x + i < MW && y - i > 0) && MapExt.at(y - i, x + i) == 100 && i * sqrt2 < minLdis) minLdis = float(i * sqrt2);
      if ((x + i < MW && y - i > 0) && MapExt.at(y - i, x + i) == 50 && i * sqrt2 < minRdis) minRdis = float(i * sqrt2);
      if ((x - i > 0 && y + i < MH) && MapExt.at(y + i, x - i) == 100 && i * sqrt2 < minLdis) minLdis = float(i * sqrt2);
      if ((x - i > 0 && y + i < MH) && MapExt.at(y + i, x - i) == 50 && i * sqrt2 < minRdis) minRdis = float(i * sqrt2);
    }
    const float blend = minLdis / (minRdis + minLdis);
    MergedDis.at(y, x) = minLdis < minRdis ? minLdis : minRdis;
    return blend;
  }

  // StitchTool.cpp:98-146.  smooth=false stops before the tile/global box blurs (:130-143).
  void GenerateBlend(bool smooth) {
    const int C = ImageL.w, Rr = ImageL.h;
    ImgF blend(C, Rr, 1);
    const int length = C / 5;
    MapExt = ImgU8(C + 2 * length, Rr, 1);
    for (int y = 0; y < Rr; ++y)
      for (int x = 0; x < MapExt.w; ++x) {
        int sx = x - length;
        if (sx < 0) sx += C; else if (sx >= C) sx -= C;
        MapExt.at(y, x) = Map.at(y, sx);
      }
    ImgF MergedExt(MapExt.w, Rr, 1);
    MergedDis = std::move(MergedExt);
    for (int y = 0; y < Rr; ++y)
      for (int x = 0; x < C; ++x) {
        const uint8_t m = MapExt.at(y, x + length);
        if (m == 100) blend.at(y, x) = 0;
        else if (m == 50) blend.at(y, x) = 1;
        else if (m == 150) blend.at(y, x) = countblend(x + length, y);
        else blend.at(y, x) = 0.5f;
      }
    ImgF md(C, Rr, 1);
    for (int y = 0; y < Rr; ++y)
      for (int x = 0; x < C; ++x) md.at(y, x) = MergedDis.at(y, x + length);
    MergedDis = std::move(md);
    if (smooth) {
      const int step = C <= Rr ? C / 200 : Rr / 200;
      const int k1 = Rr / 130, k2 = Rr / 400;
      // step==0 would loop forever and k==0 asserts in OpenCV: defined here as "skip".
      if (step > 0 && k1 > 0)
        for (int y = 0; y + step < Rr; y = y + step)
          for (int x = 0; x + step < C; x = x + step)
            if (MergedDis.at(y, x) > step) box_blur_roi(blend, x, y, step, step, k1);
      if (k2 > 0) box_blur_roi(blend, 0, 0, C, Rr, k2);
    }
    Blend = std::move(blend);
  }

  // StitchTool.cpp:7-36
  void prepare(const ImgU8& L, const ImgU8& R, bool smooth) {
    ImageL = L; ImageR = R;
    MatchImages();
    OverlappedL = ImgU8(L.w, L.h, 4); OverlappedR = ImgU8(L.w, L.h, 4);
    for (int y = 0; y < L.h; ++y)
      for (int x = 0; x < L.w; ++x) {
        const uint8_t m = Map.at(y, x) > 140 ? 1 : 0;
        for (int c = 0; c < 4; ++c) { OverlappedL.at(y, x, c) = ImageL.at(y, x, c) * m; OverlappedR.at(y, x, c) = ImageR.at(y, x, c) * m; }
      }
    GenerateBlend(smooth);
  }

  // StitchTool.cpp:52-96.  OOB probes (latent hazard) are defined as "no match".
  void Gather() {
    const int C = ImageL.w, Rr = ImageL.h;
    ImgU8 map(C, Rr, 1), result(C, Rr, 4);
    for (int y = 0; y < Rr; ++y)
      for (int x = 0; x < C; ++x) {
        const int v = Map.at(y, x) + (Mergedmiddle.at(y, x, 3) > 0 ? 75 : 0);
        map.at(y, x) = (uint8_t)std::min(v, 255);
      }
    auto M = [&](int y, int x) -> int { return (x < 0 || x >= C || y < 0 || y >= Rr) ? -1 : map.at(y, x); };
    for (int y = 0; y < Rr; ++y)
      for (int x = 0; x < C; ++x) {
        const int m = map.at(y, x);
        uint8_t* o = &result.d[(size_t(y) * C + x) * 4];
        auto cp = [&](const ImgU8& s) { for (int c = 0; c < 4; ++c) o[c] = s.at(y, x, c); };
        if (m == 100) cp(ImageL);
        else if (m == 50) cp(ImageR);
        else if (m == 225 || m == 125 || m == 175) cp(Mergedmiddle);
        else if (m == 150) {
          for (int i = 1; i < 100; i++) {
            auto any = [&](int v) {
              return M(y, x + i) == v || M(y, x - i) == v || M(y + i, x) == v || M(y - i, x) == v || M(y - i, x - i) == v ||
                     M(y - i, x + i) == v || M(y + i, x - i) == v || M(y + i, x + i) == v;
            };
            if (any(100)) { cp(ImageL); break; }
            else if (any(50)) { cp(ImageR); break; }
            else { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 255; }
          }
        } else if (m == 0) { o[0] = o[1] = o[2] = o[3] = 0; }
      }
    FinalResult = std::move(result);
  }
};

static ImgU8 wrapU8(const uint8_t* p, int w, int h, int c) { ImgU8 i(w, h, c); std::memcpy(i.d.data(), p, i.d.size()); return i; }
static ImgF wrapF(const float* p, int w, int h, int c) { ImgF i(w, h, c); std::memcpy(i.d.data(), p, i.d.size() * 4); return i; }
// PixFlow's constructor arguments (PixFlow.hpp:46-68): the factory's presets unless a test set others (orc_set_params; process-wide,
// set between calls only -- the parity tests of pf_set_solver_params are the one user)
static Params g_params;
static Params mkParams(int maxPct) { Params p = g_params; p.maxPercentage = maxPct; return p; }

}  // namespace orc

// =============================================================================================
// C entry points (ctypes in tests/, smoke(), bench cpu_baseline).  All buffers caller-owned, packed.
// =============================================================================================
using namespace orc;
extern "C" {

void orc_resize_cubic_u8(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh) {
  ImgU8 s = wrapU8(src, sw, sh, cn), d; resize_cubic_u8(s, d, dw, dh); std::memcpy(dst, d.d.data(), d.d.size());
}
void orc_resize_linear_f32(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh) {
  ImgF s = wrapF(src, sw, sh, cn), d; resize_linear_f32(s, d, dw, dh); std::memcpy(dst, d.d.data(), d.d.size() * 4);
}
void orc_resize_cubic_f32(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh) {
  ImgF s = wrapF(src, sw, sh, cn), d; resize_cubic_f32(s, d, dw, dh); std::memcpy(dst, d.d.data(), d.d.size() * 4);
}
void orc_gaussian_kernel(int n, double sigma, float* out) { auto k = gaussian_kernel(n, sigma); std::memcpy(out, k.data(), n * 4); }
void orc_gaussian_blur_f32(const float* src, int w, int h, int cn, int ksize, double sigma, float* dst) {
  ImgF s = wrapF(src, w, h, cn), d; gaussian_blur_f32(s, d, ksize, sigma); std::memcpy(dst, d.d.data(), d.d.size() * 4);
}
void orc_sobel1(const float* src, int w, int h, int dx, int dy, float* dst) {
  ImgF s = wrapF(src, w, h, 1), d; sobel1(s, d, dx, dy); std::memcpy(dst, d.d.data(), d.d.size() * 4);
}
void orc_median5(const float* src, int w, int h, int cn, float* dst) {
  ImgF s = wrapF(src, w, h, cn), d; median5(s, d); std::memcpy(dst, d.d.data(), d.d.size() * 4);
}
void orc_box_blur_roi(float* img, int w, int h, int x0, int y0, int rw, int rh, int k) {
  ImgF s = wrapF(img, w, h, 1); box_blur_roi(s, x0, y0, rw, rh, k); std::memcpy(img, s.d.data(), s.d.size() * 4);
}
void orc_gradients(const float* I, int w, int h, float* Ix, float* Iy) {
  ImgF s = wrapF(I, w, h, 1), a, b; gradients(s, a, b);
  std::memcpy(Ix, a.d.data(), a.d.size() * 4); std::memcpy(Iy, b.d.data(), b.d.size() * 4);
}
// PixFlow(pyrScaleFactor, smoothnessCoef, verticalRegularizationCoef, horizontalRegularizationCoef, gradientStepSize, ...) -- PixFlow.hpp:54-68;
// downscaleFactor stays 0.5 (the GPU path supports nothing else), directionalRegularizationCoef is read by no code of the reference
void orc_set_params(float pyrScaleFactor, float smoothnessCoef, float verticalRegularizationCoef, float horizontalRegularizationCoef, float gradientStepSize) {
  g_params.pyrScaleFactor = pyrScaleFactor; g_params.smoothnessCoef = smoothnessCoef; g_params.verticalRegularizationCoef = verticalRegularizationCoef;
  g_params.horizontalRegularizationCoef = horizontalRegularizationCoef; g_params.gradientStepSize = gradientStepSize;
}
void orc_reset_params() { g_params = Params(); }
int orc_pyramid_sizes(int w0, int h0, int* ws, int* hs, int cap) {
  std::vector<int> a, b; pyramid_sizes(w0, h0, g_params.pyrScaleFactor, a, b);
  for (int i = 0; i < (int)a.size() && i < cap; ++i) { ws[i] = a[i]; hs[i] = b[i]; }
  return (int)a.size();
}
// half-res planes of one image (PixFlow.hpp:78-103)
void orc_preprocess(const uint8_t* bgra, int cols, int rows, float* I, float* alpha) {
  ImgU8 s = wrapU8(bgra, cols, rows, 4); ImgF a, b; preprocess(s, mkParams(0), a, b);
  std::memcpy(I, a.d.data(), a.d.size() * 4); std::memcpy(alpha, b.d.data(), b.d.size() * 4);
}
// one pyramid step (PixFlow.hpp:146-148)
void orc_pyr_down(const float* src, int sw, int sh, float* dst, int dw, int dh) { orc_resize_linear_f32(src, sw, sh, 1, dst, dw, dh); }

// one raster sweep on explicit planes (for HIP sweep-kernel isolation tests)
void orc_sweep(const float* I0x, const float* I0y, const float* I1x, const float* I1y, const float* blurred, const float* a0,
               const float* a1, float* flow, int w, int h, int forward) {
  ImgF i0x = wrapF(I0x, w, h, 1), i0y = wrapF(I0y, w, h, 1), i1x = wrapF(I1x, w, h, 1), i1y = wrapF(I1y, w, h, 1),
       bl = wrapF(blurred, w, h, 2), A0 = wrapF(a0, w, h, 1), A1 = wrapF(a1, w, h, 1), f = wrapF(flow, w, h, 2);
  Params p = g_params; LevelCtx L{&i0x, &i0y, &i1x, &i1y, &bl, &p, w};
  sweep(L, A0, A1, f, forward != 0);
  std::memcpy(flow, f.d.data(), f.d.size() * 4);
}
// per-pixel error (PixFlow.hpp:427-456) at explicit flow candidates, for exactness tests of sqrt/div
void orc_error_function(const float* I0x, const float* I0y, const float* I1x, const float* I1y, const float* blurred, int w, int h,
                        const float* cand, float* err) {
  ImgF i0x = wrapF(I0x, w, h, 1), i0y = wrapF(I0y, w, h, 1), i1x = wrapF(I1x, w, h, 1), i1y = wrapF(I1y, w, h, 1), bl = wrapF(blurred, w, h, 2);
  Params p = g_params; LevelCtx L{&i0x, &i0y, &i1x, &i1y, &bl, &p, w};
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) err[size_t(y) * w + x] = errorFunction(L, x, y, cand[(size_t(y) * w + x) * 2], cand[(size_t(y) * w + x) * 2 + 1]);
}
void orc_adjust_initial_flow(const float* I0, const float* I1, const float* a0, const float* a1, int w, int h, int hint, int maxPct,
                             float* flow) {
  ImgF i0 = wrapF(I0, w, h, 1), i1 = wrapF(I1, w, h, 1), A0 = wrapF(a0, w, h, 1), A1 = wrapF(a1, w, h, 1), f(w, h, 2);
  adjustInitialFlow(i0, i1, A0, A1, f, hint, maxPct);
  std::memcpy(flow, f.d.data(), f.d.size() * 4);
}
void orc_diffusion(const float* a0, const float* a1, float* flow, int w, int h) {
  ImgF A0 = wrapF(a0, w, h, 1), A1 = wrapF(a1, w, h, 1), f = wrapF(flow, w, h, 2);
  lowAlphaFlowDiffusion(A0, A1, f); std::memcpy(flow, f.d.data(), f.d.size() * 4);
}
// one pyramid level (PixFlow.hpp:272-340); flow_in may be NULL (coarsest level)
void orc_level(const float* I0, const float* I1, const float* a0, const float* a1, int w, int h, const float* flow_in, int hint,
               int maxPct, int stage_mask, float* flow_out) {
  ImgF i0 = wrapF(I0, w, h, 1), i1 = wrapF(I1, w, h, 1), A0 = wrapF(a0, w, h, 1), A1 = wrapF(a1, w, h, 1), f;
  if (flow_in) f = wrapF(flow_in, w, h, 2);
  patchMatchPropagationAndSearch(i0, i1, A0, A1, f, hint, mkParams(maxPct), stage_mask);
  std::memcpy(flow_out, f.d.data(), f.d.size() * 4);
}
// PixFlow<P>::computeOpticalFlow (PixFlow.hpp:72-135)
void orc_compute_optical_flow(const uint8_t* bgra0, const uint8_t* bgra1, int cols, int rows, int maxPct, int hint, float* flow) {
  ImgU8 a = wrapU8(bgra0, cols, rows, 4), b = wrapU8(bgra1, cols, rows, 4); ImgF f;
  computeOpticalFlow(a, b, f, hint, mkParams(maxPct));
  std::memcpy(flow, f.d.data(), f.d.size() * 4);
}
// one direction of NovelViewGeneratorAsymmetricFlow::prepare (pad + solve + crop); dir 0 = LtoR (hint LEFT), 1 = RtoL (hint RIGHT)
void orc_flow_one_dir(const uint8_t* L, const uint8_t* R, int cols, int rows, int maxPct, int dir, float* flow) {
  ImgU8 l = wrapU8(L, cols, rows, 4), r = wrapU8(R, cols, rows, 4);
  const int length = cols / 20;
  ImgU8 nL = wrap_pad(l, length), nR = wrap_pad(r, length); ImgF f;
  if (dir == 0) computeOpticalFlow(nL, nR, f, LEFT, mkParams(maxPct)); else computeOpticalFlow(nR, nL, f, RIGHT, mkParams(maxPct));
  for (int y = 0; y < rows; ++y) std::memcpy(flow + size_t(y) * cols * 2, &f.d[(size_t(y) * f.w + length) * 2], size_t(cols) * 8);
}
// NovelViewGeneratorAsymmetricFlow::prepare (OpticalFlow.cpp:102-145)
void orc_flow_bidir(const uint8_t* L, const uint8_t* R, int cols, int rows, int maxPct, float* flowLtoR, float* flowRtoL) {
  ImgU8 l = wrapU8(L, cols, rows, 4), r = wrapU8(R, cols, rows, 4); ImgF a, b;
  flow_bidir(l, r, mkParams(maxPct), a, b);
  std::memcpy(flowLtoR, a.d.data(), a.d.size() * 4); std::memcpy(flowRtoL, b.d.data(), b.d.size() * 4);
}
// NovelViewUtil::combineNovelViews (OpticalFlow.cpp:30-92)
void orc_combine_novel_views(const uint8_t* L, const uint8_t* R, const float* flowLtoR, const float* flowRtoL, const float* blend,
                             int cols, int rows, uint8_t* out) {
  ImgU8 l = wrapU8(L, cols, rows, 4), r = wrapU8(R, cols, rows, 4), o;
  ImgF a = wrapF(flowLtoR, cols, rows, 2), b = wrapF(flowRtoL, cols, rows, 2), bl = wrapF(blend, cols, rows, 1);
  combineNovelViews(l, r, a, b, bl, o);
  std::memcpy(out, o.d.data(), o.d.size());
}
// Stitchtools::prepare (StitchTool.cpp:7-36); smooth=0 skips :130-143
void orc_stitch_prepare(const uint8_t* L, const uint8_t* R, int cols, int rows, int smooth, uint8_t* map, uint8_t* ovL, uint8_t* ovR,
                        float* blend, float* mergedDis) {
  Stitch s; s.prepare(wrapU8(L, cols, rows, 4), wrapU8(R, cols, rows, 4), smooth != 0);
  std::memcpy(map, s.Map.d.data(), s.Map.d.size());
  std::memcpy(ovL, s.OverlappedL.d.data(), s.OverlappedL.d.size());
  std::memcpy(ovR, s.OverlappedR.d.data(), s.OverlappedR.d.size());
  std::memcpy(blend, s.Blend.d.data(), s.Blend.d.size() * 4);
  std::memcpy(mergedDis, s.MergedDis.d.data(), s.MergedDis.d.size() * 4);
}
// the smoothing tail of GenerateBlend alone (StitchTool.cpp:130-143)
void orc_blend_smooth(float* blend, const float* mergedDis, int cols, int rows) {
  ImgF b = wrapF(blend, cols, rows, 1), md = wrapF(mergedDis, cols, rows, 1);
  const int step = cols <= rows ? cols / 200 : rows / 200, k1 = rows / 130, k2 = rows / 400;
  if (step > 0 && k1 > 0)
    for (int y = 0; y + step < rows; y += step)
      for (int x = 0; x + step < cols; x += step)
        if (md.at(y, x) > step) box_blur_roi(b, x, y, step, step, k1);
  if (k2 > 0) box_blur_roi(b, 0, 0, cols, rows, k2);
  std::memcpy(blend, b.d.data(), b.d.size() * 4);
}
// Stitchtools::Gather (StitchTool.cpp:52-96)
void orc_stitch_gather(const uint8_t* L, const uint8_t* R, const uint8_t* merged, const uint8_t* map, int cols, int rows, uint8_t* out) {
  Stitch s; s.ImageL = wrapU8(L, cols, rows, 4); s.ImageR = wrapU8(R, cols, rows, 4); s.Mergedmiddle = wrapU8(merged, cols, rows, 4);
  s.Map = wrapU8(map, cols, rows, 1); s.Gather();
  std::memcpy(out, s.FinalResult.d.data(), s.FinalResult.d.size());
}
const char* orc_version() { return "pixflow-oracle r1 (parity unpinned: OpenCV 3.2 semantics restated)"; }
}
