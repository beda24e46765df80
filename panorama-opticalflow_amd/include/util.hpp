// Host-side mirror of the parts of the reference's util.hpp that the hot path's API needs
// (reference: CPU/util.hpp:38-53).  No OpenCV/gflags/glog: a small Mat stands in for cv::Mat.
#ifndef PANO_UTIL_HPP_
#define PANO_UTIL_HPP_

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "../../include/panoflow.h"

namespace util {

// CPU/util.hpp:38-43
struct VrCamException : public std::exception {
  std::string msg;
  explicit VrCamException(const std::string& m) : msg(m) {}
  const char* what() const noexcept override { return msg.c_str(); }
};
// CPU/util.hpp:45-49
static inline void requireArg(const std::string& arg, const std::string& name) {
  if (arg.empty()) throw VrCamException("missing required command line argument: " + name);
}
// CPU/util.hpp:51-53
static inline double getCurrTimeSec() {
  return (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count() / 1000.0;
}

}  // namespace util

// ---------------------------------------------------------------------------------------------
// Minimal stand-in for the cv:: types the reference's public signatures use.  Same member names and
// type codes as OpenCV so call sites (CPU/main.cpp:70-95) compile unchanged against these headers.
// ---------------------------------------------------------------------------------------------
namespace panocv {

enum { CV_8UC1 = 0, CV_8UC3 = 16, CV_8UC4 = 24, CV_32FC1 = 5, CV_32FC2 = 13, CV_8U = 0, CV_32F = 5 };

struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Vec4b {
  unsigned char val[4] = {0, 0, 0, 0};
  Vec4b() {}
  Vec4b(unsigned char a, unsigned char b, unsigned char c, unsigned char d) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  unsigned char& operator[](int i) { return val[i]; }
  const unsigned char& operator[](int i) const { return val[i]; }
};

class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char* data = nullptr;

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(Size s, int type) { create(s.height, s.width, type); }
  // wrap caller-owned memory (no copy), like cv::Mat(rows, cols, type, data, step)
  Mat(int r, int c, int type, void* ext, size_t ext_step = 0) : rows(r), cols(c), step(ext_step ? ext_step : size_t(c) * elemSizeOf(type)), data((unsigned char*)ext), type_(type) {}

  static size_t elemSizeOf(int type) { return type == CV_8UC1 ? 1 : type == CV_8UC3 ? 3 : type == CV_8UC4 ? 4 : type == CV_32FC1 ? 4 : 8; }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type; step = size_t(c) * elemSizeOf(type);
    store_.reset(new unsigned char[step * size_t(r) + 16], std::default_delete<unsigned char[]>());
    data = store_.get();
    std::memset(data, 0, step * size_t(r));
  }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  static Mat zeros(Size s, int type) { return Mat(s, type); }
  int type() const { return type_; }
  size_t elemSize() const { return elemSizeOf(type_); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Size size() const { return Size(cols, rows); }
  Mat clone() const {
    Mat m;
    if (empty()) return m;
    m.create(rows, cols, type_);
    for (int y = 0; y < rows; ++y) std::memcpy(m.data + size_t(y) * m.step, data + size_t(y) * step, size_t(cols) * elemSize());
    return m;
  }
  template <typename T> T& at(int y, int x) { return *reinterpret_cast<T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
  template <typename T> const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
  template <typename T> T* ptr(int y = 0) { return reinterpret_cast<T*>(data + size_t(y) * step); }
  template <typename T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + size_t(y) * step); }

 private:
  int type_ = 0;
  std::shared_ptr<unsigned char> store_;
};

}  // namespace panocv

namespace pano {

// One panoflow context per host thread (the C ABI's ownership rule); created on first use on device
// PANOFLOW_DEVICE (default 0).  Throws util::VrCamException when no gfx950 device is usable: there is
// no CPU fallback (the reference's GPU build silently fell back to the CPU, GPU/OpticalFlow.cpp:160-189).
struct ContextHolder {
  pf_ctx* ctx = nullptr;
  ~ContextHolder() { if (ctx) pf_destroy(ctx); }
};
static inline pf_ctx* context() {
  static thread_local ContextHolder holder;
  if (!holder.ctx) {
    const char* env = std::getenv("PANOFLOW_DEVICE");
    // PANOFLOW_PRESIZE=COLSxROWS allocates every device buffer for that image size up front (pf_create's pre-sizing)
    int mc = 0, mr = 0;
    if (const char* ps = std::getenv("PANOFLOW_PRESIZE")) { if (std::sscanf(ps, "%dx%d", &mc, &mr) != 2) mc = mr = 0; }
    holder.ctx = pf_create(env ? std::atoi(env) : 0, mc, mr);
    if (!holder.ctx) throw util::VrCamException(std::string("panoflow: ") + pf_last_error(nullptr));
  }
  return holder.ctx;
}
static inline void check(int rc) {
  if (rc != 0) throw util::VrCamException(std::string("panoflow: ") + pf_last_error(context()));
}

}  // namespace pano

#endif  // PANO_UTIL_HPP_
