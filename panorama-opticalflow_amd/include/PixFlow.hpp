// Drop-in for the reference's CPU/PixFlow.hpp public surface (OpticalFlowInterface :15-26,
// PixFlow<MaxPercentage> :28-70, makeOpticalFlowByName :459-500).  The solver itself runs on the
// MI355X through the C ABI (include/panoflow.h); nothing is computed on the host.
#ifndef PixFlow_h
#define PixFlow_h

#include <string>

#include "util.hpp"

namespace optical_flow {
using namespace panocv;
using namespace util;

class OpticalFlowInterface {
 public:
  virtual ~OpticalFlowInterface() {}
  enum class DirectionHint { UNKNOWN, RIGHT, DOWN, LEFT, UP };
  virtual void computeOpticalFlow(const Mat& I0BGRA, const Mat& I1BGRA, Mat& flow, DirectionHint hint) = 0;
};

template <int MaxPercentage = 0>
struct PixFlow : public OpticalFlowInterface {
  const float pyrScaleFactor, smoothnessCoef, verticalRegularizationCoef, horizontalRegularizationCoef, gradientStepSize, downscaleFactor,
      directionalRegularizationCoef;

  PixFlow(const float pyrScaleFactor, const float smoothnessCoef, const float verticalRegularizationCoef, const float horizontalRegularizationCoef,
          const float gradientStepSize, const float downscaleFactor, const float directionalRegularizationCoef)
      : pyrScaleFactor(pyrScaleFactor), smoothnessCoef(smoothnessCoef), verticalRegularizationCoef(verticalRegularizationCoef),
        horizontalRegularizationCoef(horizontalRegularizationCoef), gradientStepSize(gradientStepSize), downscaleFactor(downscaleFactor),
        directionalRegularizationCoef(directionalRegularizationCoef) {
    // Round 6: the coefficients are run-time arguments of the device kernels (pf_set_solver_params, include/panoflow.h), as they are
    // constructor arguments in the reference (CPU/PixFlow.hpp:54-68).  The one exception is downscaleFactor: the 8-bit half-resolution
    // front end exists for 0.5 only -- rejected here instead of silently ignored; the other ranges are checked by the library at solve time.
    if (downscaleFactor != 0.5f) throw VrCamException("PixFlow: downscaleFactor must be 0.5 (the reference factory's value, CPU/PixFlow.hpp:466)");
  }
  ~PixFlow() {}

  void computeOpticalFlow(const Mat& rgba0byte, const Mat& rgba1byte, Mat& flow, DirectionHint hint) override {
    if (rgba0byte.type() != CV_8UC4 || rgba1byte.type() != CV_8UC4 || rgba0byte.rows != rgba1byte.rows || rgba0byte.cols != rgba1byte.cols ||
        rgba0byte.step != rgba1byte.step)
      throw VrCamException("computeOpticalFlow: inputs must be two CV_8UC4 images of equal size");
    Mat out(rgba0byte.rows, rgba0byte.cols, CV_32FC2);
    // this object's parameters for this solve; the shared per-thread context then returns to the factory's presets, which the other classes
    // (NovelViewGeneratorAsymmetricFlow, Stitchtools: they only know algorithm NAMES, CPU/OpticalFlow.cpp:128) rely on
    pf_ctx* ctx = pano::context();
    const pf_solver_params sp = {pyrScaleFactor, smoothnessCoef, verticalRegularizationCoef, horizontalRegularizationCoef, gradientStepSize, downscaleFactor,
                                 directionalRegularizationCoef};
    pano::check(pf_set_solver_params(ctx, &sp));
    const int rc = pf_flow(ctx, rgba0byte.data, rgba1byte.data, rgba0byte.cols, rgba0byte.rows, rgba0byte.step, MaxPercentage, int(hint), out.ptr<float>(), out.step);
    const std::string msg = rc ? pf_last_error(ctx) : "";
    pf_set_solver_params(ctx, nullptr);
    if (rc) throw VrCamException("panoflow: " + msg);
    flow = out;
  }
};

static inline OpticalFlowInterface* makeOpticalFlowByName(const std::string flowAlgName) {
  if (flowAlgName == "pixflow_low") return new PixFlow<0>(0.9f, 0.001f, 0.01f, 0.01f, 0.5f, 0.5f, 0.0f);
  if (flowAlgName == "pixflow_search_20") return new PixFlow<20>(0.9f, 0.001f, 0.01f, 0.01f, 0.5f, 0.5f, 0.0f);
  throw VrCamException("unrecognized flow algorithm name: " + flowAlgName);
}

}  // namespace optical_flow

#endif /* PixFlow_h */
