// Drop-in for the reference's CPU/OpticalFlow.hpp (NovelViewUtil :18-32, NovelViewGenerator :34-48,
// NovelViewGeneratorAsymmetricFlow :50-70).  prepare() runs both flow solves concurrently on the GPU
// (pf_flow_bidir), generateNovelView() runs the blend kernel (pf_blend).
#ifndef OpticalFlow_hpp
#define OpticalFlow_hpp

#include <string>

#include "PixFlow.hpp"
#include "util.hpp"

namespace optical_flow {
using namespace panocv;
using namespace util;

struct NovelViewUtil {
  // CPU/OpticalFlow.cpp:9-28: one warped sample (nearest, truncating; x wraps, y clamps).  Pure indexing,
  // kept for API completeness -- combineNovelViews does NOT call it, the blend kernel does the warp.
  static Vec4b generateNovelViewPoint(const Mat& srcImage, const Mat& flow, const double t, const int x, const int y) {
    const Point2f flowDir = flow.at<Point2f>(y, x);
    int srcx = int(x + flowDir.x * t);
    if (srcx > srcImage.cols - 1) srcx = srcx - srcImage.cols;
    if (srcx < 0) srcx = srcx + srcImage.cols;
    srcx %= srcImage.cols; if (srcx < 0) srcx += srcImage.cols;
    int srcy = int(y + flowDir.y * t);
    if (srcy > srcImage.rows - 1) srcy = srcImage.rows - 1;
    if (srcy < 0) srcy = 0;
    return srcImage.at<Vec4b>(srcy, srcx);
  }

  // CPU/OpticalFlow.cpp:30-92
  static Mat combineNovelViews(const Mat& imageL, const Mat& imageR, const Mat& flowLtoR, const Mat& flowRtoL, const Mat& blend) {
    if (imageL.type() != CV_8UC4 || imageR.type() != CV_8UC4 || flowLtoR.type() != CV_32FC2 || flowRtoL.type() != CV_32FC2 || blend.type() != CV_32FC1)
      throw VrCamException("combineNovelViews: unexpected Mat types");
    if (imageL.step != imageR.step || flowLtoR.step != flowRtoL.step) throw VrCamException("combineNovelViews: mismatched row steps");
    Mat blendImage(imageL.rows, imageL.cols, CV_8UC4);
    pano::check(pf_blend(pano::context(), imageL.data, imageR.data, imageL.step, flowLtoR.ptr<float>(), flowRtoL.ptr<float>(), flowLtoR.step,
                         blend.ptr<float>(), blend.step, imageL.cols, imageL.rows, blendImage.data, blendImage.step));
    return blendImage;
  }
};

class NovelViewGenerator {
 public:
  virtual ~NovelViewGenerator() {}
  virtual void prepare(const Mat& colorImageL, const Mat& colorImageR) = 0;
  virtual void generateNovelView(Mat& outNovelViewMerged) = 0;
  virtual Mat getFlowLtoR() { return Mat(); }
  virtual Mat getFlowRtoL() { return Mat(); }
  virtual void setBlend(const Mat& blend) = 0;
};

class NovelViewGeneratorAsymmetricFlow : public NovelViewGenerator {
 public:
  std::string flowAlgName;
  Mat imageL, imageR;
  Mat flowLtoR, flowRtoL;
  Mat Blend;

  NovelViewGeneratorAsymmetricFlow(const std::string flowAlgName) : flowAlgName(flowAlgName) {}
  ~NovelViewGeneratorAsymmetricFlow() {}

  // CPU/OpticalFlow.cpp:102-145
  void prepare(const Mat& colorImageL, const Mat& colorImageR) override {
    const int maxPct = pf_max_percentage_by_name(flowAlgName.c_str());
    if (maxPct < 0) throw VrCamException("unrecognized flow algorithm name: " + flowAlgName);
    if (colorImageL.type() != CV_8UC4 || colorImageR.type() != CV_8UC4 || colorImageL.rows != colorImageR.rows || colorImageL.cols != colorImageR.cols)
      throw VrCamException("prepare: inputs must be two CV_8UC4 images of equal size");
    imageL = colorImageL.clone();
    imageR = colorImageR.clone();
    flowLtoR = Mat(imageL.rows, imageL.cols, CV_32FC2);
    flowRtoL = Mat(imageL.rows, imageL.cols, CV_32FC2);
    pano::check(pf_flow_bidir(pano::context(), imageL.data, imageR.data, imageL.cols, imageL.rows, imageL.step, maxPct, flowLtoR.ptr<float>(),
                              flowRtoL.ptr<float>(), flowLtoR.step));
  }
  // CPU/OpticalFlow.cpp:94-100
  void generateNovelView(Mat& outNovelViewMerged) override {
    outNovelViewMerged = NovelViewUtil::combineNovelViews(imageL, imageR, flowLtoR, flowRtoL, Blend);
  }
  Mat getFlowLtoR() override { return flowLtoR; }
  Mat getFlowRtoL() override { return flowRtoL; }
  void setBlend(const Mat& blend) override { Blend = blend.clone(); }
};

}  // namespace optical_flow

#endif /* OpticalFlow_hpp */
