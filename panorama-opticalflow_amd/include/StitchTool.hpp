// Drop-in for the reference's CPU/StitchTool.hpp (:21-61).  Map / overlap masks / blend ramp / composite
// are computed by the StitchTool kernels through pf_stitch_prepare / pf_stitch_gather.
#ifndef StitchTool_hpp
#define StitchTool_hpp

#include <string>

#include "util.hpp"

namespace stitch_tools {
using namespace panocv;

class Stitchtools {
 public:
  Mat ImageL, ImageR;
  Mat Blend;
  Mat OverlappedL, OverlappedR;
  Mat Mergedmiddle;  // merged image in the overlap
  Mat Map;           // ImageL only: 100; ImageR only: 50; overlap: 150
  Mat FinalResult;
  Mat MergedDis;

  Stitchtools() {}
  ~Stitchtools() {}

  // CPU/StitchTool.cpp:7-36.  One fused device pass (pf_stitch_prepare) = MatchImages() + the overlap masking + GenerateBlend(): nothing can
  // edit Map between them here, so the result is what calling the two public methods in sequence gives.
  void prepare(const Mat& colorImageL, const Mat& colorImageR) {
    if (colorImageL.type() != CV_8UC4 || colorImageR.type() != CV_8UC4 || colorImageL.rows != colorImageR.rows || colorImageL.cols != colorImageR.cols)
      throw util::VrCamException("Stitchtools::prepare: inputs must be two CV_8UC4 images of equal size");
    ImageL = colorImageL.clone();
    ImageR = colorImageR.clone();
    raw_ = Mat(); rawDis_ = Mat();
    const int r = ImageL.rows, c = ImageL.cols;
    Map = Mat(r, c, CV_8UC1); OverlappedL = Mat(r, c, CV_8UC4); OverlappedR = Mat(r, c, CV_8UC4);
    Blend = Mat(r, c, CV_32FC1); MergedDis = Mat(r, c, CV_32FC1);
    pano::check(pf_stitch_prepare(pano::context(), ImageL.data, ImageR.data, c, r, ImageL.step, Map.data, Map.step, OverlappedL.data, OverlappedR.data,
                                  Blend.ptr<float>(), Blend.step, MergedDis.ptr<float>()));
  }
  // CPU/StitchTool.cpp:38-50: Map from the two alpha channels (+ the overlap-masked copies prepare() derives from it, :17-33).
  void MatchImages() {
    if (ImageL.empty()) throw util::VrCamException("Stitchtools::MatchImages: prepare first");
    const int r = ImageL.rows, c = ImageL.cols;
    Map = Mat(r, c, CV_8UC1); OverlappedL = Mat(r, c, CV_8UC4); OverlappedR = Mat(r, c, CV_8UC4);
    pano::check(pf_stitch_match(pano::context(), ImageL.data, ImageR.data, c, r, ImageL.step, Map.data, Map.step, OverlappedL.data, OverlappedR.data));
  }
  // CPU/StitchTool.cpp:98-146: the blend ramp of the CURRENT Map member (a caller may have edited it since MatchImages(), as with the reference).
  void GenerateBlend() {
    if (Map.empty()) throw util::VrCamException("Stitchtools::GenerateBlend: MatchImages first");
    Blend = Mat(Map.rows, Map.cols, CV_32FC1); MergedDis = Mat(Map.rows, Map.cols, CV_32FC1);
    pano::check(pf_stitch_generate_blend(pano::context(), Map.data, Map.step, Map.cols, Map.rows, Blend.ptr<float>(), Blend.step, MergedDis.ptr<float>()));
  }
  // CPU/StitchTool.cpp:148-191.  x is in wrap-extended map coordinates (cols/5 columns were prepended, :102-111).
  // Like the reference it returns the RAW ratio minLdis / (minRdis + minLdis) of that pixel -- not the smoothed ramp --
  // and writes min(minLdis, minRdis) into MergedDis (:185-188).  The device evaluates the 8-direction search for the
  // whole image in one pass (pf_stitch_raw_blend); the result is cached until the next prepare().
  float countblend(const int x, const int y) {
    if (ImageL.empty()) throw util::VrCamException("Stitchtools::countblend: prepare first");
    if (raw_.empty()) {
      raw_ = Mat(ImageL.rows, ImageL.cols, CV_32FC1); rawDis_ = Mat(ImageL.rows, ImageL.cols, CV_32FC1);
      pano::check(pf_stitch_raw_blend(pano::context(), ImageL.data, ImageR.data, ImageL.cols, ImageL.rows, ImageL.step, raw_.ptr<float>(), raw_.step,
                                      rawDis_.ptr<float>()));
    }
    int sx = x - ImageL.cols / 5;
    if (sx < 0) sx += ImageL.cols; else if (sx >= ImageL.cols) sx -= ImageL.cols;
    if (MergedDis.empty()) MergedDis = Mat(ImageL.rows, ImageL.cols, CV_32FC1);
    MergedDis.at<float>(y, sx) = rawDis_.at<float>(y, sx);
    return raw_.at<float>(y, sx);
  }
  // CPU/StitchTool.cpp:52-96
  void Gather() {
    if (Mergedmiddle.empty() || Mergedmiddle.step != ImageL.step) throw util::VrCamException("Stitchtools::Gather: setMergedmiddle first");
    FinalResult = Mat(ImageL.rows, ImageL.cols, CV_8UC4);
    pano::check(pf_stitch_gather(pano::context(), ImageL.data, ImageR.data, Mergedmiddle.data, ImageL.step, Map.data, Map.step, ImageL.cols, ImageL.rows,
                                 FinalResult.data, FinalResult.step));
  }

  Mat getImageL() { return ImageL; }
  Mat getImageR() { return ImageR; }
  Mat getBlend() { return Blend; }
  Mat getMap() { return Map; }
  Mat getOverlappedL() { return OverlappedL; }
  Mat getOverlappedR() { return OverlappedR; }
  Mat getFinalResult() { return FinalResult; }
  void setMergedmiddle(const Mat& image) { Mergedmiddle = image.clone(); }

 private:
  Mat raw_, rawDis_;    // unsmoothed ramp / MergedDis for countblend(), computed on first use
};

// One whole iteration of the stitch loop (CPU/main.cpp:70-95) without leaving the device (pf_stitch_step).
// colorImageR == nullptr chains on the previous call's result, which stays resident in HBM (main.cpp:64-65).
static inline Mat stitchStep(const Mat& colorImageL, const Mat* colorImageR, const std::string& flowAlgName) {
  const int maxPct = pf_max_percentage_by_name(flowAlgName.c_str());
  if (maxPct < 0) throw util::VrCamException("unrecognized flow algorithm name: " + flowAlgName);
  if (colorImageL.type() != CV_8UC4 || (colorImageR && (colorImageR->type() != CV_8UC4 || colorImageR->rows != colorImageL.rows ||
                                                         colorImageR->cols != colorImageL.cols || colorImageR->step != colorImageL.step)))
    throw util::VrCamException("stitchStep: inputs must be CV_8UC4 images of equal size");
  Mat out(colorImageL.rows, colorImageL.cols, CV_8UC4);
  pano::check(pf_stitch_step(pano::context(), colorImageL.data, colorImageR ? colorImageR->data : nullptr, colorImageL.cols, colorImageL.rows,
                             colorImageL.step, maxPct, out.data, out.step));
  return out;
}

}  // namespace stitch_tools

#endif /* StitchTool_hpp */
