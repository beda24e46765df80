// Drop-in for the reference's CLI driver (CPU/main.cpp:47-110): same flags, same file names, same
// 5-step chain (R_i = FinalResult_{i-1}, main.cpp:64-65), same timing lines; all pixel work on the MI355X.
//   pano_stitch -test_dir <dir> -top_img top.tif -flow_alg pixflow_low|pixflow_search_20 [-steps 5] [-fused 0|1]
//   pano_stitch -inputs 4 -test_dir <dir> -flow_alg ...      the one-pass 4-photo variant (CPU_4Input/main.cpp:45-120)
// reads <dir>/<top_img> and <dir>/1.tif .. 5.tif (8-bit RGB/RGBA TIFF or PNG), writes ProcessResult{i}.png and
// FinalResult.png (main.cpp:97-100).
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

#include "../include/OpticalFlow.hpp"
#include "../include/StitchTool.hpp"
#include "image_io.hpp"

using namespace panocv;
using namespace util;
using namespace optical_flow;
using namespace stitch_tools;

static std::map<std::string, std::string> parseFlags(int argc, char** argv) {   // gflags syntax: -name value | --name=value
  std::map<std::string, std::string> f;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.empty() || a[0] != '-') throw VrCamException("unexpected argument: " + a);
    a = a.substr(a.find_first_not_of('-'));
    const size_t eq = a.find('=');
    if (eq != std::string::npos) f[a.substr(0, eq)] = a.substr(eq + 1);
    else if (i + 1 < argc) f[a] = argv[++i];
    else throw VrCamException("missing value for flag: " + a);
  }
  return f;
}


// CPU_4Input/main.cpp:54-113: crop every photo to the columns where its centre row is opaque, L = 1 + 3, R = 2 + 4
// (saturating), then ONE stitch step.  The crop/sum is the driver's own image preparation (byte copies on the host,
// as in the reference); the stitch step runs on the device.
static int run4Input(const std::string& dir, const std::string& flow_alg) {
  double StartTime = getCurrTimeSec();
  Mat im[4];
  for (int i = 0; i < 4; ++i) im[i] = pano_io::imreadExceptionOnFail(dir + "/" + char(i + 49) + ".tif");
  for (int i = 1; i < 4; ++i)
    if (im[i].rows != im[0].rows || im[i].cols != im[0].cols) throw VrCamException("4-input mode: the four photos must have the same size");
  const int rows = im[0].rows, cols = im[0].cols;
  for (int i = 0; i < 4; ++i)
    for (int x = 0; x < cols; ++x)
      if (!im[i].at<Vec4b>(rows / 2, x)[3])
        for (int y = 0; y < rows; ++y) im[i].at<Vec4b>(y, x) = Vec4b(0, 0, 0, 0);
  Mat colorImageL(rows, cols, CV_8UC4), colorImageR(rows, cols, CV_8UC4);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols * 4; ++x) {
      const int l = im[0].ptr<unsigned char>(y)[x] + im[2].ptr<unsigned char>(y)[x], r = im[1].ptr<unsigned char>(y)[x] + im[3].ptr<unsigned char>(y)[x];
      colorImageL.ptr<unsigned char>(y)[x] = (unsigned char)(l > 255 ? 255 : l);   // cv::Mat + cv::Mat on CV_8U saturates
      colorImageR.ptr<unsigned char>(y)[x] = (unsigned char)(r > 255 ? 255 : r);
    }
  Mat FinalResult = stitchStep(colorImageL, &colorImageR, flow_alg);
  pano_io::imwriteExceptionOnFail(dir + "/FinalResult.png", FinalResult);
  std::cout << "TotalRunTime (sec) = " << (getCurrTimeSec() - StartTime) << std::endl;
  return EXIT_SUCCESS;
}

int main(int argc, char** argv) {
  try {
    auto flags = parseFlags(argc, argv);
    const std::string FLAGS_test_dir = flags["test_dir"], FLAGS_top_img = flags["top_img"], FLAGS_flow_alg = flags["flow_alg"];
    const int nsteps = flags.count("steps") ? atoi(flags["steps"].c_str()) : 5;
    const bool fused = !flags.count("fused") || atoi(flags["fused"].c_str()) != 0;   // -fused 0: the reference's object-by-object sequence
    if (flags.count("inputs") && atoi(flags["inputs"].c_str()) == 4) {
      requireArg(FLAGS_test_dir, "test_dir");
      requireArg(FLAGS_flow_alg, "flow_alg");
      return run4Input(FLAGS_test_dir, FLAGS_flow_alg);
    }
    double StartTime = getCurrTimeSec();
    requireArg(FLAGS_test_dir, "test_dir");
    requireArg(FLAGS_top_img, "top_img");
    requireArg(FLAGS_flow_alg, "flow_alg");

    Mat colorImageL, colorImageR, FinalResult;
    Mat colorImageT = pano_io::imreadExceptionOnFail(FLAGS_test_dir + "/" + FLAGS_top_img);
    for (int i = 1; i <= nsteps; i++) {
      double StepStart = getCurrTimeSec();
      if (i == 1) colorImageR = colorImageT; else colorImageR = FinalResult;
      colorImageL = pano_io::imreadExceptionOnFail(FLAGS_test_dir + "/" + char(i + 48) + ".tif");

      if (fused) {
        // same kernels, same results; the step's intermediates and the chained R stay in HBM
        FinalResult = stitchStep(colorImageL, i == 1 ? &colorImageR : nullptr, FLAGS_flow_alg);
      } else {
      Stitchtools Stools;
      Stools.prepare(colorImageL, colorImageR);
      Mat overlappedL = Stools.getOverlappedL();
      Mat overlappedR = Stools.getOverlappedR();
      Mat blend = Stools.getBlend();

      NovelViewGenerator* novelViewGen = new NovelViewGeneratorAsymmetricFlow(FLAGS_flow_alg);
      novelViewGen->prepare(overlappedL, overlappedR);
      novelViewGen->setBlend(blend);
      Mat novelViewMerged = Mat();
      novelViewGen->generateNovelView(novelViewMerged);

      Stools.setMergedmiddle(novelViewMerged);
      Stools.Gather();
      FinalResult = Stools.getFinalResult();
      delete novelViewGen;
      }

      if (i == nsteps) pano_io::imwriteExceptionOnFail(FLAGS_test_dir + "/" + "FinalResult.png", FinalResult);
      else pano_io::imwriteExceptionOnFail(FLAGS_test_dir + "/" + "ProcessResult" + char(i + 48) + ".png", FinalResult);
      std::cout << "Part" << i << " Finished!" << "RUNTIME (sec) = " << (getCurrTimeSec() - StepStart) << std::endl;
    }
    std::cout << "TotalRunTime (sec) = " << (getCurrTimeSec() - StartTime) << std::endl;
    return EXIT_SUCCESS;
  } catch (const VrCamException& e) {
    std::cerr << "VrCamException: " << e.what() << std::endl;   // the reference's terminate handler prints and aborts (util.cpp:59-78)
    return EXIT_FAILURE;
  }
}
