// One iteration of the reference's stitch loop (CPU/main.cpp:70-95) written against the drop-in
// headers, with raw-file I/O: the call sequence is the reference's own.
//   stitch_pair <cols> <rows> <L.bgra> <R.bgra> <flow_alg> <out_prefix>
// writes <out_prefix>.final.bgra, .merged.bgra, .blend.f32, .flowLR.f32, .flowRL.f32, .map.u8
#include <cstdio>
#include <iostream>

#include "../include/OpticalFlow.hpp"
#include "../include/StitchTool.hpp"

using namespace panocv;
using namespace util;
using namespace optical_flow;
using namespace stitch_tools;

static Mat readRaw(const std::string& path, int rows, int cols, int type) {
  Mat m(rows, cols, type);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw VrCamException("failed to load image: " + path);
  const size_t n = size_t(rows) * m.step;
  if (fread(m.data, 1, n, f) != n) { fclose(f); throw VrCamException("short read: " + path); }
  fclose(f);
  return m;
}
static void writeRaw(const std::string& path, const Mat& m) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw VrCamException("failed to write image: " + path);
  for (int y = 0; y < m.rows; ++y) fwrite(m.data + size_t(y) * m.step, 1, size_t(m.cols) * m.elemSize(), f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 7) { std::cerr << "usage: stitch_pair cols rows L.bgra R.bgra flow_alg out_prefix\n"; return 2; }
  try {
    const int cols = atoi(argv[1]), rows = atoi(argv[2]);
    const std::string flow_alg = argv[5], prefix = argv[6];
    Mat colorImageL = readRaw(argv[3], rows, cols, CV_8UC4), colorImageR = readRaw(argv[4], rows, cols, CV_8UC4);
    double StartTime = getCurrTimeSec();

    Stitchtools Stools;
    Stools.prepare(colorImageL, colorImageR);
    Mat overlappedL = Stools.getOverlappedL();
    Mat overlappedR = Stools.getOverlappedR();
    Mat blend = Stools.getBlend();

    NovelViewGenerator* novelViewGen = new NovelViewGeneratorAsymmetricFlow(flow_alg);
    novelViewGen->prepare(overlappedL, overlappedR);
    novelViewGen->setBlend(blend);
    Mat novelViewMerged = Mat();
    novelViewGen->generateNovelView(novelViewMerged);

    Stools.setMergedmiddle(novelViewMerged);
    Stools.Gather();
    Mat FinalResult = Stools.getFinalResult();

    writeRaw(prefix + ".final.bgra", FinalResult);
    writeRaw(prefix + ".merged.bgra", novelViewMerged);
    writeRaw(prefix + ".blend.f32", blend);
    writeRaw(prefix + ".flowLR.f32", novelViewGen->getFlowLtoR());
    writeRaw(prefix + ".flowRL.f32", novelViewGen->getFlowRtoL());
    writeRaw(prefix + ".map.u8", Stools.getMap());
    delete novelViewGen;
    std::cout << "Part1 Finished!RUNTIME (sec) = " << (getCurrTimeSec() - StartTime) << std::endl;
  } catch (const VrCamException& e) {
    std::cerr << "VrCamException: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
