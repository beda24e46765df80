// A solver with its OWN coefficients through the drop-in header: the reference's PixFlow<P> takes them as constructor arguments
// (CPU/PixFlow.hpp:54-68); its factory only ever passes one set (:459-497), the class accepts any.
//   custom_flow <cols> <rows> <I0.bgra> <I1.bgra> <hint 0..4> <pyrScale> <smoothness> <vReg> <hReg> <stepSize> <out.f32>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "../include/PixFlow.hpp"

using namespace panocv;
using namespace util;
using namespace optical_flow;

int main(int argc, char** argv) {
  if (argc < 12) { std::cerr << "usage: custom_flow cols rows I0.bgra I1.bgra hint pyrScale smoothness vReg hReg stepSize out.f32\n"; return 2; }
  try {
    const int cols = atoi(argv[1]), rows = atoi(argv[2]);
    Mat I0(rows, cols, CV_8UC4), I1(rows, cols, CV_8UC4);
    for (int i = 0; i < 2; ++i) {
      FILE* f = fopen(argv[3 + i], "rb");
      Mat& m = i ? I1 : I0;
      if (!f || fread(m.data, 1, size_t(rows) * m.step, f) != size_t(rows) * m.step) throw VrCamException(std::string("failed to load image: ") + argv[3 + i]);
      fclose(f);
    }
    OpticalFlowInterface* flowAlg = new PixFlow<0>(float(atof(argv[6])), float(atof(argv[7])), float(atof(argv[8])), float(atof(argv[9])), float(atof(argv[10])), 0.5f, 0.0f);
    Mat flow;
    flowAlg->computeOpticalFlow(I0, I1, flow, OpticalFlowInterface::DirectionHint(atoi(argv[5])));
    delete flowAlg;
    // the shared context is back on the factory's presets: a named algorithm right behind a custom one must not inherit its coefficients
    OpticalFlowInterface* preset = makeOpticalFlowByName("pixflow_low");
    Mat flowPreset;
    preset->computeOpticalFlow(I0, I1, flowPreset, OpticalFlowInterface::DirectionHint(atoi(argv[5])));
    delete preset;
    FILE* o = fopen(argv[11], "wb");
    if (!o) throw VrCamException("failed to write");
    for (int y = 0; y < rows; ++y) fwrite(flow.data + size_t(y) * flow.step, 1, size_t(cols) * 8, o);
    for (int y = 0; y < rows; ++y) fwrite(flowPreset.data + size_t(y) * flowPreset.step, 1, size_t(cols) * 8, o);
    fclose(o);
  } catch (const VrCamException& e) {
    std::cerr << "VrCamException: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
