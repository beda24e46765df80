// Geometry of one v2 sweep launch: which bands and which sweep-order columns cover the bounding box of the gated
// pixels.  Plain C++ (no HIP) so that the CPU test tier can check it (tests/cpp/sweep_window_test.cpp).
#pragma once

namespace pf {

struct SweepWindow {
  bool empty;     // nothing gated: the sweep is the identity
  int tr;         // 1 = bands of 8 columns stepping along y, 0 = bands of 8 rows stepping along x
  int uLo, uHi;   // sweep-order columns [uLo, uHi) along the step axis (uLo = one before the box when there is one)
  int LSv;        // uHi - uLo
  int bandLo;     // first band (absolute index, 8 rows/columns each)
  int nbands;     // bands covered
  int nwg;        // workgroups (kWaves bands each)
  int nstepsPad;  // steps per band, padded to whole chunks
};

// box = [x0,x1) x [y0,y1) in image coordinates (clipped to the image here); forward = 1 raster order, 0 mirrored
inline SweepWindow make_sweep_window(int W, int H, int forward, int x0, int y0, int x1, int y1, int rows_per_band, int bands_per_wg, int chunk) {
  SweepWindow w = {};
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (x1 > W) x1 = W;
  if (y1 > H) y1 = H;
  if (x1 <= x0 || y1 <= y0) { w.empty = true; return w; }
  w.tr = (x1 - x0) < (y1 - y0) ? 1 : 0;   // bands across the SHORTER side of the box: fewest band-to-band hand-offs
  // the same box in sweep order (mirrored for the backward sweep), as (u = along the step axis, v = across the bands)
  const int cx0 = forward ? x0 : W - x1, cx1 = forward ? x1 : W - x0, cy0 = forward ? y0 : H - y1, cy1 = forward ? y1 : H - y0;
  const int U0 = w.tr ? cy0 : cx0, U1 = w.tr ? cy1 : cx1, V0 = w.tr ? cx0 : cy0, V1 = w.tr ? cx1 : cy1;
  w.uLo = U0 > 0 ? U0 - 1 : 0;   // one column before the box: its (unchanged) flow is the first "previous pixel" proposal
  w.uHi = U1;
  w.LSv = w.uHi - w.uLo;
  w.bandLo = V0 / rows_per_band;
  w.nbands = (V1 + rows_per_band - 1) / rows_per_band - w.bandLo;
  w.nwg = (w.nbands + bands_per_wg - 1) / bands_per_wg;
  w.nstepsPad = ((w.LSv + rows_per_band - 1) + chunk - 1) / chunk * chunk;
  return w;
}

}  // namespace pf
