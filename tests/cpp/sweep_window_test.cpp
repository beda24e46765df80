// CPU check of the sweep-window geometry (panorama-opticalflow_amd/csrc/sweep_window.hpp): for random images, boxes and
// both directions, replay the (band, step, row) -> pixel mapping the prepass and the sweep kernel use and verify that
//   * every pixel of the box is visited exactly once,
//   * visited pixels outside the box only lie in the one column before it / inside the first and last band,
//   * for a non-empty box the pixel "before" each box row (previous along the step axis) is visited too, when it exists,
//   * padding steps map to no pixel, the box's bands fit the workgroup count.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../panorama-opticalflow_amd/csrc/sweep_window.hpp"

static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static int rnd(int n) { rng ^= rng << 7; rng ^= rng >> 9; rng *= 0x2545F4914F6CDD1Dull; return int((rng >> 33) % (unsigned long long)n); }

int main() {
  const int kRows = 8, kWaves = 4, kChunk = 8;
  int cases = 0;
  for (int it = 0; it < 4000; ++it) {
    const int W = 1 + rnd(140), H = 1 + rnd(140), fwd = rnd(2);
    int x0 = rnd(W), x1 = x0 + 1 + rnd(W - x0), y0 = rnd(H), y1 = y0 + 1 + rnd(H - y0);
    if (it % 17 == 0) { x0 = 0; y0 = 0; x1 = W; y1 = H; }          // whole image
    if (it % 23 == 0) { x1 = x0; }                                   // empty
    const pf::SweepWindow w = pf::make_sweep_window(W, H, fwd, x0, y0, x1, y1, kRows, kWaves, kChunk);
    if (x1 <= x0) { if (!w.empty) { printf("empty box not flagged\n"); return 1; } continue; }
    if (w.empty || w.nwg * kWaves < w.nbands || w.nstepsPad % kChunk || w.nstepsPad < w.LSv + kRows - 1) { printf("bad sizes\n"); return 1; }
    const int LS = w.tr ? H : W, LB = w.tr ? W : H;
    std::vector<int> seen(size_t(W) * H, 0);
    for (int band = 0; band < w.nbands; ++band)   // (the prepass also fills records for the padding bands of the last workgroup; no compute wave runs them)
      for (int s = 0; s < w.nstepsPad; ++s)
        for (int r = 0; r < kRows; ++r) {
          const int ia = w.uLo + s - r, ib = (w.bandLo + band) * kRows + r;          // as in k_sweep_prep
          if (!(s - r >= 0 && ia < w.uHi && ia < LS && ib < LB)) continue;
          const int cx = w.tr ? ib : ia, cy = w.tr ? ia : ib;
          const int x = fwd ? cx : W - 1 - cx, y = fwd ? cy : H - 1 - cy;
          if (x < 0 || x >= W || y < 0 || y >= H) { printf("out of image\n"); return 1; }
          ++seen[size_t(y) * W + x];
        }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const bool in = x >= x0 && x < x1 && y >= y0 && y < y1;
        const int n = seen[size_t(y) * W + x];
        if (n > 1 || (in && n != 1)) { printf("W %d H %d fwd %d box %d %d %d %d: pixel (%d,%d) visited %d times\n", W, H, fwd, x0, y0, x1, y1, x, y, n); return 1; }
        if (!in && n) {   // allowed: the column before the box along the step axis, rows/columns of the first/last band outside the box
          const int cx = fwd ? x : W - 1 - x, cy = fwd ? y : H - 1 - y;
          const int u = w.tr ? cy : cx, v = w.tr ? cx : cy;
          if (u < w.uLo || u >= w.uHi || v < w.bandLo * kRows || v >= (w.bandLo + w.nbands) * kRows) { printf("stray pixel visited\n"); return 1; }
        }
      }
    // the previous pixel along the step axis of the box's first column is visited (it hands on its unchanged flow)
    {
      const int cx0 = fwd ? x0 : W - x1, cy0 = fwd ? y0 : H - y1;
      const int U0 = w.tr ? cy0 : cx0;
      if (U0 > 0 && w.uLo != U0 - 1) { printf("uLo is not the column before the box\n"); return 1; }
      if (U0 == 0 && w.uLo != 0) { printf("uLo\n"); return 1; }
    }
    ++cases;
  }
  printf("ok %d cases\n", cases);
  return 0;
}
