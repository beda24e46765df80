// Experiment (round 2): how many parallel "chaotic relaxation" rounds does an exact sweep need?
//
// The Gauss-Seidel sweep (CPU/PixFlow.hpp:315-337) is out(p) = F(p; out(left), out(top)) on a DAG, so the
// sequence out^0 = D (the no-adoption result, known without neighbours), out^k(p) = F(p; out^{k-1}(L), out^{k-1}(T))
// reaches the sweep's exact result after at most W+H-1 rounds -- and as soon as one round changes nothing.
// A pixel has to be re-evaluated in round k only if one of its two predecessors changed in round k-1.
// This program runs the oracle's solver on a raw BGRA pair and prints, per level and sweep, the number of
// rounds and the total number of re-evaluations, and checks the fixed point equals the sequential sweep bit for bit.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off -o /tmp/jacobi_rounds tests/micro/jacobi_rounds.cpp
// run:   /tmp/jacobi_rounds L.raw R.raw cols rows [maxPct]     (raw = cols*rows*4 BGRA bytes, already wrap-padded)
#include "../../oracle/pixflow_oracle.cpp"

#include <cstdio>

using namespace orc;

struct Stat { int level, w, h, fwd, rounds; long long evals, gated; int maxdepth; };
static std::vector<Stat> g_stats;
#include <map>
static std::map<int, double> g_model; static double g_fast = 180, g_slow = 920, g_hand = 600; static int g_rows = 8;

static inline bool same(float a, float b) { uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4); return x == y; }

// F(p; L, T): exactly the body of orc::sweep for one pixel with explicit neighbour values
static inline void eval_px(const LevelCtx& L, int x, int y, float cx, float cy, bool hasA, float ax, float ay, bool hasB, float bx, float by,
                           float& ox, float& oy) {
  const float eps = Params::kGradEpsilon;
  float fx = cx, fy = cy;
  float currErr = errorFunction(L, x, y, fx, fy);
  if (hasA) { const float e = errorFunction(L, x, y, ax, ay); if (e < currErr) { fx = ax; fy = ay; currErr = e; } }
  if (hasB) { const float e = errorFunction(L, x, y, bx, by); if (e < currErr) { fx = bx; fy = by; currErr = e; } }
  const float ex = errorFunction(L, x, y, fx + eps, fy + 0.0f);
  const float ey = errorFunction(L, x, y, fx + 0.0f, fy + eps);
  const float gx = (ex - currErr) / eps, gy = (ey - currErr) / eps;
  ox = fx - L.p->gradientStepSize * gx;
  oy = fy - L.p->gradientStepSize * gy;
}

static void jacobi(const LevelCtx& L, const ImgF& alpha0, const ImgF& alpha1, const ImgF& in, const ImgF& expect, bool forward, int level) {
  const int W = in.w, H = in.h;
  const float thr = Params::kUpdateAlphaThreshold;
  const int sx = forward ? -1 : 1, sy = forward ? -1 : 1;   // where the predecessors are
  ImgF cur = in, nxt;
  std::vector<uint8_t> gate(size_t(W) * H), dirty(size_t(W) * H, 0), changed(size_t(W) * H, 0), nchanged(size_t(W) * H, 0);
  long long gated = 0;
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { gate[size_t(y) * W + x] = alpha0.at(y, x) > thr && alpha1.at(y, x) > thr; gated += gate[size_t(y) * W + x]; }
  // round 0: the result without any adoption (neighbours "absent")
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) if (gate[size_t(y) * W + x]) {
    float ox, oy; eval_px(L, x, y, in.at(y, x, 0), in.at(y, x, 1), false, 0, 0, false, 0, 0, ox, oy);
    cur.at(y, x, 0) = ox; cur.at(y, x, 1) = oy;
  }
  std::fill(changed.begin(), changed.end(), 1);   // round 1 evaluates every gated pixel
  long long evals = gated; int rounds = 0;
  for (;;) {
    ++rounds;
    nxt = cur; std::fill(nchanged.begin(), nchanged.end(), 0);
    long long nch = 0;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      const size_t i = size_t(y) * W + x;
      if (!gate[i]) continue;
      const int ax = x + sx, by = y + sy;
      const bool hasA = ax >= 0 && ax < W, hasB = by >= 0 && by < H;
      const bool d = (hasA && changed[size_t(y) * W + ax]) || (hasB && changed[size_t(by) * W + x]);
      if (!d) continue;
      ++evals;
      float ox, oy;
      eval_px(L, x, y, in.at(y, x, 0), in.at(y, x, 1), hasA, hasA ? cur.at(y, ax, 0) : 0, hasA ? cur.at(y, ax, 1) : 0, hasB, hasB ? cur.at(by, x, 0) : 0,
              hasB ? cur.at(by, x, 1) : 0, ox, oy);
      if (!same(ox, cur.at(y, x, 0)) || !same(oy, cur.at(y, x, 1))) { nchanged[i] = 1; ++nch; nxt.at(y, x, 0) = ox; nxt.at(y, x, 1) = oy; }
    }
    cur.d.swap(nxt.d); changed.swap(nchanged);
    for (int K : {2, 4, 8, 16, 32, 64}) if (rounds == K) {
      // pixel "slow" in a verifying wavefront after K rounds: its predecessors' final values differ from the ones its last evaluation used,
      // i.e. a predecessor still changes after round K-1  <=>  predecessor's value after round K-1 (= cur before... see below) != final.
      // Here: cur = out^K.  A pixel evaluated with out^{K-1} neighbours is right iff those equal the final values; approximate with out^K (one round later => slightly optimistic).
      long long wrong = 0, cells = 0, slow = 0;
      std::vector<uint8_t> bad(size_t(W) * H, 0);
      for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { const size_t i = size_t(y) * W + x; if (!same(cur.d[2 * i], expect.d[2 * i]) || !same(cur.d[2 * i + 1], expect.d[2 * i + 1])) { bad[i] = 1; ++wrong; } }
      // bands of 8 across the shorter side; step = anti-diagonal inside the band; pixel needs the slow path if a predecessor is bad
      const bool tr = W < H; const int LB = tr ? W : H, LS = tr ? H : W;
      const int nb = (LB + g_rows - 1) / g_rows, ns = LS + g_rows - 1;
      // pipeline model of the band wavefront: band b step s may start when its own step s-1 and band b-1's step s+7 (+ hand-off) are done
      const double cFast = g_fast, cSlow = g_slow, cHand = g_hand;
      std::vector<double> Tprev(ns + 8, 0.0), Tcur(ns + 8, 0.0);
      double tEnd = 0;
      for (int b = 0; b < nb; ++b) { const int b0 = b * g_rows; double t = 0;
        for (int s = 0; s < ns; ++s) {
        bool any = false, has = false;
        for (int r = 0; r < g_rows && b0 + r < LB; ++r) { const int u = s - r; if (u < 0 || u >= LS) continue; has = true;
          int x = tr ? b0 + r : u, y = tr ? u : b0 + r; if (!forward) { x = W - 1 - x; y = H - 1 - y; }
          if (!gate[size_t(y) * W + x]) continue;
          const int ax = x + sx, by = y + sy;
          if ((ax >= 0 && ax < W && bad[size_t(y) * W + ax]) || (by >= 0 && by < H && bad[size_t(by) * W + x])) any = true; }
        if (has) { ++cells; slow += any; }
        double start = t; if (b > 0) { const double dep = Tprev[std::min(s + g_rows - 1, ns - 1)] + cHand; if (dep > start) start = dep; }
        t = start + (any ? cSlow : cFast); Tcur[s] = t; }
        Tprev.swap(Tcur); tEnd = t; }
      g_model[K] += tEnd; if (K == 2) g_model[0] += double(ns + (g_rows + 0.6) * (nb - 1)) * cSlow;
      fprintf(stderr, "   level %d %s K=%d: %.3f%% pixels not final, %.2f%% of band-steps slow\n", level, forward ? "fwd" : "bwd", K, 100.0 * wrong / (double)(gated ? gated : 1), 100.0 * slow / (double)cells);
    }
    if (rounds <= 12 || rounds % 50 == 0) fprintf(stderr, "   level %d %s round %d: %lld changed\n", level, forward ? "fwd" : "bwd", rounds, nch);
    if (nch == 0) break;
    if (rounds > W + H) { fprintf(stderr, "did not converge?!\n"); break; }
  }
  { const bool tr = W < H; const int LB = tr ? W : H, LS = tr ? H : W; const int nb = (LB + g_rows - 1) / g_rows, ns = LS + g_rows - 1;
    for (int K : {2, 4, 8, 16, 32, 64}) if (rounds < K) g_model[K] += ns * g_fast + (nb - 1) * (g_rows * g_fast + g_hand); }   // converged before K rounds: every step fast
  bool ok = true;
  for (size_t i = 0; i < cur.d.size(); ++i) if (!same(cur.d[i], expect.d[i])) { ok = false; break; }
  if (!ok) fprintf(stderr, "MISMATCH vs sequential sweep at level %d\n", level);
  g_stats.push_back({level, W, H, forward ? 1 : 0, rounds, evals, gated, 0});
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s a.raw b.raw cols rows [maxPct]\n", argv[0]); return 2; }
  const int cols = atoi(argv[3]), rows = atoi(argv[4]); const int maxPct = argc > 5 ? atoi(argv[5]) : 0;
  if (argc > 6) g_fast = atof(argv[6]);
  if (argc > 7) g_rows = atoi(argv[7]);
  ImgU8 a(cols, rows, 4), b(cols, rows, 4);
  FILE* f = fopen(argv[1], "rb"); if (!f || fread(a.d.data(), 1, a.d.size(), f) != a.d.size()) return 3; fclose(f);
  f = fopen(argv[2], "rb"); if (!f || fread(b.d.data(), 1, b.d.size(), f) != b.d.size()) return 3; fclose(f);
  Params p; p.maxPercentage = maxPct;
  ImgF I0, I1, alpha0, alpha1;
  preprocess(a, p, I0, alpha0); preprocess(b, p, I1, alpha1);
  auto pI0 = buildPyramid(I0, p.pyrScaleFactor), pI1 = buildPyramid(I1, p.pyrScaleFactor), pA0 = buildPyramid(alpha0, p.pyrScaleFactor),
       pA1 = buildPyramid(alpha1, p.pyrScaleFactor);
  ImgF flow;
  for (int level = int(pI0.size()) - 1; level >= 0; --level) {
    const ImgF &i0 = pI0[level], &i1 = pI1[level], &a0 = pA0[level], &a1 = pA1[level];
    ImgF I0x, I0y, I1x, I1y; gradients(i0, I0x, I0y); gradients(i1, I1x, I1y);
    if (flow.empty()) { flow = ImgF(i0.w, i0.h, 2); if (maxPct > 0) adjustInitialFlow(i0, i1, a0, a1, flow, LEFT, maxPct); }
    ImgF blurred; gaussian_blur_f32(flow, blurred, 15, 8.0);
    LevelCtx L{&I0x, &I0y, &I1x, &I1y, &blurred, &p, i0.w};
    ImgF before = flow; sweep(L, a0, a1, flow, true); jacobi(L, a0, a1, before, flow, true, level);
    median5(flow, flow);
    before = flow; sweep(L, a0, a1, flow, false); jacobi(L, a0, a1, before, flow, false, level);
    median5(flow, flow);
    lowAlphaFlowDiffusion(a0, a1, flow);
    if (level > 0) { ImgF up; resize_cubic_f32(flow, up, pI0[level - 1].w, pI0[level - 1].h); const float s = 1.0f / p.pyrScaleFactor; for (auto& v : up.d) v = v * s + 0.0f; flow = std::move(up); }
  }
  printf("level  W    H   dir rounds  W+H-1  evals/gated\n");
  long long R = 0, S = 0;
  for (auto& s : g_stats) { printf("%3d %5d %5d %s %6d %6d  %.2f\n", s.level, s.w, s.h, s.fwd ? "fwd" : "bwd", s.rounds, s.w + s.h - 1, double(s.evals) / double(s.gated ? s.gated : 1)); R += s.rounds; S += s.w + s.h - 1; }
  for (auto& kv : g_model) printf("model: K=%d  sum of sweep times %.3f ms (2.4 GHz; fast %g slow %g handoff %g cycles)\n", kv.first, kv.second / 2.4e6, g_fast, g_slow, g_hand);
  printf("total rounds %lld vs wavefront steps %lld\n", R, S);
  return 0;
}
