// K6r: the raster sweep (CPU/PixFlow.hpp:315-324 forward, :328-337 backward) as an EVENT-DRIVEN relaxation -- experiment behind
// PANOFLOW_SWEEP=3, included by kernels_sweep2.hip (shares its exact error function).
//
// The sweep is a recurrence on a DAG: out(p) = F(p; out(left p), out(top p)) with everything else frozen, so its result is the
// UNIQUE fixed point of "every pixel satisfies its own equation".  Any evaluation order reaches it as long as a pixel is
// re-evaluated after each change of one of its two inputs.  The wavefront kernel pays one step per anti-diagonal (W+H steps
// per sweep); but F forwards a neighbour's value only when the proposal is adopted, so real dependency chains are ~4x shorter
// than the anti-diagonal count (tests/micro/jacobi_rounds.cpp).  Here a workgroup owns a 48x48 tile with ALL its state in LDS
// (current values, per-pixel constants, the gathered-plane window) and runs rounds over a work list of "dirty" pixels: evaluate,
// and if the value changed push the right/down successors.  Tiles exchange their edge values through 8-byte granules in HBM
// (data is its own flag, cdna_hip_programming.md G16/R2) WITHOUT waiting for each other; only termination is ordered: a tile
// is final once its left and top tiles are final, it has re-read their granules, and its list is empty.  Tiles are taken by
// ticket in row-major order, so a tile only ever waits for tiles that already started (no residency assumption).
//
// Bit-exactness: same F as the wavefront kernel (d_error2g's exactly rounded forms), same fixed point.

namespace {

constexpr int kT = 48;                 // tile edge (pixels)
constexpr int kTP = kT * kT;           // 2304 = 9 x 256
#ifndef PF_RX_THREADS
#define PF_RX_THREADS 1024
#endif
constexpr int kRXThreads = PF_RX_THREADS;
constexpr int kRXRad = 8;
constexpr int kRXW = kT + 2 * kRXRad + 1;   // 65: window of the gathered plane (odd stride: conflict-free rows)
constexpr int kPollEvery = 4;          // rounds between two looks at the neighbouring tiles (power of two)
constexpr int kGranPerTile = 104;      // [0,48) right column, [48,96) bottom row, [96] finished flag

struct RXSmem {
  unsigned long long P[kTP];           // current value of every pixel of the tile (sweep-order local index v*48+u)
  float2 win[kRXW * kRXW];             // (I1x,I1y) texels around the tile, image coordinates
  float2 cC[kTP], cG[kTP], cB[kTP];    // incoming flow, (I0x,I0y), blurred flow
  float cE[3][kTP];                    // E(C), E(C+dx), E(C+dy)
  unsigned long long haloL[kT], haloT[kT];
  unsigned short list[2][kTP];
  unsigned bits[2][kTP / 32];
  unsigned gate[kTP / 32];
  int n[3];
  int tile, finL, finT, abort;
  long long deadline;
};

__device__ __forceinline__ float d_error_rx(const float2* __restrict__ g1, const float2* win, int wx0, int wy0, int W, float wm2, float hm2, float fW,
                                            float rW, int x, int y, float i0x, float i0y, float bx, float by, float fdx, float fdy) {
  const float matchX = float(x) + fdx, matchY = float(y) + fdy;
  float cx = (0.0f < matchX) ? matchX : 0.0f; cx = (cx < wm2) ? cx : wm2;
  float cy = (0.0f < matchY) ? matchY : 0.0f; cy = (cy < hm2) ? cy : hm2;
  const int x0 = int(cx), y0 = int(cy);
  const float xR = cx - float(x0), yR = cy - float(y0);
  const int lx = x0 - wx0, ly = y0 - wy0;
  float2 q00, q10, q01, q11;
  if ((unsigned)lx < (unsigned)(kRXW - 1) && (unsigned)ly < (unsigned)(kRXW - 1)) {
    const float2* p = win + ly * kRXW + lx;
    q00 = p[0]; q10 = p[1]; q01 = p[kRXW]; q11 = p[kRXW + 1];
  } else {
    const float2* p = g1 + size_t(y0) * W + x0;
    q00 = p[0]; q10 = p[1]; q01 = p[W]; q11 = p[W + 1];
  }
  float i1x, i1y;
  {
    const float f00 = q00.x, f10 = q10.x, f01 = q01.x, f11 = q11.x;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1x = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  {
    const float f00 = q00.y, f10 = q10.y, f01 = q01.y, f11 = q11.y;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1y = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  const float dfx = bx - fdx, dfy = by - fdy;
  const float s2 = dfx * dfx + dfy * dfy;
  const float d2 = (i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y);
  const float av = kVerticalRegularizationCoef * fabsf(fdy), ah = kHorizontalRegularizationCoef * fabsf(fdx);
  if (fast_range_ok(s2, d2, av, ah))
    return sqrt_core(d2) + sqrt_core(s2) * kSmoothnessCoef + div_core(av, fW, rW) + div_core(ah, fW, rW);
  return sqrtf(d2) + sqrtf(s2) * kSmoothnessCoef + av / fW + ah / fW;
}

// Two evaluations of the error function at once (same pixel, two candidate flows): straight-line code in the common case, so the
// two dependency chains interleave; the rare cases (a texel outside the LDS window, an operand outside the exact range of the
// cheap sqrt/division forms) are fixed up under wave-level branches.  Same operations per value as d_error2g.
struct RXPix { int x, y; float i0x, i0y, bx, by; };
__device__ __forceinline__ void d_error_rx2(const float2* __restrict__ g1, const float2* win, int wx0, int wy0, int W, float wm2, float hm2, float fW,
                                            float rW, const RXPix& p, float ax, float ay, float bx_, float by_, float& ea, float& eb) {
  float cxa = float(p.x) + ax, cya = float(p.y) + ay, cxb = float(p.x) + bx_, cyb = float(p.y) + by_;
  cxa = (0.0f < cxa) ? cxa : 0.0f; cxa = (cxa < wm2) ? cxa : wm2; cya = (0.0f < cya) ? cya : 0.0f; cya = (cya < hm2) ? cya : hm2;
  cxb = (0.0f < cxb) ? cxb : 0.0f; cxb = (cxb < wm2) ? cxb : wm2; cyb = (0.0f < cyb) ? cyb : 0.0f; cyb = (cyb < hm2) ? cyb : hm2;
  const int xa = int(cxa), ya = int(cya), xb = int(cxb), yb = int(cyb);
  const float xRa = cxa - float(xa), yRa = cya - float(ya), xRb = cxb - float(xb), yRb = cyb - float(yb);
  const int lxa = xa - wx0, lya = ya - wy0, lxb = xb - wx0, lyb = yb - wy0;
  const bool ina = (unsigned)lxa < (unsigned)(kRXW - 1) && (unsigned)lya < (unsigned)(kRXW - 1);
  const bool inb = (unsigned)lxb < (unsigned)(kRXW - 1) && (unsigned)lyb < (unsigned)(kRXW - 1);
  const float2* pa = win + (ina ? lya * kRXW + lxa : 0);
  const float2* pb = win + (inb ? lyb * kRXW + lxb : 0);
  float2 a00 = pa[0], a10 = pa[1], a01 = pa[kRXW], a11 = pa[kRXW + 1];
  float2 b00 = pb[0], b10 = pb[1], b01 = pb[kRXW], b11 = pb[kRXW + 1];
  if (!(ina && inb)) {
    if (!ina) { const float2* q = g1 + size_t(ya) * W + xa; a00 = q[0]; a10 = q[1]; a01 = q[W]; a11 = q[W + 1]; }
    if (!inb) { const float2* q = g1 + size_t(yb) * W + xb; b00 = q[0]; b10 = q[1]; b01 = q[W]; b11 = q[W + 1]; }
  }
  const float i1xa = a00.x + (a10.x - a00.x) * xRa + (a01.x - a00.x) * yRa + (a00.x + a11.x - a10.x - a01.x) * xRa * yRa;
  const float i1ya = a00.y + (a10.y - a00.y) * xRa + (a01.y - a00.y) * yRa + (a00.y + a11.y - a10.y - a01.y) * xRa * yRa;
  const float i1xb = b00.x + (b10.x - b00.x) * xRb + (b01.x - b00.x) * yRb + (b00.x + b11.x - b10.x - b01.x) * xRb * yRb;
  const float i1yb = b00.y + (b10.y - b00.y) * xRb + (b01.y - b00.y) * yRb + (b00.y + b11.y - b10.y - b01.y) * xRb * yRb;
  const float dfxa = p.bx - ax, dfya = p.by - ay, dfxb = p.bx - bx_, dfyb = p.by - by_;
  const float s2a = dfxa * dfxa + dfya * dfya, s2b = dfxb * dfxb + dfyb * dfyb;
  const float d2a = (p.i0x - i1xa) * (p.i0x - i1xa) + (p.i0y - i1ya) * (p.i0y - i1ya);
  const float d2b = (p.i0x - i1xb) * (p.i0x - i1xb) + (p.i0y - i1yb) * (p.i0y - i1yb);
  const float ava = kVerticalRegularizationCoef * fabsf(ay), aha = kHorizontalRegularizationCoef * fabsf(ax);
  const float avb = kVerticalRegularizationCoef * fabsf(by_), ahb = kHorizontalRegularizationCoef * fabsf(bx_);
  ea = sqrt_core(d2a) + sqrt_core(s2a) * kSmoothnessCoef + div_core(ava, fW, rW) + div_core(aha, fW, rW);
  eb = sqrt_core(d2b) + sqrt_core(s2b) * kSmoothnessCoef + div_core(avb, fW, rW) + div_core(ahb, fW, rW);
  const bool oka = fast_range_ok(s2a, d2a, ava, aha), okb = fast_range_ok(s2b, d2b, avb, ahb);
  if (!(oka && okb)) {
    if (!oka) ea = sqrtf(d2a) + sqrtf(s2a) * kSmoothnessCoef + ava / fW + aha / fW;
    if (!okb) eb = sqrtf(d2b) + sqrtf(s2b) * kSmoothnessCoef + avb / fW + ahb / fW;
  }
}

__device__ __forceinline__ void rx_push(RXSmem& sm, int pn, int cn, int q) {
  const unsigned m = 1u << (q & 31);
  const unsigned old = atomicOr(&sm.bits[pn][q >> 5], m);
  if (!(old & m)) {
    const int pos = atomicAdd(&sm.n[cn], 1);
    sm.list[pn][pos] = (unsigned short)q;
  }
}

}  // namespace

#ifdef PF_RX_STATS
__device__ long long g_rxdbg[4096 * 16];
__device__ int g_rxdbg_sel[2];
#endif

template <bool FWD>
__global__ __launch_bounds__(kRXThreads) void k_sweep_relax(const float2* __restrict__ g0, const float2* __restrict__ g1, const float2* __restrict__ blurred,
                                                            const uint8_t* __restrict__ gate, float2* __restrict__ flow,
                                                            unsigned long long* __restrict__ gran, int* __restrict__ ctrl, int W, int H, int U0, int V0,
                                                            int U1, int V1, int ntx, int nty, float rW, long long budgetTicks) {
  __shared__ RXSmem sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W);
  const int ntiles = ntx * nty;
  if (tid == 0) { sm.deadline = (long long)wall_clock64() + budgetTicks; sm.abort = 0; }

  for (;;) {
    __syncthreads();   // the previous tile's LDS state is dead
    if (tid == 0) {
      sm.tile = atomicAdd(&ctrl[0], 1);
      sm.n[0] = 0; sm.n[1] = 0; sm.n[2] = 0; sm.finL = 0; sm.finT = 0;
    }
    if (tid < kTP / 32) { sm.bits[0][tid] = 0; sm.bits[1][tid] = 0; }
    __syncthreads();
    const int t = sm.tile;
    if (t >= ntiles || sm.abort) break;
#ifdef PF_RX_STATS
    const long long stT0 = wall_clock64();
    long long stTLastBusy = 0, stTFin = 0;
    int stBusy = 0, stEvals = 0, stLastBusy = -1, stDenseT = 0, stDenseN = 0, stSparseT = 0, stSparseN = 0, stSparseE = 0;
#endif
    const int ti = t % ntx, tj = t / ntx;
    const int Ub = U0 + ti * kT, Vb = V0 + tj * kT;
    const int tw = min(kT, U1 - Ub), th = min(kT, V1 - Vb);
    const int xlo = FWD ? Ub : W - Ub - tw, ylo = FWD ? Vb : H - Vb - th;   // image rectangle of the tile
    const int wx0 = xlo - kRXRad, wy0 = ylo - kRXRad;

    const bool hasL = ti > 0, hasT = tj > 0;
    const bool pubR = ti + 1 < ntx, pubB = tj + 1 < nty;
    unsigned long long* gMine = gran + size_t(t) * kGranPerTile;

    // ---- tile state: incoming flow, gate bits, initial work list (every gated pixel) ----
    bool anyGated = false;
    for (int q = tid; q < kTP; q += kRXThreads) {   // (whole waves: kTP and kRXThreads are multiples of 64)
      const int u = q % kT, v = q / kT;
      const bool valid = u < tw && v < th;
      const int cu = Ub + u, cv = Vb + v;
      const int x = FWD ? cu : W - 1 - cu, y = FWD ? cv : H - 1 - cv;
      float2 C = make_float2(0.f, 0.f);
      bool gated = false;
      if (valid) {
        const size_t idx = size_t(y) * W + x;
        C = flow[idx];
        gated = gate[idx] != 0;
        if (gated) { sm.cG[q] = g0[idx]; sm.cB[q] = blurred[idx]; }
      }
      sm.cC[q] = C;
      sm.P[q] = pack2(C);
      const unsigned long long m = __ballot(gated);
      if (lane == 0) { sm.gate[(q >> 5)] = (unsigned)m; sm.gate[(q >> 5) + 1] = (unsigned)(m >> 32); }
      if (m) {
        anyGated = true;
        int base = 0;
        if (lane == 0) base = atomicAdd(&sm.n[0], __popcll(m));
        base = __shfl(base, 0);
        if (gated) sm.list[0][base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)q;
        if (lane == 0) { sm.bits[0][q >> 5] = (unsigned)m; sm.bits[0][(q >> 5) + 1] = (unsigned)(m >> 32); }
      }
    }
    // first guess for the neighbouring tiles' edge values: their incoming flow (static and final when the neighbour is outside the window)
    if (tid < kT) {
      const int v = tid, cu = Ub - 1, cv = Vb + v;
      unsigned long long h = 0;
      if (v < th && cu >= 0) { const int x = FWD ? cu : W - 1 - cu, y = FWD ? cv : H - 1 - cv; h = pack2(flow[size_t(y) * W + x]); }
      sm.haloL[v] = h;
    } else if (tid >= 64 && tid < 64 + kT) {
      const int u = tid - 64, cu = Ub + u, cv = Vb - 1;
      unsigned long long h = 0;
      if (u < tw && cv >= 0) { const int x = FWD ? cu : W - 1 - cu, y = FWD ? cv : H - 1 - cv; h = pack2(flow[size_t(y) * W + x]); }
      sm.haloT[u] = h;
    }
    const bool tileGated = __syncthreads_or(anyGated);
    if (tileGated) {
      for (int i = tid; i < kRXW * kRXW; i += kRXThreads) {
        int wx = wx0 + i % kRXW, wy = wy0 + i / kRXW;
        wx = wx < 0 ? 0 : (wx > W - 1 ? W - 1 : wx); wy = wy < 0 ? 0 : (wy > H - 1 ? H - 1 : wy);
        sm.win[i] = g1[size_t(wy) * W + wx];
      }
      __syncthreads();
      // own-flow terms of every gated pixel; first value = the update without neighbours (what most pixels end up with)
      const int n0 = sm.n[0];
      for (int i = tid; i < n0; i += kRXThreads) {
        const int q = sm.list[0][i], u = q % kT, v = q / kT;
        const int cu = Ub + u, cv = Vb + v;
        const float2 C = sm.cC[q], g = sm.cG[q], bl = sm.cB[q];
        const RXPix px = {FWD ? cu : W - 1 - cu, FWD ? cv : H - 1 - cv, g.x, g.y, bl.x, bl.y};
        float e0, e1, e2, e3;
        d_error_rx2(g1, sm.win, wx0, wy0, W, wm2, hm2, fW, rW, px, C.x, C.y, C.x + kGradEpsilon, C.y + 0.0f, e0, e1);
        d_error_rx2(g1, sm.win, wx0, wy0, W, wm2, hm2, fW, rW, px, C.x + 0.0f, C.y + kGradEpsilon, C.x, C.y, e2, e3);
        sm.cE[0][q] = e0; sm.cE[1][q] = e1; sm.cE[2][q] = e2;
        const float gx = (e1 - e0) / kGradEpsilon, gy = (e2 - e0) / kGradEpsilon;
        const unsigned long long d = pack2(make_float2(C.x - kGradientStepSize * gx, C.y - kGradientStepSize * gy));
        sm.P[q] = d;
      }
    }
    __syncthreads();

    // ---- rounds ----
    // this thread's polling role: 0 none, 1 left tile's right column, 2 top tile's bottom row, 3 left flag, 4 top flag
    int role = 0;
    const unsigned long long* src = nullptr;
    if (hasL && tid < th) { role = 1; src = gran + size_t(t - 1) * kGranPerTile + tid; }
    else if (hasT && tid >= 64 && tid < 64 + tw) { role = 2; src = gran + size_t(t - ntx) * kGranPerTile + kT + (tid - 64); }
    else if (hasL && tid == 128) { role = 3; src = gran + size_t(t - 1) * kGranPerTile + 2 * kT; }
    else if (hasT && tid == 129) { role = 4; src = gran + size_t(t - ntx) * kGranPerTile + 2 * kT; }
#ifdef PF_RX_STATS
    const long long stT1 = wall_clock64();
#endif
    // publisher threads: ONE writer per granule (two waves storing to one address are not ordered), one round behind the values
    int pubQ = -1;
    unsigned long long* pubDst = nullptr;
    if (kRXThreads >= 512) {
      if (pubR && tid >= 256 && tid < 256 + th) { pubQ = (tid - 256) * kT + tw - 1; pubDst = gMine + (tid - 256); }
      else if (pubB && tid >= 320 && tid < 320 + tw) { pubQ = (th - 1) * kT + (tid - 320); pubDst = gMine + kT + (tid - 320); }
    } else {
      if (pubR && tid >= 136 && tid < 136 + th) { pubQ = (tid - 136) * kT + tw - 1; pubDst = gMine + (tid - 136); }
      else if (pubB && tid >= 192 && tid < 192 + tw) { pubQ = (th - 1) * kT + (tid - 192); pubDst = gMine + kT + (tid - 192); }
    }
    unsigned long long lastPub = pubQ >= 0 ? pack2(sm.cC[pubQ]) : 0ull;   // what the neighbour assumes before it hears anything
    unsigned long long pend = kNotReady;
    int finSeen = (hasL || hasT) ? 0x3fffffff : -2;
    int idle = 0;
    for (int r = 0;; ++r) {
      const int pc = r & 1, pn = pc ^ 1, c = r % 3, cn = (r + 1) % 3, cz = (r + 2) % 3;
      // (a) look at the predecessors' edge granules / finished flags: issued here, absorbed after the evaluation (the load's latency
      // hides behind it)
      // -- one look every kPollEvery rounds, absorbed kPollEvery-1 rounds later: a sparse round is shorter than the load's latency
      if (role && (r & (kPollEvery - 1)) == 0) pend = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pubQ >= 0) { const unsigned long long cur = sm.P[pubQ]; if (cur != lastPub) { __hip_atomic_store(pubDst, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); lastPub = cur; } }
      // (c) evaluate the current list
      const int n = sm.n[c];
#ifdef PF_RX_STATS
      if (n > 0) { ++stBusy; stEvals += n; stLastBusy = r; }
      const long long stR0 = wall_clock64();
#endif
      if (n <= kRXThreads / 8) {
        // few dirty pixels: 8 lanes per pixel, the six error evaluations of one update side by side (as the wavefront kernel does)
        const int gi = tid >> 3, k = tid & 7;
        if (((tid & ~63) >> 3) < n) {   // wave-uniform
          const bool act = gi < n;
          const int q = act ? sm.list[pc][gi] : 0, u = q % kT, v = q / kT;
          const int cu = Ub + u, cv = Vb + v;
          const bool okL = cu > 0, okT = cv > 0;
          const float2 C = sm.cC[q], g = sm.cG[q], bl = sm.cB[q];
          const float2 L = okL ? unpack2(u > 0 ? sm.P[q - 1] : sm.haloL[v]) : C;
          const float2 T = okT ? unpack2(v > 0 ? sm.P[q - kT] : sm.haloT[u]) : C;
          const float2 base = k < 4 ? L : T;   // lane roles of select_step: 0-2 the L proposal (+0, +dx, +dy), 4-6 the T proposal
          const int kk = k & 3;
          const float fx = base.x + (kk == 1 ? kGradEpsilon : 0.0f), fy = base.y + (kk == 2 ? kGradEpsilon : 0.0f);
          float e = 0.0f;
          if (act && kk < 3) e = d_error_rx(g1, sm.win, wx0, wy0, W, wm2, hm2, fW, rW, FWD ? cu : W - 1 - cu, FWD ? cv : H - 1 - cv, g.x, g.y, bl.x, bl.y, fx, fy);
          int emin = 0; float vmax = 0.0f;
          const float2 o = select_step<false, false>(e, sm.cE[0][q], sm.cE[0][q], own_gradient_step(C, sm.cE[0][q], sm.cE[1][q], sm.cE[2][q], kGradientStepSize), base, okL, okT, 0.0f, kGradientStepSize, emin, vmax);   // (this rejected experiment knows the factory's presets only: pf_set_solver_params refuses other sets for sweep_impl 3)
          if (act && k == 0) {
            atomicAnd(&sm.bits[pc][q >> 5], ~(1u << (q & 31)));
            const unsigned long long nv = pack2(o);
            if (nv != sm.P[q]) {
              sm.P[q] = nv;
              if (u + 1 < tw && (sm.gate[(q + 1) >> 5] >> ((q + 1) & 31) & 1)) rx_push(sm, pn, cn, q + 1);
              if (v + 1 < th && (sm.gate[(q + kT) >> 5] >> ((q + kT) & 31) & 1)) rx_push(sm, pn, cn, q + kT);
            }
          }
        }
      } else
      for (int i = tid; i < n; i += kRXThreads) {
        const int q = sm.list[pc][i], u = q % kT, v = q / kT;
        atomicAnd(&sm.bits[pc][q >> 5], ~(1u << (q & 31)));
        const int cu = Ub + u, cv = Vb + v;
        const bool okL = cu > 0, okT = cv > 0;
        const float2 C = sm.cC[q], g = sm.cG[q], bl = sm.cB[q];
        const float2 L = okL ? unpack2(u > 0 ? sm.P[q - 1] : sm.haloL[v]) : C;
        const float2 T = okT ? unpack2(v > 0 ? sm.P[q - kT] : sm.haloT[u]) : C;
        const RXPix px = {FWD ? cu : W - 1 - cu, FWD ? cv : H - 1 - cv, g.x, g.y, bl.x, bl.y};
        float cur = sm.cE[0][q], ex = sm.cE[1][q], ey = sm.cE[2][q];
        float2 f = C;
        bool adopted = false;
        float eL, eT;
        d_error_rx2(g1, sm.win, wx0, wy0, W, wm2, hm2, fW, rW, px, L.x + 0.0f, L.y + 0.0f, T.x + 0.0f, T.y + 0.0f, eL, eT);
        if (okL && eL < cur) { cur = eL; f = L; adopted = true; }
        if (okT && eT < cur) { cur = eT; f = T; adopted = true; }
        if (adopted) d_error_rx2(g1, sm.win, wx0, wy0, W, wm2, hm2, fW, rW, px, f.x + kGradEpsilon, f.y + 0.0f, f.x + 0.0f, f.y + kGradEpsilon, ex, ey);
        const float gx = (ex - cur) / kGradEpsilon, gy = (ey - cur) / kGradEpsilon;
        const unsigned long long nv = pack2(make_float2(f.x - kGradientStepSize * gx, f.y - kGradientStepSize * gy));
        if (nv != sm.P[q]) {
          sm.P[q] = nv;
          if (u + 1 < tw && (sm.gate[(q + 1) >> 5] >> ((q + 1) & 31) & 1)) rx_push(sm, pn, cn, q + 1);
          if (v + 1 < th && (sm.gate[(q + kT) >> 5] >> ((q + kT) & 31) & 1)) rx_push(sm, pn, cn, q + kT);
        }
      }
      if (role && (r & (kPollEvery - 1)) == kPollEvery - 1 && pend != kNotReady) {
        if (role == 1) { if (pend != sm.haloL[tid]) { sm.haloL[tid] = pend; const int q = tid * kT; if (sm.gate[q >> 5] >> (q & 31) & 1) rx_push(sm, pn, cn, q); } }
        else if (role == 2) { const int u = tid - 64; if (pend != sm.haloT[u]) { sm.haloT[u] = pend; if (sm.gate[u >> 5] >> (u & 31) & 1) rx_push(sm, pn, cn, u); } }
        else if (pend == 1ull) { if (role == 3) sm.finL = 1; else sm.finT = 1; }
      }
      if (tid == 0) sm.n[cz] = 0;
      __syncthreads();
#ifdef PF_RX_STATS
      { const int dt = int((long long)wall_clock64() - stR0); if (n > kRXThreads) { stDenseT += dt; ++stDenseN; } else if (n > 0) { stSparseT += dt; ++stSparseN; stSparseE += n; } if (n > 0) stTLastBusy = wall_clock64(); }
#endif
      // (d) final?  The predecessors' flags were absorbed in round finSeen; polls issued after that round's barrier (round
      // finSeen+1, a multiple of kPollEvery) see their final edge values and are absorbed in round finSeen+kPollEvery.
      const int nNext = sm.n[cn];
      const bool fin = (!hasL || sm.finL) && (!hasT || sm.finT);
      if (fin && finSeen == 0x3fffffff) {
        finSeen = r;
#ifdef PF_RX_STATS
        stTFin = wall_clock64();
#endif
      }   // (no acquire fence: every later look at the predecessors is an agent-scope atomic load)
      if (nNext == 0) {
        if (r >= finSeen + kPollEvery) {
#ifdef PF_RX_STATS
          if (tid == 0 && W == g_rxdbg_sel[0] && (int)FWD == g_rxdbg_sel[1] && t < 4096) {
            long long* d = g_rxdbg + size_t(t) * 16;
            d[0] = stT0; d[1] = stT1; d[2] = stTLastBusy; d[3] = stTFin; d[4] = wall_clock64(); d[5] = r + 1; d[6] = stBusy; d[7] = stEvals; d[8] = stLastBusy;
            d[9] = finSeen; d[10] = stDenseN; d[11] = stDenseT; d[12] = stSparseN; d[13] = stSparseT; d[14] = stSparseE; d[15] = ntx;
          }
#endif
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        if ((++idle & 255) == 0 && (long long)wall_clock64() > sm.deadline) { if (tid == 0) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } break; }
      }
    }
    // ---- the tile is final: every published edge value is in memory before the flag; then the flow plane ----
    if (pubQ >= 0) { const unsigned long long cur = sm.P[pubQ]; if (cur != lastPub) __hip_atomic_store(pubDst, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(gMine + 2 * kT, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (relaxed: every edge store was drained by s_waitcnt vmcnt(0) + the barrier above)
    if (tileGated) {
      for (int q = tid; q < kTP; q += kRXThreads) {
        const int u = q % kT, v = q / kT;
        if (u < tw && v < th && (sm.gate[q >> 5] >> (q & 31) & 1)) {
          const int cu = Ub + u, cv = Vb + v;
          const int x = FWD ? cu : W - 1 - cu, y = FWD ? cv : H - 1 - cv;
          flow[size_t(y) * W + x] = unpack2(sm.P[q]);
        }
      }
    }
  }
}

// ---- host side ----
size_t sweep_relax_boundary_elems(int W, int H) { return size_t((W + kT - 1) / kT) * size_t((H + kT - 1) / kT) * kGranPerTile; }

bool launch_sweep_relax(hipStream_t st, const SweepArgs& a) {
  int x0 = a.ax0 < 0 ? 0 : a.ax0, y0 = a.ay0 < 0 ? 0 : a.ay0, x1 = a.ax1 > a.W ? a.W : a.ax1, y1 = a.ay1 > a.H ? a.H : a.ay1;
  if (x1 <= x0 || y1 <= y0) return false;
  const int U0 = a.forward ? x0 : a.W - x1, U1 = a.forward ? x1 : a.W - x0, V0 = a.forward ? y0 : a.H - y1, V1 = a.forward ? y1 : a.H - y0;
  const int ntx = (U1 - U0 + kT - 1) / kT, nty = (V1 - V0 + kT - 1) / kT;
  static const int cap = [] { const char* e = getenv("PANOFLOW_RX_WGS"); const int v = e ? atoi(e) : 128; return v < 1 ? 1 : v; }();
  const int ntiles = ntx * nty, grid = ntiles < cap ? ntiles : cap;
  const float rW = (float)(1.0 / (double)(float)a.W);
  const long long budget = 200000000ll + 100000ll * (long long)(a.W + a.H);
  if (a.forward)
    hipExtLaunchKernelGGL((k_sweep_relax<true>), dim3(grid), dim3(kRXThreads), 0, st, a.ev_start, a.ev_stop, 0, a.g0, a.g1, a.blurred, a.gate, a.flow, a.boundary,
                          a.ctrl, a.W, a.H, U0, V0, U1, V1, ntx, nty, rW, budget);
  else
    hipExtLaunchKernelGGL((k_sweep_relax<false>), dim3(grid), dim3(kRXThreads), 0, st, a.ev_start, a.ev_stop, 0, a.g0, a.g1, a.blurred, a.gate, a.flow, a.boundary,
                          a.ctrl, a.W, a.H, U0, V0, U1, V1, ntx, nty, rW, budget);
  return true;
}

#ifdef PF_RX_STATS
}  // namespace pf
// per-tile trace of one selected sweep launch (level width W, direction): tests/micro/rx_trace.py
extern "C" int pf_debug_rx_select(int W, int fwd) { int v[2] = {W, fwd}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pf::g_rxdbg_sel), v, sizeof v); }
extern "C" int pf_debug_rx_dump(long long* out, int ntiles) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_rxdbg), size_t(ntiles) * 16 * sizeof(long long)); }
namespace pf {
#endif
