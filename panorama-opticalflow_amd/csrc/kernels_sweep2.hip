// K6 v2: exact Gauss-Seidel sweeps (CPU/PixFlow.hpp:315-337) restructured for CDNA4 latency.
//
// The sweep is dependency-latency bound (W+H-1 anti-diagonal steps per sweep), so the design minimises
// the time of ONE step instead of bytes moved:
//   * 8 lanes per pixel.  Everything about a pixel that does not depend on its neighbours -- E(C) and
//     the gradient step its own incoming flow C would take, rC -- is computed by a fully parallel prepass
//     (k_sweep_prep) which also packs the static per-pixel inputs into records (32 bytes in memory, 48 in LDS) laid out in the
//     order the wavefront consumes them.  In the sequential kernel six lanes evaluate
//     E(L), E(L+dx), E(L+dy), E(T), E(T+dx), E(T+dy) for the two proposals (L = previous column,
//     T = previous row) AT THE SAME TIME: one gather round per step instead of five dependent ones --
//     and each proposal takes its own gradient step in its own lane (the same instructions for all lanes)
//     before the selection, which follows the reference order (current, then L, then T, strict '<') and
//     only picks one of three finished results.  Same arithmetic, same order => bit-identical.
//   * one compute wave = a band of 8 rows (lane = 8*row + role).  Four compute waves (one per SIMD)
//     + seven helper waves (four loaders, publisher, poller, drainer) form a workgroup = 32 rows.  Rows inside a
//     wave hand their result to the next row by DPP moves; waves inside a workgroup through the LDS result
//     ring plus a step counter (producer: value, then counter; consumer: counter, then value, both one step
//     ahead of use -- the readiness test is two scalar instructions); workgroups through 8-byte
//     data-is-flag granules in HBM (agent-scope relaxed atomics; cdna_hip_programming.md G16/R2).
//   * compute waves never touch HBM except for the rare bilinear gather outside the LDS window: the helper
//     waves stream records and window texels HBM->LDS ahead of the wavefront, drain results LDS->HBM,
//     publish the granules and poll the previous workgroup's granules, so the long-latency traffic sits in
//     THEIR in-order memory queues, not in the compute waves'.
//   * a lone wave issues ONE instruction per ~4.46 cycles here whatever it is -- dependent or not, vector, scalar, DPP or an s_nop state
//     (tests/micro/slot_model.py, profiles/r06_slot_model.txt) -- so the step time is its count of issue SLOTS: 103 per step since round 6
//     (packed fp32 math, exact cheap forms of sqrt and division -- exact_forms.hpp --, one range guard per step, no per-step address arithmetic
//     that a loop-carried register or an immediate can replace, 64-bit DPP moves for the proposals, the across proposal's moves folded into
//     v_cndmask_b32_dpp / v_sub_f32_dpp, the gather torus addressed by image coordinates), and the product's assembly goes through
//     tools/asm_sched.py, which re-derives the wait states and fills them with independent instructions (hipcc leaves 7-9 per step, the pass 0-1).
//     DESIGN.md 3.2-3.3 and docs/history.md have the history (154 -> 103), profiles/r06_sweep_step_isa.txt the listing.
//   * the gather window in LDS follows the flow (round 5): centred per chunk of 8 steps on pixel + the rounded blurred
//     flow, a torus addressed by the texel's image coordinates (the loader of k_sweep2 below).
// Workgroups take their band index from an atomic ticket (a band only waits for bands already
// running), every spin is bounded and raises ctrl[1] instead of hanging.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <stddef.h>
#include <algorithm>

#include "pf_common.hpp"
#include "sweep_window.hpp"
#include "exact_forms.hpp"

namespace pf {

namespace {
constexpr int kRows = 8;     // rows per compute wave
constexpr int kBS = 256;     // boundary ring (columns)
constexpr float kKeepEnergy = -1.0f;   // below every real energy: E(C), E(C+dx), E(C+dy) of a pixel the sweep must not change
constexpr int kChunk = 8;    // steps streamed per helper iteration and wave
constexpr int kRad = 8;      // LDS window of the gathered plane: +-kRad texels around the band
constexpr int kWC = 64;      // window ring along the step axis (columns)
constexpr int kWRing = 32;   // FOLLOW: window ring across the band (rows)
constexpr int kRadT = 6;     // throughput form: its window reaches +-kRadT texels around pixel + offset (the test admits kRadT - 1): with the window following
                             // the flow a narrower one serves, and its skewed ring then has room for the loader to run three chunks ahead
// Two workgroup shapes of the SAME step (compute_band is one function; every test holds both to the same bits):
//  * SwLatency -- ONE pair: 4 compute waves (one per SIMD: alone on its SIMD a wave is offered an issue slot every ~5.5 cycles,
//    and the step's dependency chain cannot use a second wave) + 4 loaders + publisher + poller + drainer = 704 threads, 113 KB of
//    LDS: one workgroup owns a CU.  Everything is sized for the latency of one band chain (32-step rings, the loader three
//    chunks ahead, one private gather window per band).
//  * SwWide -- a BATCH of pairs (throughput mode): more sweep workgroups are asked for than the chip has CUs, so what counts is
//    steps per CU and second, and a SIMD whose lone compute wave leaves ~3/5 of its issue slots empty is the waste.  8 compute
//    waves (two per SIMD, 64 rows per workgroup) share the same seven helper roles: 960 threads, 4 waves per SIMD (128 VGPRs),
//    126 KB of LDS -- two adjacent bands share one 33-row gather window and one loader, rings are 16 steps.
template <int NW, int BPW>
struct SwGeom {
  static constexpr int kWaves = NW;              // compute waves (bands) per workgroup
  static constexpr int kBPW = BPW;               // adjacent bands that share one gather window and one loader wave
  static constexpr int kLoaders = NW / BPW;
  static constexpr int kRS = BPW == 1 ? 32 : 24; // record ring (steps)
  static constexpr int kRQ = BPW == 1 ? 3 : 2;   // float4 quads per record in the LDS ring: two quads from memory + (x, y, window offset) formed by the loader, or -- wide form, where
                                                 // LDS is what limits the rings -- its first two quads, the (x, y) of the third in a ring of its own (40 bytes)
  static constexpr int kLoadAhead = BPW == 1 ? 3 : 1;   // chunks per band the loader fetches in one round when the ring has room
  static constexpr int kOS = 32;                 // result ring (steps)
  static constexpr int kWA = BPW * kRows + 2 * kRad + 1;   // window extent across the band(s): 25 / 33
  // FOLLOW (latency form, round 5): the gather window is a TORUS -- 64 ring columns along the step axis as before, and kWRing = 32 ring
  // rows across the band -- addressed by the texel's ABSOLUTE sweep-order coordinates, and its loader centres it on pixel + (rounded
  // blurred flow of the chunk) instead of on the pixel.  See the loader in k_sweep2.  The wide form (lab build) keeps the fixed window.
  static constexpr bool kFollow = BPW == 1;
  static constexpr int kWRows = kFollow ? 32 + 1 : kWA;    // rows of LDS per window (torus: ring row 0 again behind row 31)
  static constexpr int kWavesTotal = NW + kLoaders + 3;    // + publisher, poller, drainer
  static constexpr int kThreads = 64 * kWavesTotal;
  static constexpr int kScratch = BPW == 1 ? NW : 1;       // scratch areas for the non-publishing lanes (write-only: may be shared)
};
using SwLatency = SwGeom<4, 1>;
using SwWide = SwGeom<8, 2>;
constexpr int kPrioCompute = 3, kPrioHelper = 1;   // s_setprio of the compute waves / of the helper waves (above another kernel's waves on the same CU)
constexpr int kWCp = kWC + 3; // row stride of the window: ring column 0 is stored a second time behind column 63, so that the
                              // right-hand texel of a bilinear footprint is always the NEXT slot (no second ring wrap on the address chain).
                              // 67, not 65: row r of a band sits at column s - r, so with a stride of 65 float2 the gathers of all 8 rows of a
                              // step (equal flows) fall on the SAME LDS banks (64 r float2 apart); 67 puts them 4 banks apart (a texel pair is
                              // 4 banks wide).  profiles/r04_sweep_helpers_ab.txt
// Every wait is bounded by WALL-CLOCK time, not by a spin count: the deadline (kernel entry + a budget the host scales with
// the launch: 2 s + 1000x the expected duration) sits in LDS and is only looked at every 1024 polls.  A band that is merely
// slow -- several contexts oversubscribing the GPU, a predecessor workgroup not scheduled yet -- therefore never raises
// PF_ERR_TIMEOUT; a genuinely stuck one still does instead of hanging the GPU.
// Hand-off parameters (each swept on the hardware, profiles/r04_sweep_helpers_ab.txt, r05 same-box A/Bs; the rejected alternatives -- a last band that
// stores its granules itself, asm address / sum-of-squares blocks in the throughput form, other priorities / sleeps / window strides -- are kept as
// profiles/r06_removed_experiment_macros.patch, not as macros in the product source):
constexpr int kMarginAcross = 0;    // extra columns a latency-form band falls behind after it has waited for a granule ACROSS workgroups (1 until round 4)
constexpr int kMarginAcrossT = 1;   // ... the throughput form's
constexpr int kLoaderIdleSleep = 8; // a loader whose ring is full sleeps this long instead of walking its predicated-off body
constexpr int kDrainSleep = 8;      // drainer: s_sleep between two looks at the bands' step counters when no chunk was complete
constexpr int kPubSleep = 1;        // publisher wave: s_sleep between two looks at the last band's step counter
constexpr int kPollSleep = 1;       // poller wave: s_sleep between two polls of the previous workgroup's granules
#define PF_STR2(x) #x
#define PF_STR(x) PF_STR2(x)   // ~0.2 s of polling: a stuck band raises ctrl[1] instead of hanging the GPU

struct F2x2 { float a, b, c, d; } __attribute__((aligned(8)));
#ifdef PF_EXPERIMENTS
// lab build only (PANOFLOW_POISON_LDS=1, tests/test_gpu_hygiene.py): every sweep workgroup fills its gather windows with NaN / inf / +-1e38 before the
// loaders start -- a result that depends on a window slot its loader never wrote (or on what a pixel that is not updated reads there) then differs
// from the oracle instead of depending on what the previous kernel happened to leave in LDS
__device__ int g_poison_lds;
__device__ __forceinline__ void poison_window(float2* win, int n) {
  if (g_poison_lds == 0) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int k = i & 3;
    const float v = k == 0 ? __builtin_nanf("") : (k == 1 ? __builtin_inff() : (k == 2 ? 1.0e38f : -1.0e38f));
    win[i] = make_float2(v, (i & 4) ? -v : v);
  }
}
#endif
#ifdef PF_SWEEP_STATS
// diagnostics build only (var_libs/lib_stats.so): [0] wave-steps of the latency form, [1] its gather rounds (one per step) in which >= 1 lane of a
// pixel that is updated left the LDS gather window (the whole wave then takes the HBM path), [2] / [3] the same for the throughput form (two
// gather rounds per step)
__device__ unsigned long long g_sweep_stats[4];
#ifdef PF_SWEEP_TRACE   // (with PF_SWEEP_STATS; tests/micro/band_trace.py) per band: [0] edge waits, [1] edge spins, [2] slow chunk starts, [3] end, [4 + i] start of chunk 64 i (wall clock, 10 ns)
__device__ long long g_band_trace[1024][40];
#endif
#endif

// errorFunction (PixFlow.hpp:427-456); identical operation order to kernels_sweep.hip / the oracle.
__device__ __forceinline__ float d_error2(const float2* __restrict__ g1, int W, float wm2, float hm2, float fW, const SolverCoef& cf, int x, int y, float i0x, float i0y,
                                          float bx, float by, float fdx, float fdy) {
  const float matchX = float(x) + fdx, matchY = float(y) + fdy;
  float cx = (0.0f < matchX) ? matchX : 0.0f; cx = (cx < wm2) ? cx : wm2;
  float cy = (0.0f < matchY) ? matchY : 0.0f; cy = (cy < hm2) ? cy : hm2;
  const int x0 = int(cx), y0 = int(cy);
  const float xR = cx - float(x0), yR = cy - float(y0);
  const float2* p = g1 + size_t(y0) * W + x0;
  const F2x2 t0 = *reinterpret_cast<const F2x2*>(p);
  const F2x2 t1 = *reinterpret_cast<const F2x2*>(p + W);
  float i1x, i1y;
  {
    const float f00 = t0.a, f10 = t0.c, f01 = t1.a, f11 = t1.c;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1x = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  {
    const float f00 = t0.b, f10 = t0.d, f01 = t1.b, f11 = t1.d;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1y = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  const float dfx = bx - fdx, dfy = by - fdy;
  const float smoothness = sqrtf(dfx * dfx + dfy * dfy);
  return sqrtf((i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y)) + smoothness * cf.smooth +
         cf.vreg * fabsf(fdy) / fW + cf.hreg * fabsf(fdx) / fW;
}

// all of v0..v3 (>= 0) are zero or inside [2^-95, 2^100]: the range where sqrt_core/div_core are exact
__device__ __forceinline__ bool fast_range_ok(float v0, float v1, float v2, float v3) {
  const int e0 = __builtin_amdgcn_frexp_expf(v0), e1 = __builtin_amdgcn_frexp_expf(v1), e2 = __builtin_amdgcn_frexp_expf(v2),
            e3 = __builtin_amdgcn_frexp_expf(v3);   // 0 for zero input; <= -125 for denormals
  const int emin = min(min(e0, e1), min(e2, e3));
  const float vmax = __builtin_fmaxf(__builtin_fmaxf(v0, v1), __builtin_fmaxf(v2, v3));
  return emin >= -94 && vmax <= 0x1p100f;
}

// v (>= 0) is zero or inside [2^-95, 2^100]
__device__ __forceinline__ bool fast_range_ok1(float v) {
  return __builtin_amdgcn_frexp_expf(v) >= -94 && v <= 0x1p100f;
}

// errorFunction for the prepass: identical values to d_error2, with the two square roots and two divisions in their
// cheap exact forms whenever every operand is inside the valid range (per-thread test; otherwise the IEEE sequence).
__device__ __forceinline__ float d_error2g(const float2* __restrict__ g1, int W, float wm2, float hm2, float fW, float rW, const SolverCoef& cf, int x, int y, float i0x,
                                           float i0y, float bx, float by, float fdx, float fdy) {
  const float matchX = float(x) + fdx, matchY = float(y) + fdy;
  float cx = (0.0f < matchX) ? matchX : 0.0f; cx = (cx < wm2) ? cx : wm2;
  float cy = (0.0f < matchY) ? matchY : 0.0f; cy = (cy < hm2) ? cy : hm2;
  const int x0 = int(cx), y0 = int(cy);
  const float xR = cx - float(x0), yR = cy - float(y0);
  const float2* p = g1 + size_t(y0) * W + x0;
  const F2x2 t0 = *reinterpret_cast<const F2x2*>(p);
  const F2x2 t1 = *reinterpret_cast<const F2x2*>(p + W);
  float i1x, i1y;
  {
    const float f00 = t0.a, f10 = t0.c, f01 = t1.a, f11 = t1.c;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1x = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  {
    const float f00 = t0.b, f10 = t0.d, f01 = t1.b, f11 = t1.d;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    i1y = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  const float dfx = bx - fdx, dfy = by - fdy;
  const float s2 = dfx * dfx + dfy * dfy;
  const float d2 = (i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y);
  const float av = cf.vreg * fabsf(fdy), ah = cf.hreg * fabsf(fdx);
  if (fast_range_ok(s2, d2, av, ah))
    return sqrt_core(d2) + sqrt_core(s2) * cf.smooth + div_core(av, fW, rW) + div_core(ah, fW, rW);
  return sqrtf(d2) + sqrtf(s2) * cf.smooth + av / fW + ah / fW;
}

// lane i reads lane i+N of its row of 16; lanes whose source falls outside the row read 0 (bound_ctrl), no 'old' operand to set up
template <int N>
__device__ __forceinline__ float dpp_shl0(float src) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), 0x100 + N, 0xF, 0xF, true));
}
// errorFunction with the fast exact forms.  They are only exact inside a range (see sqrt_core / div_core): instead of
// branching at every use, the caller keeps a running (min exponent, max magnitude) of every operand that went through
// them and redoes the whole step with the IEEE sequence if the range was ever left (practically never).
// The four bilinear texels come from the band's LDS window (filled by the loader wave) when the proposal points
// within +-(kRad-1) texels of the pixel, otherwise from HBM.  Scheduled in three phases: (A) addresses + issue of
// the texel reads, (B) every term that does not need the texels (smoothness, the two regularisers) in the shadow
// of their latency, (C) bilinear + data term.  sched_barrier keeps the compiler from sinking B below the wait.
// SKEW (throughput form, 32 rows per band): the window ring is indexed by u + (window row) instead of u -- at any step all 32 rows of a
// band then touch the same ~31 ring slots although their pixels are 32 columns apart (a ring of 64 serves; unskewed it would take 128).
// Moving one window row down also moves one slot on, and the first TWO ring columns are stored again behind column 63 (kWCPT).
// FOLLOW: the window is a torus addressed by the texel's absolute coordinates and centred by its loader on the pixel + an integer offset pc
// (the rounded blurred flow of the chunk; the offset is kept with the record): resident are the texels within 8 of the centre in both axes, so the test is on
// the flow against pc -- a flow of any size within 7 of the offset is served from LDS.  A pixel that is not updated carries pc = NaN: its (discarded) evaluations never
// send the wave through the HBM path -- they read a valid LDS slot with whatever it holds ("not > 7" is true for a NaN distance).
// FOLLOW = 2 (throughput form): the same torus in skewed coordinates -- ring row = the texel's row mod kWA (tRV = 48; `ob` = a multiple of 48 chosen by
// the loader per chunk so that the chunk's rows fall in [ob, ob + 96): one conditional subtraction, as v_min3_u32), ring column = (u + v) & 63; pc = the
// chunk's OFFSET (wave-uniform, image axes) and the test is on the flow relative to it, |fd - pc| <= kRadT - 1 = 5 (the window holds [centre - 6,
// centre + 6], so a difference that rounds onto +-5 is covered; centres are kept inside the image by the loader); `live` = the pixel is
// updated -- one that is not never sends the wave to HBM.
template <bool TR, bool FWD, int kWA, int WCP = kWCp, bool SKEW = false, int FOLLOW = 0>
__device__ __forceinline__ float d_error_fast(const float2* __restrict__ g1, __attribute__((address_space(3))) const float2* win, int ob, int W, int H, float wm2, float hm2,
                                              float fW, float rW, const SolverCoef& cf, f2p pos, float i0x, float i0y, float bx, float by, f2p fd,
                                              int& emin, float& vmax, f2p pc = f2p{0.f, 0.f}, bool live = true) {
  // ---- A ----  (pos = the pixel's (x, y), fd = the candidate flow: packed fp32 wherever both components take the same operation)
  const float fdx = fd.x, fdy = fd.y;
  const f2p match = pos + fd;
  const float matchX = match.x, matchY = match.y;
  const float cx = __builtin_amdgcn_fmed3f(matchX, 0.0f, wm2);   // min(w-2, max(0, v)) incl. NaN -> 0 (med3 with a NaN input returns min3)
  const float cy = __builtin_amdgcn_fmed3f(matchY, 0.0f, hm2);
  const int x0 = int(cx), y0 = int(cy);
  // |flow| <= kRad-1 in both components keeps the four texels inside the window: the clamps only move (cx,cy) towards the
  // pixel, and int(c), int(c)+1 then lie within [f-7, f+8].  Tested on the flow itself (known at the start of the step),
  // not on the clamped position: one max + one compare, off the address chain.  (NaN flows compare false -> HBM path.)
  bool inwin;
  if (FOLLOW == 2) {
    const f2p dm = fd - pc;
    inwin = !(__builtin_fmaxf(fabsf(dm.x), fabsf(dm.y)) > float(kRadT - 1)) || !live;
  } else if (FOLLOW == 1) {
    // pc = the window's OFFSET o (the record's third quad, z / w; centre = pixel + o).  The loader keeps every window centre INSIDE the image
    // (it clamps the chunk's offset), so the test may be made on the flow relative to the offset, before the sample's clamp to the image -- off
    // the address chain, from the proposal alone (it issues beside the position sum and fills the wait state behind that packed add): a sample
    // within 7 of a centre in [0, W - 1] that the clamp moves to 0 or W - 2 is still within 7 of it.  The subtraction is exact to well under a
    // texel, and the window holds one texel more than the test admits on either side ([centre - 8, centre + 8]): a difference that rounds onto
    // +-7 is covered.
    const f2p dm = fd - pc;
    inwin = !(__builtin_fmaxf(fabsf(dm.x), fabsf(dm.y)) > float(kRad - 1));   // (NaN offset = a pixel that is not updated: "inside")
  } else inwin = __builtin_fmaxf(fabsf(fdx), fabsf(fdy)) <= float(kRad - 1);
  // sweep-order coordinates of texel (x0,y0): u along the step axis, v across the bands
  // The 2x2 footprint in sweep order is (v0, v0 + sg) x (u0, u0 + sg), sg = +1 forward / -1 backward.  Addressed from its LOWER
  // corner (vlo, ulo) it is two adjacent slots in two adjacent window rows -- the slot after ring column 63 holds column 0 again
  // (kWCp) -- so one ring wrap and one address serve all four texels (immediate offsets 0, 1, kWCp, kWCp + 1).
  // FOLLOW == 1 (the product's latency form, round 6): the torus is addressed by the texel's IMAGE coordinates instead -- the footprint's lower corner
  // is (x0, y0) whatever the sweep's direction, so neither the mirror subtractions of a backward sweep nor the window-origin subtraction stand
  // on the address chain (1 instruction per step forward, 2 backward), and the +1 neighbours are the next slots in both directions.
  constexpr bool kAbs = FOLLOW == 1;
  const int cxl = (FWD || kAbs) ? x0 : W - 2 - x0, cyl = (FWD || kAbs) ? y0 : H - 2 - y0;
  const int ulo = TR ? cyl : cxl, vlo = TR ? cxl : cyl;
  // out-of-window lanes are clamped to a valid window row (they read garbage that the HBM path below overwrites)
  int alo;   // ring / window row of texel row vlo (always a valid slot)
  if (FOLLOW == 2) { const unsigned a = unsigned(vlo - ob); alo = int(min(min(a, a - unsigned(kWA)), unsigned(kWA - 1))); }   // a in [0, 2 * 48) -> a mod 48; anything else (a lane outside the window, overwritten by the HBM path): row 47, so that its footprint stays inside this band's own window rows
  else if (FOLLOW == 1) alo = (kAbs ? vlo : vlo - ob) & (kWRing - 1);
  else alo = min(max(vlo - ob, 0), kWA - 2);
  auto q = [&](int dr, int dc) { return ((FWD || kAbs) ? dr : 1 - dr) * (WCP + (SKEW ? 1 : 0)) + ((FWD || kAbs) ? dc : 1 - dc); };   // texel (v0 + sg*dr, u0 + sg*dc), relative to the corner
  const int o00 = q(0, 0);                            // texel (x0, y0)
  const int o10 = TR ? q(1, 0) : q(0, 1);             // texel (x0+1, y0)
  const int o01 = TR ? q(0, 1) : q(1, 0);             // texel (x0, y0+1)
  const int o11 = q(1, 1);                            // texel (x0+1, y0+1)
  // LDS reads are unconditional (out-of-window lanes read a clamped slot and are overwritten below) so that the two
  // address spaces never meet in one pointer -- a merged pointer would turn every access into a slow flat_load.
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const f2v lds_f2;
  // explicit LDS address space: ds_read2_b64, never a flat access.  The corner's byte address as (ring column << 3) + window, then
  // + row * stride in one 24-bit multiply-add: five instructions from (x0, y0) to the address.
  // = ((column & 63) << 3) + window base + alo * row stride.  Written out as ONE block of three instructions (round 5): the compiler's own form is
  // v_lshl + v_and + v_add + v_mul_u32_u24 + v_add3 (two more), and pinning single instructions with asm statements costs an s_nop behind each
  // statement whose result the next instruction reads (4 cycles for a lone wave: as much as the instruction saved).
  const unsigned colSum = unsigned(SKEW ? ulo + (FOLLOW == 2 ? vlo : alo) : ulo);
  unsigned cornerAddr;
  if (!SKEW) {
    asm("v_and_b32 %0, %5, %1\n\tv_lshl_add_u32 %0, %0, 3, %2\n\tv_mad_u32_u24 %0, %3, %4, %0"
        : "=&v"(cornerAddr) : "v"(colSum), "s"((unsigned)(size_t)win), "v"(alo), "s"(unsigned(WCP * sizeof(float2))), "n"(kWC - 1));
  } else {
    unsigned ringCol = colSum & (kWC - 1);
    asm("" : "+v"(ringCol));   // (x & 63) << 3 + base as v_and + v_lshl_add, not the canonical v_lshl + v_and + v_add
    const unsigned cornerCol = (ringCol << 3) + (unsigned)(size_t)win;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cornerAddr) : "v"(alo), "s"(unsigned(WCP * sizeof(float2))), "v"(cornerCol));
  }
  lds_f2* win3 = (lds_f2*)(size_t)cornerAddr;
  const f2v w00 = win3[o00], w10 = win3[o10], w01 = win3[o01], w11 = win3[o11];
  float2 t00 = make_float2(w00.x, w00.y), t10 = make_float2(w10.x, w10.y), t01 = make_float2(w01.x, w01.y), t11 = make_float2(w11.x, w11.y);
  __builtin_amdgcn_sched_barrier(0);
  // ---- B ----
  const float xR = __builtin_amdgcn_fractf(cx), yR = __builtin_amdgcn_fractf(cy);   // cx, cy >= 0: exactly cx - float(int(cx))
  float s2;
  if (PF_PK_ASM && !SKEW) s2 = sumsq_diff2(f2p{bx, by}, fd);   // |blurred - flow|^2: one block (exact_forms.hpp)
  else {
    const f2p df = f2p{bx, by} - fd;
    const f2p df2 = df * df;
    s2 = df2.x + df2.y;
    asm volatile("" : "+v"(s2));   // a finished scalar here: otherwise the SLP vectoriser packs this add with d2's below behind two register moves (3 instructions for 2)
  }
  float av = cf.vreg * fabsf(fdy);
  asm volatile("" : "+v"(av));   // keeps the two products scalar (|x| is a free source modifier there); packed, they need two v_and for the abs: 3 instructions for 2
  const float ah = cf.hreg * fabsf(fdx);
  const f2p reg = div_core2(f2p{av, ah}, fW, rW);
  const float rv = reg.x, rh = reg.y;
  emin = min(min(__builtin_amdgcn_frexp_expf(s2), __builtin_amdgcn_frexp_expf(av)), __builtin_amdgcn_frexp_expf(ah));   // 0 for a zero operand
  vmax = __builtin_fmaxf(__builtin_fmaxf(s2, av), ah);
  // The window test is taken HERE, behind phase B, not right behind the texel reads (round 5): the wave cannot issue past the branch before
  // the test's result is there, and with the flow-following window the test hangs on one more (packed) subtraction -- behind B its latency is
  // covered.  Wave-uniform test first: the common "every lane inside the window" case costs a compare + one scalar branch, not an exec-mask
  // save / restore around an empty block (two instructions of ~150 per step; a step is issue-bound, profiles/r03_sweep_step_isa.txt)
  if (__builtin_expect(__any(!inwin), 0)) {
#ifdef PF_SWEEP_STATS
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_sweep_stats[SKEW ? 3 : 1], 1ull);   // gather rounds of a wave that left the LDS window
#endif
    if (!inwin) {
      const float2* p = g1 + (y0 * W + x0);
      t00 = p[0]; t10 = p[1]; t01 = p[W]; t11 = p[W + 1];
    }
    // (throughput form: its compute wave has record loads in flight; the wait for these four must sit INSIDE this cold branch, or the
    // in-order memory counter would make every step wait for the newest record request)
    if (SKEW) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a real instruction, so that the compiler's counter tracking knows nothing is pending at the join
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- C ----
  // i1 = a1 + a2 * xR + a3 * yR + a4 * xR * yR per channel, left to right: the last addition (p + l) is made inside the block that also takes
  // the difference to I0, its square and the sum of the two channels (exact_forms.hpp: no wait states between the four)
  float px, py, lx, ly;
  {
    const float f00 = t00.x, f10 = t10.x, f01 = t01.x, f11 = t11.x;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    px = a1 + a2 * xR + a3 * yR; lx = a4 * xR * yR;
  }
  {
    const float f00 = t00.y, f10 = t10.y, f01 = t01.y, f11 = t11.y;
    const float a1 = f00, a2 = f10 - f00, a3 = f01 - f00, a4 = f00 + f11 - f10 - f01;
    py = a1 + a2 * xR + a3 * yR; ly = a4 * xR * yR;
  }
  float d2;
  if (PF_PK_ASM && !SKEW) d2 = sumsq_diff2_sum(f2p{i0x, i0y}, f2p{px, py}, f2p{lx, ly});
  else {
    const float i1x = px + lx, i1y = py + ly;
    f2p di, di2;   // (i0 - i1)^2 per channel: one asm block, see exact_forms.hpp
#if PF_PK_ASM
    asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %1, %0, %0" : "=&v"(di), "=&v"(di2) : "v"(f2p{i0x, i0y}), "v"(f2p{i1x, i1y}));
#else
    di = f2p{i0x, i0y} - f2p{i1x, i1y}; di2 = di * di;
#endif
    d2 = di2.x + di2.y;
    asm volatile("" : "+v"(d2));   // a scalar add into the register next to s2 (not a packed add + a move of s2)
  }
  vmax = __builtin_fmaxf(vmax, d2);
  int ed2;
  const f2p sq = sqrt_core2(f2p{d2, s2}, ed2);   // both square roots of the step as one packed sequence (+ d2's exponent for the guard)
  emin = min(emin, ed2);
  return sq.x + sq.y * cf.smooth + rv + rh;
}

// The tail of a step.  Every proposal takes its OWN gradient step, in its own lane, before anybody knows which one wins: lane 0
// of a group holds the along-axis proposal's E, lanes 1 and 2 its E(+dx), E(+dy) (lanes 4, 5, 6: the across proposal's), so two
// DPP moves give lanes 0 and 4 their finite differences, and ONE pass through the division / update -- the same instructions
// for all lanes -- produces both candidates' results from the proposal flow each lane already holds.  The current flow's own
// step, rC = C - 0.5 * grad E(C) / eps, does not depend on the neighbours: it comes with the record (prepass).  Selection in
// the reference's order (current, then L, then T, strict '<') then only has to pick one of three finished results in lane 0.
// (Before: gather all six energies in lane 0, select (E, E+dx, E+dy, flow) in two stages of five v_cndmask, then one gradient
// step: two DPP moves and five v_cndmask more per step, all on the dependency chain.)
// FAST uses div_core and extends the running range guard; !FAST is the IEEE sequence.
// CROSS (round 6): the cross-lane neighbour exists for every lane (any band but a sweep's first), i.e. the comparison that reads the across
// proposal needs no mask.  Then the three DPP moves that fetch the across proposal's energy and result (lane + 4) and the selects that use them are
// ONE instruction each: a VOP2 instruction takes a DPP source itself -- v_cndmask_b32_dpp D = vcc ? own : across -- 6 issue slots instead of 8
// (the compiler cannot form them: its selects carry their mask in an SGPR pair, the DPP encoding reads VCC; gfx950 has no DPP form of VOPC, so the
// across energy still takes its own move).  Same comparisons, same operands: same bits (the lab build keeps the plain form; every product-vs-lab
// test holds the two to each other).  The block's DPP reads stand >= 2 slots behind the writers of their sources by construction (rp is written
// before the block and first read through DPP in its fourth slot; tools/asm_sched.py knows the instructions and re-checks).
#define PF_DPP4 " row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
// MID: a hook of the caller's that is issued between the gradient step and the selection -- compute_band puts its read-ahead of the producer's
// step counter and top value there (round 6): as late in the step as the LDS latency before the step's publishing wait allows, so that a band
// follows the band above ~0.25 step closer (same-box: lone dense pair 43.70 -> 43.51 ms; profiles/r06_step_block_ab.txt)
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <bool FAST, bool TR, bool CROSS = false, class MID = NoMid>
__device__ __forceinline__ float2 select_step(float e, float eC, float eCL, float2 rC, float2 cnd, bool okL, bool okT, float rEps, float step, int& emin, float& vmax, MID mid = MID()) {
  [[maybe_unused]] constexpr bool kFuse = FAST && CROSS && PF_PK_ASM;
  float2 rp;                                               // this lane's proposal after its gradient step (meaningful in lanes 0 and 4)
  if (FAST) {
    // (E+dx, E+dy) - E as two scalar subtractions whose first operand comes through DPP (row_shl:n reads lane+n): v_sub_f32_dpp, no separate move
    // Range guard of the division by eps, taken on the ENERGIES (round 6) instead of on their differences: every lane holds an energy (lanes 1, 2
    // / 5, 6 the ones at +dx, +dy), the guard is wave-wide, and two energies that are each zero or of magnitude in [2^-71, 2^99] differ by zero or by
    // at least an ulp of 2^-71 = 2^-94 and by at most 2^100 -- inside div_core's proven range [2^-95, 2^100].  One exponent (biased by 24 so that it
    // shares the guard's threshold: exponent >= -94 <=> value >= 2^-95) and one operand of the max3 instead of two exponents, a minimum and a maximum.
    emin = min(emin, __builtin_amdgcn_frexp_expf(e) - 24);
    vmax = __builtin_fmaxf(vmax, fabsf(e));
    f2p dg;
#if PF_PK_ASM
    if (kFuse) {
      // (E+dx, E+dy) - E as two subtractions whose first operand comes through DPP (row_shl:n reads lane+n): v_sub_f32_dpp, no separate moves + packed
      // subtraction.  (s_nop 1: the DPP source is the energy's last addition, one slot up, and the compiler does not see a DPP read inside an asm
      // statement -- 2 wait states; tools/asm_sched.py re-derives them and fills them with independent instructions)
      float dx_, dy_;
      asm("s_nop 1\n\tv_sub_f32_dpp %0, %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_sub_f32_dpp %1, %2, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
          : "=&v"(dx_), "=&v"(dy_) : "v"(e));
      dg = f2p{dx_, dy_};
    } else
#endif
    {
      const float g1 = dpp_shl0<1>(e), g2 = dpp_shl0<2>(e);   // row_shl:n reads lane+n
      dg = f2p{g1, g2} - f2p{e, e};
    }
    // packed fp32 (a step is issue-bound: one v_pk_* per pair of operations): / eps, flow - 0.5 * g
    const f2p gq = div_core2(dg, kGradEpsilon, rEps);
    // flow - step * g as ONE fused multiply-add: the product with a power of two is exact (|g| >= 2^-84 or 0 inside the guard's
    // range, so no underflow for step >= 2^-16), hence the single rounding of the FMA is the rounding of the reference's subtraction -- bit for
    // bit, signs of zero included (g = +0: f + (-0) = f) -- and one instruction less on the step's dependency chain.
    // (`step` is a kernel argument since round 6: the host only lets the fast form's result stand when step is a power of two in
    // [2^-16, 2^16] -- SolverCoef::guard_min sends every other step size through the IEEE branch below.)
    const f2p r = __builtin_elementwise_fma(gq, f2p{-step, -step}, f2p{cnd.x, cnd.y});
    rp = make_float2(r.x, r.y);
  } else {
    const float g1 = dpp_shl0<1>(e), g2 = dpp_shl0<2>(e);   // row_shl:n reads lane+n
    const float gx = (g1 - e) / kGradEpsilon, gy = (g2 - e) / kGradEpsilon;
    rp = make_float2(cnd.x - step * gx, cnd.y - step * gy);
  }
  __builtin_amdgcn_sched_barrier(0);   // (the barriers keep the compiler from hoisting the two independent LDS reads back up)
  mid();
  __builtin_amdgcn_sched_barrier(0);
#if PF_PK_ASM
  if (kFuse && !TR) {
    // L = the along proposal (own lane), T = the across proposal (lane + 4); okL and okT are true here
    const bool pickL = e < eCL;
    const float cur = pickL ? e : eC;
    const float eT = dpp_shl0<4>(e);
    const unsigned long long mL = __builtin_amdgcn_ballot_w64(pickL);
    float fx, fy;
    asm("v_cmp_nlt_f32_e32 vcc, %4, %5\n\t"                      // vcc = !(E(T) < cur)
        "v_cndmask_b32_e64 %0, %6, %2, %8\n\t"                   // (pickL ? rL : rC).x
        "v_cndmask_b32_e64 %1, %7, %3, %8\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc" PF_DPP4 "\n\t"       // vcc ? that : rT.x  (rT = rp of lane + 4)
        "v_cndmask_b32_dpp %1, %3, %1, vcc" PF_DPP4
        : "=&v"(fx), "=&v"(fy) : "v"(rp.x), "v"(rp.y), "v"(eT), "v"(cur), "v"(rC.x), "v"(rC.y), "s"(mL) : "vcc");
    return make_float2(fx, fy);
  }
  if (kFuse && TR) {
    // transposed: L = the across proposal (lane + 4; okL is true here), T = the along proposal (own lane; okT per lane)
    const float eL = dpp_shl0<4>(e);
    float fx = rC.x, fy = rC.y, cur;
    asm("v_cmp_nlt_f32_e32 vcc, %3, %7\n\t"                      // vcc = !(E(L) < E(C) as L sees it)
        "s_nop 1\n\t"                                            // (VALU write of VCC -> VALU read: 2 wait states; asm_sched.py fills them)
        "v_cndmask_b32_e32 %2, %3, %6, vcc\n\t"                  // cur = vcc ? E(C) : E(L)
        "v_cndmask_b32_dpp %0, %4, %0, vcc" PF_DPP4 "\n\t"       // vcc ? rC : rL  (rL = rp of lane + 4)
        "v_cndmask_b32_dpp %1, %5, %1, vcc" PF_DPP4
        : "+v"(fx), "+v"(fy), "=&v"(cur) : "v"(eL), "v"(rp.x), "v"(rp.y), "v"(eC), "v"(eCL) : "vcc");
    const bool pickT = okT && (e < cur);
    return make_float2(pickT ? rp.x : fx, pickT ? rp.y : fy);
  }
#endif
  // lane 0: the across proposal's energy and result from lane 4; L is the along-axis proposal unless the sweep is transposed
  const float eX = dpp_shl0<4>(e);
  const float2 rX = make_float2(dpp_shl0<4>(rp.x), dpp_shl0<4>(rp.y));
  const float eL = TR ? eX : e, eT = TR ? e : eX;
  const float2 rL = TR ? rX : rp, rT = TR ? rp : rX;
  const bool pickL = okL && (eL < eCL);   // eCL = eC, or below every energy where L does not exist (see the records)
  const float cur = pickL ? eL : eC;
  const bool pickT = okT && (eT < cur);
  float2 f; f.x = pickL ? rL.x : rC.x; f.y = pickL ? rL.y : rC.y;
  f.x = pickT ? rT.x : f.x; f.y = pickT ? rT.y : f.y;
  return f;
}

__device__ __forceinline__ unsigned long long pack2(float2 f) {
  return (unsigned long long)__float_as_uint(f.x) | ((unsigned long long)__float_as_uint(f.y) << 32);
}
__device__ __forceinline__ float2 unpack2(unsigned long long v) {
  return make_float2(__uint_as_float((unsigned)(v & 0xffffffffu)), __uint_as_float((unsigned)(v >> 32)));
}

// DPP cross-lane moves (wave64, rows of 16 lanes).  row_shl:n -> lane i reads lane i+n.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}
typedef float f2d __attribute__((ext_vector_type(2)));
template <class G>
struct SmemT {
  float4 rec[G::kWaves][G::kRS][kRows][G::kRQ];
  float2 recxy[G::kRQ == 3 ? 1 : G::kWaves][G::kRQ == 3 ? 1 : G::kRS][kRows];   // wide form: the records' (x, y)
  float2 out[G::kWaves][G::kOS][kRows];
  float2 win[G::kLoaders][G::kWRows][kWCp];  // (I1x,I1y) texels around each band (pair of bands), sweep-order coordinates, ring along the step axis (+ column 0 again)
  float2 scratch[G::kScratch][128];       // where lanes 1-7 of a group "store" in the publishing step (lane + 8 * step-in-chunk)
  unsigned long long bnd[kBS];            // granules of the previous workgroup's last row (poller -> wave 0), valid below bndHead
  int recHead[G::kWaves];   // steps of records available to wave w        (stream helper -> compute)
  int outHead[G::kWaves];   // steps completed by wave w                    (compute -> helpers, next wave)
  int outTail[G::kWaves];   // steps of wave w written to the flow plane    (stream helper -> compute)
  int pubTail;           // steps of the last wave published as granules (granule helper -> compute)
  int bndHead;           // boundary columns available to wave 0        (granule helper -> compute)
  int abort;
  int wg;
  long long deadline;        // wall_clock64() value after which a waiting wave gives up
  int statHits, statSpins;   // -DPF_SWEEP_STATS only
  long long statEntry;
};

template <class SM>
__device__ __forceinline__ bool spin_expired(int& spins, const SM& sm) {
  if ((++spins & 1023) != 0) return false;
  return (long long)wall_clock64() > *(const volatile long long*)&sm.deadline;
}

// LDS counters: a wave's LDS operations are executed in issue order by the CU's LDS unit, so "write data,
// then write counter" / "read counter, then read data" needs no s_waitcnt between them -- only the
// compiler must not reorder them (the empty asm is a compiler-only barrier).
__device__ __forceinline__ int ld_cnt(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void st_cnt(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

// One compute wave: a band of 8 rows, lane = 8*row + role.  TOP: 0 = image border above, 1 = previous wave
// of this workgroup (its LDS result ring), 2 = previous workgroup (granule ring filled by the poller wave).
// Hand-off protocol: the producer writes its result, then its step counter; the consumer reads the counter,
// then the value (one wave's LDS operations execute in issue order), one step ahead of use.  The counter is
// wave-uniform, so the per-step "is my top neighbour there" test is two scalar instructions.
template <class G, int TOP, bool TR, bool FWD, bool SPARSE>
__device__ __forceinline__ bool compute_band(SmemT<G>& sm, const float2* __restrict__ g1, int W, int H, int nsteps, int w, int band,
                                             int nact, bool publishes, float rW, float rEps, const SolverCoef cf, int uLo, int LSv) {
  constexpr int kWaves = G::kWaves, kRS = G::kRS, kOS = G::kOS, kWA = G::kWA;
  constexpr int transposed = TR ? 1 : 0, forward = FWD ? 1 : 0;
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, k = lane & 7;
  const int ib = band * kRows + r;           // index across the bands: row (normal) or column (transposed)
  const int LS = transposed ? H : W;         // extent along the step axis
  // the cross-lane neighbour (row/column before this one) exists: always, for a band that has a band before it (known at
  // compile time there: the selection then needs no mask on that comparison -- one scalar instruction + one wait state per step)
  const bool hasCross = (TOP != 0) || ib > 0;
  // The band's window as an LDS address the compiler cannot take apart (an SGPR): with the struct offset visible it splits the
  // four texel reads over two address registers instead of one address + immediate offsets.
  typedef __attribute__((address_space(3))) const float2 lds_cf2;
  lds_cf2* win = (lds_cf2*)&sm.win[w / G::kBPW][0][0];   // (wide form: the window this band shares with its neighbour)
  asm volatile("" : "+s"(win));
  int ob = (band - w % G::kBPW) * kRows - kRad;   // window origin across the bands (the workgroup's first band is a multiple of kBPW)
  asm volatile("" : "+s"(ob));                // opaque: otherwise the compiler splits it into (v - band*8) + 8, one more instruction on the address chain
  const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W);
  // Lane roles inside a pixel's group of 8: lanes 0-2 evaluate the proposal of the previous pixel ALONG the step axis (this
  // row's own last result) at +0, +dx, +dy; lanes 4-6 the proposal of the pixel ACROSS (the row above); lanes 3 and 7 repeat
  // lanes 0 and 4.  Quads, because DPP bank masks select quads: each lane's proposal arrives by DPP straight from the lane
  // that computed it (lane 0 of a group), no select.
  const int kk = k & 3;
  const float addx = (kk == 1) ? kGradEpsilon : 0.0f, addy = (kk == 2) ? kGradEpsilon : 0.0f;
  const bool lastPub = publishes && (w == kWaves - 1);
  const bool hasNext = (w + 1 < nact);
  // where row 0's top neighbour of column c lives, and the counter that says how many columns are there
  const int wp = (w > 0) ? w - 1 : 0;
  const int* topHead = (TOP == 1) ? &sm.outHead[wp] : &sm.bndHead;
  constexpr int kBias = (TOP == 1) ? kRows - 1 : 0;   // TOP==1: column c is the producer's step c + kRows - 1
  auto top_slot = [&](int c) -> const unsigned long long* {
    return (TOP == 1) ? reinterpret_cast<const unsigned long long*>(&sm.out[wp][(c + kRows - 1) % kOS][kRows - 1]) : &sm.bnd[c & (kBS - 1)];
  };
  float2 prev = make_float2(0.f, 0.f);
  bool dead = false;
  // The step counter's LDS address lives in a register for the whole band (the empty asm keeps it opaque, so the compiler cannot
  // rematerialise it from an SGPR with a v_mov in every step -- a step is issue-bound, every instruction is ~0.65 % of it:
  // profiles/r03_sweep_step_isa.txt).  (Measured and rejected: counting with an LDS atomic add of a register-resident 1 -- two
  // instructions fewer, but the 64 lanes of the wave serialise on the one address: sweeps +17 %.)
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* cntp = (lds_int*)&sm.outHead[w];
  asm volatile("" : "+v"(cntp));
#ifdef PF_SWEEP_STATS
  int statHits = 0, statSpins = 0;
  long long statT0 = 0, statWait = 0, statR8 = 0, statC0 = 0;
  int statRedo = 0, statOOW = 0;
  int statSlowChunks = 0, statChunkSpins = 0, statFailRec = 0, statFailTail = 0, statFailPub = 0, statFailNext = 0;
  const long long statB = wall_clock64();
#endif
  // image coordinates of this lane's pixel: across the bands (constant) and along the step axis (s - r in sweep order)
  const float fLast = float(LS - 1);
  // Does the next step have to wait for its top value?  Decided at the end of each step from the producer's counter read
  // during the step, with a vector compare straight on the loaded register (v_cmp + branch on vcc: two instructions fewer
  // per step than moving the counter to an SGPR and comparing there).
  bool waitTop = true;
  unsigned long long tv = 0;           // raw top value for the current step (read during the previous one)
  // flow-control counters for the NEXT chunk, read one chunk ahead (they only grow, a stale value is conservative)
  int fcRec = 0, fcTail = 0, fcPub = 0, fcNext = 0;
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;   // this step's record (read one step ahead)
  float2 rc = make_float2(0.f, 0.f);                          // the record's third quad: the pixel's (x, y) ...
  float2 rp = make_float2(0.f, 0.f);                          // ... and, FOLLOW, the offset (ox, oy) of its gather window, written by the loader
  for (int s0 = 0; s0 < nsteps; s0 += kChunk) {
    // ---- once per 8 steps: records of the chunk present, result-ring slots of the chunk free ----
    if (dead) return false;
#ifdef PF_SWEEP_STATS
    const long long tw0 = __builtin_readcyclecounter();
    if (s0 == 0) statT0 = tw0;
#endif
    {
      int spins = 0;
      for (;;) {   // slot t%kOS may be reused once step t-kOS was written out, published and consumed by the next wave
        const int rec = __builtin_amdgcn_readfirstlane(fcRec);
        int lim = __builtin_amdgcn_readfirstlane(fcTail) + kOS;
        if (lastPub) { const int c1 = __builtin_amdgcn_readfirstlane(fcPub) + kOS; lim = lim < c1 ? lim : c1; }
        if (hasNext) { const int c2 = __builtin_amdgcn_readfirstlane(fcNext) + kOS + kRows - 1; lim = lim < c2 ? lim : c2; }
        if (__builtin_expect(rec >= s0 + kChunk && lim >= s0 + kChunk, 1)) break;
#ifdef PF_SWEEP_STATS
        if (spins == 0) {
          if (rec < s0 + kChunk) ++statFailRec;
          if (__builtin_amdgcn_readfirstlane(fcTail) + kOS < s0 + kChunk) ++statFailTail;
          if (lastPub && __builtin_amdgcn_readfirstlane(fcPub) + kOS < s0 + kChunk) ++statFailPub;
          if (hasNext && __builtin_amdgcn_readfirstlane(fcNext) + kOS + kRows - 1 < s0 + kChunk) ++statFailNext;
        }
#endif
        if (spins) __builtin_amdgcn_s_sleep(1);
        fcRec = ld_cnt(&sm.recHead[w]); fcTail = ld_cnt(&sm.outTail[w]);
        if (lastPub) fcPub = ld_cnt(&sm.pubTail);
        if (hasNext) fcNext = ld_cnt(&sm.outHead[w + 1]);
        if (spin_expired(spins, sm) || (((spins & 255) == 0) && ld_cnt(&sm.abort))) return false;
      }
#ifdef PF_SWEEP_STATS
      if (spins) { ++statSlowChunks; statChunkSpins += spins; }
#endif
      if (__builtin_expect(spins != 0, 0)) {
        // The last step of the previous chunk read this chunk's first record ahead; that read was only good if the
        // record was already there, which the one-chunk-old counter just confirmed unless we had to wait.
        const float4* rp0 = &sm.rec[w][s0 % kRS][r][0];
        ra = rp0[0]; rb = rp0[1]; rc = (G::kRQ == 3) ? *reinterpret_cast<const float2*>(rp0 + 2) : sm.recxy[G::kRQ == 3 ? 0 : w][G::kRQ == 3 ? 0 : s0 % kRS][r];
        if (G::kFollow) rp = *(reinterpret_cast<const float2*>(rp0 + 2) + 1);
        asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w), "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w), "+v"(rc.x), "+v"(rc.y), "+v"(rp.x), "+v"(rp.y));
      }
    }
    // read the counters again for the next chunk; the loads complete in the shadow of this chunk's steps
    fcRec = ld_cnt(&sm.recHead[w]); fcTail = ld_cnt(&sm.outTail[w]);
    if (lastPub) fcPub = ld_cnt(&sm.pubTail);
    if (hasNext) fcNext = ld_cnt(&sm.outHead[w + 1]);
#ifdef PF_SWEEP_STATS
    statWait += __builtin_readcyclecounter() - tw0;
#endif
#ifdef PF_SWEEP_STATS
    if (s0 == 8) statR8 = wall_clock64();
#ifdef PF_SWEEP_TRACE
    if (((s0 >> 3) & 63) == 0 && band < 1024 && (s0 >> 9) < 34 && lane == 0) g_band_trace[band][4 + (s0 >> 9)] = wall_clock64();
#endif
    if (s0 == 0) statC0 = wall_clock64();
#endif
    // Ring positions of the chunk (nsteps is a whole number of chunks; every ring length is a multiple of the chunk, so a
    // chunk never wraps inside a ring): with the step loop unrolled, the per-step ring addresses are base + constant.
    // They are LDS pointers held in VGPRs across the chunk (opaque to the compiler, which otherwise recomputes each of them
    // from s in every step: 3 + 1 + 1 instructions of a step that is issue-bound).
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f2w __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const f4v lds_f4;
    typedef __attribute__((address_space(3))) f2w lds_wf2;
    typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
    lds_f4* recChunk = (lds_f4*)&sm.rec[w][s0 % kRS][r][0];                 // record of step s0 + j: recChunk + j * kRows * kRQ
    lds_f4* recNext = (lds_f4*)&sm.rec[w][(s0 + kChunk) % kRS][r][0];       // first record of the next chunk
    typedef __attribute__((address_space(3))) const f2w lds_cf2w;
    lds_cf2w* xyChunk = (lds_cf2w*)&sm.recxy[G::kRQ == 3 ? 0 : w][G::kRQ == 3 ? 0 : s0 % kRS][r];              // wide form: (x, y) of step s0 + j: xyChunk + j * kRows
    lds_cf2w* xyNext = (lds_cf2w*)&sm.recxy[G::kRQ == 3 ? 0 : w][G::kRQ == 3 ? 0 : (s0 + kChunk) % kRS][r];
    if (G::kRQ != 3) asm volatile("" : "+v"(xyChunk), "+v"(xyNext));
    lds_wf2* outChunk = (lds_wf2*)((k == 0) ? &sm.out[w][s0 % kOS][r] : &sm.scratch[G::kScratch == 1 ? 0 : w][lane]);   // result slot of step s0 + j: outChunk + j * kRows (lanes 1-7 of a group hold no result: scratch)
    lds_u64* topChunk = (lds_u64*)((TOP == 1) ? top_slot(s0 + 1) : &sm.bnd[s0 & (kBS - 1)] + 1);   // top value of column s0 + j + 1
    lds_u64* topNext = (lds_u64*)top_slot(s0 + kChunk);                     // ... of the next chunk's first column
    asm volatile("" : "+v"(recChunk), "+v"(recNext), "+v"(outChunk));
    if (TOP != 0) asm volatile("" : "+v"(topChunk), "+v"(topNext));
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = s0 + j;
      // ---- this lane's proposal: lanes 0-3 of a group <- the group's own last result (its lane 0), lanes 4-7 <- the result of
      // the row above (lane 0 of the group before; row 0 of the band: the ring value).  prev is valid in lane 0 of a group only. ----
      // lanes no DPP move writes (lanes 4-7 of the band's row 0) keep this: the ring value (there is one unless this is the first band)
      float2 cnd = prev;
      if (TOP != 0) {
        if (__builtin_expect(waitTop, 0)) {
          if (!dead && s < LSv) {
            // at the edge of the producer: wait for this column (across workgroups: and the next one, i.e. fall one more
            // column behind, so that the following steps find their top value already read despite the HBM hop's jitter).
            // No early exit from the hot loop: a timeout only marks the band dead (checked once per chunk).
            const int need = (s + 1 + (TOP == 2 ? kMarginAcross : 0) < LSv) ? s + 1 + (TOP == 2 ? kMarginAcross : 0) : LSv;
            int spins = 0;
#ifdef PF_SWEEP_STATS
            ++statHits;
#endif
            for (;;) {
              // counter, then value, issued together (one wave's LDS operations execute in issue order: a counter that says "there" makes the
              // value read behind it good) -- one LDS round trip per poll instead of two dependent ones at the exit
              const int cnt = ld_cnt(topHead);
              tv = __hip_atomic_load(top_slot(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              const int avail = __builtin_amdgcn_readfirstlane(cnt) - kBias;   // columns [0, avail) of the row above are in the ring
              if (avail >= need) break;
              if (spin_expired(spins, sm) || (((spins & 255) == 0) && ld_cnt(&sm.abort))) { dead = true; break; }
            }
#ifdef PF_SWEEP_STATS
            statSpins += spins;
#endif
            // leave nothing in flight at the join with the fast path (the compiler would wait there on every step)
            unsigned tlo = unsigned(tv), thi = unsigned(tv >> 32);
            asm volatile("" : "+v"(tlo), "+v"(thi));
            tv = (unsigned long long)tlo | ((unsigned long long)thi << 32);
          }
        }
        cnd = unpack2(tv);
      }
      // Every move reads prev itself (one DPP hop from the lane that computed it; only the hop into the next row of 16 lanes
      // needs a second one): the chain prev -> proposal is 1-2 dependent DPP moves, not 4 and a select.
      // Both components of a flow move in ONE 64-bit DPP instruction (row_newbcast only: the two row_bcast:15 moves stay 32-bit): 5 issue slots
      // instead of 8.  t15 first -- its readers, the two row_bcast:15 moves, then stand two slots behind it (the scheduling barrier ties the order down).
      long p64 = __builtin_bit_cast(long, f2d{prev.x, prev.y}), c64 = __builtin_bit_cast(long, f2d{cnd.x, cnd.y});
      // row_newbcast:8 -> lane 15 (lanes 12-15; the other lanes are not used): the odd group's result, for the next row of 16
      long t64 = __builtin_amdgcn_mov_dpp(p64, 0x158, 0xF, 0x8, true);
      __builtin_amdgcn_sched_barrier(0);
      if (TOP != 0) {
        c64 = __builtin_amdgcn_update_dpp(c64, p64, 0x150, 0xF, 0x9, false);   // row_newbcast:0 -> lanes 0-3 (own) and 12-15 (the row above of the odd group)
      } else {
        // first band: no ring value.  Lanes 4-7 of row 0 take the row's own last result as well (masked out of the selection, but
        // evaluated: prev is only meaningful in lane 0 of a group, a wild value would send the wave through the out-of-window path
        // in every step); with every lane written by one of the moves there is no previous value to set up.
        c64 = __builtin_amdgcn_mov_dpp(p64, 0x150, 0xF, 0xB, false);
      }
      c64 = __builtin_amdgcn_update_dpp(c64, p64, 0x158, 0xF, 0x4, false);     // row_newbcast:8 -> lanes 8-11 (own, odd group)
      const f2d c2 = __builtin_bit_cast(f2d, c64), t2 = __builtin_bit_cast(f2d, t64);
      cnd = make_float2(c2.x, c2.y);
      const float2 t15 = make_float2(t2.x, t2.y);
      cnd.x = dpp<0x142, 0xE, 0x2>(cnd.x, t15.x);            // row_bcast:15 -> lanes 4-7 of rows 1..3 (lane 15 of the row of 16 above)
      cnd.y = dpp<0x142, 0xE, 0x2>(cnd.y, t15.y);
      // ---- the six proposal evaluations, one per lane ----
      // the pixel's image coordinates (exact small integers in fp32) come with its record
      const f2p posv = f2p{rc.x, rc.y};
      const float fpos = transposed ? rc.y : rc.x;   // along the step axis
      const float eC = rb.x, eCa = rb.w;
      const float2 rC = make_float2(rb.y, rb.z);   // the current flow C after its own gradient step (prepass); C itself where the pixel is not updated
      const bool gated = eC >= 0.0f;   // the pixel is updated (alpha0, alpha1 > 0.9): otherwise its record holds kKeepEnergy
      // previous pixel along the step axis = own result of the previous step; previous pixel across = DPP/ring.
      // Reference order is always "previous column, then previous row" (PixFlow.hpp:319-320 / :332-333).
      float2 fin = rC;   // (a step in which no pixel of the wave is updated: rC = C everywhere)
      float4 na, nb; float2 nc, np_ = rp;
      int hN = 0; unsigned long long tvN = tv;
      // next step's inputs (LDS): records (unconditional: past the chunk it reads a slot that is reloaded at the chunk
      // start anyway), producer counter, then the top value.  Issued behind the gather inside the evaluation below.
      lds_f4* rpn = (j + 1 < kChunk) ? recChunk + (j + 1) * (kRows * G::kRQ) : recNext;
      lds_cf2w* xyn = (G::kRQ == 3) ? (lds_cf2w*)(rpn + 2) : ((j + 1 < kChunk) ? xyChunk + (j + 1) * kRows : xyNext);
      lds_u64* tpn = (j + 1 < kChunk) ? topChunk + j * ((TOP == 1) ? kRows : 1) : topNext;
      // Sparse overlap (full-canvas inputs, CPU/StitchTool.cpp:17-33): when no pixel of this anti-diagonal is gated
      // the whole step is bookkeeping only (wave-uniform branch; an ungated pixel keeps its flow, PixFlow.hpp:317).
      if (!SPARSE || __any(gated)) {
      // a missing neighbour is evaluated anyway (its slot holds a finite stale flow) and masked out of the selection
      // The proposal of the previous pixel ALONG the step axis is missing at the first pixel of a row.  When that proposal is
      // L (not transposed) the record says so by itself: L is compared with eCa = kKeepEnergy there (no compare, no mask).
      // Transposed, it is T, which is compared with a value computed in the step: masked as before.
      const bool hasAlong = transposed ? (forward ? (fpos > 0.0f) : (fpos < fLast)) : true;
      const float eCL = transposed ? eC : eCa;
      const bool okL = transposed ? hasCross : hasAlong, okT = transposed ? hasAlong : hasCross;
      const float2 cand = cnd;
      int emin; float vmax;
      const f2p fdv = f2p{cand.x, cand.y} + f2p{addx, addy};
      float e = d_error_fast<TR, FWD, kWA, kWCp, false, G::kFollow ? 1 : 0>(g1, win, ob, W, H, wm2, hm2, fW, rW, cf, posv, ra.x, ra.y, ra.z, ra.w, fdv, emin, vmax, f2p{rp.x, rp.y});
      { // Only what the step uses is loaded: a loaded register nothing reads is handed out again by the register allocator at once,
        // and the hardware must then wait for the load in flight before the new value may be written (s_waitcnt right behind the
        // loads, ~50 cycles per step).  Transposed sweeps do not use Ea, the fourth float of the second quad.
        typedef float f3v __attribute__((ext_vector_type(3)));
        const f4v q0 = rpn[0];
        if (G::kFollow) { const f4v q2 = rpn[2]; nc = make_float2(q2.x, q2.y); np_ = make_float2(q2.z, q2.w); }
        else { const f2w q2 = *xyn; nc = make_float2(q2.x, q2.y); }
        na = make_float4(q0.x, q0.y, q0.z, q0.w);
        if (transposed) { const f3v q1 = *(__attribute__((address_space(3))) const f3v*)(rpn + 1); nb = make_float4(q1.x, q1.y, q1.z, 0.f); }
        else { const f4v q1 = rpn[1]; nb = make_float4(q1.x, q1.y, q1.z, q1.w); } }
      // the producer's counter and the top value of the NEXT step, read inside select_step (MID): ~16 slots in front of the step's publishing wait
      auto read_top_ahead = [&]() { if (TOP != 0) { hN = ld_cnt(topHead); tvN = __hip_atomic_load(tpn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } };
      fin = select_step<true, TR, TOP != 0>(e, eC, eCL, rC, cnd, okL, okT, rEps, cf.step, emin, vmax, read_top_ahead);
      asm volatile("" : "+v"(fin.x), "+v"(fin.y));   // finish the fast result before the branch: the range test then runs beside the division, not before it
      // Only pixels that will be updated count: the lanes of a pixel without data (gate <= 0) still run the arithmetic, and
      // there the inputs are blur tails of black borders (operands ~1e-40) -- their result is discarded two lines below.
      if (__builtin_expect(__any((emin < cf.guard_min || !(vmax <= 0x1p99f)) && gated), 0)) {
#ifdef PF_SWEEP_STATS
        ++statRedo;
#endif
        // an operand left the range where the fast forms are exact: the whole wave redoes the step with IEEE sqrt and division
        e = d_error2(g1, W, wm2, hm2, fW, cf, int(posv.x), int(posv.y), ra.x, ra.y, ra.z, ra.w, cand.x + addx, cand.y + addy);
        fin = select_step<false, TR>(e, eC, eCL, rC, cnd, okL, okT, rEps, cf.step, emin, vmax);
      }
      // (a pixel that is not updated keeps C through its record: kKeepEnergy, see d_make_record)
      } else {
      { // Only what the step uses is loaded: a loaded register nothing reads is handed out again by the register allocator at once,
        // and the hardware must then wait for the load in flight before the new value may be written (s_waitcnt right behind the
        // loads, ~50 cycles per step).  Transposed sweeps do not use Ea, the fourth float of the second quad.
        typedef float f3v __attribute__((ext_vector_type(3)));
        const f4v q0 = rpn[0];
        if (G::kFollow) { const f4v q2 = rpn[2]; nc = make_float2(q2.x, q2.y); np_ = make_float2(q2.z, q2.w); }
        else { const f2w q2 = *xyn; nc = make_float2(q2.x, q2.y); }
        na = make_float4(q0.x, q0.y, q0.z, q0.w);
        if (transposed) { const f3v q1 = *(__attribute__((address_space(3))) const f3v*)(rpn + 1); nb = make_float4(q1.x, q1.y, q1.z, 0.f); }
        else { const f4v q1 = rpn[1]; nb = make_float4(q1.x, q1.y, q1.z, q1.w); } }
      if (TOP != 0) { hN = ld_cnt(topHead); tvN = __hip_atomic_load(tpn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      }
      if (TOP != 0) {
        // Take the values read ahead BEFORE the stores below: LDS operations return in order, so a wait for them
        // at the top of the next step would also wait for this step's publishing stores.
        waitTop = __any(s + 1 + kBias >= hN);   // column s + 1 not yet there (every lane holds the same counter value)
        asm volatile("" : "+v"(tvN));   // the register PAIR as one operand: two 32-bit operands cost two v_mov per step to split and rejoin it
        tv = tvN;
      }
      prev = fin;   // valid in lane 0 of each group.  "no pixel" steps hand on their zero record: never used as a neighbour (masked / outside the image)
      // ---- publish: result ring (lane 0 of a group stores to the slot, the other lanes to a scratch slot of their own), then the step counter ----
      outChunk[j * kRows] = f2w{fin.x, fin.y};
      asm volatile("" ::: "memory");   // after the result (LDS operations of one wave execute in issue order)
      __hip_atomic_store(cntp, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" : "+v"(cntp));
      ra = na; rb = nb; rc = nc; rp = np_;
    }
  }
#ifdef PF_SWEEP_STATS
  if (lane == 0) {
    atomicAdd(&sm.statHits, statHits); atomicAdd(&sm.statSpins, statSpins);
    atomicAdd(&g_sweep_stats[0], (unsigned long long)nsteps);   // ([1] is counted where it happens: d_error_fast)
#ifdef PF_SWEEP_TRACE
    if (band < 1024) { g_band_trace[band][0] = statHits; g_band_trace[band][1] = statSpins; g_band_trace[band][2] = statSlowChunks; g_band_trace[band][3] = wall_clock64(); }
#endif
#ifdef PF_SWEEP_STATS_PRINT
    if (band < 8 || band % 32 == 1 || (!hasNext && !publishes)) printf("band %d xcc %d: %lld cycles, %lld in chunk-start waits, %d edge waits, %d spins, nsteps %d, IEEE-redo steps %d, out-of-window steps %d, slow chunk starts %d (%d spins; first check failed on rec %d tail %d pub %d next %d), step8 at %lld, end at %lld; since kernel entry: band start %lld, first chunk %lld, end %lld (10ns)\n", band, (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF),
           (long long)__builtin_readcyclecounter() - statT0, statWait, statHits, statSpins, nsteps, statRedo, statOOW, statSlowChunks, statChunkSpins, statFailRec, statFailTail, statFailPub, statFailNext, statR8, (long long)wall_clock64(), statB - sm.statEntry, statC0 - sm.statEntry, (long long)wall_clock64() - sm.statEntry);
#endif
  }
#endif
  return !dead;
}

// The current flow's own gradient step (PixFlow.hpp:322-341 with the pixel's own energies), IEEE operations in the reference's order.
__device__ __forceinline__ float2 own_gradient_step(float2 f, float e0, float ex, float ey, float step) {
  const float gx = (ex - e0) / kGradEpsilon, gy = (ey - e0) / kGradEpsilon;
  return make_float2(f.x - step * gx, f.y - step * gy);
}

// One record of the prepass (slot = linear index in wavefront order, see k_sweep_prep): shared by the prepass kernel and by
// the prepass blocks that ride inside the sweep launch (k_sweep2, MODE 2).
// ROWS = rows per band (8: latency / wide form, 32: throughput form).  RC: the record carries rC, the current flow after its own gradient
// step (three energies per pixel: what the 8-lane step needs); !RC (throughput form): it carries C itself and E(C) only -- that step
// takes the winner's gradient step itself, the current flow's included (one energy per pixel here instead of three).
// (band, s, r) = the record's band (counted from bandLo), step and row; `inside` = the slot exists (band < nbandsPad, s < nstepsPad).
template <int ROWS = kRows, bool RC = true>
__device__ __forceinline__ void d_make_record_at(int band, int s, int r, bool inside, const float2* __restrict__ g0, const float2* __restrict__ g1,
                                                 const float2* __restrict__ blurred, const uint8_t* __restrict__ gate, const float2* __restrict__ flow, int W,
                                                 int H, int forward, int transposed, float rW, const SolverCoef& cf, int uLo, int uHi, int bandLo, float4& a,
                                                 float4& b, float4& c) {
  const int LS = transposed ? H : W, LB = transposed ? W : H;   // extent along the step axis / across the bands
  const int ia = uLo + s - r, ib = (bandLo + band) * ROWS + r;
  // A pixel that is not updated (gate <= 0) carries E(C) = kKeepEnergy and rC = C: every proposal's energy is >= 0 (or NaN), so
  // the selection keeps rC = C -- the sweep's step needs no "if not gated keep C" of its own (two v_cndmask per step).
  // (third quad: (x, y) and -- for the latency form's flow-following window -- z = w = 0 for a pixel that is updated, NaN otherwise: the sweep's
  // loader adds the window offset (ox, oy) there, and a NaN offset means "never leaves the window", see d_error_fast<FOLLOW>)
  a = make_float4(0.f, 0.f, 0.f, 0.f); b = make_float4(kKeepEnergy, 0.f, 0.f, kKeepEnergy); c = make_float4(0.f, 0.f, __builtin_nanf(""), __builtin_nanf(""));
  if (inside && s - r >= 0 && ia < uHi && ia < LS && ib < LB) {
    const int cx = transposed ? ib : ia, cy = transposed ? ia : ib;   // position in sweep order
    const int x = forward ? cx : W - 1 - cx, y = forward ? cy : H - 1 - cy;
    const size_t idx = size_t(y) * W + x;
    const float2 f = flow[idx];
    b.y = f.x; b.z = f.y;             // rC = C unless the pixel is updated (below)
    c.x = float(x); c.y = float(y);   // the pixel's image coordinates (exact small integers)
    if (gate[idx]) {
      const float2 g = g0[idx], bl = blurred[idx];
      const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W);
      a = make_float4(g.x, g.y, bl.x, bl.y);
      c.z = 0.f; c.w = 0.f;
      const float e0 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, x, y, g.x, g.y, bl.x, bl.y, f.x, f.y);
      b.x = e0;
      if (RC) {
        const float e1 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, x, y, g.x, g.y, bl.x, bl.y, f.x + kGradEpsilon, f.y + 0.0f);
        const float e2 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, x, y, g.x, g.y, bl.x, bl.y, f.x + 0.0f, f.y + kGradEpsilon);
        const float2 rc0 = own_gradient_step(f, e0, e1, e2, cf.step);
        b.y = rc0.x; b.z = rc0.y;
      }
      b.w = (ia > 0) ? e0 : kKeepEnergy;   // E(C) as the proposal from the previous pixel ALONG the step axis sees it: unbeatable at the first pixel of a row (there is none)
    }
  }
}
// the same from a linear slot index (row fastest, then step, then band): two 64-bit divisions -- only the lab build's prepass blocks
// inside the sweep launch (MODE 2) still address their records this way
template <int ROWS = kRows, bool RC = true>
__device__ __forceinline__ void d_make_record(size_t tid, size_t total, const float2* __restrict__ g0, const float2* __restrict__ g1,
                                              const float2* __restrict__ blurred, const uint8_t* __restrict__ gate, const float2* __restrict__ flow, int W,
                                              int H, int forward, int transposed, int nstepsPad, float rW, const SolverCoef& cf, int uLo, int uHi, int bandLo, float4& a,
                                              float4& b, float4& c) {
  const int r = int(tid % ROWS);
  const int s = int((tid / ROWS) % nstepsPad);
  const int band = int(tid / (size_t(ROWS) * nstepsPad));
  d_make_record_at<ROWS, RC>(band, s, r, tid < total, g0, g1, blurred, gate, flow, W, H, forward, transposed, rW, cf, uLo, uHi, bandLo, a, b, c);
}

// ------------------------------------------------------------------------------------------------
// prepass: records in wavefront order for the ACTIVE window of the sweep -- the product's latency form (SOA): per band and chunk of 8 steps
// [64 first quads][64 second quads], record (s % 8) * 8 + r; the lab forms: rec[((band*nstepsPad + s)*8 + r)*3 + j] with the third quad (x, y, -, -):
// band counts from bandLo, step s handles sweep-order column uLo + s - r (columns [uLo, uHi)).
//   j=0: (I0x, I0y, blurred.x, blurred.y)   j=1: (E(C), rC.x, rC.y, Ea)   j=2: (x, y, -, -)     (the step reads 16 + 16 + 8 bytes)
//   rC = C after its own gradient step, C - 0.5 * ((E(C+dx), E(C+dy)) - E(C)) / eps (IEEE operations: what the sweep's exact fast
//   forms reproduce bit for bit for the two proposals) -- the result of the pixel if neither proposal beats E(C);
//   (x, y) = the pixel's image coordinates as floats; Ea = E(C), or kKeepEnergy at the first pixel of a row in sweep order
//   (what the proposal of the previous pixel along the axis is compared with: no such pixel there, no mask needed).
//   A pixel that is not updated (alpha <= 0.9: keep C) and a (step,row) slot without a pixel carry kKeepEnergy in both
//   energies and rC = C; "updated" is E(C) >= 0.
// When the window does not start at the first band, the row above it never changes during this sweep: its flow
// is written as the granule row the first workgroup's poller reads (top0).
// ------------------------------------------------------------------------------------------------
// SOA (the product's latency form, round 5): the record stream holds TWO quads per record -- the third, (x, y, window offset), is a function of
// the slot and of E(C)'s sign and is formed by the sweep's loader -- laid out per chunk of 8 steps x 8 rows as [64 first quads][64 second
// quads]: a wave of this kernel IS one chunk, so its two stores are 1 KB runs without the LDS stage, and a loader lane reads its own record's
// two quads with two coalesced loads.  32 instead of 48 bytes per record written here and read there (the prepass is bandwidth-bound at the large levels).
template <int ROWS, bool RC, bool SOA = false>
__global__ __launch_bounds__(256) void k_sweep_prep(const float2* __restrict__ g0, const float2* __restrict__ g1, const float2* __restrict__ blurred,
                                                    const uint8_t* __restrict__ gate, const float2* __restrict__ flow, int W, int H, int forward,
                                                    int transposed, int nstepsPad, int nbandsPad, float rW, float4* __restrict__ rec, int uLo, int uHi,
                                                    int bandLo, unsigned long long* __restrict__ top0, size_t bstride, SolverCoef cf) {
  {   // blockIdx.z = pair of a batched launch (every pointer is pair 0's)
    const size_t bo = size_t(blockIdx.z) * bstride;
    PF_BOFF(g0, bo); PF_BOFF(g1, bo); PF_BOFF(blurred, bo); PF_BOFF(gate, bo); PF_BOFF(flow, bo); PF_BOFF(rec, bo);
    if (top0 != nullptr) PF_BOFF(top0, bo);
  }
  // Which of the block's 256 record slots this thread computes.  Slots are in wavefront order (row fastest).  With 8 rows per band a
  // wave's 64 slots are 8 rows x 8 consecutive columns: 64-byte runs of every input plane.  With 32 rows they would be 32 rows x 2
  // columns (16-byte runs: the throughput form's first prepass ran 1.5x LONGER than the latency form's with a third of the arithmetic),
  // so the block's 8 steps x 32 rows are dealt out column-fastest instead; the LDS stage below puts the records back in slot order.
  // Grid: x = the blocks of one band (256 / ROWS steps each), y = band -- until round 4 the grid was linear and every thread took its
  // (band, step, row) out of a 64-bit slot index with two emulated divisions, a third of the kernel's vector instructions.
  constexpr int kStepsPerBlock = 256 / ROWS;
  const unsigned lt = (ROWS == 32) ? (threadIdx.x & 7) * 32 + (threadIdx.x >> 3) : threadIdx.x;
  const int band = blockIdx.y, s = int(blockIdx.x) * kStepsPerBlock + int(lt / ROWS), r = int(lt % ROWS);
  const size_t tid = (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + lt;   // any numbering of the launch's threads serves top0
  if (top0 != nullptr && tid < size_t(uHi - uLo)) {
    const int ia = uLo + int(tid), ib = bandLo * ROWS - 1;
    const int cx = transposed ? ib : ia, cy = transposed ? ia : ib;
    const int x = forward ? cx : W - 1 - cx, y = forward ? cy : H - 1 - cy;
    top0[tid] = pack2(flow[size_t(y) * W + x]);
  }
  // (no early return: the block stages its 256 records in LDS so that they leave as three fully coalesced 4 KB stores instead
  // of 16-byte pieces at a 48-byte stride)
  // (throughput form: 32-byte records -- the step forms the pixel's coordinates itself, the record stream is what bounds this kernel there)
  constexpr int kQuads = RC ? 3 : 2;
  float4 a, b, c;
  d_make_record_at<ROWS, RC>(band, s, r, band < nbandsPad && s < nstepsPad, g0, g1, blurred, gate, flow, W, H, forward, transposed, rW, cf, uLo, uHi, bandLo, a, b, c);
  if (SOA) {
    static_assert(!SOA || (ROWS == 8 && RC), "chunk layout of the latency form");
    // float4 index in the band's stream: chunk * 128 + quad * 64 + (record in chunk); lt = 64 * (chunk in block) + (record in chunk)
    const size_t bandBase2 = size_t(band) * nstepsPad * ROWS * 2;
    const unsigned o = (unsigned(blockIdx.x) * 4u + (lt >> 6)) * 128u + (lt & 63u);
    if (s < nstepsPad) { rec[bandBase2 + o] = a; rec[bandBase2 + o + 64u] = b; }   // (nstepsPad is a whole number of chunks; s is wave-uniform up to the chunk)
    return;
  }
  __shared__ float4 stage[256 * kQuads];
  stage[lt * kQuads + 0] = a; stage[lt * kQuads + 1] = b;
  if (kQuads == 3) stage[lt * kQuads + 2] = c;
  __syncthreads();
  // the block's records are contiguous in the band's stream (the last block of a band may reach past its end: nstepsPad is a multiple of 8, not of 32)
  const size_t bandBase = size_t(band) * nstepsPad * ROWS * kQuads;   // in float4 units
  const unsigned inBand = unsigned(blockIdx.x) * 256u * kQuads, limBand = unsigned(nstepsPad) * ROWS * kQuads;
#pragma unroll
  for (int k = 0; k < kQuads; ++k) {
    const unsigned o = inBand + unsigned(k) * 256u + threadIdx.x;
    if (o < limBand) rec[bandBase + o] = stage[k * 256 + threadIdx.x];
  }
}


// ------------------------------------------------------------------------------------------------
// Loader wave of the WIDE form (SwWide): serves the PAIR of adjacent bands A = 2l, B = 2l + 1 of its workgroup -- their record
// streams (one ring each) and the ONE gather window they share (33 = 8 + 8 + 2 * 8 + 1 texel rows, the same 64-column ring
// along the step axis).  Window batch b = the 8 texel columns [8b-16, 8b-8) (relative to uLo) x 33 rows; a band working on
// chunk j (steps 8j .. 8j+7) reads batches j .. j+4.  The window follows the LEADING band A: batch j + 4 is loaded together
// with A's chunk j (batches 0..3 in the first round), which overwrites batch j - 4.  Hence two rules beyond the rings' room:
//   * A's chunk j is loaded only when B has finished every step of its chunks <= j - 4 (B still reads batch j - 4 otherwise),
//     i.e. outHead[B] >= 8j - 24.  B runs 8-9 steps behind A by construction and A's records are at most 16 steps ahead of A,
//     so in steady state outHead[B] ~ 8j - 17: the rule never binds unless B is held up.
//   * B's chunk j is published only after A's chunk j (its batches <= j + 4 are then in the window).
// No deadlock: A blocked by the first rule holds records up to step 8j - 1, B needs A's results up to its own step + 8 only.
// One chunk per band and round (16-step rings: two chunks): a round is one HBM round trip for both bands.
// ------------------------------------------------------------------------------------------------
template <class G, bool TR, bool FWD>
__device__ __forceinline__ void sweep_loader_pair(SmemT<G>& sm, const float4* __restrict__ rec, const float2* __restrict__ g1, int* __restrict__ ctrl, int W, int H,
                                                  int nsteps, int nstepsPad, int l, int nact, int band0, int bandLo, int uLo, int LSv) {
  static_assert(G::kBPW == 2, "pair loader");
  constexpr int kRS = G::kRS, kWA = G::kWA;
  constexpr int kKT = (8 * kWA + 63) / 64;   // window texels per lane and batch (5)
  const int lane = threadIdx.x & 63;
  const int wA = 2 * l, wB = 2 * l + 1;
  if (wA >= nact) return;
  const bool hasB = wB < nact;
  const int LS = TR ? H : W, LB = TR ? W : H;
  int tc[kKT], ta[kKT]; bool tvalid[kKT];
#pragma unroll
  for (int k = 0; k < kKT; ++k) {
    const int t = lane + 64 * k;
    tc[k] = TR ? t / kWA : (t & 7); ta[k] = TR ? t % kWA : (t >> 3); tvalid[k] = t < 8 * kWA;
  }
  float2* winw = &sm.win[l][0][0];
  const int v0 = (bandLo + band0 + wA) * kRows - kRad;   // window origin across the bands
  auto win_addr = [&](int b, int k, int& slot) -> const float2* {   // texel k of this lane in batch b
    const int u = uLo + 8 * b - 16 + tc[k], v = v0 + ta[k];   // absolute sweep-order texel
    slot = ta[k] * kWCp + (u & (kWC - 1));
    if (!tvalid[k] || u < 0 || u >= LS || v < 0 || v >= LB) return nullptr;
    const int cxc = TR ? v : u, cyc = TR ? u : v;
    const int x = FWD ? cxc : W - 1 - cxc, y = FWD ? cyc : H - 1 - cyc;
    return g1 + (y * W + x);
  };
  auto win_store = [&](int slot, float2 v) {   // ring column 0 also goes to the slot behind column 63 (kWCp)
    winw[slot] = v;
    if (slot % kWCp == 0) winw[slot + kWC] = v;
  };
  const float4* recA = rec + size_t(band0 + wA) * nstepsPad * (kRows * 3);
  const float4* recB = recA + size_t(nstepsPad) * (kRows * 3);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // a chunk's records arrive as 192 float4 (64 records x 3 quads, lane i holds quads i, i + 64, i + 128); in the LDS ring quads 0 and 1
  // of record n go to rec[..][n][q], the first half of quad 2 to recxy[..][n]
  int dn[3], dq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int i = lane + 64 * k; dn[k] = i / 3; dq[k] = i % 3; }
  auto rec_store = [&](int wv_, int ring, const float4& q0_, const float4& q1_, const float4& q2_) {
    float4* d4 = &sm.rec[wv_][ring][0][0];
    float2* d2 = &sm.recxy[wv_][ring][0];
    const float4 v[3] = {q0_, q1_, q2_};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (dq[k] < 2) d4[dn[k] * 2 + dq[k]] = v[k];
      else d2[dn[k]] = make_float2(v[k].x, v[k].y);
    }
  };
  int rhA = 0, rhB = 0, idle = 0;
  bool first = true;
  for (;;) {
    const int ohA = first ? 0 : ld_cnt(&sm.outHead[wA]);
    const int ohB = (first || !hasB) ? 0 : ld_cnt(&sm.outHead[wB]);
    const bool ldA = rhA < nsteps && (rhA + kChunk - ohA <= kRS) && (!hasB || ohB >= rhA - 24);
    const int rhA2 = rhA + (ldA ? kChunk : 0);
    const bool ldB = hasB && rhB < nsteps && (rhB + kChunk - ohB <= kRS) && (rhB + kChunk <= rhA2);
    float4 a0 = z4, a1 = z4, a2 = z4, b0 = z4, b1 = z4, b2 = z4;
    float2 wv[kKT]; int ws[kKT]; bool wok[kKT];
#pragma unroll
    for (int k = 0; k < kKT; ++k) { wv[k] = make_float2(0.f, 0.f); ws[k] = 0; wok[k] = false; }
    if (ldA) {
      const float4* src = recA + size_t(rhA) * (kRows * 3);
      a0 = src[lane]; a1 = src[lane + 64]; a2 = src[lane + 128];
      const int b = rhA / kChunk + 4;
#pragma unroll
      for (int k = 0; k < kKT; ++k) {
        const float2* q = win_addr(b, k, ws[k]);
        wok[k] = q != nullptr;
        if (wok[k]) wv[k] = *q;
      }
    }
    if (ldB) {
      const float4* src = recB + size_t(rhB) * (kRows * 3);
      b0 = src[lane]; b1 = src[lane + 64]; b2 = src[lane + 128];
    }
    if (first) {   // batches 0..3 (0 and 1 lie before the window: they exist when the window does not start at the image border)
      float2 pv[4][kKT]; int ps[4][kKT]; bool pk[4][kKT];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < kKT; ++k) {
          pv[b][k] = make_float2(0.f, 0.f);
          const float2* q = win_addr(b, k, ps[b][k]);
          pk[b][k] = q != nullptr;
          if (pk[b][k]) pv[b][k] = *q;
        }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < kKT; ++k) if (pk[b][k]) win_store(ps[b][k], pv[b][k]);
      first = false;
    }
    bool progress = false;
    if (ldA) {
      rec_store(wA, rhA % kRS, a0, a1, a2);
#pragma unroll
      for (int k = 0; k < kKT; ++k) if (wok[k]) win_store(ws[k], wv[k]);
      rhA += kChunk;
      st_cnt(&sm.recHead[wA], rhA);
      progress = true;
    }
    if (ldB) {
      rec_store(wB, rhB % kRS, b0, b1, b2);
      rhB += kChunk;
      st_cnt(&sm.recHead[wB], rhB);
      progress = true;
    }
    if (rhA >= nsteps && (!hasB || rhB >= nsteps)) break;
    if (progress) idle = 0;
    else {
      __builtin_amdgcn_s_sleep(4);
      if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SwLatency, 704 threads: waves 0-3 compute (one band of 8 rows each), waves 4-7 load records and window texels (one per
// compute wave), wave 8 publishes the workgroup's last row as granules, wave 9 polls the previous workgroup's
// granules, wave 10 drains results.  Waves land on SIMD (wave % 4): every compute wave shares its SIMD with its
// own loader; three waves per SIMD cap the kernel at 168 VGPRs.
// SwWide, 960 threads: waves 0-7 compute (two per SIMD), waves 8-11 load for a PAIR of adjacent bands each (their records,
// and the one 33-row window the two share), waves 12 / 13 / 14 publish, poll, drain: four waves per SIMD, 128 VGPRs.
// ------------------------------------------------------------------------------------------------
// MODE 0: records come from k_sweep_prep (launched in front).  MODE 1: the loader waves compute them (experiment, slower).
// MODE 2: the prepass rides INSIDE this launch -- blocks [nwgSweep, gridDim.x) compute the records in wavefront order (sweep
// workgroup 0's first) and hand them over with the write-through + counter + acquire protocol of cdna_hip_programming.md G16/R1;
// the first bands start a few microseconds after the launch instead of after a whole prepass kernel.
template <class G, bool TR, bool FWD, bool SPARSE, int MODE>
__global__ __launch_bounds__(G::kThreads) void k_sweep2(const float4* __restrict__ rec, const float2* __restrict__ g1, float2* __restrict__ flow,
                                                unsigned long long* __restrict__ boundary, int* __restrict__ ctrl, int W, int H,
                                                int nstepsPad, int nbands, float rW, float rEps, int uLo, int LSv, int bandLo, long long budgetTicks,
                                                const float2* __restrict__ g0, const float2* __restrict__ blurred, const uint8_t* __restrict__ gate,
                                                int nwgSweep, int* __restrict__ prepcnt, size_t bstride, SolverCoef cf) {
  {   // blockIdx.z = pair of a batched launch: an independent sweep with its own ticket, granules and records
    const size_t bo = size_t(blockIdx.z) * bstride;
    PF_BOFF(rec, bo); PF_BOFF(g1, bo); PF_BOFF(flow, bo); PF_BOFF(boundary, bo); PF_BOFF(ctrl, bo); PF_BOFF(g0, bo); PF_BOFF(blurred, bo); PF_BOFF(gate, bo);
    if (prepcnt != nullptr) PF_BOFF(prepcnt, bo);
  }
  constexpr int kWaves = G::kWaves, kRS = G::kRS, kOS = G::kOS, kLoadAhead = G::kLoadAhead, kLoaders = G::kLoaders;
  static_assert(MODE == 0 || G::kBPW == 1, "the experimental record paths exist in the latency form only");
  using Smem = SmemT<G>;
  if (MODE == 2 && int(blockIdx.x) >= nwgSweep) {
    // ======================= prepass block (MODE 2): 704 records, written through to memory, then counted in =======================
    const int slotsPerWG = kWaves * nstepsPad * kRows;
    const size_t total = size_t(nwgSweep) * slotsPerWG;
    const size_t first = size_t(int(blockIdx.x) - nwgSweep) * blockDim.x, slot = first + threadIdx.x;
    float4 a, b, c;
    d_make_record(slot, total, g0, g1, blurred, gate, flow, W, H, FWD ? 1 : 0, TR ? 1 : 0, nstepsPad, rW, cf, uLo, uLo + LSv, bandLo, a, b, c);
    if (slot < total) {
      // sc1 (write-through) 16-byte stores through a buffer descriptor built from wave-uniform values (G16 R1)
      typedef unsigned int u4v __attribute__((ext_vector_type(4)));
      const unsigned long long base = reinterpret_cast<unsigned long long>(rec);
      const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(base)), hi = __builtin_amdgcn_readfirstlane(unsigned(base >> 32));
      void* ubase = reinterpret_cast<void*>((unsigned long long)lo | ((unsigned long long)hi << 32));
      const unsigned nbytes = __builtin_amdgcn_readfirstlane(unsigned(total * 48));
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, int(nbytes), 0x00020000);
      const unsigned off = unsigned(slot) * 48u;
      __builtin_amdgcn_raw_buffer_store_b128(u4v{__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)}, rsrc, off, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(u4v{__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)}, rsrc, off + 16u, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(u4v{__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w)}, rsrc, off + 32u, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                    // ... before ONE lane counts the block in
    if (threadIdx.x == 0 && first < total) {
      const size_t last = (first + blockDim.x < total ? first + blockDim.x : total) - 1;
      const int k0 = int(first / slotsPerWG), k1 = int(last / slotsPerWG);
      const int n = int(last - first + 1), n0 = k0 == k1 ? n : int(size_t(k0 + 1) * slotsPerWG - first);
      __hip_atomic_fetch_add(&prepcnt[k0], n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k1 != k0) __hip_atomic_fetch_add(&prepcnt[k1], n - n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // FUSED (MODE 1): FUSED PREPASS -- the loader waves compute the records themselves (same expressions as k_sweep_prep) while they
  // run ahead of the wavefront: no prepass launch in front of the sweep, and the 2 x 48 B per level-pixel of record traffic
  // through HBM (written by the prepass, read back here) disappears.  A pixel's own flow C is read before its step is
  // computed and overwritten (by the drainer) only afterwards, so reading it from the plane being updated is safe.
  // Active window of this sweep (everything outside it holds pixels that are not updated and keeps its flow):
  // nbands bands starting at band bandLo, sweep-order columns [uLo, uLo + LSv) along the step axis.  Steps, ring
  // indices and granule columns are relative to uLo; image coordinates are formed from uLo + relative column.
  constexpr int transposed = TR ? 1 : 0, forward = FWD ? 1 : 0;
  __shared__ Smem sm;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
#ifdef PF_SWEEP_STATS
  const long long tEntry = wall_clock64();
#endif
  if (tid == 0) {
    sm.wg = atomicAdd(&ctrl[0], 1);
    sm.bndHead = 0; sm.abort = 0; sm.pubTail = 0; sm.statHits = 0; sm.statSpins = 0;
    sm.deadline = (long long)wall_clock64() + budgetTicks;
  }
  if (tid < kWaves) { sm.recHead[tid] = 0; sm.outHead[tid] = 0; sm.outTail[tid] = 0; }
#ifdef PF_EXPERIMENTS
  poison_window(&sm.win[0][0][0], int(sizeof(sm.win) / sizeof(float2)));
#endif
  __syncthreads();
  const int wg = sm.wg;
  const int LS = transposed ? H : W, LB = transposed ? W : H;   // extent along the step axis / across the bands
  const int nsteps = nstepsPad;   // LS + kRows - 1 rounded up to whole chunks: the padding steps carry "no pixel" records
  const int band0 = wg * kWaves;                                          // first band of this workgroup, relative to bandLo
  const int nact = (nbands - band0) < kWaves ? (nbands - band0) : kWaves;  // active compute waves in this workgroup
  const bool publishes = band0 + kWaves < nbands;                         // another workgroup follows (then nact == kWaves)
  const bool staticTop = bandLo > 0;                                      // the row above the window exists and never changes: prepass wrote it as granule row 0

  if (wave < kWaves) {
    // ======================= compute wave: band of 8 rows =======================
    if (wave >= nact) return;
    __builtin_amdgcn_s_setprio(kPrioCompute);   // the helper waves share SIMDs with compute waves: compute wins issue arbitration
#ifdef PF_SWEEP_STATS
    sm.statEntry = tEntry;
#endif
    const int top = (wave > 0) ? 1 : ((wg > 0 || staticTop) ? 2 : 0);   // where row 0's top neighbour comes from
    const int band = bandLo + band0 + wave;                               // absolute band index
    bool ok;
    if (top == 1) ok = compute_band<G, 1, TR, FWD, SPARSE>(sm, g1, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    else if (top == 2) ok = compute_band<G, 2, TR, FWD, SPARSE>(sm, g1, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    else ok = compute_band<G, 0, TR, FWD, SPARSE>(sm, g1, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    if (!ok) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifdef PF_SWEEP_STATS_PRINT   // (stage entry only: ctrl[2..3] belong to the next sweep in a whole solve)
    if (lane == 0) { atomicAdd(&ctrl[2], atomicExch(&sm.statHits, 0)); atomicAdd(&ctrl[3], atomicExch(&sm.statSpins, 0)); }
#endif
    return;
  }

  // helper waves: below the compute waves (3), above another kernel's waves that share this CU -- since the two directions run
  // out of phase, the other direction's Gaussians / medians / prepass sit on the sweep's SIMDs (strip -0.12 ms, 9000x4000 pair -0.4 ms)
  __builtin_amdgcn_s_setprio(kPrioHelper);
  if constexpr (G::kBPW == 2) {
    if (wave >= kWaves && wave < kWaves + kLoaders) {
      sweep_loader_pair<G, TR, FWD>(sm, rec, g1, ctrl, W, H, nsteps, nstepsPad, wave - kWaves, nact, band0, bandLo, uLo, LSv);
      return;
    }
  }
  if constexpr (G::kBPW == 1) if (wave >= kWaves && wave < 2 * kWaves) {
    // ======================= loader of compute wave w: records + gather window HBM -> LDS, up to kRS steps ahead =======================
    // THE WINDOW FOLLOWS THE FLOW (round 5).  Until round 4 the window held the texels within +-8 of the band's pixels, so a proposal
    // was served from LDS only while |flow| <= 7 -- every larger one sent its whole wave to HBM (x8 displacement: 63 % of the steps, dense
    // pair 46 -> 68 ms).  Now the window of chunk j (steps 8j .. 8j + 7) is centred on pixel + o(j), o(j) = the blurred flow at the chunk's
    // centre pixel rounded to integers (the flow is median-filtered and diffused at every level: inside 8 rows x 15 columns it stays
    // within a pixel or two of that), moving by at most one texel per chunk and axis:
    //   * the LDS window is a TORUS addressed by the texel's IMAGE coordinates (round 6; until then: sweep-order coordinates relative to the band):
    //     slot = ((image coordinate across the bands) & 31) * stride + ((image coordinate along the step axis) & 63)
    //     (ring row 0 again behind row 31, ring column 0 again behind column 63: a 2 x 2 footprint never wraps);
    //   * chunk j needs columns [uLo + 8j - 15 + ou, uLo + 8j + 15 + ou] x rows [vb - 8 + ov, vb + 15 + ov] ((ou, ov) = o(j) in sweep order):
    //     the loader keeps a column front and loads the 7 / 8 / 9 new columns of the chunk's 24 rows (8 + the change of ou), and, when ov
    //     moved, the ONE new row over the columns already present; what it overwrites (column - 64, row -+ 32) left every active chunk's
    //     rectangle long ago (the loader is at most 4 chunks ahead: 31 + 24 + 3 columns < 64, 24 + 3 rows < 32);
    //   * the window's offset (ox, oy) travels with the pixel's record (third quad, z / w: patched in here), so the step's test is
    //     |flow - offset| <= 7 (d_error_fast<FOLLOW>): same texels as the HBM path => same bits, whatever the offsets are.
    // One loader per compute wave: its in-order memory queue holds nothing but this band's loads, and a round
    // (issue up to kLoadAhead chunks, wait once, registers -> LDS, publish) costs one HBM round trip.
    const int w = wave - kWaves;
    if (w >= nact) return;
    float2* winw = &sm.win[w][0][0];
    const int vb = (bandLo + band0 + w) * kRows;   // the band's first row (column, transposed) in sweep order
    auto tex_ptr = [&](int u, int v) -> const float2* {   // texel (u, v) in sweep order; nullptr outside the image (never sampled: samples are clamped)
      if (u < 0 || u >= LS || v < 0 || v >= LB) return nullptr;
      const int cxc = TR ? v : u, cyc = TR ? u : v;
      const int x = FWD ? cxc : W - 1 - cxc, y = FWD ? cyc : H - 1 - cyc;
      return g1 + (y * W + x);
    };
    // ring slot of texel (u, v) (sweep order): by its IMAGE coordinates along the step axis / across the bands (d_error_fast<FOLLOW = 1>, kAbs)
    auto ring_rc = [&](int u, int v, int& rr, int& cc) {
      rr = (FWD ? v : LB - 1 - v) & (kWRing - 1); cc = (FWD ? u : LS - 1 - u) & (kWC - 1);
    };
    auto win_store = [&](int u, int v, float2 val) {
      int rr, cc; ring_rc(u, v, rr, cc);
      float2* q = winw + rr * kWCp + cc;
      q[0] = val;
      if (cc == 0) q[kWC] = val;
      if (rr == 0) { q[kWRing * kWCp] = val; if (cc == 0) q[kWRing * kWCp + kWC] = val; }
    };
    // the same for the block of new columns, whose duplicates are rare per texel (1 in 64 / 1 in 32): wave-uniform guards around them
    // (dupc: the block's columns contain ring column 0; dupr: its rows contain ring row 0) keep the usual store at two address instructions + one write
    auto win_store_block = [&](int u, int v, float2 val, bool dupc, bool dupr) {
      int rr, cc; ring_rc(u, v, rr, cc);
      float2* q = winw + rr * kWCp + cc;
      q[0] = val;
      if (dupc) { if (cc == 0) q[kWC] = val; }
      if (dupr) { if (rr == 0) { q[kWRing * kWCp] = val; if (cc == 0) q[kWRing * kWCp + kWC] = val; } }
    };
    // image index of texel (u, v) = iC + iA * u + iB * v (mirrored for the backward sweep): the block's addresses are one wave-uniform base + a
    // per-lane constant, and a block that lies inside the image (nearly all do) needs no bounds tests
    const int iA = TR ? (FWD ? W : -W) : (FWD ? 1 : -1), iB = TR ? (FWD ? 1 : -1) : (FWD ? W : -W), iC = FWD ? 0 : W * H - 1;
    // a block of 8 columns x 24 rows: texel t = lane + 64 k (k < 3), dealt out so that a wave's loads run along memory
    constexpr int kRectRows = kRows + 2 * kRad;   // 24
    int tdc[3], tdr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const int t = lane + 64 * k; tdc[k] = TR ? t / kRectRows : (t & 7); tdr[k] = TR ? t % kRectRows : (t >> 3); }
    int tio[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) tio[k] = iA * tdc[k] + iB * tdr[k];
    // rounded blurred flow at the centre pixel of chunk (obase + lane): one gather serves 64 chunks
    // (kept as integers: everything per chunk below is wave-uniform integer arithmetic, i.e. work for the scalar unit, not for the vector
    // pipe this wave shares with its band's compute wave)
    int ocx = 0, ocy = 0; int obase = -(1 << 20);
    auto refill_offsets = [&](int jb) {
      obase = jb;
      int ia = uLo + 8 * (jb + lane) + 3 - kRows / 2;
      ia = ia < uLo ? uLo : ia; ia = ia > uLo + LSv - 1 ? uLo + LSv - 1 : ia; ia = ia > LS - 1 ? LS - 1 : ia; ia = ia < 0 ? 0 : ia;
      int ibc = vb + kRows / 2; ibc = ibc > LB - 1 ? LB - 1 : ibc;
      const int cxc = TR ? ibc : ia, cyc = TR ? ia : ibc;
      const int x = FWD ? cxc : W - 1 - cxc, y = FWD ? cyc : H - 1 - cyc;
      const float2 bl = blurred[y * W + x];
      const float rx = __builtin_rintf(bl.x), ry = __builtin_rintf(bl.y);
      ocx = (fabsf(rx) < 1.0e6f) ? int(rx) : 0;   // NaN / inf / absurd: no offset
      ocy = (fabsf(ry) < 1.0e6f) ? int(ry) : 0;
    };
    // image-axis offset (ox, oy) -> sweep order (mirrored for the backward sweep), cut back so that EVERY window centre of chunk j lies inside
    // the image (d_error_fast<FOLLOW> tests the sample before its clamp to the image): the chunk's pixels sit in columns [uLo + 8j - 7,
    // uLo + 8j + 7] (clipped to the image) and rows [vb, vb + 7].  The cut only binds at the image borders; there the offset may move by more
    // than one texel per chunk, towards the pixels: columns that are present.
    auto sweep_offsets = [&](int j, int& ox, int& oy, int& ou, int& ov) {
      ou = TR ? (FWD ? oy : -oy) : (FWD ? ox : -ox);
      ov = TR ? (FWD ? ox : -ox) : (FWD ? oy : -oy);
      int plo = uLo + 8 * j - (kRows - 1), phi = uLo + 8 * j + (kChunk - 1);
      plo = plo < 0 ? 0 : (plo > LS - 1 ? LS - 1 : plo); phi = phi < 0 ? 0 : (phi > LS - 1 ? LS - 1 : phi);
      const int vhi = vb + kRows - 1 > LB - 1 ? LB - 1 : vb + kRows - 1;
      ou = ou < -plo ? -plo : ou; ou = ou > LS - 1 - phi ? LS - 1 - phi : ou;
      ov = ov < -vb ? -vb : ov; ov = ov > LB - 1 - vhi ? LB - 1 - vhi : ov;
      const int sx = TR ? ov : ou, sy = TR ? ou : ov;   // back to image axes
      ox = FWD ? sx : -sx; oy = FWD ? sy : -sy;
    };
    constexpr bool fused = MODE == 1;
    constexpr bool soa = MODE == 0;   // the product's record stream: 32-byte records, [64 first quads][64 second quads] per chunk (k_sweep_prep<.., SOA>)
    const float4* recw = fused ? nullptr : rec + size_t(band0 + w) * nstepsPad * (kRows * (soa ? 2 : 3));
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W);
    // fused prepass: this lane's slot inside a chunk is (step offset lane >> 3, row lane & 7)
    const int lj = lane >> 3, lr = lane & 7;
    const int lib = (bandLo + band0 + w) * kRows + lr;       // position across the bands (absolute)
    // record stream: lane i of a chunk's three loads holds quads i, i + 64, i + 128 of its 192; quad q is part q % 3 of record q / 3 -- the
    // third part (x, y, -, -) gets the window OFFSET (ox, oy) in z / w
    bool isC[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) isC[k] = (lane + 64 * k) % 3 == 2;
    int rh = 0, idle = 0;
    if (MODE == 2) {
      // the records of this workgroup's four bands are complete when its counter has reached their number: ONE word polled
      // relaxed, ONE agent acquire after the match (drops this CU's stale L1 lines), then plain loads (G16 R1)
      const int need = kWaves * nstepsPad * kRows;
      int spins = 0;
      while (__hip_atomic_load(&prepcnt[wg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != need) {
        __builtin_amdgcn_s_sleep(2);
        if (spin_expired(spins, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- chunk 0's whole rectangle (31 columns x 24 rows, four blocks in one round trip) ----
    int pox, poy; int front, pov;
    {
      refill_offsets(0);
      pox = __builtin_amdgcn_readfirstlane(ocx); poy = __builtin_amdgcn_readfirstlane(ocy);
      int ou, ov; sweep_offsets(0, pox, poy, ou, ov);
      const int c0 = uLo - 15 + ou, c1 = uLo + 16 + ou, r0 = vb - kRad + ov;
      float2 pv[4][3]; int pu[4][3], pvv[4][3]; bool pk[4][3];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          pu[b][k] = c0 + 8 * b + tdc[k]; pvv[b][k] = r0 + tdr[k];
          const float2* q = pu[b][k] < c1 ? tex_ptr(pu[b][k], pvv[b][k]) : nullptr;
          pk[b][k] = q != nullptr; pv[b][k] = make_float2(0.f, 0.f);
          if (pk[b][k]) pv[b][k] = *q;
        }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) if (pk[b][k]) win_store(pu[b][k], pvv[b][k], pv[b][k]);
      front = c1; pov = ov;
    }
    bool first = true;
    for (;;) {
      const int oh = first ? 0 : ld_cnt(&sm.outHead[w]);
      // ring full (the usual state: the loader runs kRS steps ahead): a SHORT idle iteration -- the body below costs a few hundred
      // predicated-off instructions per pass, issued on the compute wave's own SIMD
      if (!first && rh + kChunk - oh > kRS) {
        __builtin_amdgcn_s_sleep(kLoaderIdleSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        continue;
      }
      first = false;
      float4 va[kLoadAhead], vb4[kLoadAhead], vc[kLoadAhead];
      bool ld[kLoadAhead];
      float cox[kLoadAhead], coy[kLoadAhead]; bool cdupc[kLoadAhead], cdupr[kLoadAhead], cwhole[kLoadAhead];
      // window work of a chunk: the block of new columns (3 texels per lane), a ninth column (lanes 0-23), one new row (one texel per lane)
      float2 wv[kLoadAhead][5]; int wu[kLoadAhead][5], wvv[kLoadAhead][5]; bool wok[kLoadAhead][5];
      // fused prepass, phase 1: the inputs of up to kLoadAhead chunks are requested together (one round trip)
      float2 qf[kLoadAhead], qg[kLoadAhead], qb[kLoadAhead]; int qgate[kLoadAhead], qx[kLoadAhead], qy[kLoadAhead], ia_of[kLoadAhead]; bool qvalid[kLoadAhead];
#pragma unroll
      for (int c = 0; c < kLoadAhead; ++c) {
        const int r0 = rh + c * kChunk;
        va[c] = z4; vb4[c] = z4; vc[c] = z4; cox[c] = 0.f; coy[c] = 0.f; cdupc[c] = true; cdupr[c] = true; cwhole[c] = false;
        ld[c] = r0 < nsteps && (r0 + kChunk - oh <= kRS);
        qf[c] = make_float2(0.f, 0.f); qg[c] = qf[c]; qb[c] = qf[c]; qgate[c] = 0; qx[c] = 0; qy[c] = 0; ia_of[c] = 0; qvalid[c] = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) { wv[c][k] = make_float2(0.f, 0.f); wu[c][k] = 0; wvv[c][k] = 0; wok[c][k] = false; }
        if (ld[c]) {
          if (!fused) {
            if (soa) {
              const float4* src = recw + size_t(r0) * (kRows * 2);   // chunk r0 / 8: 128 quads
              va[c] = src[lane]; vb4[c] = src[lane + 64];            // this lane's own record (step r0 + lane / 8, row lane % 8)
              // its third quad: the pixel's coordinates (0, 0 for a slot without a pixel, as the prepass wrote them until round 5)
              const int sstep = r0 + lj, ia = uLo + sstep - lr;
              const bool inside = sstep - lr >= 0 && ia < uLo + LSv && ia < LS && lib < LB;
              const int cxs = TR ? lib : ia, cys = TR ? ia : lib;   // position in sweep order
              const int px = FWD ? cxs : W - 1 - cxs, py = FWD ? cys : H - 1 - cys;
              vc[c] = make_float4(inside ? float(px) : 0.f, inside ? float(py) : 0.f, 0.f, 0.f);
            } else {
              const float4* src = recw + size_t(r0) * (kRows * 3);
              va[c] = src[lane]; vb4[c] = src[lane + 64]; vc[c] = src[lane + 128];
            }
          } else {
            const int sstep = r0 + lj, ia = uLo + sstep - lr;
            ia_of[c] = ia;
            qvalid[c] = sstep - lr >= 0 && ia < uLo + LSv && ia < LS && lib < LB;
            const int cxs = TR ? lib : ia, cys = TR ? ia : lib;   // position in sweep order
            qx[c] = qvalid[c] ? (FWD ? cxs : W - 1 - cxs) : 0; qy[c] = qvalid[c] ? (FWD ? cys : H - 1 - cys) : 0;
            const int idx = qy[c] * W + qx[c];
            qf[c] = flow[idx]; qgate[c] = gate[idx]; qg[c] = g0[idx]; qb[c] = blurred[idx];
          }
          // ---- the chunk's window: offset (at most one texel from the previous chunk's), new columns, new row ----
          const int j = r0 / kChunk;
          if (j - obase >= 64 || j < obase) refill_offsets(j);
          int tx = __builtin_amdgcn_readlane(ocx, j - obase), ty = __builtin_amdgcn_readlane(ocy, j - obase);
          tx = tx < pox - 1 ? pox - 1 : (tx > pox + 1 ? pox + 1 : tx);
          ty = ty < poy - 1 ? poy - 1 : (ty > poy + 1 ? poy + 1 : ty);
          int ou, ov; sweep_offsets(j, tx, ty, ou, ov);
          const int need_lo = uLo + 8 * j - 15 + ou, need_front = uLo + 8 * j + 16 + ou, row0 = vb - kRad + ov;
          int n = need_front - front; n = n < 0 ? 0 : n;   // 7 / 8 / 9 (0 for chunk 0, whose rectangle is in place)
          const bool inside = front >= 0 && front + 8 <= LS && row0 >= 0 && row0 + kRectRows <= LB;   // (wave-uniform) the block lies inside the image
          cwhole[c] = inside && n == 8;
          if (cwhole[c]) {   // the usual block -- eight new columns, inside the image: unconditional loads and stores (no exec-mask work per texel)
            const int ibase = iC + iA * front + iB * row0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { wu[c][k] = front + tdc[k]; wvv[c][k] = row0 + tdr[k]; wok[c][k] = true; wv[c][k] = g1[ibase + tio[k]]; }
          } else if (inside) {
            const int ibase = iC + iA * front + iB * row0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              wu[c][k] = front + tdc[k]; wvv[c][k] = row0 + tdr[k];
              wok[c][k] = tdc[k] < n;
              if (wok[c][k]) wv[c][k] = g1[ibase + tio[k]];
            }
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              wu[c][k] = front + tdc[k]; wvv[c][k] = row0 + tdr[k];
              const float2* q = (tdc[k] < n) ? tex_ptr(wu[c][k], wvv[c][k]) : nullptr;
              wok[c][k] = q != nullptr;
              if (wok[c][k]) wv[c][k] = *q;
            }
          }
          {   // does the block (columns [front, front + 8), rows [row0, row0 + 24)) contain ring column 0 / ring row 0?  (lowest ring coordinate of the block)
            const int clo = (FWD ? front : LS - 8 - front) & (kWC - 1);
            const int rlo = (FWD ? row0 : LB - kRectRows - row0) & (kWRing - 1);
            cdupc[c] = clo + 8 > kWC || clo == 0;
            cdupr[c] = rlo + kRectRows > kWRing || rlo == 0;
          }
          if (n > 8) {   // the ninth column (wave-uniform: the offset along the step axis grew)
            wu[c][3] = front + 8; wvv[c][3] = row0 + lane;
            const float2* q = lane < kRectRows ? tex_ptr(wu[c][3], wvv[c][3]) : nullptr;
            wok[c][3] = q != nullptr;
            if (wok[c][3]) wv[c][3] = *q;
          }
          if (ov != pov) {   // the row that entered the rectangle, over the columns that are already there (wave-uniform: the offset across moved)
            wu[c][4] = need_lo + lane; wvv[c][4] = ov > pov ? row0 + kRectRows - 1 : row0;
            const float2* q = wu[c][4] < front ? tex_ptr(wu[c][4], wvv[c][4]) : nullptr;
            wok[c][4] = q != nullptr;
            if (wok[c][4]) wv[c][4] = *q;
          }
          front = need_front > front ? need_front : front; pov = ov; pox = tx; poy = ty; cox[c] = float(tx); coy[c] = float(ty);
        }
      }
      // fused prepass, phase 2: own-flow terms E(C), E(C+dx), E(C+dy) -- the expressions of k_sweep_prep -- straight into the record
      // ring (the slots are free: ld[c] checked the ring room; they are published further down)
      if (fused) {
#pragma unroll
        for (int c = 0; c < kLoadAhead; ++c) {
          if (ld[c]) {
            const float2 f = qf[c], g = qg[c], bl = qb[c];
            const bool on = qvalid[c] && qgate[c] != 0;
            const float e0 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, qx[c], qy[c], g.x, g.y, bl.x, bl.y, f.x, f.y);
            const float e1 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, qx[c], qy[c], g.x, g.y, bl.x, bl.y, f.x + kGradEpsilon, f.y + 0.0f);
            const float e2 = d_error2g(g1, W, wm2, hm2, fW, rW, cf, qx[c], qy[c], g.x, g.y, bl.x, bl.y, f.x + 0.0f, f.y + kGradEpsilon);
            float4* dst = &sm.rec[w][(rh + c * kChunk) % kRS][0][0] + lane * 3;   // slot (lane >> 3, lane & 7) = linear slot `lane`
            dst[0] = on ? make_float4(g.x, g.y, bl.x, bl.y) : z4;
            const float2 fv = make_float2(qvalid[c] ? f.x : 0.f, qvalid[c] ? f.y : 0.f);
            const float2 rc0 = on ? own_gradient_step(f, e0, e1, e2, cf.step) : fv;
            dst[1] = make_float4(on ? e0 : kKeepEnergy, rc0.x, rc0.y, (on && ia_of[c] > 0) ? e0 : kKeepEnergy);
            dst[2] = make_float4(float(qx[c]), float(qy[c]), on ? cox[c] : __builtin_nanf(""), on ? coy[c] : __builtin_nanf(""));
          }
        }
      }
      bool progress = false;
#pragma unroll
      for (int c = 0; c < kLoadAhead; ++c) {
        if (ld[c]) {
          float4* dst = &sm.rec[w][rh % kRS][0][0];
          if (!fused && soa) {
            // "updated" is E(C) != kKeepEnergy (d_make_record_at): the window offset for such a pixel, NaN ("never leaves the window") otherwise
            const bool on = vb4[c].x != kKeepEnergy;
            float4 q2 = vc[c];
            q2.z = on ? cox[c] : __builtin_nanf(""); q2.w = on ? coy[c] : __builtin_nanf("");
            dst[3 * lane] = va[c]; dst[3 * lane + 1] = vb4[c]; dst[3 * lane + 2] = q2;
          } else if (!fused) {
            float4 v3[3] = {va[c], vb4[c], vc[c]};
#pragma unroll
            for (int k = 0; k < 3; ++k) if (isC[k]) { v3[k].z = cox[c] + v3[k].z; v3[k].w = coy[c] + v3[k].w; }   // the chunk's window offset (+ 0, or + NaN where the pixel is not updated)
            dst[lane] = v3[0]; dst[lane + 64] = v3[1]; dst[lane + 128] = v3[2];
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) if (cwhole[c] || wok[c][k]) win_store_block(wu[c][k], wvv[c][k], wv[c][k], cdupc[c], cdupr[c]);
          if (__any(wok[c][3])) { if (wok[c][3]) win_store(wu[c][3], wvv[c][3], wv[c][3]); }
          if (__any(wok[c][4])) { if (wok[c][4]) win_store(wu[c][4], wvv[c][4], wv[c][4]); }
          rh += kChunk;
          st_cnt(&sm.recHead[w], rh);
          progress = true;
        }
      }
      if (rh >= nsteps) break;
      if (progress) idle = 0;
      else {
        __builtin_amdgcn_s_sleep(4);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    return;
  }

  if (wave == kWaves + kLoaders + 2) {
    // ======================= drainer: results LDS ring -> flow plane (stores only: never waits on HBM) =======================
    int idle = 0;
    for (;;) {
      bool progress = false, done = true;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        if (w < nact) {
          int ot = sm.outTail[w];
          const int oh = ld_cnt(&sm.outHead[w]);
          int n = oh - ot; n = n > 8 ? 8 : n;
          if (n == 8 || (n > 0 && oh >= nsteps)) {   // whole chunks only (one full-wave store per band and chunk; the result ring holds four)
            const int j = lane >> 3, r = lane & 7, t = ot + j;
            if (j < n) {
              const float2 val = sm.out[w][t % kOS][r];
              const int ia = uLo + t - r, ib = (bandLo + band0 + w) * kRows + r;
              if (t - r >= 0 && t - r < LSv && ib < LB) {
                const int cx = transposed ? ib : ia, cy = transposed ? ia : ib;
                const int x = forward ? cx : W - 1 - cx, y = forward ? cy : H - 1 - cy;
                flow[size_t(y) * W + x] = val;
              }
            }
            ot += n;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ring slots are read before they are released
            st_cnt(&sm.outTail[w], ot);
            progress = true;
          }
          if (ot < nsteps) done = false;
        }
      }
      if (done) break;
      if (progress) idle = 0;
      else {
        __builtin_amdgcn_s_sleep(kDrainSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    return;
  }

  if (wave == kWaves + kLoaders) {
    // ======================= publisher: last row of the workgroup -> granules in HBM (tight loop, never waits on HBM) =======================
    if (!publishes) return;
    unsigned long long* bnd_out = boundary + size_t(wg + 1) * LSv;   // granule row 0 belongs to the static row above the window
    const int wl = kWaves - 1;
    int pt = 0, idle = 0;
    while (pt < nsteps) {
      const int ohl = ld_cnt(&sm.outHead[wl]);
      int n = ohl - pt; n = n > kOS ? kOS : n;
      if (n > 0) {
        const int t = pt + lane;
        if (lane < n) {
          const float2 val = sm.out[wl][t % kOS][kRows - 1];
          const int cx = t - (kRows - 1);
          if (cx >= 0 && cx < LSv) __hip_atomic_store(bnd_out + cx, pack2(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pt += n;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st_cnt(&sm.pubTail, pt);
        idle = 0;
      } else {
        __builtin_amdgcn_s_sleep(kPubSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    return;
  }

  // ======================= poller: previous workgroup's granules HBM -> LDS ring =======================
  {
    if ((wg == 0 && !staticTop) || wave != kWaves + kLoaders + 1) return;
    const unsigned long long* bnd_in = boundary + size_t(wg) * LSv;
    const bool topFromPlane = MODE != 0 && wg == 0;   // (wg == 0 only gets here with a static top row)
    int bh = 0, idle = 0;
    while (bh < LSv) {
      const int oh0 = ld_cnt(&sm.outHead[0]);
      if (bh + 64 - oh0 <= kBS) {
        unsigned long long g = kNotReady;
        if (bh + lane < LSv) {
          if (topFromPlane) {
            // fused prepass: the row above the window never changes during this sweep -- read it from the flow plane itself
            const int ia = uLo + bh + lane, ib = bandLo * kRows - 1;
            const int cxs = TR ? ib : ia, cys = TR ? ia : ib;
            const int x = FWD ? cxs : W - 1 - cxs, y = FWD ? cys : H - 1 - cys;
            g = pack2(flow[y * W + x]);
          } else {
            g = __hip_atomic_load(bnd_in + bh + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        const bool ready = (g != kNotReady) || (bh + lane >= LSv);
        const unsigned long long m = __ballot(ready);
        const int n = (m == ~0ull) ? 64 : __builtin_ctzll(~m);
        if (n > 0) {
          if (lane < n && bh + lane < LSv) sm.bnd[(bh + lane) % kBS] = g;
          bh += n;
          st_cnt(&sm.bndHead, bh);
          idle = 0;
          continue;
        }
        __builtin_amdgcn_s_sleep(kPollSleep);
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
      if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Load-time gate of the asm-block packed chains (exact_forms.hpp, PF_PK_ASM): the product's own two blocks -- the packed square-root
// sequence and (i0 - i1)^2 -- against the compiler-scheduled forms of the same arithmetic, dependent chains of them back to back,
// from one wave per CU up to 16 waves per SIMD.  pf_create runs it once per device and process (~0.1 ms) and refuses to create a
// context if a single bit differs: the assumption "a v_pk_*_f32 result may be read by the next instruction" is then checked on the
// very chip the library runs on, not only on the one the round-3 probe (tests/micro/pk_hazard_probe.hip) ran on.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_pk_probe(int rounds, unsigned* __restrict__ bad) {
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  // operands inside the range guard: 2^-90 .. 2^90, both signs for the difference
  auto val = [&]() { const unsigned u = rnd(); return __uint_as_float((u & 0x007fffffu) | ((37u + (u >> 23) % 180u) << 23)); };
  float a = val(), b = val();
  unsigned diff = 0;
  for (int r = 0; r < rounds; ++r) {
    const float c = val(), d = val();
    int e0, e1;
    const f2p q = sqrt_core2(f2p{a, b}, e0), qs = sqrt_core2_safe(f2p{a, b}, e1);
    f2p di, di2;
#if PF_PK_ASM
    asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %1, %0, %0" : "=&v"(di), "=&v"(di2) : "v"(f2p{c, q.x}), "v"(f2p{q.y, d}));
#else
    di = f2p{c, q.x} - f2p{q.y, d}; di2 = di * di;
#endif
    const f2p ds = f2p{c, qs.x} - f2p{qs.y, d}, ds2 = ds * ds;
    // the step's two sum-of-squares blocks (round 5), fed with results that are one instruction old
    const float u0 = sumsq_diff2(f2p{c, q.x}, f2p{q.y, d}), u1 = sumsq_diff2_safe(f2p{c, qs.x}, f2p{qs.y, d});
    const float w0 = sumsq_diff2_sum(f2p{c, d}, f2p{q.x, q.y}, di2), w1 = sumsq_diff2_sum_safe(f2p{c, d}, f2p{qs.x, qs.y}, ds2);
    diff |= (__float_as_uint(q.x) ^ __float_as_uint(qs.x)) | (__float_as_uint(q.y) ^ __float_as_uint(qs.y)) | unsigned(e0 ^ e1) |
            (__float_as_uint(di2.x) ^ __float_as_uint(ds2.x)) | (__float_as_uint(di2.y) ^ __float_as_uint(ds2.y)) |
            (__float_as_uint(u0) ^ __float_as_uint(u1)) | (__float_as_uint(w0) ^ __float_as_uint(w1));
    // the next operands depend on this round's results (a dependent chain, like a sweep step), folded back into the guard's range
    const unsigned ua = __float_as_uint(di2.x), ub = __float_as_uint(q.y);
    a = __uint_as_float((ua & 0x007fffffu) | ((37u + ((ua >> 23) & 0xffu) % 180u) << 23));
    b = __uint_as_float((ub & 0x007fffffu) | ((37u + ((ub >> 23) & 0xffu) % 180u) << 23));
  }
  if (diff) atomicAdd(bad, 1u);
}
// The second hardware assumption of the product's sweep TU (round 6; tools/asm_sched.py --dpp-old-wait 0): a DPP move whose masks leave lanes
// unwritten may DIRECTLY follow the instruction that wrote its destination -- those lanes keep the old value, nothing reads it through the
// cross-lane network.  The step's own sequence (VALU producer of the pair, 64-bit row_newbcast moves, 32-bit row_bcast:15 moves) back to back
// against the same with wait states, on changing data; a single differing bit refuses the device like the packed chains above
// (tests/micro/dpp_old_probe.hip is the long form).
#define PF_DPP_OLD_SEQ(N)                                                                                        \
  "v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n s_nop 4\n" \
  "v_mov_b64_dpp v[14:15], v[12:13] row_newbcast:8 row_mask:0xf bank_mask:0x8 bound_ctrl:1\n s_nop 4\n"       \
  "v_pk_add_f32 v[10:11], v[16:17], v[18:19]\n" N                                                               \
  "v_mov_b64_dpp v[10:11], v[12:13] row_newbcast:0 row_mask:0xf bank_mask:0x9\n" N                              \
  "v_mov_b64_dpp v[10:11], v[12:13] row_newbcast:8 row_mask:0xf bank_mask:0x4\n" N                              \
  "v_mov_b32_dpp v10, v14 row_bcast:15 row_mask:0xe bank_mask:0x2\n" N                                          \
  "v_mov_b32_dpp v11, v15 row_bcast:15 row_mask:0xe bank_mask:0x2\n s_nop 4\n"                                \
  "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n"
__global__ __launch_bounds__(1024) void k_dpp_old_probe(int rounds, unsigned* __restrict__ bad) {
#if defined(__gfx950__)
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2246822519u + 777u;
  auto val = [&]() { s = s * 1664525u + 1013904223u; return float(int(s >> 8) % 20001 - 10000) / 256.0f; };
  float p0 = val(), p1 = val(), a0 = val(), a1 = val(), b0 = val(), b1 = val();
  unsigned diff = 0;
  for (int r = 0; r < rounds; ++r) {
    float x, y, xs, ys;
    asm volatile(PF_DPP_OLD_SEQ("") : "=&v"(x), "=&v"(y) : "v"(p0), "v"(p1), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
    asm volatile(PF_DPP_OLD_SEQ("s_nop 4\n") : "=&v"(xs), "=&v"(ys) : "v"(p0), "v"(p1), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
    diff |= (__float_as_uint(x) ^ __float_as_uint(xs)) | (__float_as_uint(y) ^ __float_as_uint(ys));
    const float fx = float(int(__float_as_uint(x) >> 9) % 4001 - 2000) / 64.0f, fy = float(int(__float_as_uint(y) >> 9) % 4001 - 2000) / 64.0f;
    p0 = a1 + fx; p1 = b0 - fy; a0 = fy + float(r & 7); a1 = fx * 0.5f + 1.0f; b0 = p0 * 0.25f + 2.0f; b1 = fabsf(fy) + 0.5f;
  }
  if (diff) atomicAdd(bad, 1u);
#endif
}
// 0 = the asm-block forms and the compiler-scheduled forms agree bit for bit on this device (or the build has no asm blocks); bad = scratch word on the device
int sweep_pk_probe(hipStream_t st, unsigned* bad) {
#if defined(PF_SAFE_PK)
  (void)st; (void)bad;
  return 0;
#else
  if (hipMemsetAsync(bad, 0, 4, st) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_pk_probe, dim3(256), dim3(64), 0, st, 256, bad);          // one wave per CU
  hipLaunchKernelGGL(k_pk_probe, dim3(256 * 4), dim3(1024), 0, st, 24, bad);      // 16 waves per SIMD
  hipLaunchKernelGGL(k_dpp_old_probe, dim3(256), dim3(64), 0, st, 128, bad);
  hipLaunchKernelGGL(k_dpp_old_probe, dim3(256 * 4), dim3(1024), 0, st, 16, bad);
  unsigned h = 1;
  if (hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
  return int(h);
#endif
}

#ifdef PF_SWEEP_STATS
}  // namespace pf
// diagnostics build only: read (and optionally clear) the out-of-window counters
#ifdef PF_SWEEP_TRACE
extern "C" __attribute__((visibility("default"))) int pf_debug_band_trace(long long* out, int nbands) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::g_band_trace), size_t(nbands) * 40 * 8) == hipSuccess ? 0 : -1;
}
#endif
extern "C" __attribute__((visibility("default"))) int pf_debug_sweep_stats(unsigned long long* out4, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(pf::g_sweep_stats), 32) != hipSuccess) return -1;
  if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pf::g_sweep_stats), z, 32) != hipSuccess) return -1; }
  return 0;
}
namespace pf {
#endif
#include "kernels_sweep_t.inl"   // the throughput form (32 rows per wave, 2 lanes per pixel): k_sweep_t, launch_sweep_t

// ---- host side ----
// Bands run across the SHORTER side of the active window (fewer band-to-band hand-offs on the critical path):
// normal = bands of 8 rows stepping along x; transposed = bands of 8 columns stepping along y.
// (buffers are sized for either workgroup shape: the latency form has the most workgroups, the wide form pads the bands to a multiple of 8)
static inline int wgs_for(int LB, int waves = SwLatency::kWaves) { const int nbands = (LB + kRows - 1) / kRows; return (nbands + waves - 1) / waves; }
static inline int steps_pad(int LS) { return ((LS + kRows - 1) + kChunk - 1) / kChunk * kChunk; }
int sweep2_num_wgs(int H) { return wgs_for(H); }
int sweep2_num_wgs_max(int W, int H) { const int a = wgs_for(W), b = wgs_for(H); return a > b ? a : b; }
size_t sweep2_boundary_elems(int W, int H) {   // hand-off granules of one sweep launch, either orientation (+1 row: the static row above the window)
  const size_t a = size_t(wgs_for(H) + 1) * W, b = size_t(wgs_for(W) + 1) * H;
  return a > b ? a : b;
}
size_t sweep_boundary_elems(int W, int H) {   // hand-off granules a sweep launch on a W x H level may need (every sweep implementation of this build)
  size_t v = sweep2_boundary_elems(W, H);
#ifdef PF_EXPERIMENTS
  const size_t v1 = sweep1_boundary_elems(W, H), v3 = sweep_relax_boundary_elems(W, H);
  if (v1 > v) v = v1;
  if (v3 > v) v = v3;
#endif
  return v;
}
size_t sweep2_rec_bytes(int W, int H) {
  const size_t a = size_t(wgs_for(H, SwWide::kWaves)) * SwWide::kWaves * steps_pad(W), b = size_t(wgs_for(W, SwWide::kWaves)) * SwWide::kWaves * steps_pad(H);
  // throughput form: bands of 32 rows, three per workgroup, tRows - 1 more steps per band
  // (12 = a multiple of both workgroup sizes, 3 and 4 bands; + tPre steps that the record-prefetching form reads past the last band's end)
  auto t_records = [](int LB, int LS) { const size_t nb = (size_t(LB) + tRows - 1) / tRows, nbp = (nb + 11) / 12 * 12; return nbp * tRows * (size_t(LS + tRows - 1 + kChunk - 1) / kChunk * kChunk) + size_t(tPre) * tRows; };
  const size_t t = std::max(t_records(H, W), t_records(W, H));
  return std::max((a > b ? a : b) * kRows * 48, t * 32);
}
template <class G>
static bool launch_sweep2_form(hipStream_t st, const SweepArgs& a, float* rec) {
  constexpr int kWaves = G::kWaves;
  // active window: bounding box of the gated pixels (pixels outside it are not updated by this sweep and keep their flow)
  const SweepWindow win = make_sweep_window(a.W, a.H, a.forward, a.ax0, a.ay0, a.ax1, a.ay1, kRows, kWaves, kChunk);
  if (win.empty) return false;   // nothing to update: the sweep is the identity
  const int tr = win.tr, uLo = win.uLo, uHi = win.uHi, LSv = win.LSv, bandLo = win.bandLo, nbands = win.nbands;
  const int nwg = win.nwg, nbandsPad = nwg * kWaves, nstepsPad = win.nstepsPad;
  const size_t total = size_t(nbandsPad) * nstepsPad * kRows;
  const float rW = (float)(1.0 / (double)(float)a.W), rEps = (float)(1.0 / (double)kGradEpsilon);
  // How the records reach the sweep.  0 (the product): k_sweep_prep in front of the sweep.  Two measured and rejected alternatives
  // exist in the lab build only (-DPF_EXPERIMENTS, SweepArgs::prep_mode; latency form only), both bit-identical: 1 = the loader waves
  // compute them (no record traffic through HBM, but 8 % slower: profiles/r02_fused_prepass_ab.txt); 2 = prepass blocks inside the
  // sweep launch with a write-through + counter + acquire hand-off (the first bands start microseconds after the launch, yet the
  // launch as a whole is not shorter: same file).
#ifdef PF_EXPERIMENTS
  const int mode = (G::kBPW != 1 || (a.prep_mode == 2 && (a.prepcnt == nullptr || total * 48 >= (size_t(1) << 31)))) ? 0 : a.prep_mode;
#else
  constexpr int mode = 0;
#endif
  if (mode == 0 && G::kBPW == 1)
    hipExtLaunchKernelGGL((k_sweep_prep<kRows, true, true>), dim3((unsigned)((nstepsPad + 256 / kRows - 1) / (256 / kRows)), (unsigned)nbandsPad, a.bt.n), dim3(256), 0, st, a.ev_start, nullptr, 0, a.g0, a.g1, a.blurred, a.gate, a.flow,
                          a.W, a.H, a.forward, tr, nstepsPad, nbandsPad, rW, reinterpret_cast<float4*>(rec), uLo, uHi, bandLo,
                          bandLo > 0 ? a.boundary : (unsigned long long*)nullptr, a.bt.stride, a.cf);
  else if (mode == 0)
    hipExtLaunchKernelGGL((k_sweep_prep<kRows, true>), dim3((unsigned)((nstepsPad + 256 / kRows - 1) / (256 / kRows)), (unsigned)nbandsPad, a.bt.n), dim3(256), 0, st, a.ev_start, nullptr, 0, a.g0, a.g1, a.blurred, a.gate, a.flow,
                          a.W, a.H, a.forward, tr, nstepsPad, nbandsPad, rW, reinterpret_cast<float4*>(rec), uLo, uHi, bandLo,
                          bandLo > 0 ? a.boundary : (unsigned long long*)nullptr, a.bt.stride, a.cf);
  hipEvent_t evs = mode == 0 ? nullptr : a.ev_start;   // without a prepass kernel the sweep launch carries both events
  // wall-clock budget of every wait inside the launch, in 100 MHz ticks: 2 s + 1000 x the expected duration (~0.5 us per step)
  const long long budget = 200000000ll + 1000ll * 50ll * (long long)(nstepsPad + 9 * nbands);
  const unsigned nthreads = G::kThreads;
  const dim3 grid(mode == 2 ? nwg + (unsigned)((total + nthreads - 1) / nthreads) : nwg, 1, a.bt.n), block(nthreads);
  const float4* r4 = mode == 1 ? nullptr : reinterpret_cast<const float4*>(rec);
#define PF_LAUNCH_SWEEP2_(TRV, FWV, SPV, MDV) hipExtLaunchKernelGGL((k_sweep2<G, TRV, FWV, SPV, MDV>), grid, block, 0, st, evs, a.ev_stop, 0, r4, a.g1, a.flow, a.boundary, a.ctrl, a.W, a.H, nstepsPad, nbands, rW, rEps, uLo, LSv, bandLo, budget, a.g0, a.blurred, a.gate, nwg, a.prepcnt, a.bt.stride, a.cf)
#ifdef PF_EXPERIMENTS
#define PF_LAUNCH_SWEEP2__(TRV, FWV, SPV) do { if constexpr (G::kBPW == 1) { if (mode == 2) { PF_LAUNCH_SWEEP2_(TRV, FWV, SPV, 2); break; } if (mode == 1) { PF_LAUNCH_SWEEP2_(TRV, FWV, SPV, 1); break; } } PF_LAUNCH_SWEEP2_(TRV, FWV, SPV, 0); } while (0)
#else
#define PF_LAUNCH_SWEEP2__(TRV, FWV, SPV) PF_LAUNCH_SWEEP2_(TRV, FWV, SPV, 0)   /* the product library ships the MODE-0 variants only: 8 per workgroup shape */
#endif
#define PF_LAUNCH_SWEEP2(TRV, FWV) do { if (a.sparse) PF_LAUNCH_SWEEP2__(TRV, FWV, true); else PF_LAUNCH_SWEEP2__(TRV, FWV, false); } while (0)
  if (tr) { if (a.forward) PF_LAUNCH_SWEEP2(true, true); else PF_LAUNCH_SWEEP2(true, false); }
  else { if (a.forward) PF_LAUNCH_SWEEP2(false, true); else PF_LAUNCH_SWEEP2(false, false); }
#undef PF_LAUNCH_SWEEP2__
  return true;
#undef PF_LAUNCH_SWEEP2_
#undef PF_LAUNCH_SWEEP2
}
// Which workgroup shape: the wide form (two compute waves per SIMD) when the launch asks for more workgroups than the chip can
// hold at once -- a batch of pairs at the large levels -- and the latency form otherwise (a.wide: 0 / 1 forced, -1 = by size).
// `concurrent` = sweep launches that run beside this one (the other direction's): the host's estimate, 2 for a bidirectional solve.
bool launch_sweep2(hipStream_t st, const SweepArgs& a, float* rec) {
#ifdef PF_EXPERIMENTS
  static const int poison = [] { const int v = getenv("PANOFLOW_POISON_LDS") ? 1 : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_poison_lds), &v, sizeof(int)); return v; }();
  (void)poison;
#endif
  int wide = a.wide;
  if (wide < 0) {
    // oversubscribed launches of a batch: the throughput form where it exists (dense, bands stepping along x), else the latency form
    const SweepWindow win = make_sweep_window(a.W, a.H, a.forward, a.ax0, a.ay0, a.ax1, a.ay1, kRows, SwLatency::kWaves, kChunk);
    wide = (!win.empty && long(win.nwg) * a.concurrent_sweeps > long(a.wide_threshold_wgs) && !a.sparse && (!win.tr || a.wide_tr)) ? 2 : 0;
  }
  if (wide == 2 && !a.sparse) return launch_sweep_t(st, a, rec);
#ifdef PF_EXPERIMENTS
  // measured-and-rejected form, lab build only (profiles/r04_wide_sweep.txt): the latency step with two compute waves per SIMD
  if (wide == 1) return launch_sweep2_form<SwWide>(st, a, rec);
#endif
  return launch_sweep2_form<SwLatency>(st, a, rec);
}

#ifdef PF_EXPERIMENTS
#include "kernels_relax.inl"   // rejected experiment (41 % slower), kept bit-exact under test in the lab build only
#endif

}  // namespace pf
