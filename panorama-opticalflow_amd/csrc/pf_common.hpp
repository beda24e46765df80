// Shared host/device declarations for the panoflow HIP library (gfx950 only).
// All kernels are built with -ffp-contract=off: the solver makes strict '<' decisions on nearly
// equal floats, so every expression keeps the reference's evaluation order without FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

// ---- constants of the solver (reference: CPU/PixFlow.hpp:32-44 and the factory :459-497) ----
constexpr int kPyrMinImageSize = 24;
constexpr int kPyrMaxLevels = 1000;
constexpr float kGradEpsilon = 0.001f;
constexpr float kUpdateAlphaThreshold = 0.9f;
constexpr float kPyrScaleFactor = 0.9f;
constexpr float kSmoothnessCoef = 0.001f;
constexpr float kVerticalRegularizationCoef = 0.01f;
constexpr float kHorizontalRegularizationCoef = 0.01f;
constexpr float kGradientStepSize = 0.5f;
constexpr float kDownscaleFactor = 0.5f;
// The reference's PixFlow takes these as constructor arguments (CPU/PixFlow.hpp:46-68); its factory only ever passes the values above
// (:459-497).  The sweep kernels take them as scalar kernel arguments (round 6, pf_set_solver_params): a multiply by a coefficient costs
// the same from an SGPR as from a literal.  `step`: the fast step folds `flow - step * g` into ONE fused multiply-add, which is the
// reference's two roundings only when the product is exact, i.e. when step is a power of two; for any other step size the host sets
// `guard_min` above every exponent, so that the range guard sends EVERY step of an updated pixel through the IEEE sequence
// (select_step<false>, which multiplies and subtracts): no second kernel variant, bit-exact, slower.
struct SolverCoef {
  float smooth = kSmoothnessCoef, vreg = kVerticalRegularizationCoef, hreg = kHorizontalRegularizationCoef, step = kGradientStepSize;
  int guard_min = -94;   // an operand whose frexp exponent lies below this (and is not zero) leaves the exact forms' range; INT_MAX = always
};

// sentinel for "boundary flow not published yet" (a NaN payload arithmetic cannot produce)
constexpr unsigned long long kNotReady = 0x7FFFDEAD7FFFDEADull;

struct Gauss {  // separable kernel taps (host-computed, [OpenCV] getGaussianKernel CV_32F)
  float k[15];
  int ksize;
};

// ---- device helpers ----
__device__ __forceinline__ int d_reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
  return p;
}
__device__ __forceinline__ int d_replicate(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); }
__device__ __forceinline__ int d_floor(float v) { int i = (int)v; return i - (v < (float)i); }

// [OpenCV imgwarp.cpp] interpolateCubic, A = -0.75, float arithmetic in this exact order
__device__ __forceinline__ void d_cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
// source coordinate of destination index d: (float)((d+0.5)*scale-0.5), floor and fraction
__device__ __forceinline__ void d_src_coord(int d, double scale, int& s, float& f) {
  f = (float)((d + 0.5) * scale - 0.5);
  s = d_floor(f);
  f -= s;
}

// [OpenCV imgwarp.cpp] INTER_LINEAR float: HResizeLinear (fx forced to 0 at the borders, tail pixels
// copy S[sx]*1) then VResizeLinear (rows clipped per tap, weights untouched).
template <int CN>
__device__ __forceinline__ void d_resize_linear_px(const float* __restrict__ src, int sw, int sh, int dw, int dh, double scale_x,
                                                   double scale_y, int dx, int dy, float* out) {
  int sx, sy; float fx, fy;
  d_src_coord(dx, scale_x, sx, fx);
  d_src_coord(dy, scale_y, sy, fy);
  if (sx < 0) { fx = 0; sx = 0; }
  const bool tail = (sx + 1 >= sw);
  if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const float* r0 = src + size_t(d_replicate(sy, sh)) * sw * CN;
  const float* r1 = src + size_t(d_replicate(sy + 1, sh)) * sw * CN;
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    float h0, h1;
    if (tail) { h0 = r0[sx * CN + c] * 1.0f; h1 = r1[sx * CN + c] * 1.0f; }
    else { h0 = r0[sx * CN + c] * a0 + r0[(sx + 1) * CN + c] * a1; h1 = r1[sx * CN + c] * a0 + r1[(sx + 1) * CN + c] * a1; }
    out[c] = h0 * b0 + h1 * b1;
  }
}

// ---- batch dimension -------------------------------------------------------------------------------------------------------
// A launch may cover `n` same-size pairs at once (throughput mode: one kernel boundary for n pairs instead of n).  The pairs'
// working buffers lie in identically laid-out slabs `stride` bytes apart, so a kernel reaches pair z's copy of ANY internal buffer
// by adding z * stride to the pointer it was given for pair 0 (blockIdx.z = pair; kernels that already use z for planes keep the
// plane in its low bits).  Caller-owned buffers (input images, blend ramp, outputs) come as per-pair pointer tables instead.
constexpr int kMaxBatch = 16;
struct Batch { int n = 1; size_t stride = 0; };
struct ExtPtrs { const void* p[kMaxBatch]; };   // caller-owned buffers of the pairs of a batch (read-only or written, by use)
// (pointer arithmetic on bytes, NOT a round trip through uintptr_t: an inttoptr hides the kernel argument from the compiler's address-space
// inference and every access through the pointer becomes a FLAT instruction -- counted in lgkmcnt as well, completion order unknown, so
// each wait on one is s_waitcnt vmcnt(0) lgkmcnt(0).  Rounds 3-4 shipped that: found in round 4, tests/micro/isa_flat_count.sh)
#define PF_BOFF(ptr, off) (ptr = (decltype(ptr))((char*)(ptr) + (off)))

// ---- launch wrappers (defined in the kernels_*.hip files) ----
// preprocessing
void launch_downscale_gray(hipStream_t st, const uint8_t* bgra, int cols, int rows, int pad, float* gray, float* alpha, int dw, int dh, Batch bt = Batch(),
                           const ExtPtrs* imgs = nullptr /* batched: the pairs' input images instead of bgra */);
void launch_gauss_small(hipStream_t st, const float* src, float* dst, int w, int h, int cn, const Gauss& g, Batch bt = Batch());
void launch_resize_linear(hipStream_t st, const float* src, int sw, int sh, float* dst, int dw, int dh, int cn, float mul, bool do_mul);
void launch_pyr_down4(hipStream_t st, const float* s0, const float* s1, const float* s2, const float* s3, int sw, int sh, float* d0,
                      float* d1, float* d2, float* d3, int dw, int dh, Batch bt = Batch());
void launch_pyr_chain4(hipStream_t st, float* p0, float* p1, float* p2, float* p3, const int* ws, const int* hs, const size_t* off, int first, int k,
                       Batch bt = Batch());
void launch_pyr_down2(hipStream_t st, const float* s0, const float* s1, int sw, int sh, float* d0, float* d1, int dw, int dh);
// per level
void launch_gradients(hipStream_t st, const float* img, int w, int h, float* gxy, const Gauss& g3);
// offsets/sizes of the pyramid levels inside one pyramid plane, passed by value to kernels that cover all levels at once
constexpr int kLevelTableMax = 96;
struct LevelTable { int n; int w[kLevelTableMax]; int h[kLevelTableMax]; unsigned off[kLevelTableMax]; };
void launch_gate_bbox(hipStream_t st, const uint8_t* gate, const LevelTable& t, size_t total, int* box);   // box[4*l..]: min x, min y, max x, max y
// elements [first, total) of the pyramid planes (levels lie back to back, level 0 first)
void launch_gradients_all(hipStream_t st, const float* pyr0, const float* pyr1, float* grad0, float* grad1, const LevelTable& t, size_t first,
                          size_t total, const Gauss& g3, int max_blocks = 0, Batch bt = Batch());
void launch_gate(hipStream_t st, const float* a0, const float* a1, int n, uint8_t* gate);
// gate + per-level bounding boxes + level-0 count in one launch, published to mapped pinned host memory behind an epoch flag
// (work: 4*kLevelTableMax + 2 ints, initialised once to (INT_MAX, INT_MAX, -1, -1)*, 0, 0; host_mapped: same size)
void launch_gate_bbox_all(hipStream_t st, const float* a0, const float* a1, uint8_t* gate, const LevelTable& t, size_t total, int* work, int* host_mapped,
                          int epoch, Batch bt = Batch(), size_t host_stride = 0 /* bytes between the pairs' mapped host areas */);
void launch_count_gate(hipStream_t st, const uint8_t* gate, int n, unsigned* count /* zeroed by the caller */);
void launch_gauss15(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, const Gauss& g15, Batch bt = Batch());
void launch_median_gauss15_mix(hipStream_t st, const float* flow, const float* a0, const float* a1, int w, int h, const Gauss& g15, float* out, Batch bt = Batch());
void launch_gauss15_upsample(hipStream_t st, const float* coarse, int sw, int sh, float mul, float* up, float* dst, int w, int h, const Gauss& g15, Batch bt = Batch());
void launch_gauss15_mix(hipStream_t st, float* flow, float* tmp, const float* a0, const float* a1, int w, int h, const Gauss& g15,
                        float* out, Batch bt = Batch());
void launch_median5(hipStream_t st, const float* src, float* dst, int w, int h, Batch bt = Batch());   // direct form below 3 Mpix, LDS-tiled form above
void launch_median5_form(hipStream_t st, const float* src, float* dst, int w, int h, bool tiled, Batch bt = Batch());   // a given form at any size (tests)
void launch_upsample_cubic(hipStream_t st, const float* src, int sw, int sh, float* dst, int dw, int dh, float mul, Batch bt = Batch());
void launch_final_flow(hipStream_t st, const float* flow0, int sw, int sh, int pad_cols, int rows, int pad, float mul, const Gauss& g3,
                       float* out, Batch bt = Batch(), const ExtPtrs* outs = nullptr /* batched: the pairs' output planes instead of out */);
// sweep
struct SweepArgs {
  const float2* g0;       // (I0x,I0y) at the pixel
  const float2* g1;       // (I1x,I1y) gathered bilinearly
  const float2* blurred;  // frozen blurred flow
  const uint8_t* gate;    // alpha0>thr && alpha1>thr
  float2* flow;           // in/out
  unsigned long long* boundary;  // [nbands][W] hand-off rows, pre-filled with kNotReady
  int* ctrl;              // [0] ticket, [1] abort/timeout flag
  int W, H, forward;
  int sparse;             // few pixels gated (full-canvas inputs): use the kernel variant that skips ungated anti-diagonals
  // optional timing events (v2 sweep): attached to the launches themselves (hipExtLaunchKernel: start of the prepass / end of the
  // sweep kernel come from the dispatch packets' own timestamps), so that timing a sweep puts no marker packets on its stream
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int* prepcnt = nullptr; // v2 sweep, prepass inside the launch: one counter per sweep workgroup (sweep2_num_wgs_max ints), ZEROED before the launch
  Batch bt;               // v2 sweep: bt.n same-size pairs in one launch (every pointer above is pair 0's; pair z's lies z * bt.stride bytes on)
  int prep_mode = 0;      // lab build only (-DPF_EXPERIMENTS): 1 / 2 = the two rejected record paths (pf_config::record_path)
  int ax0 = 0, ay0 = 0, ax1 = 1 << 30, ay1 = 1 << 30;   // bounding box [ax0,ax1) x [ay0,ay1) of the gated pixels (default: everything); v2 sweep only
  int wide = 0;           // v2 sweep form: 0 = latency form (8 lanes per pixel, 4 bands of 8 rows per workgroup, one compute wave per SIMD), 1 = the same step
                          // with 8 bands per workgroup (two compute waves per SIMD), 2 = throughput form (2 lanes per pixel, bands of 32 rows,
                          // kernels_sweep_t.inl), -1 = throughput form when the launch oversubscribes the chip (dense, bands along x), else latency
  int wide_threshold_wgs = 512;   // wide = -1: "oversubscribed" means more latency-form workgroups x concurrent_sweeps than this
  int wide_tr = 0;                // wide = -1: transposed sweeps (bands along y) may take the throughput form too (its window loads do not coalesce there)
  int concurrent_sweeps = 2;      // sweeps that run on the chip at the same time as this launch's (this launch's pairs x 2 directions x lanes)
  SolverCoef cf;                  // the energy's coefficients and the gradient step size (defaults = the reference factory's presets)
};
size_t sweep_boundary_elems(int W, int H);   // hand-off granules needed per sweep launch (covers every sweep kernel of this build)
size_t sweep1_boundary_elems(int W, int H);                     // lab build only (-DPF_EXPERIMENTS)
void launch_sweep(hipStream_t st, const SweepArgs& a);          // lab build only: v1, 64 rows per wave, the independent cross-check (pf_config::sweep_impl = 1)
int sweep2_num_wgs(int H);
int sweep2_num_wgs_max(int W, int H);          // workgroups a sweep launch on a W x H level can have (either band orientation)
size_t sweep2_boundary_elems(int W, int H);   // granules one sweep launch may need (either band orientation)
size_t sweep2_rec_bytes(int W, int H);
int sweep_pk_probe(hipStream_t st, unsigned* d_scratch);   // 0 = the sweep's asm-block packed chains give the compiler forms' bits on this device; > 0 mismatching threads; < 0 HIP error
bool launch_sweep2(hipStream_t st, const SweepArgs& a, float* rec);  // v2: prepass + 8 lanes/pixel + helper waves; false = empty window, nothing launched
size_t sweep_relax_boundary_elems(int W, int H);
bool launch_sweep_relax(hipStream_t st, const SweepArgs& a);   // lab build only (pf_config::sweep_impl = 3): event-driven relaxation on LDS-resident tiles, kernels_relax.inl
// coarsest-level search
void launch_adjust_initial_flow(hipStream_t st, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, int hint,
                                int max_pct, float* i1eq_tmp, float* flow, Batch bt = Batch());
// blend
void launch_blend_tables(hipStream_t st);   // once per device before the first blend (create_ctx)
void launch_blend(hipStream_t st, const uint8_t* L, const uint8_t* R, const float* flowLR, const float* flowRL, const float* blend, int cols,
                  int rows, uint8_t* out);
struct BlendPtrs { const uint8_t* L[kMaxBatch]; const uint8_t* R[kMaxBatch]; const float* fLR[kMaxBatch]; const float* fRL[kMaxBatch]; const float* blend[kMaxBatch]; uint8_t* out[kMaxBatch]; };
void launch_blend_batch(hipStream_t st, const BlendPtrs& p, int n, int cols, int rows);
// stitch
void launch_match_images(hipStream_t st, const uint8_t* L, const uint8_t* R, int cols, int rows, uint8_t* map, uint8_t* ovL, uint8_t* ovR);
void launch_countblend(hipStream_t st, const uint8_t* map, int cols, int rows, float* blend, float* mergedDis);
void launch_box_blur(hipStream_t st, const float* src, float* dst, double* rowsum_tmp, int cols, int rows, int k);
size_t tile_blur_work_bytes(int cols, int rows, int step, int k);   // device scratch of launch_tile_blur (diagonal counts + barrier word)
size_t tile_blur_lds_bytes(int step, int k);
void launch_tile_blur(hipStream_t st, float* blend, const float* mergedDis, int cols, int rows, int step, int k, void* work);
void launch_gather(hipStream_t st, const uint8_t* L, const uint8_t* R, const uint8_t* merged, const uint8_t* map, int cols, int rows,
                   uint8_t* out);
void launch_fill_u64(hipStream_t st, unsigned long long* p, size_t n, unsigned long long v, Batch bt = Batch());
void launch_fill_u32(hipStream_t st, unsigned* p, size_t n, unsigned v, Batch bt = Batch());   // memset that knows the batch dimension
void launch_checksum64(hipStream_t st, const void* p, size_t bytes, unsigned long long* acc /* zeroed by the caller */);
void launch_count_diff_u32(hipStream_t st, const uint32_t* a, const uint32_t* b, size_t n, int* count /* zeroed by the caller */);
void launch_collect_status(hipStream_t st, const int* ctrl, int nwords, int* status_mapped, int bit, Batch bt = Batch());

}  // namespace pf
