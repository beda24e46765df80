// C-ABI implementation (include/panoflow.h): context, HBM arena, orchestration of the kernel families
// on HIP streams.  Host code only launches kernels and moves data; all pixel arithmetic is in the
// kernels_*.hip files.  There is no CPU fallback anywhere in this library.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/panoflow.h"
#include "pf_common.hpp"

using namespace pf;

namespace {

thread_local std::string g_err;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct ProfEntry { double ms = 0; int n = 0; };
constexpr int kGateWords = 4 * pf::kLevelTableMax + 2;   // per pair: boxes of every level, level-0 count, epoch flag (k_gate_bbox_all)
struct ProfPending { int id; hipEvent_t a, b; };

}  // namespace

struct pf_ctx {
  int device = 0;
  hipStream_t s_main = nullptr, s_dir[2] = {nullptr, nullptr}, s_aux = nullptr;
  hipEvent_t ev_alpha = nullptr, ev_gate = nullptr;
  hipEvent_t ev_aux_go = nullptr, ev_aux_done = nullptr;
  hipEvent_t ev_pre = nullptr, ev_dir[2] = {nullptr, nullptr};
  hipEvent_t ev_stagger = nullptr;
  hipEvent_t ev_fine2 = nullptr;  // ... of the finest levels (narrow launch)
  hipEvent_t ev_fine = nullptr;   // gradients of the fine levels done (the directions start on the coarse ones before that)
  std::string err;
  std::string warn; int warn_count = 0;   // pf_last_warning / pf_warning_count: conditions that cost performance, never results
  std::map<std::string, DevBuf> bufs;  // named grow-only arena: everything a solve needs stays resident
  Gauss g5, g3_05, g3_1, g15;
  int prof = 0;   // 0 off, 1 every kernel family, 2 only the dominant family (the sweeps): fewer events in a timed region
  pf_config cfg;  // scheduling knobs (pf_create_cfg); results never depend on them
  bool is_lane = false;   // one of several lanes of pf_novel_view_batch_dev running side by side
  int lanes_running = 1;  // throughput mode: lanes (this one included) solving batches side by side on the device right now
  long fuse_ups_px = 0;   // levels up to this many pixels get their incoming flow upsampled inside their first Gaussian (0 = never)
  int chain_cols = 0, chain_rows = 0;   // size of the stitch-chain result resident in "ch_final"
  long long last_swept_steps = 0;       // wavefront steps of one direction of the last solve (both sweeps, all levels, gated windows)
  // pf_stitch_prefetch: `hint` = the image announced for the NEXT step (one-shot: the next pf_stitch_step latches and clears it, uploads
  // it into "ch_l_next" while its own kernels run, and records it as `ready`); `ready` = what sits in "ch_l_next" for the step after
  // (one-shot as well: that step either consumes it or drops it -- a stale host pointer is never dereferenced or matched later)
  struct HostImage { const uint8_t* src = nullptr; int cols = 0, rows = 0; size_t step = 0; uint64_t sig = 0; /* content signature at upload time (host_image_sig) */ };
  HostImage hint, ready;
  hipStream_t s_copy = nullptr;         // uploads that overlap compute (created on first use, like s_aux: a context that only solves
                                        // pairs drives three streams, so that six lanes of the throughput mode fit the hardware queues)
  bool drained = true;                  // false between "work enqueued" and finish(): what CallGuard looks at
  size_t slab_stride = 0, slab_work_off = 0; int slab_pairs = 0;   // layout the "batch_slab" buffer was last initialised for (alloc_solve_batch)
  std::vector<pf_ctx*> lanes;           // throughput mode: further stream/buffer sets on the same device (pf_novel_view_batch_dev)
  int* h_gate = nullptr; int* d_gate = nullptr; int gate_epoch = 0;   // mapped pinned: per-level gate boxes + count + epoch flag (k_gate_bbox_all)
  int* h_status = nullptr;              // mapped pinned host word: bit d set = a sweep band of direction d timed out
  int* d_status = nullptr;              // the same word as the device sees it
  std::vector<std::string> prof_names;
  std::vector<ProfEntry> prof_tot;
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> ev_pool;
  std::mutex prof_mu;   // the two directions may be enqueued from two host threads
};

namespace {

int fail(pf_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf;
  g_err = buf;
  return code;
}
// A call that drives `needed` HIP streams at once on a runtime that maps streams onto fewer hardware queues runs them partly one after the
// other -- correct, slower, and silent.  The runtime sizes its queue pool from GPU_MAX_HW_QUEUES (default 4) when it is initialised; the
// library cannot change that any more, but it can say so.  (The only environment variable this library looks at, and only to report.)
void check_hw_queues(pf_ctx* c, int needed, const char* what) {
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  const int queues = e ? atoi(e) : 4;
  if (queues <= 0 || needed <= queues) return;
  char buf[512];
  snprintf(buf, sizeof buf, "%s drives %d HIP streams, but the HIP runtime maps streams onto %d hardware queues (GPU_MAX_HW_QUEUES %s): streams share queues "
           "and their kernels serialise; set GPU_MAX_HW_QUEUES >= %d in the environment before the first HIP call of the process", what, needed, queues,
           e ? "as set" : "is unset: the runtime's default", needed);
  c->warn = buf; c->warn_count += 1;
}
#define HIPCHK(c, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(c, PF_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// [OpenCV smooth.cpp] getGaussianKernel(n, sigma, CV_32F)
Gauss make_gauss(int n, double sigma) {
  Gauss g; memset(&g, 0, sizeof g); g.ksize = n;
  const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; g.k[i] = (float)exp(scale2X * x * x); sum += g.k[i]; }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) g.k[i] = (float)(g.k[i] * sum);
  return g;
}

void* ensure(pf_ctx* c, const char* name, size_t bytes) {
  DevBuf& b = c->bufs[name];
  if (b.cap >= bytes && b.p) return b.p;
  if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
  const size_t cap = (bytes + 255) & ~size_t(255);
  if (hipMalloc(&b.p, cap) != hipSuccess) { b.p = nullptr; fail(c, PF_ERR_NOMEM, "hipMalloc(%zu) for '%s' failed", cap, name); return nullptr; }
  b.cap = cap;
#ifdef PF_EXPERIMENTS
  // debugging aid (lab build only): PANOFLOW_POISON=all | <buffer name> fills fresh allocations with 0xFF bytes (NaNs / -1): a result
  // that depends on it reads memory it never wrote
  if (const char* po = getenv("PANOFLOW_POISON")) {
    if (strcmp(po, "all") == 0 || strstr(po, name) != nullptr) { hipMemset(b.p, 0xFF, cap); hipDeviceSynchronize(); }
    else if (strcmp(po, "zero") == 0 || po[0] == '!') { hipMemset(b.p, (po[0] == '!' && strstr(po + 1, name) != nullptr) ? 0xFF : 0x00, cap); hipDeviceSynchronize(); }
  }
#endif
  return b.p;
}

// ---- profiling: HIP events on the stream each kernel family is launched on ----
int prof_id(pf_ctx* c, const char* name) {
  for (size_t i = 0; i < c->prof_names.size(); ++i) if (c->prof_names[i] == name) return (int)i;
  c->prof_names.push_back(name); c->prof_tot.push_back(ProfEntry());
  return (int)c->prof_names.size() - 1;
}
hipEvent_t prof_event(pf_ctx* c) {
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e;   // timing only: no system-scope release at the marker
}
struct ProfScope {
  pf_ctx* c; hipStream_t st; ProfPending p; bool on;
  ProfScope(pf_ctx* c_, hipStream_t st_, const char* name) : c(c_), st(st_), on(c_->prof == 1 || (c_->prof == 2 && strncmp(name, "sweep", 5) == 0)) {
    if (!on) return;
    { std::lock_guard<std::mutex> lk(c->prof_mu); p.id = prof_id(c, name); p.a = prof_event(c); p.b = prof_event(c); }
    hipEventRecord(p.a, st);
  }
  ~ProfScope() { if (on) { hipEventRecord(p.b, st); std::lock_guard<std::mutex> lk(c->prof_mu); c->prof_pending.push_back(p); } }
};
void prof_collect(pf_ctx* c) {
  for (auto& p : c->prof_pending) {
    float ms = 0; hipEventSynchronize(p.b); hipEventElapsedTime(&ms, p.a, p.b);
    c->prof_tot[p.id].ms += ms; c->prof_tot[p.id].n += 1;
    c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
  }
  c->prof_pending.clear();
}
#define PROF(c, st, name) ProfScope prof_scope_##__LINE__(c, st, name)

// ---- pyramid geometry (PixFlow.hpp:137-151) ----
struct Geometry {
  int cols, rows, pad, ce, w0, h0, n;
  std::vector<int> ws, hs;
  std::vector<size_t> off;  // element offset of each level inside a pyramid plane
  size_t P;                 // total level pixels (padded to 64 per level)
  size_t Pexact;
};
Geometry make_geometry(int cols, int rows, int pad) {
  Geometry g; g.cols = cols; g.rows = rows; g.pad = pad; g.ce = cols + 2 * pad;
  g.w0 = int(g.ce * kDownscaleFactor); g.h0 = int(rows * kDownscaleFactor);
  g.ws = {g.w0}; g.hs = {g.h0};
  while ((int)g.ws.size() < kPyrMaxLevels) {
    const int nw = int(g.ws.back() * kPyrScaleFactor + 0.5f), nh = int(g.hs.back() * kPyrScaleFactor + 0.5f);
    if (nh <= kPyrMinImageSize || nw <= kPyrMinImageSize) break;
    g.ws.push_back(nw); g.hs.push_back(nh);
  }
  g.n = (int)g.ws.size();
  size_t o = 0, pe = 0;
  for (int l = 0; l < g.n; ++l) { g.off.push_back(o); const size_t px = size_t(g.ws[l]) * g.hs[l]; pe += px; o += (px + 63) & ~size_t(63); }
  g.P = o; g.Pexact = pe;
  return g;
}

// plain image entry points (blend / stitch): positive size, pixel count inside the kernels' 32-bit indexing
int check_image(pf_ctx* c, int cols, int rows) {
  if (cols <= 0 || rows <= 0) return fail(c, PF_ERR_ARG, "bad image size %dx%d", cols, rows);
  if ((double)cols * rows > 2.0e9) return fail(c, PF_ERR_ARG, "image too large");
  return 0;
}

int check_dims(pf_ctx* c, int cols, int rows, int pad) {
  if (cols <= 0 || rows <= 0) return fail(c, PF_ERR_ARG, "bad image size %dx%d", cols, rows);
  const int w0 = int((cols + 2 * pad) * kDownscaleFactor), h0 = int(rows * kDownscaleFactor);
  if (w0 < 2 || h0 < 2) return fail(c, PF_ERR_ARG, "image %dx%d too small for the half-res solver", cols, rows);
  if ((double)cols * rows > 2.0e9) return fail(c, PF_ERR_ARG, "image too large");
  return 0;
}

// Levels up to this many pixels trade launches for longer kernels (upsample inside the next Gaussian, second median inside the
// diffusion): 0 for a lone pair, set by the throughput mode for its lanes; pf_config::fuse_small_level_px overrides both.
int64_t fuse_small_px(const pf_ctx* c) { return c->cfg.fuse_small_level_px >= 0 ? c->cfg.fuse_small_level_px : c->fuse_ups_px; }

// the product library ships ONE sweep (k_sweep_prep + k_sweep2); the lab build (-DPF_EXPERIMENTS, libpanoflow_exp.so) adds the
// cross-check implementations the test-suite holds it against
inline bool launch_sweep_any(hipStream_t st, const SweepArgs& a, float* rec, bool relax) {
#ifdef PF_EXPERIMENTS
  if (relax) return launch_sweep_relax(st, a);
#endif
  (void)relax;
  return launch_sweep2(st, a, rec);
}

// One level of one direction (PixFlow.hpp:272-340, gradients excluded: they are precomputed for all levels).
// flow_a holds the incoming flow and receives the level's result (flow_b, blurred, tmp are scratch).
struct LevelBufs { float *flow_a, *flow_b, *blurred, *tmp, *rec; };
// box = bounding box (min x, min y, max x, max y) of the gated pixels of this level, or nullptr for "everything"
void run_level(pf_ctx* c, hipStream_t st, const float* g0, const float* g1, const float* a0, const float* a1, const uint8_t* gate, int w, int h, int sparse,
               const int* box, const LevelBufs& b, unsigned long long* bnd_fwd, unsigned long long* bnd_bwd, int* ctrl_fwd, int* ctrl_bwd, float** result,
               int* pc_fwd = nullptr, int* pc_bwd = nullptr, const float* ups_src = nullptr, int ups_w = 0, int ups_h = 0, Batch bt = Batch()) {
  // ups_src: flow_a does not hold this level's incoming flow yet -- it is the upsample of the coarser level's result (ups_w x ups_h),
  // computed by the Gaussian's tile loader on the way (small levels: one launch instead of two)
  if (ups_src) { PROF(c, st, "gauss15_blurredFlow"); launch_gauss15_upsample(st, ups_src, ups_w, ups_h, 1.0f / kPyrScaleFactor, b.flow_a, b.blurred, w, h, c->g15, bt); }
  else { PROF(c, st, "gauss15_blurredFlow"); launch_gauss15(st, b.flow_a, b.tmp, b.blurred, w, h, c->g15, bt); }
  SweepArgs sa;
  sa.bt = bt;
  sa.g0 = reinterpret_cast<const float2*>(g0); sa.g1 = reinterpret_cast<const float2*>(g1);
  sa.blurred = reinterpret_cast<const float2*>(b.blurred); sa.gate = gate; sa.W = w; sa.H = h; sa.sparse = sparse;
  if (box) { sa.ax0 = box[0]; sa.ay0 = box[1]; sa.ax1 = box[2] + 1; sa.ay1 = box[3] + 1; }   // empty (max < min): the sweeps are the identity
  // workgroup shape (pf_config::sweep_wide): a lone pair never oversubscribes the chip (126 workgroups at 9000x4000) and keeps the latency form
  sa.wide = c->cfg.sweep_wide < 0 ? (bt.n > 1 ? -1 : 0) : c->cfg.sweep_wide;
  sa.wide_threshold_wgs = c->cfg.sweep_wide_threshold;
  sa.wide_tr = c->cfg.sweep_throughput_transposed;
  sa.concurrent_sweeps = 2 * bt.n * c->lanes_running;   // both directions of every pair of every lane's batch sweep at the same time
  // Timing a sweep (profile mode 1 or 2) attaches the two events to the launches themselves (hipExtLaunchKernel) instead of
  // recording markers around them.  Same-box A/B, ms per step: no timing 27.38, markers 27.65, attached events 27.60 -- bench.py's
  // roofline needs per-launch HIP events inside its timed region, so ~0.2 ms of every timed step is the measurement itself.
  auto sweep = [&](SweepArgs& a) {
#ifdef PF_EXPERIMENTS
    if (c->cfg.sweep_impl == 1) { PROF(c, st, "sweep"); launch_sweep(st, a); return; }
    const bool relax = c->cfg.sweep_impl == 3;
    a.prep_mode = c->cfg.record_path;
#else
    const bool relax = false;
#endif
    if (!c->prof) { launch_sweep_any(st, a, b.rec, relax); return; }
    ProfPending p;
    { std::lock_guard<std::mutex> lk(c->prof_mu); p.id = prof_id(c, "sweep"); p.a = prof_event(c); p.b = prof_event(c); }
    a.ev_start = p.a; a.ev_stop = p.b;
    const bool launched = launch_sweep_any(st, a, b.rec, relax);
    a.ev_start = nullptr; a.ev_stop = nullptr;
    std::lock_guard<std::mutex> lk(c->prof_mu);
    if (launched) c->prof_pending.push_back(p); else { c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b); }
  };
  { sa.flow = reinterpret_cast<float2*>(b.flow_a); sa.boundary = bnd_fwd; sa.ctrl = ctrl_fwd; sa.prepcnt = pc_fwd; sa.forward = 1; sweep(sa); }
  { PROF(c, st, "median5"); launch_median5(st, b.flow_a, b.flow_b, w, h, bt); }
  { sa.flow = reinterpret_cast<float2*>(b.flow_b); sa.boundary = bnd_bwd; sa.ctrl = ctrl_bwd; sa.prepcnt = pc_bwd; sa.forward = 0; sweep(sa); }
  if ((long)w * h <= fuse_small_px(c)) {
    // throughput mode, small levels: the second median rides in the diffusion's tile loader (one launch fewer; result in b.tmp,
    // which nothing else uses: it must not be flow_a, the plane the next level's incoming flow is written to)
    PROF(c, st, "gauss15_diffusion"); launch_median_gauss15_mix(st, b.flow_b, a0, a1, w, h, c->g15, b.tmp, bt);
    *result = b.tmp;
    return;
  }
  { PROF(c, st, "median5"); launch_median5(st, b.flow_b, b.flow_a, w, h, bt); }
  { PROF(c, st, "gauss15_diffusion"); launch_gauss15_mix(st, b.flow_a, b.tmp, a0, a1, w, h, c->g15, b.flow_b, bt); }
  *result = b.flow_b;
}

// bounding boxes of the gated pixels of the levels described by t (device gate plane) -> host; one stream sync
int gate_boxes_to_host(pf_ctx* c, hipStream_t st, const uint8_t* gate, const LevelTable& t, size_t total, std::vector<int>& box) {
  box.assign(size_t(t.n) * 4, 0);
  for (int l = 0; l < t.n; ++l) { box[4 * l] = 0x7fffffff; box[4 * l + 1] = 0x7fffffff; box[4 * l + 2] = -1; box[4 * l + 3] = -1; }
  int* d_box = (int*)ensure(c, "gate_box", size_t(kLevelTableMax) * 4 * sizeof(int));
  if (!d_box) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemcpyAsync(d_box, box.data(), box.size() * sizeof(int), hipMemcpyHostToDevice, st));
  launch_gate_bbox(st, gate, t, total, d_box);
  HIPCHK(c, hipMemcpyAsync(box.data(), d_box, box.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return 0;
}

// Everything one solve keeps in HBM (named grow-only arena): shared pyramids / gradients / gate, per-direction flow
// planes, record buffers, hand-off granules and control words.  Also used by pf_create's pre-sizing.
struct SolveBufs {
  float* pyrI[2]; float* pyrA[2]; float* grad[2];
  uint8_t* gate; float* half_tmp;
  std::vector<size_t> bnd_off; size_t bnd_total;
  LevelBufs lb[2]; unsigned long long* bnd[2]; int* ctrl[2]; float* ratio[2];
  int* prepcnt[2]; std::vector<size_t> pc_off; size_t pc_total;   // per sweep launch: one "records ready" counter per sweep workgroup
  int* gate_work = nullptr;     // batch slabs only: this pair's work area of k_gate_bbox_all (a lone solve uses the context's "gate_work")
  float* nv_flow[2] = {nullptr, nullptr};   // batch slabs only: internal flow planes for pairs whose caller does not want the flows
};
// Where a solve's buffers come from: the context's named grow-only arena (a lone solve), or -- for a batch of pairs solved by the same
// launches -- one slab per pair, all with the same layout and `stride` bytes apart, so that a kernel reaches pair z's copy of any buffer
// by adding z * stride to pair 0's pointer (pf_common.hpp: Batch).  base == nullptr is the sizing pass.
struct Carver {
  pf_ctx* c; bool slab; char* base; size_t off;
  void* get(const char* name, size_t bytes) {
    if (!slab) return ensure(c, name, bytes);
    const size_t o = off;
    off += (bytes + 255) & ~size_t(255);
    return base ? static_cast<void*>(base + o) : reinterpret_cast<void*>(size_t(256));   // sizing pass: any non-null value
  }
};
int alloc_solve(Carver& cv, const Geometry& g, int ndirs, SolveBufs& b) {
  pf_ctx* c = cv.c;
  const size_t n0 = size_t(g.w0) * g.h0;
  const char* nI[2] = {"pyrI0", "pyrI1"}; const char* nA[2] = {"pyrA0", "pyrA1"}; const char* nG[2] = {"grad0", "grad1"};
  for (int i = 0; i < 2; ++i) {
    b.pyrI[i] = (float*)cv.get(nI[i], g.P * 4); b.pyrA[i] = (float*)cv.get(nA[i], g.P * 4); b.grad[i] = (float*)cv.get(nG[i], g.P * 8);
    if (!b.pyrI[i] || !b.pyrA[i] || !b.grad[i]) return PF_ERR_NOMEM;
  }
  b.gate = (uint8_t*)cv.get("gate", g.P);
  b.half_tmp = (float*)cv.get("half_tmp", n0 * 4);
  if (!b.gate || !b.half_tmp) return PF_ERR_NOMEM;
  // hand-off rows + control words of every sweep launch of this solve
  b.bnd_off.assign(g.n, 0);
  b.bnd_total = 0;
  for (int l = 0; l < g.n; ++l) { b.bnd_off[l] = b.bnd_total; b.bnd_total += sweep_boundary_elems(g.ws[l], g.hs[l]); }
  b.pc_off.assign(g.n, 0);
  b.pc_total = 0;
  for (int l = 0; l < g.n; ++l) { b.pc_off[l] = b.pc_total; b.pc_total += 2 * size_t(sweep2_num_wgs_max(g.ws[l], g.hs[l])); }   // forward + backward sweep
  const char* nb[2][8] = {{"d0_flow_a", "d0_flow_b", "d0_blurred", "d0_tmp", "d0_bnd", "d0_ctrl", "d0_ratio", "d0_rec"},
                          {"d1_flow_a", "d1_flow_b", "d1_blurred", "d1_tmp", "d1_bnd", "d1_ctrl", "d1_ratio", "d1_rec"}};
  for (int d = 0; d < ndirs; ++d) {
    b.lb[d].flow_a = (float*)cv.get(nb[d][0], n0 * 8); b.lb[d].flow_b = (float*)cv.get(nb[d][1], n0 * 8);
    b.lb[d].blurred = (float*)cv.get(nb[d][2], n0 * 8); b.lb[d].tmp = (float*)cv.get(nb[d][3], n0 * 8);
    b.bnd[d] = (unsigned long long*)cv.get(nb[d][4], b.bnd_total * 2 * 8);
    b.ctrl[d] = (int*)cv.get(nb[d][5], size_t(g.n) * 2 * 2 * sizeof(int));
    b.ratio[d] = (float*)cv.get(nb[d][6], 256);
    b.prepcnt[d] = (int*)cv.get(d == 0 ? "d0_prepcnt" : "d1_prepcnt", b.pc_total * sizeof(int));
    if (!b.prepcnt[d]) return PF_ERR_NOMEM;
    b.lb[d].rec = (float*)cv.get(nb[d][7], sweep2_rec_bytes(g.w0, g.h0));
    if (!b.lb[d].rec) return PF_ERR_NOMEM;
    if (!b.lb[d].flow_a || !b.lb[d].flow_b || !b.lb[d].blurred || !b.lb[d].tmp || !b.bnd[d] || !b.ctrl[d] || !b.ratio[d]) return PF_ERR_NOMEM;
  }
  if (cv.slab) {
    b.gate_work = (int*)cv.get("gate_work", (4 * kLevelTableMax + 2) * sizeof(int));
    for (int d = 0; d < 2; ++d) b.nv_flow[d] = (float*)cv.get(d ? "nv_flow_r2l" : "nv_flow_l2r", size_t(g.cols) * g.rows * 8);
  } else if (!ensure(c, "gate_box", size_t(kLevelTableMax) * 4 * sizeof(int)) || !ensure(c, "gate_count", 256)) return PF_ERR_NOMEM;
  (void)c;
  return 0;
}
int alloc_solve(pf_ctx* c, const Geometry& g, int ndirs, SolveBufs& b) { Carver cv{c, false, nullptr, 0}; return alloc_solve(cv, g, ndirs, b); }
// slabs of a batch of nb pairs: returns pair 0's buffers and the slab stride
int alloc_solve_batch(pf_ctx* c, const Geometry& g, int nb, SolveBufs& b, size_t& stride) {
  Carver sizing{c, true, nullptr, 0};
  if (int e = alloc_solve(sizing, g, 2, b)) return e;
  stride = (sizing.off + 4095) & ~size_t(4095);
  const bool fresh = c->bufs.find("batch_slab") == c->bufs.end() || c->bufs["batch_slab"].cap < stride * size_t(nb);
  char* base = (char*)ensure(c, "batch_slab", stride * size_t(nb));
  if (!base) return PF_ERR_NOMEM;
  Carver cv{c, true, base, 0};
  if (int e = alloc_solve(cv, g, 2, b)) return e;
  const size_t work_off = size_t(reinterpret_cast<char*>(b.gate_work) - base);
  if (fresh || c->slab_stride != stride || c->slab_work_off != work_off || c->slab_pairs < nb) {   // new memory or a new layout: (re)initialise the self-resetting work areas
    std::vector<int> init(4 * kLevelTableMax + 2, 0);
    for (int l = 0; l < kLevelTableMax; ++l) { init[4 * l] = 0x7fffffff; init[4 * l + 1] = 0x7fffffff; init[4 * l + 2] = -1; init[4 * l + 3] = -1; }
    for (int p = 0; p < nb; ++p)
      if (hipMemcpy(reinterpret_cast<char*>(b.gate_work) + size_t(p) * stride, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(c, PF_ERR_DEVICE, "initialising the batch slabs failed");
    c->slab_stride = stride; c->slab_work_off = work_off; c->slab_pairs = nb;
  }
  return 0;
}

// device work area of k_gate_bbox_all (self-resetting: initialised once)
int* gate_work(pf_ctx* c) {
  const bool fresh = c->bufs.find("gate_work") == c->bufs.end() || !c->bufs["gate_work"].p;
  int* w = (int*)ensure(c, "gate_work", (4 * kLevelTableMax + 2) * sizeof(int));
  if (w && fresh) {
    std::vector<int> init(4 * kLevelTableMax + 2, 0);
    for (int l = 0; l < kLevelTableMax; ++l) { init[4 * l] = 0x7fffffff; init[4 * l + 1] = 0x7fffffff; init[4 * l + 2] = -1; init[4 * l + 3] = -1; }
    if (hipMemcpy(w, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  }
  return w;
}
// Host side of k_gate_bbox_all: poll the epoch flag in mapped pinned memory (microseconds) instead of synchronising the
// stream; boxes (4 ints per level) and the level-0 count are then already in host memory.
int wait_gate_boxes(pf_ctx* c, hipStream_t st, int epoch, int nlevels, std::vector<int>& box, unsigned& count0, int pair = 0) {
  const int* hg = c->h_gate + size_t(pair) * kGateWords;
  volatile const int* flag = hg + 4 * kLevelTableMax + 1;
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != epoch) {
    // the wait is microseconds long: stay on the core, but leave the pipeline to its sibling thread
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
    if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
      HIPCHK(c, hipStreamSynchronize(st));   // surfaces a launch failure, if that is what happened
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != epoch) return fail(c, PF_ERR_DEVICE, "gate bounding boxes never arrived");
    }
  }
  box.assign(hg, hg + size_t(nlevels) * 4);
  count0 = (unsigned)hg[4 * kLevelTableMax];
  return 0;
}

// The whole solver for 1 or 2 directions on device-resident packed BGRA images, for nb same-size pairs at once.
// dir 0: I0 = img0, I1 = img1, hint0;  dir 1: I0 = img1, I1 = img0, hint1.  out[p * 2 + d]: cols x rows float2 (pad cropped).
// nb == 1: the context's arena.  nb > 1 (throughput mode): the pairs' buffers are slabs of one layout, every kernel covers all pairs
// (blockIdx.z = pair) -- ONE kernel boundary per step of the algorithm for nb pairs; the sweeps of a level share one window, the
// union of the pairs' bounding boxes (a sweep over a larger window gives the same result: pixels outside a pair's own box are not
// gated and keep their flow).
int solve_n(pf_ctx* c, int nb, const uint8_t* const* d_img0, const uint8_t* const* d_img1, int cols, int rows, int pad, int max_pct, int ndirs, const int* hints,
            float* const* d_out, float** used_out = nullptr /* [nb * 2]: where each flow went (a NULL d_out entry of a batch = a plane inside the pair's slab) */) {
  if (int e = check_dims(c, cols, rows, pad)) return e;
  if (max_pct < 0 || max_pct > 100) return fail(c, PF_ERR_ARG, "max_percentage %d out of range", max_pct);
  if (nb < 1 || nb > kMaxBatch) return fail(c, PF_ERR_ARG, "batch of %d pairs (1..%d)", nb, kMaxBatch);
  const Geometry g = make_geometry(cols, rows, pad);
  SolveBufs sb;
  Batch bt;
  if (nb == 1) { if (int e = alloc_solve(c, g, ndirs, sb)) return e; }
  else {
    if (g.n > kLevelTableMax || g.P >= (size_t(1) << 31)) return fail(c, PF_ERR_ARG, "image too large for a batched solve");
    size_t stride = 0;
    if (int e = alloc_solve_batch(c, g, nb, sb, stride)) return e;
    bt.n = nb; bt.stride = stride;
  }
  float** pyrI = sb.pyrI; float** pyrA = sb.pyrA; float** grad = sb.grad;
  uint8_t* gate = sb.gate; float* half_tmp = sb.half_tmp;
  const std::vector<size_t>& bnd_off = sb.bnd_off; const size_t bnd_total = sb.bnd_total;
  LevelBufs* lb = sb.lb; unsigned long long** bnd = sb.bnd; int** ctrl = sb.ctrl; float** ratio = sb.ratio;
  *c->h_status = 0;
  hipStream_t sm = c->s_main;
  // --- shared front end on the main stream: half-res planes, pyramids, gradients + gate of ALL levels.
  // (Measured and rejected: the alpha path on a second stream -- alpha pyramids, gate, boxes beside the grey path: +0.3 ms
  // per pair with 72 instead of 36 small pyramid launches in front of the boxes; profiles/r02_frontend_ab.txt.) ---
  hipStream_t sg = sm;
  for (int i = 0; i < 2; ++i) {
    ExtPtrs imgs{};
    for (int p = 0; p < nb; ++p) imgs.p[p] = i ? d_img1[p] : d_img0[p];
    { PROF(c, sm, "downscale_gray"); launch_downscale_gray(sm, nullptr, cols, rows, pad, half_tmp, pyrA[i], g.w0, g.h0, bt, &imgs); }
    { PROF(c, sm, "preblur5"); launch_gauss_small(sm, half_tmp, pyrI[i], g.w0, g.h0, 1, c->g5, bt); }
  }
  // pyramids: one launch per level while the levels are large, then two and three levels per launch (the chain of dependent
  // ~5 us launches is otherwise ~0.2 ms in front of everything; kernels_pre.hip: k_pyr_chain)
  const int chainMode = c->cfg.pyramid_chaining;
  for (int l = 1; l < g.n;) {
    PROF(c, sm, "pyr_down");
    const size_t px = size_t(g.ws[l]) * g.hs[l];
    int k = 1;
    if (chainMode) { if (px <= 40000 && l + 2 < g.n) k = 3; else if (px <= 160000 && l + 1 < g.n) k = 2; }
    if (k == 1)
      launch_pyr_down4(sm, pyrI[0] + g.off[l - 1], pyrI[1] + g.off[l - 1], pyrA[0] + g.off[l - 1], pyrA[1] + g.off[l - 1], g.ws[l - 1], g.hs[l - 1],
                       pyrI[0] + g.off[l], pyrI[1] + g.off[l], pyrA[0] + g.off[l], pyrA[1] + g.off[l], g.ws[l], g.hs[l], bt);
    else
      launch_pyr_chain4(sm, pyrI[0], pyrI[1], pyrA[0], pyrA[1], g.ws.data(), g.hs.data(), g.off.data(), l - 1, k, bt);
    l += k;
  }
  // The host needs the per-level bounding boxes of the gate (they size the sweep launches) and the level-0 gate count (dense
  // or sparse sweep variant; full-canvas inputs, CPU/StitchTool.cpp:17-33): one fused kernel computes gate, boxes and count
  // and publishes them into mapped pinned memory; the host polls its epoch flag (microseconds, no blocking sync, no pageable
  // copies) while the gradients of all levels and the hand-off initialisation are still running behind it.
  bool have_table = false; LevelTable table;
  const int split = g.n > 10 ? 8 : 0;   // levels [0, split) are "fine": 80 % of the pixels of a 0.9x pyramid
  unsigned h_cnt = 0;
  std::vector<int> boxes;
  int epoch = 0;
  if (g.n <= kLevelTableMax && g.P < (size_t(1) << 31)) {
    LevelTable t; t.n = g.n;
    for (int l = 0; l < g.n; ++l) { t.w[l] = g.ws[l]; t.h[l] = g.hs[l]; t.off[l] = (unsigned)g.off[l]; }
    int* work = nb == 1 ? gate_work(c) : sb.gate_work;
    if (!work) return PF_ERR_NOMEM;
    epoch = ++c->gate_epoch;
    { PROF(c, sg, "gate"); launch_gate_bbox_all(sg, pyrA[0], pyrA[1], gate, t, g.P, work, c->d_gate, epoch, bt, kGateWords * sizeof(int)); }
    // gradients: the coarse levels first (a few percent of the pixels) -- the directions start on those -- the fine levels in a
    // second launch that runs while the coarse levels are already being solved (ev_fine, waited for at level split - 1)
    { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], t, g.off[split], g.P, c->g3_05, 0, bt); }
    have_table = true; table = t;
  } else {
    for (int l = 0; l < g.n; ++l) {
      PROF(c, sm, "gradients");
      launch_gradients(sm, pyrI[0] + g.off[l], g.ws[l], g.hs[l], grad[0] + 2 * g.off[l], c->g3_05);
      launch_gradients(sm, pyrI[1] + g.off[l], g.ws[l], g.hs[l], grad[1] + 2 * g.off[l], c->g3_05);
      launch_gate(sg, pyrA[0] + g.off[l], pyrA[1] + g.off[l], g.ws[l] * g.hs[l], gate + g.off[l]);
    }
  }
  for (int d = 0; d < ndirs; ++d) {
    PROF(c, sm, "init_handoff");
    launch_fill_u64(sm, bnd[d], bnd_total * 2, kNotReady, bt);
    launch_fill_u32(sm, reinterpret_cast<unsigned*>(ctrl[d]), size_t(g.n) * 2 * 2, 0u, bt);
    launch_fill_u32(sm, reinterpret_cast<unsigned*>(sb.prepcnt[d]), sb.pc_total, 0u, bt);
  }
  HIPCHK(c, hipEventRecord(c->ev_pre, sm));
  // Fine levels in two launches behind the coarse ones: levels [split2, split) (needed first, a quarter of the fine pixels), then the
  // finest levels [0, split2).  For a lone pair BOTH are NARROW launches: they run beside the sweeps of ~30 coarser levels and are not
  // needed for milliseconds, while at full width they take every wave slot of the chip -- and a sweep workgroup needs 11 free wave
  // slots and 115 KB of LDS on ONE CU: the first sweep of the first direction used to wait ~150 us for the full-width launch of
  // [split2, split) to drain (kernel timeline, tests/micro/pair_timeline.py), and the late direction, which starts k levels behind the
  // first and ends the call, with it (dense pair 46.23 -> 46.11 ms, profiles/r04_sweep_helpers_ab.txt 8).
  // (a batch keeps every CU busy anyway -- there is nothing to hide a narrow launch behind, and at 64 blocks per image it would run
  // for the whole solve: full width, pf_config::full_width_batch_gradients)
  const int fineBlocks = (nb > 1 && c->cfg.full_width_batch_gradients) ? 0 : c->cfg.fine_gradient_blocks;
  const int split2 = split > 4 ? 4 : 0;
  if (have_table && split > 0) { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], table, g.off[split2], g.off[split], c->g3_05, fineBlocks, bt); }
  HIPCHK(c, hipEventRecord(c->ev_fine, sm));
  if (have_table && split2 > 0) { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], table, 0, g.off[split2], c->g3_05, fineBlocks, bt); }
  HIPCHK(c, hipEventRecord(c->ev_fine2, sm));
  double area0 = (double)g.ws[0] * g.hs[0];   // the sweeps only cover the window of gated pixels: density inside that window is what counts
  if (have_table) {
    // one set of boxes per pair; a batch sweeps the union (a superset of each pair's own window: same results)
    for (int p = 0; p < nb; ++p) {
      std::vector<int> bp; unsigned cnt = 0;
      if (int e = wait_gate_boxes(c, sg, epoch, g.n, bp, cnt, p)) return e;
      h_cnt += cnt;
      if (p == 0) boxes = bp;
      else for (int l = 0; l < g.n; ++l) {
        if (bp[4 * l + 2] < bp[4 * l] || bp[4 * l + 3] < bp[4 * l + 1]) continue;                          // this pair gates nothing at level l
        if (boxes[4 * l + 2] < boxes[4 * l] || boxes[4 * l + 3] < boxes[4 * l + 1]) { for (int k = 0; k < 4; ++k) boxes[4 * l + k] = bp[4 * l + k]; continue; }
        boxes[4 * l] = std::min(boxes[4 * l], bp[4 * l]); boxes[4 * l + 1] = std::min(boxes[4 * l + 1], bp[4 * l + 1]);
        boxes[4 * l + 2] = std::max(boxes[4 * l + 2], bp[4 * l + 2]); boxes[4 * l + 3] = std::max(boxes[4 * l + 3], bp[4 * l + 3]);
      }
    }
    if (!c->cfg.sweep_window) boxes.clear();
  } else {
    unsigned* d_cnt = (unsigned*)ensure(c, "gate_count", 256);
    if (!d_cnt) return PF_ERR_NOMEM;
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 4, sg));
    launch_count_gate(sg, gate, g.ws[0] * g.hs[0], d_cnt);
    HIPCHK(c, hipMemcpyAsync(&h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, sg));
    HIPCHK(c, hipStreamSynchronize(sg));
  }
  if (!boxes.empty() && boxes[2] >= boxes[0] && boxes[3] >= boxes[1]) area0 = double(boxes[2] - boxes[0] + 1) * double(boxes[3] - boxes[1] + 1);
  int sparse = (double)h_cnt < 0.5 * area0 * nb ? 1 : 0;
  if (c->cfg.sparse_sweep >= 0) sparse = c->cfg.sparse_sweep ? 1 : 0;   // forced variant: results are identical either way
  // critical path of the exact sweeps given the windows: (w + h - 1) anti-diagonals per sweep, two sweeps per level
  c->last_swept_steps = 0;
  for (int l = 0; l < g.n; ++l) {
    int bw = g.ws[l], bh = g.hs[l];
    if (!boxes.empty()) { bw = boxes[4 * l + 2] - boxes[4 * l] + 1; bh = boxes[4 * l + 3] - boxes[4 * l + 1] + 1; }
    if (bw > 0 && bh > 0) c->last_swept_steps += 2 * (long long)(bw + bh - 1);
  }

  // --- the two directions are independent (OpticalFlow.cpp:130-139): one stream each.  The host enqueues them level by
  // level in turn (a direction's ~430 launches take the host >1 ms: enqueued one after the other, the second
  // direction's stream would sit idle that long) ---
  for (int d = 0; d < ndirs; ++d) HIPCHK(c, hipStreamWaitEvent(c->s_dir[d], c->ev_pre, 0));
  // Fewer launches or shorter launches?  Alone, a pair is faster with the separate upsample kernel (strip 27.36 vs 27.44 ms); with
  // several pairs in flight the time between a stream's kernels dominates and one launch fewer per level wins (+3 %): the
  // throughput mode turns the fusion on for its lanes (pf_novel_view_batch_dev).  pf_config::fuse_small_level_px overrides both.
  const long fuseUpsPx = fuse_small_px(c);
  auto fuse_ups = [&](int level) { return (long)g.ws[level] * g.hs[level] <= fuseUpsPx; };   // level whose incoming flow is upsampled inside its Gaussian
  float* prev_res[2] = {nullptr, nullptr};
  auto enqueue_level = [&](int d, int level) {
    hipStream_t st = c->s_dir[d];
    if (level == split - 1) hipStreamWaitEvent(st, c->ev_fine, 0);   // first level whose gradients come from the second launch
    if (split2 > 0 && level == split2 - 1) hipStreamWaitEvent(st, c->ev_fine2, 0);   // ... from the third (narrow) launch
    const int i0 = d, i1 = 1 - d;
    LevelBufs& b = lb[d];
    const int w = g.ws[level], h = g.hs[level];
    const size_t o = g.off[level];
    if (level == g.n - 1) {
      launch_fill_u32(st, reinterpret_cast<unsigned*>(b.flow_a), size_t(w) * h * 2, 0u, bt);  // PixFlow.hpp:298
      if (max_pct > 0 && hints[d] != PF_HINT_UNKNOWN) {
        PROF(c, st, "adjust_initial_flow");
        launch_adjust_initial_flow(st, pyrI[i0] + o, pyrI[i1] + o, pyrA[i0] + o, pyrA[i1] + o, w, h, hints[d], max_pct, ratio[d], b.flow_a, bt);
      }
    }
    float* res = nullptr;
    // small levels: the upsample of the previous (coarser) level's result rides in this level's first Gaussian
    const bool upsHere = level < g.n - 1 && fuse_ups(level);
    run_level(c, st, grad[i0] + 2 * o, grad[i1] + 2 * o, pyrA[i0] + o, pyrA[i1] + o, gate + o, w, h, sparse, boxes.empty() ? nullptr : &boxes[4 * level], b,
              bnd[d] + bnd_off[level],
              bnd[d] + bnd_total + bnd_off[level], ctrl[d] + level * 4, ctrl[d] + level * 4 + 2, &res,
              sb.prepcnt[d] + sb.pc_off[level], sb.prepcnt[d] + sb.pc_off[level] + sweep2_num_wgs_max(w, h),
              upsHere ? prev_res[d] : nullptr, upsHere ? g.ws[level + 1] : 0, upsHere ? g.hs[level + 1] : 0, bt);
    prev_res[d] = res;
    if (level > 0) {
      if (!fuse_ups(level - 1)) {
        PROF(c, st, "upsample_cubic");
        launch_upsample_cubic(st, res, w, h, b.flow_a, g.ws[level - 1], g.hs[level - 1], 1.0f / kPyrScaleFactor, bt);
      }
    } else {
      PROF(c, st, "final_flow");
      ExtPtrs outs{};
      for (int p = 0; p < nb; ++p) {
        float* o = d_out[p * 2 + d];
        if (!o && nb > 1) o = reinterpret_cast<float*>(reinterpret_cast<char*>(sb.nv_flow[d]) + size_t(p) * bt.stride);
        outs.p[p] = o;
        if (used_out) used_out[p * 2 + d] = o;
      }
      launch_final_flow(st, res, w, h, g.ce, rows, pad, 1.0f / kDownscaleFactor, c->g3_1, nullptr, bt, &outs);
    }
  };
  // (Measured and rejected: one host thread per direction -- +0.1 ms per pair; the GPU, not the host, paces the launches.)
  // Direction 1 starts when direction 0 has finished its k coarsest levels.  Started together, the two directions stay in lockstep:
  // their throughput kernels (Gaussians, medians, prepass) run beside each other, each at half speed, and their sweeps -- which
  // leave most CUs idle -- run beside each other too.  A small offset puts one direction's throughput kernels beside the other's
  // sweeps.  The late direction finishes k coarse levels later, which is what limits k: measured (profiles/r02_frontend_ab.txt)
  // strip 27.36 -> 27.18 ms at k = 2, 9000x4000 pair 59.1 -> 57.7 ms at k = 4-6.  Not for the lanes of the throughput mode (they are
  // out of phase with each other anyway: -1 %).  pf_config::stagger_levels overrides.
  const int stagger = c->cfg.stagger_levels >= 0 ? c->cfg.stagger_levels : (c->is_lane ? 0 : (size_t(g.w0) * g.h0 >= 5000000 ? 4 : 2));
  if (stagger > 0 && ndirs == 2 && g.n > 1) {
    const int k = stagger < g.n ? stagger : g.n - 1;
    for (int t = 0; t < g.n + k; ++t) {
      const int l0 = g.n - 1 - t, l1 = g.n - 1 - (t - k);
      if (l0 >= 0) {
        enqueue_level(0, l0);
        if (t == k - 1) HIPCHK(c, hipEventRecord(c->ev_stagger, c->s_dir[0]));
      }
      if (t >= k && l1 >= 0) {
        if (t == k) HIPCHK(c, hipStreamWaitEvent(c->s_dir[1], c->ev_stagger, 0));
        enqueue_level(1, l1);
      }
    }
  } else {
    for (int level = g.n - 1; level >= 0; --level)
      for (int d = 0; d < ndirs; ++d) enqueue_level(d, level);
  }
  for (int d = 0; d < ndirs; ++d) {
    launch_collect_status(c->s_dir[d], ctrl[d], g.n * 4, c->d_status, 1 << d, bt);
    HIPCHK(c, hipEventRecord(c->ev_dir[d], c->s_dir[d]));
    HIPCHK(c, hipStreamWaitEvent(sm, c->ev_dir[d], 0));
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
int solve(pf_ctx* c, const uint8_t* d_img0, const uint8_t* d_img1, int cols, int rows, int pad, int max_pct, int ndirs, const int* hints,
          float* const* d_out) {
  float* outs[2] = {d_out[0], ndirs > 1 ? d_out[1] : nullptr};
  return solve_n(c, 1, &d_img0, &d_img1, cols, rows, pad, max_pct, ndirs, hints, outs);
}

// after the streams have drained: did any sweep band give up?  (the word lives in mapped pinned host memory and was
// written by k_collect_status at the end of each direction's stream: no copy, no further sync)
int check_sweeps(pf_ctx* c) {
  const int st = __atomic_load_n(c->h_status, __ATOMIC_ACQUIRE);
  if (st) { *c->h_status = 0; return fail(c, PF_ERR_TIMEOUT, "an in-kernel wait timed out (bits 0/1: sweep band of direction 0/1, bit 2: blend-ramp grid barrier; mask %d)", st); }
  return 0;
}

int finish(pf_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->s_main));
  HIPCHK(c, hipStreamSynchronize(c->s_dir[0]));
  HIPCHK(c, hipStreamSynchronize(c->s_dir[1]));
  if (c->s_aux) HIPCHK(c, hipStreamSynchronize(c->s_aux));
  if (c->s_copy) HIPCHK(c, hipStreamSynchronize(c->s_copy));
  c->drained = true;
  if (c->prof) prof_collect(c);
  return 0;
}

// Every entry point that enqueues work owns one of these: whichever way the call returns (also on an early error,
// with copies from the caller's buffers or kernels still in flight), all four streams are idle afterwards, so the
// caller may free or reuse its buffers and the next call starts from a clean pipeline.
struct CallGuard {
  pf_ctx* c;
  explicit CallGuard(pf_ctx* c_) : c(c_) { if (c) c->drained = false; }
  ~CallGuard() {
    if (!c || c->drained) return;   // the normal exit went through finish(): nothing is in flight
    c->drained = true;
    hipStreamSynchronize(c->s_main); hipStreamSynchronize(c->s_dir[0]); hipStreamSynchronize(c->s_dir[1]);
    if (c->s_aux) hipStreamSynchronize(c->s_aux);
    if (c->s_copy) hipStreamSynchronize(c->s_copy);
  }
};

int use(pf_ctx* c) {
  if (!c) return fail(nullptr, PF_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}

// packed rows on both sides (the usual case): one linear copy -- the DMA engines at the link rate when the host side is pinned
int up2d(pf_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, int rows) {
  if (dpitch == width_bytes && spitch == width_bytes) HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * size_t(rows), hipMemcpyHostToDevice, c->s_main));
  else HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyHostToDevice, c->s_main));
  return 0;
}
int down2d(pf_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, int rows) {
  if (dpitch == width_bytes && spitch == width_bytes) HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * size_t(rows), hipMemcpyDeviceToHost, c->s_main));
  else HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToHost, c->s_main));
  return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int pf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#ifdef PF_EXPERIMENTS
const char* pf_version(void) { return "panoflow-mi355x r3 (gfx950, lab build with the cross-check sweeps)"; }
#else
const char* pf_version(void) { return "panoflow-mi355x r3 (gfx950)"; }
#endif

void pf_config_init(pf_config* cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (int)sizeof *cfg;
  cfg->stagger_levels = -1; cfg->fuse_small_level_px = -1; cfg->fine_gradient_blocks = 64; cfg->pyramid_chaining = 1;
  cfg->sweep_window = 1; cfg->sparse_sweep = -1; cfg->sweep_impl = 2; cfg->record_path = 0; cfg->batch_pairs = -1;
  cfg->sweep_wide = -1; cfg->sweep_wide_threshold = 512; cfg->sweep_throughput_transposed = 1; cfg->full_width_batch_gradients = 1;
}

pf_ctx* pf_create(int device, int max_cols, int max_rows) {
  pf_config cfg; pf_config_init(&cfg);
  cfg.device = device; cfg.max_cols = max_cols; cfg.max_rows = max_rows;
  return pf_create_cfg(&cfg);
}

}  // extern "C"

namespace {
// lane = one of the extra stream / buffer sets of the throughput mode: it only ever runs pf_novel_view_dev, so it is pre-sized
// for a solve and the two internal flow planes, not for the stitch chain and the host-staging buffers (~93 B/px it would never use)
pf_ctx* create_ctx(const pf_config& cfg, bool lane) {
  const int device = cfg.device, max_cols = cfg.max_cols, max_rows = cfg.max_rows;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fail(nullptr, PF_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)"); return nullptr; }
  if (device < 0 || device >= n) { fail(nullptr, PF_ERR_ARG, "device %d out of range (0..%d)", device, n - 1); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "hipSetDevice(%d) failed", device); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "hipGetDeviceProperties failed"); return nullptr; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fail(nullptr, PF_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName); return nullptr; }
  pf_ctx* c = new pf_ctx();
  c->device = device;
  bool ok = hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking) == hipSuccess;
  for (int d = 0; d < 2 && ok; ++d) ok = hipStreamCreateWithFlags(&c->s_dir[d], hipStreamNonBlocking) == hipSuccess;
  // The runtime hands hardware queues to streams round-robin in creation order.  A context's five streams are created together so
  // that they land on five DIFFERENT queues: created on first use (after other contexts' streams), the blend-ramp stream ended
  // up sharing a queue with one of the flow directions and a 9000x4000 stitch step took 7 ms longer.  Lanes of the throughput
  // mode never stitch: three streams each, so that six lanes fit GPU_MAX_HW_QUEUES = 24.
  if (!lane) ok = ok && hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_alpha, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_aux_go, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_fine, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&c->ev_fine2, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming) == hipSuccess;
  for (int d = 0; d < 2 && ok; ++d) ok = hipEventCreateWithFlags(&c->ev_dir[d], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_status, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer((void**)&c->d_status, c->h_status, 0) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_gate, kMaxBatch * kGateWords * sizeof(int), hipHostMallocMapped) == hipSuccess &&   // one area per pair of a batch
       hipHostGetDevicePointer((void**)&c->d_gate, c->h_gate, 0) == hipSuccess;
  if (ok) { *c->h_status = 0; memset(c->h_gate, 0, kMaxBatch * kGateWords * sizeof(int)); }
  if (!ok) { fail(nullptr, PF_ERR_DEVICE, "stream/event creation failed"); delete c; return nullptr; }
  c->cfg = cfg;
  {
    // Once per device and process, under one mutex: (i) the sweep's asm-block packed chains (csrc/exact_forms.hpp) against the
    // compiler-scheduled forms of the same arithmetic, on THIS device -- a mismatch means the hardware assumption behind them does not
    // hold here: refuse, rather than compute wrong flows (a -DPF_SAFE_PK build has no such blocks and passes trivially); (ii) the
    // blend's two small tables (kernels_misc.hip: device globals, the same values for every context -- written once, so that no later
    // context rewrites them under a blend another context has in flight).  Only SUCCESS is cached: a probe that could not run
    // (a transient allocation / launch failure) is tried again by the next pf_create.
    struct DeviceInit { bool probed = false; bool tables = false; };
    static std::mutex init_mu;
    static std::map<int, DeviceInit> init_done;
    std::lock_guard<std::mutex> lk(init_mu);
    DeviceInit& di = init_done[device];
    if (!di.probed) {
      unsigned* scratch = nullptr;
      int r = -1;
      if (hipMalloc((void**)&scratch, 256) == hipSuccess) { r = sweep_pk_probe(c->s_main, scratch); hipFree(scratch); }
      if (r != 0) {
        fail(nullptr, PF_ERR_DEVICE, r < 0 ? "the packed-fp32 probe could not run on device %d"
                                             : "device %d: the sweep's asm-block packed-fp32 chains do not reproduce the compiler-scheduled forms (%d threads differ); rebuild with -DPF_SAFE_PK",
             device, r);
        pf_destroy(c);
        return nullptr;
      }
      di.probed = true;
    }
    if (!di.tables) {
      launch_blend_tables(c->s_main);
      if (hipStreamSynchronize(c->s_main) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "device %d: the blend tables could not be initialised", device); pf_destroy(c); return nullptr; }
      di.tables = true;
    }
  }
  c->g5 = make_gauss(5, 0.25); c->g3_05 = make_gauss(3, 0.5); c->g3_1 = make_gauss(3, 1.0); c->g15 = make_gauss(15, 8.0);
  // Pre-sizing: every buffer a bidirectional solve / a stitch step on max_cols x max_rows needs is
  // allocated now, so that the first call does not pay ~40 hipMallocs.  0 x 0 = allocate lazily (the arena only grows).
  if (max_cols > 0 && max_rows > 0) {
    const int pad = max_cols / 20;
    bool ok2 = check_dims(c, max_cols, max_rows, pad) == 0;
    if (ok2) { SolveBufs sb; ok2 = alloc_solve(c, make_geometry(max_cols, max_rows, pad), 2, sb) == 0; }
    const size_t n = size_t(max_cols) * max_rows;
    const struct { const char* name; size_t bytes; } io[] = {
        {"nv_flow_l2r", n * 8}, {"nv_flow_r2l", n * 8}, {"h_img0", n * 4}, {"h_img1", n * 4}, {"h_flow0", n * 8}, {"h_flow1", n * 8}, {"h_blend", n * 4}, {"h_out", n * 4},
        {"ch_l", n * 4}, {"ch_r", n * 4}, {"ch_final", n * 4}, {"st_map", n}, {"st_ovl", n * 4}, {"st_ovr", n * 4}, {"st_blend", n * 4}, {"st_md", n * 4},
        {"st_merged", n * 4}, {"st_rowsum", n * 8}, {"st_blur_tmp", n * 4}};
    for (const auto& e : io) if (ok2 && (!lane || strncmp(e.name, "nv_", 3) == 0)) ok2 = ensure(c, e.name, e.bytes) != nullptr;
    if (!ok2) { g_err = c->err; pf_destroy(c); return nullptr; }
  }
  return c;
}
}  // namespace

extern "C" {

pf_ctx* pf_create_cfg(const pf_config* user) {
  if (!user || user->struct_size != (int)sizeof(pf_config)) { fail(nullptr, PF_ERR_ARG, "pf_create_cfg: struct_size does not match this library's pf_config"); return nullptr; }
  pf_config cfg = *user;
#ifdef PF_EXPERIMENTS
  // lab build only: the diagnostics under tests/micro select variants per process through the environment
  auto env_int = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
  env_int("PANOFLOW_SWEEP", cfg.sweep_impl); env_int("PANOFLOW_PREP", cfg.record_path); env_int("PANOFLOW_STAGGER", cfg.stagger_levels);
  env_int("PANOFLOW_PYR_CHAIN", cfg.pyramid_chaining); env_int("PANOFLOW_FINE_GRAD_BLOCKS", cfg.fine_gradient_blocks);
  env_int("PANOFLOW_SPARSE", cfg.sparse_sweep); env_int("PANOFLOW_WIDE", cfg.sweep_wide); env_int("PANOFLOW_WIDE_THRESHOLD", cfg.sweep_wide_threshold);
  env_int("PANOFLOW_BATCH_GRAD_FULL", cfg.full_width_batch_gradients);
  if (getenv("PANOFLOW_NO_WINDOW")) cfg.sweep_window = 0;
  if (const char* e = getenv("PANOFLOW_FUSE_UPS_PX")) cfg.fuse_small_level_px = atol(e);
  if (cfg.sweep_impl != 1 && cfg.sweep_impl != 3) cfg.sweep_impl = 2;
  if (cfg.record_path < 0 || cfg.record_path > 2) cfg.record_path = 0;
#else
  if (cfg.sweep_impl != 2 || cfg.record_path != 0 || cfg.sweep_wide == 1) {
    fail(nullptr, PF_ERR_ARG, "sweep_impl / record_path / sweep_wide 1 select cross-check implementations that only the -DPF_EXPERIMENTS build (libpanoflow_exp.so) contains");
    return nullptr;
  }
#endif
  if (cfg.batch_pairs == 0 || cfg.batch_pairs < -1 || cfg.batch_pairs > kMaxBatch) { fail(nullptr, PF_ERR_ARG, "pf_create_cfg: batch_pairs must be -1 or 1..%d", kMaxBatch); return nullptr; }
  if (cfg.fine_gradient_blocks < 1 || cfg.stagger_levels < -1 || cfg.fuse_small_level_px < -1 || cfg.sparse_sweep < -1 || cfg.sparse_sweep > 1 ||
      cfg.sweep_wide < -1 || cfg.sweep_wide > 2 || cfg.sweep_wide_threshold < 0) {
    fail(nullptr, PF_ERR_ARG, "pf_create_cfg: knob out of range");
    return nullptr;
  }
  return create_ctx(cfg, false);
}

void pf_destroy(pf_ctx* c) {
  if (!c) return;
  for (pf_ctx* l : c->lanes) pf_destroy(l);
  c->lanes.clear();
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (auto& kv : c->bufs) if (kv.second.p) hipFree(kv.second.p);
  for (auto e : c->ev_pool) hipEventDestroy(e);
  for (auto& p : c->prof_pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  if (c->ev_pre) hipEventDestroy(c->ev_pre);
  if (c->ev_fine) hipEventDestroy(c->ev_fine);
  if (c->ev_fine2) hipEventDestroy(c->ev_fine2);
  if (c->ev_stagger) hipEventDestroy(c->ev_stagger);
  for (int d = 0; d < 2; ++d) { if (c->ev_dir[d]) hipEventDestroy(c->ev_dir[d]); if (c->s_dir[d]) hipStreamDestroy(c->s_dir[d]); }
  if (c->ev_aux_go) hipEventDestroy(c->ev_aux_go);
  if (c->ev_aux_done) hipEventDestroy(c->ev_aux_done);
  if (c->s_aux) hipStreamDestroy(c->s_aux);
  if (c->ev_alpha) hipEventDestroy(c->ev_alpha);
  if (c->ev_gate) hipEventDestroy(c->ev_gate);
  if (c->s_copy) hipStreamDestroy(c->s_copy);
  if (c->s_main) hipStreamDestroy(c->s_main);
  if (c->h_status) hipHostFree(c->h_status);
  if (c->h_gate) hipHostFree(c->h_gate);
  delete c;
}

const char* pf_last_error(const pf_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
const char* pf_last_warning(const pf_ctx* c) { return c ? c->warn.c_str() : ""; }
int pf_warning_count(const pf_ctx* c) { return c ? c->warn_count : 0; }

int pf_max_percentage_by_name(const char* name) {
  if (name && strcmp(name, "pixflow_low") == 0) return 0;
  if (name && strcmp(name, "pixflow_search_20") == 0) return 20;
  return fail(nullptr, PF_ERR_ARG, "unrecognized flow algorithm name: %s", name ? name : "(null)");
}

// ---- device memory helpers ----
void* pf_dev_alloc(pf_ctx* c, size_t bytes) {
  if (use(c)) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(c, PF_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void pf_dev_free(pf_ctx* c, void* p) { if (!use(c) && p) hipFree(p); }
// page-locked host memory for the caller's images: copies to and from it run at the link's DMA rate (a pageable destination is
// staged through the runtime's bounce buffers: 17.5 GB/s instead of ~55 GB/s for the 144 MB composite of a 9000x4000 step)
void* pf_host_alloc(pf_ctx* c, size_t bytes) {
  if (use(c)) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(c, PF_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void pf_host_free(pf_ctx* c, void* p) { if (!use(c) && p) hipHostFree(p); }
int pf_upload(pf_ctx* c, void* dst, const void* src, size_t bytes) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int pf_download(pf_ctx* c, void* dst, const void* src, size_t bytes) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int pf_sync(pf_ctx* c) { if (int e = use(c)) return e; return finish(c); }
int pf_selftest_packed_chains(pf_ctx* c) {
  if (int e = use(c)) return e;
  unsigned* scratch = (unsigned*)ensure(c, "pk_probe", 256);
  if (!scratch) return PF_ERR_NOMEM;
  const int r = sweep_pk_probe(c->s_main, scratch);
  return r < 0 ? fail(c, PF_ERR_DEVICE, "the packed-fp32 probe could not run") : r;
}
// 64-bit content checksum of `bytes` bytes at d_ptr (8-byte aligned), computed on the device: results that live in HBM -- on this
// GPU or gathered from others -- are compared without a trip through the host.  ~25 us per 144 MB strip.
int pf_checksum_dev(pf_ctx* c, const void* d_ptr, size_t bytes, uint64_t* out) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_ptr || !out || (reinterpret_cast<uintptr_t>(d_ptr) & 7)) return fail(c, PF_ERR_ARG, "pf_checksum_dev: null or misaligned pointer");
  unsigned long long* acc = (unsigned long long*)ensure(c, "checksum_acc", 256);
  if (!acc) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(acc, 0, 8, c->s_main));
  launch_checksum64(c->s_main, d_ptr, bytes, acc);
  unsigned long long h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, acc, 8, hipMemcpyDeviceToHost, c->s_main));
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  *out = h;
  return 0;
}

// ---- device-resident entry points ----
int pf_flow_bidir_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_pct, float* d_l2r, float* d_r2l) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_l2r || !d_r2l) return fail(c, PF_ERR_ARG, "null device pointer");
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};  // OpticalFlow.cpp:134,139
  float* outs[2] = {d_l2r, d_r2l};
  const int pad = cols / 20;                            // OpticalFlow.cpp:113
  if (int e = solve(c, d_l, d_r, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_blend_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, const float* d_l2r, const float* d_r2l, const float* d_blend, int cols,
                 int rows, uint8_t* d_out) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_l2r || !d_r2l || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, d_l, d_r, d_l2r, d_r2l, d_blend, cols, rows, d_out); }
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_novel_view_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_pct, const float* d_blend, uint8_t* d_out,
                      float* d_l2r, float* d_r2l) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "null device pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  float* f0 = d_l2r ? d_l2r : (float*)ensure(c, "nv_flow_l2r", size_t(cols) * rows * 8);
  float* f1 = d_r2l ? d_r2l : (float*)ensure(c, "nv_flow_r2l", size_t(cols) * rows * 8);
  if (!f0 || !f1) return PF_ERR_NOMEM;
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};
  float* outs[2] = {f0, f1};
  const int pad = cols / 20;
  if (int e = solve(c, d_l, d_r, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, d_l, d_r, f0, f1, d_blend, cols, rows, d_out); }
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// ---- throughput mode ----
// One pair keeps ~70 workgroups of a sweep busy (two directions x ~35 bands-of-4): the exact sweeps are a dependency chain, so
// most of the 256 CUs idle.  When pairs are plentiful, `in_flight` of them are on the GPU at the same time, in two ways that combine:
//   * BATCHES (round 3): B pairs go through the SAME launches (blockIdx.z = pair, solve_n): one kernel boundary per step of the
//     algorithm for B pairs.  With several independent streams the kernels themselves barely slow down, but the gap between a
//     stream's dependent kernels grows with the number of busy hardware queues (2.7 -> 23 us per launch from 1 to 2 pairs in
//     flight, profiles/r02_throughput_mode.txt); a batch pays each gap once for B pairs;
//   * LANES: further stream / buffer sets on the same device ("lane", created on first use and kept), each driven by its own host
//     thread and each working through its own batches, out of phase with the others.
// in_flight = lanes x pairs per batch; pf_config::batch_pairs picks the split (-1: see batch_split()).  Results are identical to
// n_pairs calls of pf_novel_view_dev.  Set GPU_MAX_HW_QUEUES >= 3 * lanes + 2 before the first HIP call.
namespace {
void batch_split(const pf_ctx* c, int in_flight, int& lanes, int& per_batch) {
  // measured (24 strips of 2000x4000, Mpix/s, tests/micro/tp_batch_sweep.sh): 6 in flight as 6 lanes 838, 3 x 2 911, 2 x 3 974, one batch of 6 1071;
  // 8 in flight as 4 x 2 1032, 2 x 4 1131, one batch of 8 1237; 12 = 2 lanes x 6 1326; 16 = 2 x 8 1368: the fewest lanes win
  per_batch = c->cfg.batch_pairs > 0 ? c->cfg.batch_pairs : (in_flight <= kMaxBatch ? in_flight : (in_flight + 1) / 2);
  if (per_batch > kMaxBatch) per_batch = kMaxBatch;
  if (per_batch > in_flight) per_batch = in_flight;
  lanes = (in_flight + per_batch - 1) / per_batch;
}
// one batch: pairs [first, first + count) of the arrays through one set of launches on `lane`
int novel_view_group(pf_ctx* lane, int first, int count, const uint8_t* const* d_l, const uint8_t* const* d_r, int cols, int rows, int max_pct,
                     const float* const* d_blend, uint8_t* const* d_out, float* const* d_l2r, float* const* d_r2l) {
  if (count == 1)
    return pf_novel_view_dev(lane, d_l[first], d_r[first], cols, rows, max_pct, d_blend[first], d_out[first], d_l2r ? d_l2r[first] : nullptr, d_r2l ? d_r2l[first] : nullptr);
  if (int e = use(lane)) return e;
  CallGuard guard_(lane);
  if (int e = check_dims(lane, cols, rows, cols / 20)) return e;
  float* outs[2 * kMaxBatch]; float* used[2 * kMaxBatch];
  for (int p = 0; p < count; ++p) {
    if (!d_l[first + p] || !d_r[first + p] || !d_blend[first + p] || !d_out[first + p]) return fail(lane, PF_ERR_ARG, "null device pointer (pair %d)", first + p);
    outs[2 * p] = d_l2r ? d_l2r[first + p] : nullptr; outs[2 * p + 1] = d_r2l ? d_r2l[first + p] : nullptr;
  }
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};
  if (int e = solve_n(lane, count, d_l + first, d_r + first, cols, rows, cols / 20, max_pct, 2, hints, outs, used)) return e;
  BlendPtrs bp{};
  for (int p = 0; p < count; ++p) { bp.L[p] = d_l[first + p]; bp.R[p] = d_r[first + p]; bp.fLR[p] = used[2 * p]; bp.fRL[p] = used[2 * p + 1]; bp.blend[p] = d_blend[first + p]; bp.out[p] = d_out[first + p]; }
  { PROF(lane, lane->s_main, "blend"); launch_blend_batch(lane->s_main, bp, count, cols, rows); }
  HIPCHK(lane, hipGetLastError());
  if (int e = finish(lane)) return e;
  return check_sweeps(lane);
}
}  // namespace
int pf_novel_view_batch_dev(pf_ctx* c, int n_pairs, const uint8_t* const* d_l, const uint8_t* const* d_r, int cols, int rows, int max_pct,
                            const float* const* d_blend, uint8_t* const* d_out, float* const* d_l2r, float* const* d_r2l, int in_flight) {
  if (int e = use(c)) return e;
  if (n_pairs < 0 || !d_l || !d_r || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "bad argument");
  if (in_flight < 1) in_flight = 1;
  if (in_flight > 2 * kMaxBatch) in_flight = 2 * kMaxBatch;
  if (in_flight > n_pairs) in_flight = n_pairs > 0 ? n_pairs : 1;
  int nlanes = 1, per_batch = 1;
  batch_split(c, in_flight, nlanes, per_batch);
  check_hw_queues(c, 3 * nlanes + 2, "pf_novel_view_batch_dev");   // three streams per lane + this context's blend-ramp and copy streams
  while ((int)c->lanes.size() < nlanes - 1) {
    pf_config lc = c->cfg; lc.max_cols = per_batch > 1 ? 0 : cols; lc.max_rows = per_batch > 1 ? 0 : rows;   // a batching lane lives in its slabs: nothing to pre-size
    pf_ctx* l = create_ctx(lc, true);
    if (!l) return fail(c, PF_ERR_NOMEM, "cannot create lane %d: %s", (int)c->lanes.size() + 1, g_err.c_str());
    c->lanes.push_back(l);
  }
  for (pf_ctx* l : c->lanes) l->prof = c->prof;   // profiling covers every lane (collected into the lane's own totals)
  const int ngroups = (n_pairs + per_batch - 1) / per_batch;
  std::vector<int> rc(nlanes, 0);
  std::vector<std::string> msg(nlanes);
  auto run = [&](int k) {
    pf_ctx* lane = k == 0 ? c : c->lanes[k - 1];
    struct Restore { pf_ctx* l; long v; bool b; ~Restore() { l->fuse_ups_px = v; l->is_lane = b; l->lanes_running = 1; } } restore{lane, lane->fuse_ups_px, lane->is_lane};
    lane->lanes_running = nlanes;
    // lanes side by side: launches count more than their length, the small levels fold two kernels into their neighbours (see solve_n()).
    // A batch pays every launch once for all its pairs, and there the separate (shorter) kernels win again: 8 in one batch 1374 vs 1347 Mpix/s.
    if (in_flight > 1) { lane->fuse_ups_px = per_batch > 1 ? 0 : 262144; lane->is_lane = true; }
    for (int gidx = k; gidx < ngroups; gidx += nlanes) {
      const int first = gidx * per_batch, count = std::min(per_batch, n_pairs - first);
      const int e = novel_view_group(lane, first, count, d_l, d_r, cols, rows, max_pct, d_blend, d_out, d_l2r, d_r2l);
      if (e) { rc[k] = e; msg[k] = lane->err; return; }
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < nlanes; ++k) th.emplace_back(run, k);
  run(0);
  for (auto& t : th) t.join();
  for (int k = 0; k < nlanes; ++k) if (rc[k]) return fail(c, rc[k], "lane %d: %s", k, msg[k].c_str());
  if (c->prof)   // per-kernel-family times of the lanes are reported with the owning context's
    for (pf_ctx* l : c->lanes)
      for (size_t i = 0; i < l->prof_names.size(); ++i) {
        const int id = prof_id(c, l->prof_names[i].c_str());
        c->prof_tot[id].ms += l->prof_tot[i].ms; c->prof_tot[id].n += l->prof_tot[i].n;
        l->prof_tot[i] = ProfEntry();
      }
  return 0;
}

// ---- host-buffer entry points ----
int pf_flow(pf_ctx* c, const uint8_t* i0, const uint8_t* i1, int cols, int rows, size_t step, int max_pct, int hint, float* flow, size_t fstep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!i0 || !i1 || !flow) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, 0)) return e;
  if (step < size_t(cols) * 4 || fstep < size_t(cols) * 8) return fail(c, PF_ERR_ARG, "row step too small");
  if (hint < 0 || hint > 4) return fail(c, PF_ERR_ARG, "unexpected direction %d", hint);
  const size_t ib = size_t(cols) * rows * 4;
  uint8_t* d0 = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* d1 = (uint8_t*)ensure(c, "h_img1", ib);
  float* df = (float*)ensure(c, "h_flow0", size_t(cols) * rows * 8);
  if (!d0 || !d1 || !df) return PF_ERR_NOMEM;
  if (int e = up2d(c, d0, size_t(cols) * 4, i0, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, d1, size_t(cols) * 4, i1, step, size_t(cols) * 4, rows)) return e;
  const int hints[2] = {hint, hint}; float* outs[2] = {df, nullptr};
  if (int e = solve(c, d0, d1, cols, rows, 0, max_pct, 1, hints, outs)) return e;
  if (int e = down2d(c, flow, fstep, df, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_novel_view(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, const float* blend, size_t bstep,
                  uint8_t* out, size_t ostep, float* f_l2r, float* f_r2l, size_t fstep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  if (step < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  if (out && !blend) return fail(c, PF_ERR_ARG, "blend is required when out_bgra is given");
  const size_t ib = size_t(cols) * rows * 4, fb = size_t(cols) * rows * 8;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", ib);
  float* d0 = (float*)ensure(c, "h_flow0", fb); float* d1 = (float*)ensure(c, "h_flow1", fb);
  if (!dl || !dr || !d0 || !d1) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT}; float* outs[2] = {d0, d1};
  const int pad = cols / 20;
  if (int e = solve(c, dl, dr, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  if (out) {
    float* db = (float*)ensure(c, "h_blend", size_t(cols) * rows * 4); uint8_t* dout = (uint8_t*)ensure(c, "h_out", ib);
    if (!db || !dout) return PF_ERR_NOMEM;
    if (int e = up2d(c, db, size_t(cols) * 4, blend, bstep, size_t(cols) * 4, rows)) return e;
    { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, dl, dr, d0, d1, db, cols, rows, dout); }
    if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  }
  if (f_l2r) if (int e = down2d(c, f_l2r, fstep, d0, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (f_r2l) if (int e = down2d(c, f_r2l, fstep, d1, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_flow_bidir(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, float* f_l2r, float* f_r2l,
                  size_t fstep) {
  return pf_novel_view(c, l, r, cols, rows, step, max_pct, nullptr, 0, nullptr, 0, f_l2r, f_r2l, fstep);
}

int pf_blend(pf_ctx* c, const uint8_t* l, const uint8_t* r, size_t step, const float* f_l2r, const float* f_r2l, size_t fstep, const float* blend,
             size_t bstep, int cols, int rows, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !f_l2r || !f_r2l || !blend || !out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || ostep < size_t(cols) * 4 || bstep < size_t(cols) * 4 || fstep < size_t(cols) * 8) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t ib = size_t(cols) * rows * 4, fb = size_t(cols) * rows * 8;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", ib); uint8_t* dout = (uint8_t*)ensure(c, "h_out", ib);
  float* d0 = (float*)ensure(c, "h_flow0", fb); float* d1 = (float*)ensure(c, "h_flow1", fb); float* db = (float*)ensure(c, "h_blend", ib);
  if (!dl || !dr || !dout || !d0 || !d1 || !db) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, d0, size_t(cols) * 8, f_l2r, fstep, size_t(cols) * 8, rows)) return e;
  if (int e = up2d(c, d1, size_t(cols) * 8, f_r2l, fstep, size_t(cols) * 8, rows)) return e;
  if (int e = up2d(c, db, size_t(cols) * 4, blend, bstep, size_t(cols) * 4, rows)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, dl, dr, d0, d1, db, cols, rows, dout); }
  if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

static int blend_smooth_dev(pf_ctx* c, float* d_blend, const float* d_md, int cols, int rows, hipStream_t sm = nullptr) {
  const int step = cols <= rows ? cols / 200 : rows / 200, k1 = rows / 130, k2 = rows / 400;
  if (!sm) sm = c->s_main;
  if (step > 0 && k1 > 0) {
    // the tile kernel keeps a (step+k1-1)^2 window and (step+k1-1) x step row sums in LDS: 160 KB per CU bound the canvas at ~15000 rows
    if (tile_blur_lds_bytes(step, k1) > 160 * 1024) return fail(c, PF_ERR_ARG, "canvas %dx%d too large for the blend-ramp tile smoothing (LDS)", cols, rows);
    void* work = ensure(c, "st_tile_work", tile_blur_work_bytes(cols, rows, step, k1) + 256);
    if (!work) return PF_ERR_NOMEM;
    { PROF(c, sm, "tile_blur"); launch_tile_blur(sm, d_blend, d_md, cols, rows, step, k1, work); }
    launch_collect_status(sm, static_cast<const int*>(work), 2, c->d_status, 4);   // word 1 = a grid barrier of the tile smoothing gave up
  }
  if (k2 > 0) {
    double* rs = (double*)ensure(c, "st_rowsum", size_t(cols) * rows * 8);
    float* tmp = (float*)ensure(c, "st_blur_tmp", size_t(cols) * rows * 4);
    if (!rs || !tmp) return PF_ERR_NOMEM;
    PROF(c, sm, "box_blur");
    launch_box_blur(sm, d_blend, tmp, rs, cols, rows, k2);
    HIPCHK(c, hipMemcpyAsync(d_blend, tmp, size_t(cols) * rows * 4, hipMemcpyDeviceToDevice, sm));
  }
  return 0;
}

int pf_stitch_prepare(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, uint8_t* map_out, size_t mstep, uint8_t* ovl,
                      uint8_t* ovr, float* blend_out, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || (map_out && mstep < size_t(cols)) || (blend_out && bstep < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dl || !dr || !dm || !dol || !dor || !db || !dmd) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  { PROF(c, sm, "countblend"); launch_countblend(sm, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (map_out) if (int e = down2d(c, map_out, mstep, dm, cols, cols, rows)) return e;
  if (ovl) if (int e = down2d(c, ovl, step, dol, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (ovr) if (int e = down2d(c, ovr, step, dor, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (blend_out) if (int e = down2d(c, blend_out, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// Stitchtools::MatchImages (StitchTool.cpp:38-50) + the overlap masking of prepare() (:17-33) alone: map and the two masked images.
int pf_stitch_match(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, uint8_t* map_out, size_t mstep, uint8_t* ovl, uint8_t* ovr) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || (map_out && mstep < size_t(cols))) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  if (!dl || !dr || !dm || !dol || !dor) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, c->s_main, "match_images"); launch_match_images(c->s_main, dl, dr, cols, rows, dm, dol, dor); }
  if (map_out) if (int e = down2d(c, map_out, mstep, dm, cols, cols, rows)) return e;
  if (ovl) if (int e = down2d(c, ovl, step, dol, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (ovr) if (int e = down2d(c, ovr, step, dor, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

// Stitchtools::GenerateBlend (StitchTool.cpp:98-146) from a GIVEN map -- the reference reads its public `Map` member there, so a
// caller that edits the map between MatchImages() and GenerateBlend() gets the ramp of the edited map.
int pf_stitch_generate_blend(pf_ctx* c, const uint8_t* map, size_t mstep, int cols, int rows, float* blend_out, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!map || !blend_out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (mstep < size_t(cols) || bstep < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dm || !db || !dmd) return PF_ERR_NOMEM;
  if (int e = up2d(c, dm, cols, map, mstep, cols, rows)) return e;
  { PROF(c, c->s_main, "countblend"); launch_countblend(c->s_main, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (int e = down2d(c, blend_out, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// GenerateBlend's per-pixel part alone (StitchTool.cpp:113-125 with countblend :148-191): the ramp BEFORE the tile / global
// box smoothing, i.e. what Stitchtools::countblend(x, y) returns for overlap pixels, and MergedDis.
int pf_stitch_raw_blend(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, float* raw_blend, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !raw_blend) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || bstep < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dl || !dr || !dm || !dol || !dor || !db || !dmd) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  { PROF(c, sm, "countblend"); launch_countblend(sm, dm, cols, rows, db, dmd); }
  if (int e = down2d(c, raw_blend, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_stitch_gather(pf_ctx* c, const uint8_t* l, const uint8_t* r, const uint8_t* merged, size_t step, const uint8_t* map, size_t mstep, int cols,
                     int rows, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !merged || !map || !out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || ostep < size_t(cols) * 4 || mstep < size_t(cols)) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4); uint8_t* dg = (uint8_t*)ensure(c, "st_merged", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dout = (uint8_t*)ensure(c, "h_out", n * 4);
  if (!dl || !dr || !dg || !dm || !dout) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dg, size_t(cols) * 4, merged, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dm, cols, map, mstep, cols, rows)) return e;
  { PROF(c, c->s_main, "gather"); launch_gather(c->s_main, dl, dr, dg, dm, cols, rows, dout); }
  if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}


// One whole iteration of the reference's stitch loop (CPU/main.cpp:70-95) without leaving the device:
// Stitchtools::prepare -> NovelViewGeneratorAsymmetricFlow::prepare/generateNovelView -> Gather.
// r_bgra == NULL chains on the previous call's result, which stays resident in HBM (main.cpp:64-65).
// Content signature of a host image: 16 evenly spaced rows, 8 bytes at a time (~0.1 ms at 9000x4000).  The prefetched device copy of
// an image is only used if the caller's buffer still carries the signature it had when it was uploaded: pointer, size and step alone
// cannot tell a buffer from another image that an allocator later placed at the same address (the intended use is one cv::Mat freed and
// re-read per image).
static uint64_t host_image_sig(const uint8_t* p, int cols, int rows, size_t step) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  const size_t rb = size_t(cols) * 4;
  for (int i = 0; i < 16; ++i) {
    const uint8_t* row = p + size_t((long long)(rows - 1) * i / 15) * step;
    for (size_t o = 0; o + 8 <= rb; o += 8) { uint64_t v; memcpy(&v, row + o, 8); h = (h ^ v) * 0xBF58476D1CE4E5B9ull; h ^= h >> 29; }
  }
  return h;
}

int pf_stitch_step(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  if (step < size_t(cols) * 4 || (out && ostep < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "row step too small");
  check_hw_queues(c, 5, "pf_stitch_step");   // front end, two flow directions, blend ramp, prefetch copy
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "ch_l", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "ch_r", n * 4); uint8_t* dfin = (uint8_t*)ensure(c, "ch_final", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4); uint8_t* dmerged = (uint8_t*)ensure(c, "st_merged", n * 4);
  float* f0 = (float*)ensure(c, "nv_flow_l2r", n * 8); float* f1 = (float*)ensure(c, "nv_flow_r2l", n * 8);
  if (!dl || !dr || !dfin || !dm || !dol || !dor || !db || !dmd || !dmerged || !f0 || !f1) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  uint8_t* dnext = (uint8_t*)ensure(c, "ch_l_next", n * 4);
  if (!dnext) return PF_ERR_NOMEM;
  // both prefetch records are one-shot: latched and cleared here, whatever this step does with them
  const pf_ctx::HostImage ready = c->ready, hint = c->hint;
  c->ready = pf_ctx::HostImage(); c->hint = pf_ctx::HostImage();
  if (ready.src == l && ready.cols == cols && ready.rows == rows && ready.step == step && ready.sig == host_image_sig(l, cols, rows, step)) {
    // this step's left image was uploaded while the previous step computed: the two buffers trade places (no copy; the old
    // "ch_l" is free -- the previous call drained every stream -- and receives the next prefetch)
    std::swap(c->bufs["ch_l"], c->bufs["ch_l_next"]);
    std::swap(dl, dnext);
  } else {
    if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  }
  if (r) { if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e; }
  else {
    if (c->chain_cols != cols || c->chain_rows != rows) return fail(c, PF_ERR_ARG, "pf_stitch_step: no previous result of this size to chain on");
    HIPCHK(c, hipMemcpyAsync(dr, dfin, n * 4, hipMemcpyDeviceToDevice, sm));
  }
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  // The blend ramp (GenerateBlend + countblend + smoothing, StitchTool.cpp:98-191) only depends on the map and is only
  // needed by the final blend: it runs on its own stream beside the two flow solves.  Its launches (a dozen since the tile smoothing
  // became ONE persistent launch in round 3; ~850 before) are enqueued AFTER the solver's, so that the solver's first kernel is not
  // kept waiting by them.
  if (!c->s_aux) HIPCHK(c, hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking));
  if (!c->s_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking));
  hipStream_t sa = c->s_aux;
  HIPCHK(c, hipEventRecord(c->ev_aux_go, sm));
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT}; float* outs[2] = {f0, f1};
  const int pad = cols / 20;
  if (int e = solve(c, dol, dor, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  HIPCHK(c, hipStreamWaitEvent(sa, c->ev_aux_go, 0));
  { PROF(c, sa, "countblend"); launch_countblend(sa, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows, sa)) return e;
  HIPCHK(c, hipEventRecord(c->ev_aux_done, sa));
  HIPCHK(c, hipStreamWaitEvent(sm, c->ev_aux_done, 0));
  { PROF(c, sm, "blend"); launch_blend(sm, dol, dor, f0, f1, db, cols, rows, dmerged); }
  { PROF(c, sm, "gather"); launch_gather(sm, dl, dr, dmerged, dm, cols, rows, dfin); }
  if (out) if (int e = down2d(c, out, ostep, dfin, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  // everything of this step is enqueued: upload the NEXT step's left image now (announced with pf_stitch_prefetch); the
  // host-side staging of a pageable source runs while the GPU computes
  if (hint.src && hint.src != l && hint.cols == cols && hint.rows == rows) {
    if (hint.step == size_t(cols) * 4) HIPCHK(c, hipMemcpyAsync(dnext, hint.src, n * 4, hipMemcpyHostToDevice, c->s_copy));
    else HIPCHK(c, hipMemcpy2DAsync(dnext, size_t(cols) * 4, hint.src, hint.step, size_t(cols) * 4, rows, hipMemcpyHostToDevice, c->s_copy));
    HIPCHK(c, hipStreamSynchronize(c->s_copy));
    c->ready = hint;
    c->ready.sig = host_image_sig(hint.src, cols, rows, hint.step);
  }
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  c->chain_cols = cols; c->chain_rows = rows;
  return check_sweeps(c);
}

// Announce the left image of the pf_stitch_step call AFTER the coming one: the coming step uploads it while its own kernels run
// (the copy is issued after they are enqueued).  One-shot: the hint is consumed by the coming step; the buffer must stay valid
// and unchanged until the step after it has returned, and that step must pass the same pointer / size / step -- anything else
// simply uploads as usual and the prefetched copy is dropped.  NULL cancels.
int pf_stitch_prefetch(pf_ctx* c, const uint8_t* next_l, int cols, int rows, size_t step) {
  if (!c) return fail(nullptr, PF_ERR_ARG, "null context");
  if (next_l && (cols <= 0 || rows <= 0 || step < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "bad argument");
  c->hint.src = next_l; c->hint.cols = cols; c->hint.rows = rows; c->hint.step = step;
  return 0;
}

// ---- stage-level entry points (tests) ----
#define STAGE_BEGIN(c) if (int e_ = use(c)) return e_; CallGuard guard_(c); hipStream_t sm = c->s_main; (void)sm
static void* stage_up(pf_ctx* c, const char* name, const void* host, size_t bytes) {
  void* d = ensure(c, name, bytes);
  if (d && host) hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->s_main);
  return d;
}
static int stage_down(pf_ctx* c, void* host, const void* dev, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->s_main));
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_stage_preprocess(pf_ctx* c, const uint8_t* bgra, int cols, int rows, int pad, float* gray_half, float* alpha_half) {
  STAGE_BEGIN(c);
  if (int e = check_dims(c, cols, rows, pad)) return e;
  const int dw = int((cols + 2 * pad) * kDownscaleFactor), dh = int(rows * kDownscaleFactor);
  uint8_t* d = (uint8_t*)stage_up(c, "sg_a", bgra, size_t(cols) * rows * 4);
  float* t = (float*)ensure(c, "sg_b", size_t(dw) * dh * 4); float* g = (float*)ensure(c, "sg_c", size_t(dw) * dh * 4); float* a = (float*)ensure(c, "sg_d", size_t(dw) * dh * 4);
  if (!d || !t || !g || !a) return PF_ERR_NOMEM;
  launch_downscale_gray(sm, d, cols, rows, pad, t, a, dw, dh);
  launch_gauss_small(sm, t, g, dw, dh, 1, c->g5);
  HIPCHK(c, hipMemcpyAsync(gray_half, g, size_t(dw) * dh * 4, hipMemcpyDeviceToHost, sm));
  return stage_down(c, alpha_half, a, size_t(dw) * dh * 4);
}
int pf_stage_pyr_down(pf_ctx* c, const float* src, int sw, int sh, float* dst, int dw, int dh) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", src, size_t(sw) * sh * 4); float* d = (float*)ensure(c, "sg_b", size_t(dw) * dh * 4);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_resize_linear(sm, s, sw, sh, d, dw, dh, 1, 1.f, false);
  return stage_down(c, dst, d, size_t(dw) * dh * 4);
}
int pf_stage_gradients(pf_ctx* c, const float* img, int w, int h, float* gxy) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", img, size_t(w) * h * 4); float* d = (float*)ensure(c, "sg_b", size_t(w) * h * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_gradients(sm, s, w, h, d, c->g3_05);
  return stage_down(c, gxy, d, size_t(w) * h * 8);
}
int pf_stage_gauss(pf_ctx* c, const float* src, int w, int h, int cn, int ksize, double sigma, float* dst) {
  STAGE_BEGIN(c);
  if (!((ksize == 3 || ksize == 5) && (cn == 1 || cn == 2)) && !(ksize == 15 && cn == 2)) return fail(c, PF_ERR_ARG, "unsupported gaussian %d/%d", ksize, cn);
  const size_t nb = size_t(w) * h * cn * 4;
  float* s = (float*)stage_up(c, "sg_a", src, nb); float* d = (float*)ensure(c, "sg_b", nb); float* t = (float*)ensure(c, "sg_c", nb);
  if (!s || !d || !t) return PF_ERR_NOMEM;
  const Gauss g = make_gauss(ksize, sigma);
  if (ksize == 15) launch_gauss15(sm, s, t, d, w, h, g); else launch_gauss_small(sm, s, d, w, h, cn, g);
  return stage_down(c, dst, d, nb);
}
int pf_stage_median5(pf_ctx* c, const float* flow, int w, int h, float* out) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(w) * h * 8); float* d = (float*)ensure(c, "sg_b", size_t(w) * h * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  // the stage entry runs BOTH forms of the kernel (direct and LDS-tiled; the solver picks by level size) and requires identical bits
  float* d2 = (float*)ensure(c, "sg_c", size_t(w) * h * 8); int* neq = (int*)ensure(c, "sg_d", 256);
  if (!d2 || !neq) return PF_ERR_NOMEM;
  launch_median5_form(sm, s, d, w, h, false);
  launch_median5_form(sm, s, d2, w, h, true);
  HIPCHK(c, hipMemsetAsync(neq, 0, 4, sm));
  launch_count_diff_u32(sm, reinterpret_cast<const uint32_t*>(d), reinterpret_cast<const uint32_t*>(d2), size_t(w) * h * 2, neq);
  int hneq = 0;
  HIPCHK(c, hipMemcpyAsync(&hneq, neq, 4, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, out, d, size_t(w) * h * 8)) return e;
  if (hneq) return fail(c, PF_ERR_DEVICE, "median5: the direct and the LDS-tiled kernel disagree in %d words", hneq);
  return 0;
}
int pf_stage_sweep(pf_ctx* c, const float* g0, const float* g1, const float* blurred, const float* a0, const float* a1, float* flow, int w, int h, int forward) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* dg0 = (float*)stage_up(c, "sg_a", g0, n * 8); float* dg1 = (float*)stage_up(c, "sg_b", g1, n * 8); float* dbl = (float*)stage_up(c, "sg_c", blurred, n * 8);
  float* da0 = (float*)stage_up(c, "sg_d", a0, n * 4); float* da1 = (float*)stage_up(c, "sg_e", a1, n * 4); float* df = (float*)stage_up(c, "sg_f", flow, n * 8);
  uint8_t* gate = (uint8_t*)ensure(c, "sg_g", n);
  const size_t nb = sweep_boundary_elems(w, h);
  unsigned long long* bnd = (unsigned long long*)ensure(c, "sg_h", nb * 8); int* ctrl = (int*)ensure(c, "sg_i", 16);
  if (!dg0 || !dg1 || !dbl || !da0 || !da1 || !df || !gate || !bnd || !ctrl) return PF_ERR_NOMEM;
  launch_gate(sm, da0, da1, (int)n, gate);
  launch_fill_u64(sm, bnd, nb, kNotReady);
  HIPCHK(c, hipMemsetAsync(ctrl, 0, 16, sm));
  int* pcnt = (int*)ensure(c, "sg_pc", 2 * size_t(sweep2_num_wgs_max(w, h)) * sizeof(int));
  if (!pcnt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(pcnt, 0, 2 * size_t(sweep2_num_wgs_max(w, h)) * sizeof(int), sm));
  SweepArgs sa; sa.prepcnt = pcnt; sa.g0 = (const float2*)dg0; sa.g1 = (const float2*)dg1; sa.blurred = (const float2*)dbl; sa.gate = gate; sa.flow = (float2*)df;
  sa.boundary = bnd; sa.ctrl = ctrl; sa.W = w; sa.H = h; sa.forward = forward; sa.sparse = (w * h) % 2;   // stage test: exercise both variants
  sa.wide = c->cfg.sweep_wide > 0 ? c->cfg.sweep_wide : 0;   // the sweep form the context was created for (auto = latency form: one pair)
  if (sa.wide == 2) sa.sparse = 0;            // (the throughput form has no sparse variant)
  {
    std::vector<int> box; LevelTable t; t.n = 1; t.w[0] = w; t.h[0] = h; t.off[0] = 0;
    if (int e = gate_boxes_to_host(c, sm, gate, t, n, box)) return e;
    sa.ax0 = box[0]; sa.ay0 = box[1]; sa.ax1 = box[2] + 1; sa.ay1 = box[3] + 1;
  }
  float* rec = (float*)ensure(c, "sg_rec", sweep2_rec_bytes(w, h));
  if (!rec) return PF_ERR_NOMEM;
#ifdef PF_EXPERIMENTS
  sa.prep_mode = c->cfg.record_path;
  if (c->cfg.sweep_impl == 1) { PROF(c, sm, "sweep"); launch_sweep(sm, sa); } else
#endif
  { PROF(c, sm, "sweep"); (void)launch_sweep_any(sm, sa, rec, c->cfg.sweep_impl == 3); }
  int hc[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(hc, ctrl, 16, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, flow, df, n * 8)) return e;
  if (hc[1]) return fail(c, PF_ERR_TIMEOUT, "sweep band timed out");
#ifdef PF_SWEEP_STATS_PRINT
  fprintf(stderr, "[panoflow] sweep %dx%d: edge waits %d, spin iterations %d\n", w, h, hc[2], hc[3]);
#endif
  return 0;
}
int pf_stage_diffusion(pf_ctx* c, const float* a0, const float* a1, float* flow, int w, int h) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* da0 = (float*)stage_up(c, "sg_a", a0, n * 4); float* da1 = (float*)stage_up(c, "sg_b", a1, n * 4); float* df = (float*)stage_up(c, "sg_c", flow, n * 8);
  float* t = (float*)ensure(c, "sg_d", n * 8); float* o = (float*)ensure(c, "sg_e", n * 8);
  if (!da0 || !da1 || !df || !t || !o) return PF_ERR_NOMEM;
  launch_gauss15_mix(sm, df, t, da0, da1, w, h, c->g15, o);
  return stage_down(c, flow, o, n * 8);
}
int pf_stage_upsample_cubic(pf_ctx* c, const float* flow, int sw, int sh, float* out, int dw, int dh, float scale) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(sw) * sh * 8); float* d = (float*)ensure(c, "sg_b", size_t(dw) * dh * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_upsample_cubic(sm, s, sw, sh, d, dw, dh, scale);
  return stage_down(c, out, d, size_t(dw) * dh * 8);
}
int pf_stage_final(pf_ctx* c, const float* flow, int sw, int sh, int pad_cols, int rows, int pad, float scale, float* out) {
  STAGE_BEGIN(c);
  const int cols = pad_cols - 2 * pad;
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(sw) * sh * 8); float* d = (float*)ensure(c, "sg_b", size_t(cols) * rows * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_final_flow(sm, s, sw, sh, pad_cols, rows, pad, scale, c->g3_1, d);
  return stage_down(c, out, d, size_t(cols) * rows * 8);
}
int pf_stage_adjust_initial_flow(pf_ctx* c, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, int hint, int max_pct, float* flow_out) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* d0 = (float*)stage_up(c, "sg_a", i0, n * 4); float* d1 = (float*)stage_up(c, "sg_b", i1, n * 4); float* da0 = (float*)stage_up(c, "sg_c", a0, n * 4);
  float* da1 = (float*)stage_up(c, "sg_d", a1, n * 4); float* df = (float*)ensure(c, "sg_e", n * 8); float* rt = (float*)ensure(c, "sg_f", 256);
  if (!d0 || !d1 || !da0 || !da1 || !df || !rt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(df, 0, n * 8, sm));
  if (max_pct > 0) launch_adjust_initial_flow(sm, d0, d1, da0, da1, w, h, hint, max_pct, rt, df);
  return stage_down(c, flow_out, df, n * 8);
}
int pf_stage_level(pf_ctx* c, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, const float* flow_in, int hint, int max_pct,
                   float* flow_out) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* d0 = (float*)stage_up(c, "sg_a", i0, n * 4); float* d1 = (float*)stage_up(c, "sg_b", i1, n * 4); float* da0 = (float*)stage_up(c, "sg_c", a0, n * 4);
  float* da1 = (float*)stage_up(c, "sg_d", a1, n * 4);
  float* g0 = (float*)ensure(c, "sg_e", n * 8); float* g1 = (float*)ensure(c, "sg_f", n * 8); uint8_t* gate = (uint8_t*)ensure(c, "sg_g", n);
  LevelBufs b; b.rec = (float*)ensure(c, "sg_rec", sweep2_rec_bytes(w, h)); if (!b.rec) return PF_ERR_NOMEM;
  b.flow_a = (float*)ensure(c, "sg_h", n * 8); b.flow_b = (float*)ensure(c, "sg_i", n * 8); b.blurred = (float*)ensure(c, "sg_j", n * 8); b.tmp = (float*)ensure(c, "sg_k", n * 8);
  const size_t nb = sweep_boundary_elems(w, h);
  unsigned long long* bnd = (unsigned long long*)ensure(c, "sg_l", nb * 16); int* ctrl = (int*)ensure(c, "sg_m", 16); float* rt = (float*)ensure(c, "sg_n", 256);
  if (!d0 || !d1 || !da0 || !da1 || !g0 || !g1 || !gate || !b.flow_a || !b.flow_b || !b.blurred || !b.tmp || !bnd || !ctrl || !rt) return PF_ERR_NOMEM;
  launch_gradients(sm, d0, w, h, g0, c->g3_05);
  launch_gradients(sm, d1, w, h, g1, c->g3_05);
  launch_gate(sm, da0, da1, (int)n, gate);
  launch_fill_u64(sm, bnd, nb * 2, kNotReady);
  HIPCHK(c, hipMemsetAsync(ctrl, 0, 16, sm));
  if (flow_in) HIPCHK(c, hipMemcpyAsync(b.flow_a, flow_in, n * 8, hipMemcpyHostToDevice, sm));
  else {
    HIPCHK(c, hipMemsetAsync(b.flow_a, 0, n * 8, sm));
    if (max_pct > 0 && hint != PF_HINT_UNKNOWN) launch_adjust_initial_flow(sm, d0, d1, da0, da1, w, h, hint, max_pct, rt, b.flow_a);
  }
  float* res = nullptr;
  std::vector<int> box;
  { LevelTable t; t.n = 1; t.w[0] = w; t.h[0] = h; t.off[0] = 0; if (int e = gate_boxes_to_host(c, sm, gate, t, n, box)) return e; }
  const size_t npc = size_t(sweep2_num_wgs_max(w, h));
  int* pcnt = (int*)ensure(c, "sg_pc", 2 * npc * sizeof(int));
  if (!pcnt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(pcnt, 0, 2 * npc * sizeof(int), sm));
  run_level(c, sm, g0, g1, da0, da1, gate, w, h, (w + h) % 2, box.data(), b, bnd, bnd + nb, ctrl, ctrl + 2, &res, pcnt, pcnt + npc);
  int hc[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(hc, ctrl, 16, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, flow_out, res, n * 8)) return e;
  if (hc[1] || hc[3]) return fail(c, PF_ERR_TIMEOUT, "sweep band timed out");
  return 0;
}
int pf_stage_blend_smooth(pf_ctx* c, float* blend, const float* md, int cols, int rows) {
  STAGE_BEGIN(c);
  const size_t n = size_t(cols) * rows;
  float* db = (float*)stage_up(c, "st_blend", blend, n * 4); float* dmd = (float*)stage_up(c, "st_md", md, n * 4);
  if (!db || !dmd) return PF_ERR_NOMEM;
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (int e = stage_down(c, blend, db, n * 4)) return e;
  return check_sweeps(c);
}

// ---- profiling ----
int pf_profile_enable(pf_ctx* c, int on) { if (!c) return PF_ERR_ARG; c->prof = on < 0 ? 0 : (on > 2 ? 1 : on); return 0; }
int pf_profile_reset(pf_ctx* c) { if (!c) return PF_ERR_ARG; for (auto& t : c->prof_tot) t = ProfEntry(); return 0; }
// (a context that has raised warnings lists them as one more entry, "warnings": 0 ms, launches = their number)
int pf_profile_count(pf_ctx* c) { return c ? (int)c->prof_names.size() + (c->warn_count > 0 ? 1 : 0) : 0; }
int pf_profile_get(pf_ctx* c, int idx, char* name, int cap, double* ms, int* launches) {
  if (c && c->warn_count > 0 && idx == (int)c->prof_names.size()) {
    if (name && cap > 0) { strncpy(name, "warnings", cap - 1); name[cap - 1] = 0; }
    if (ms) *ms = 0.0;
    if (launches) *launches = c->warn_count;
    return 0;
  }
  if (!c || idx < 0 || idx >= (int)c->prof_names.size()) return PF_ERR_ARG;
  if (name && cap > 0) { strncpy(name, c->prof_names[idx].c_str(), cap - 1); name[cap - 1] = 0; }
  if (ms) *ms = c->prof_tot[idx].ms;
  if (launches) *launches = c->prof_tot[idx].n;
  return 0;
}

long long pf_last_swept_steps(pf_ctx* c) { return c ? c->last_swept_steps : 0; }

long long pf_level_pixels(int cols, int rows, int* n_levels, long long* sweep_steps) {
  const Geometry g = make_geometry(cols, rows, cols / 20);
  long long steps = 0;
  for (int l = 0; l < g.n; ++l) steps += g.ws[l] + g.hs[l] - 1;
  if (n_levels) *n_levels = g.n;
  if (sweep_steps) *sweep_steps = 2 * steps;
  return (long long)g.Pexact;
}
double pf_algorithmic_bytes(int cols, int rows) {  // B_alg = 472.75*P + 102.4*C*R
  return 472.75 * (double)pf_level_pixels(cols, rows, nullptr, nullptr) + 102.4 * (double)cols * rows;
}

}  // extern "C"
