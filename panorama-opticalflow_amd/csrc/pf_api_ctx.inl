// Part of pf_api.hip (one translation unit, split along its seams in round 5): the context, its error / warning strings, the named grow-only HBM arena, profiling events, pyramid geometry and argument checks.
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
  int warned_streams = 0;                 // the largest stream count the hardware-queue warning was already raised for (once per condition, not per call)
  std::map<std::string, DevBuf> bufs;  // named grow-only arena: everything a solve needs stays resident
  Gauss g5, g3_05, g3_1, g15;
  int prof = 0;   // 0 off, 1 every kernel family, 2 only the dominant family (the sweeps): fewer events in a timed region
  pf_config cfg;  // scheduling knobs (pf_create_cfg); results never depend on them
  pf_solver_params sp = {kPyrScaleFactor, kSmoothnessCoef, kVerticalRegularizationCoef, kHorizontalRegularizationCoef, kGradientStepSize, kDownscaleFactor, 0.0f};   // PixFlow's constructor arguments (pf_set_solver_params)
  SolverCoef cf;  // ... as the sweep kernels take them (guard_min: see pf_common.hpp)
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
// The variable is read ONCE per process (the runtime reads it once, too -- at its initialisation; a value set later changes nothing), and a
// context raises the warning once per condition (a stream count it has not warned about yet), not once per call.
void check_hw_queues(pf_ctx* c, int needed, const char* what) {
  static const char* const e = getenv("GPU_MAX_HW_QUEUES");
  static const int queues = e ? atoi(e) : 4;
  if (queues <= 0 || needed <= queues || needed <= c->warned_streams) return;
  c->warned_streams = needed;
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
Geometry make_geometry(int cols, int rows, int pad, float pyrScale = kPyrScaleFactor) {
  Geometry g; g.cols = cols; g.rows = rows; g.pad = pad; g.ce = cols + 2 * pad;
  g.w0 = int(g.ce * kDownscaleFactor); g.h0 = int(rows * kDownscaleFactor);
  g.ws = {g.w0}; g.hs = {g.h0};
  while ((int)g.ws.size() < kPyrMaxLevels) {
    const int nw = int(g.ws.back() * pyrScale + 0.5f), nh = int(g.hs.back() * pyrScale + 0.5f);
    if (nw >= g.ws.back() && nh >= g.hs.back()) break;   // (a scale that does not shrink the image any more: pf_set_solver_params keeps it below 1, this keeps the loop finite regardless)
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
